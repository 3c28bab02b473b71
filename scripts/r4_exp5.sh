R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04g; mkdir -p $O
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline --phases 20"
B5="python bench.py --workload frcnn_r50_coco --dtype f16 --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --phases 20"
cd $R
for m in 0 3 2 4 0 3; do
  LMH_OPT_NMS_STAGE_MULT=$m $B > $O/f32_nms$m.json 2>/dev/null
  python - $O/f32_nms$m.json $m <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phases_ms']
print('f32 mult', sys.argv[2], '%.3f ms' % d['ms_per_step'], 'proposals %.3f' % (p['aux:proposals_done']-p['rpn_heads_done']), 'join %.3f' % p['joined'])
P
done
for m in 0 3 0 3; do
  LMH_OPT_NMS_STAGE_MULT=$m $B5 > $O/f16_nms$m.json 2>/dev/null
  python - $O/f16_nms$m.json $m <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phases_ms']
print('f16 mult', sys.argv[2], '%.3f ms' % d['ms_per_step'], 'proposals %.3f' % (p['aux:proposals_done']-p['rpn_heads_done']), 'join %.3f' % p['joined'])
P
done
