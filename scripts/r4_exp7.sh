R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ref_tf_golden.py tests/test_gpu_ssd.py tests/test_gpu_predict.py -x -q -m gpu -k "proposal or nms or detect or predict or ssd" 2>&1 | tail -n 3
python scripts/bench_nms.py 2>&1 | grep -v amdgpu.ids | tail -n 2
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline --phases 20"
B5="python bench.py --workload frcnn_r50_coco --dtype f16 --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --phases 20"
for rep in 1 2; do
  $B > /tmp/o.json 2>/dev/null
  python - /tmp/o.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phases_ms']
print('f32 %.3f ms' % d['ms_per_step'], 'proposals %.3f' % (p['aux:proposals_done']-p['rpn_heads_done']), 'rcnn %.3f' % (p['aux:rcnn_bwd_done']-p['aux:proposals_done']), 'join %.3f' % p['joined'], 'prefix %.3f' % p['next_prefix_done'])
P
  $B5 > /tmp/o.json 2>/dev/null
  python - /tmp/o.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phases_ms']
print('f16 %.3f ms' % d['ms_per_step'], 'proposals %.3f' % (p['aux:proposals_done']-p['rpn_heads_done']), 'rcnn %.3f' % (p['aux:rcnn_bwd_done']-p['aux:proposals_done']), 'join %.3f' % p['joined'], 'prefix %.3f' % p['next_prefix_done'])
P
done
