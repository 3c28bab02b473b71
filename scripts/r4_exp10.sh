R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv_fwd_bwd or loss or rcnn" 2>&1 | tail -n 3
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline --phases 20"
B5="python bench.py --workload frcnn_r50_coco --dtype f16 --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --phases 20"
run() {
  $B > /tmp/o.json 2>/tmp/o.err || tail -n 5 /tmp/o.err
  python - /tmp/o.json "$1 f32" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phases_ms']
print(sys.argv[2], '%.3f ms (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']), 'proposals %.3f' % (p['aux:proposals_done']-p['rpn_heads_done']), 'rcnn %.3f' % (p['aux:rcnn_bwd_done']-p['aux:proposals_done']), 'join %.3f' % p['joined'], 'prefix %.3f' % p['next_prefix_done'])
P
  $B5 > /tmp/o.json 2>/tmp/o.err || tail -n 5 /tmp/o.err
  python - /tmp/o.json "$1 f16" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phases_ms']
print(sys.argv[2], '%.3f ms (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']), 'proposals %.3f' % (p['aux:proposals_done']-p['rpn_heads_done']), 'rcnn %.3f' % (p['aux:rcnn_bwd_done']-p['aux:proposals_done']), 'join %.3f' % p['joined'], 'prefix %.3f' % p['next_prefix_done'])
P
}
run new
LMH_OPT_ROI_MEAN_CS=4 run roi_mean_cs4
run new
LMH_OPT_ROI_MEAN_CS=4 run roi_mean_cs4
bash scripts/r4_trace.sh r04u
