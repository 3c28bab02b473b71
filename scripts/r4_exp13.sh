# Round-4 A/B (4): RCNN loss gradients decoupled from the sums (lmh_rcnn_loss_grad where the loss sits, the one-block-per-image
# sum kernel behind the join; LUMINOTH_AMD_RCNN_LOSS_LATE=0 = sums in front of the RCNN backward as before), same library.
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_plan.py -x -q -m gpu -k "loss or plan" 2>&1 | tail -n 3
python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "test_train_step_matches_oracle or fused_two_stream" 2>&1 | tail -n 3
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline --phases 20"
B5="python bench.py --workload frcnn_r50_coco --dtype f16 --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --phases 20"
run() {
  $B > /tmp/o.json 2>/tmp/o.err || tail -n 5 /tmp/o.err
  python - /tmp/o.json "$1 f32" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phases_ms']
print(sys.argv[2], '%.3f ms (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']), 'proposals %.3f' % (p['aux:proposals_done']-p['rpn_heads_done']), 'rcnn %.3f' % (p['aux:rcnn_bwd_done']-p['aux:proposals_done']), 'join %.3f' % p['joined'], 'prefix %.3f' % p['next_prefix_done'], 'bwd_done %.3f tails %.3f' % (p['trunk_bwd_data_done'], p['tails_done']))
P
  $B5 > /tmp/o.json 2>/tmp/o.err || tail -n 5 /tmp/o.err
  python - /tmp/o.json "$1 f16" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phases_ms']
print(sys.argv[2], '%.3f ms (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']), 'proposals %.3f' % (p['aux:proposals_done']-p['rpn_heads_done']), 'rcnn %.3f' % (p['aux:rcnn_bwd_done']-p['aux:proposals_done']), 'join %.3f' % p['joined'], 'prefix %.3f' % p['next_prefix_done'], 'bwd_done %.3f tails %.3f' % (p['trunk_bwd_data_done'], p['tails_done']))
P
}
run late
LUMINOTH_AMD_RCNN_LOSS_LATE=0 run early
run late
LUMINOTH_AMD_RCNN_LOSS_LATE=0 run early
