# Round-4 A/B: pipelined NMS scan, branch-free NMS mask loop, epilogues with every load in front of the first store
# (conv_fast / conv_hs / Winograd output / Linear head / predicated kernels) against the library of the previous commit.
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ref_tf_golden.py tests/test_gpu_hs.py tests/test_gpu_plan.py -x -q -m gpu 2>&1 | tail -n 15
python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "matches_oracle" 2>&1 | tail -n 6
# (the nms_pipe option existed only at that commit: 0 selected the round-3 scan kernel, deleted afterwards)
for p in 1 0; do LMH_OPT_NMS_PIPE=$p python scripts/bench_nms.py 2>&1 | tail -n 1 | sed "s/^/nms_pipe=$p /"; done
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline --phases 20"
B5="python bench.py --workload frcnn_r50_coco --dtype f16 --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --phases 20"
L=luminoth_amd/csrc
cp $L/libluminoth_hip.so $L/libluminoth_hip_new.so
run() {
  $B > /tmp/o.json 2>/tmp/o.err || tail -n 5 /tmp/o.err
  python - /tmp/o.json "$1 f32" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phases_ms']
print(sys.argv[2], '%.3f ms (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']), 'fwd %.3f' % p.get('trunk_fwd_done', 0), 'proposals %.3f' % (p['aux:proposals_done']-p['rpn_heads_done']), 'rcnn %.3f' % (p['aux:rcnn_bwd_done']-p['aux:proposals_done']), 'join %.3f' % p['joined'], 'prefix %.3f' % p['next_prefix_done'])
P
  $B5 > /tmp/o.json 2>/tmp/o.err || tail -n 5 /tmp/o.err
  python - /tmp/o.json "$1 f16" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phases_ms']
print(sys.argv[2], '%.3f ms (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']), 'fwd %.3f' % p.get('trunk_fwd_done', 0), 'proposals %.3f' % (p['aux:proposals_done']-p['rpn_heads_done']), 'rcnn %.3f' % (p['aux:rcnn_bwd_done']-p['aux:proposals_done']), 'join %.3f' % p['joined'], 'prefix %.3f' % p['next_prefix_done'])
P
}
run new
cp $L/libluminoth_hip_base.so $L/libluminoth_hip.so
run base
cp $L/libluminoth_hip_new.so $L/libluminoth_hip.so
run new
LMH_OPT_NMS_PIPE=0 run new_oldscan
cp $L/libluminoth_hip_base.so $L/libluminoth_hip.so
run base
cp $L/libluminoth_hip_new.so $L/libluminoth_hip.so
