R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_x3.py -x -q -m gpu -k "conv or stem or head or x3" 2>&1 | tail -n 3
python -m pytest tests/test_gpu_model.py tests/test_gpu_plan.py -x -q -m gpu -k "(matches_oracle and not vgg) or plan" 2>&1 | tail -n 3
python scripts/bench_conv.py "conv1" 2>&1 | grep -v amdgpu.ids | tail -n 2
python scripts/bench_conv.py "rpn 1x1" 2>&1 | grep -v amdgpu.ids | tail -n 2
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline --phases 20"
B5="python bench.py --workload frcnn_r50_coco --dtype f16 --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --phases 20"
L=luminoth_amd/csrc
cp $L/libluminoth_hip.so $L/libluminoth_hip_new.so
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
cp $L/libluminoth_hip_base.so $L/libluminoth_hip.so
python scripts/bench_conv.py "conv1" 2>&1 | grep -v amdgpu.ids | tail -n 2
python scripts/bench_conv.py "rpn 1x1" 2>&1 | grep -v amdgpu.ids | tail -n 2
run base
cp $L/libluminoth_hip_new.so $L/libluminoth_hip.so
run new
cp $L/libluminoth_hip_base.so $L/libluminoth_hip.so
run base
cp $L/libluminoth_hip_new.so $L/libluminoth_hip.so
