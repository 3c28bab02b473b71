# Round-4 A/B (3): half-storage kernels — LDS fragment reads one k-step ahead of the MFMAs (k_conv_hs, k_wgrad_hs_tr), column-sum MFMAs with VGPR accumulators (no AGPR round trip per k-step)

R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python -m pytest tests/test_gpu_hs.py tests/test_gpu_half.py -x -q -m gpu 2>&1 | tail -n 4

B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline --phases 20"
B5="python bench.py --workload frcnn_r50_coco --dtype f16 --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --phases 20"
L=luminoth_amd/csrc
cp $L/libluminoth_hip.so $L/libluminoth_hip_new.so
run() {
  $B > /tmp/o.json 2>/tmp/o.err || tail -n 5 /tmp/o.err
  python - /tmp/o.json "$1 f32" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phases_ms']
print(sys.argv[2], '%.3f ms (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']), 'fwd %.3f' % p.get('trunk_fwd_done', 0), 'proposals %.3f' % (p['aux:proposals_done']-p['rpn_heads_done']), 'rcnn %.3f' % (p['aux:rcnn_bwd_done']-p['aux:proposals_done']), 'join %.3f' % p['joined'], 'prefix %.3f' % p['next_prefix_done'], 'bwd_done %.3f tails %.3f' % (p['trunk_bwd_data_done'], p['tails_done']))
P
  $B5 > /tmp/o.json 2>/tmp/o.err || tail -n 5 /tmp/o.err
  python - /tmp/o.json "$1 f16" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phases_ms']
print(sys.argv[2], '%.3f ms (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']), 'fwd %.3f' % p.get('trunk_fwd_done', 0), 'proposals %.3f' % (p['aux:proposals_done']-p['rpn_heads_done']), 'rcnn %.3f' % (p['aux:rcnn_bwd_done']-p['aux:proposals_done']), 'join %.3f' % p['joined'], 'prefix %.3f' % p['next_prefix_done'], 'bwd_done %.3f tails %.3f' % (p['trunk_bwd_data_done'], p['tails_done']))
P
}
run new
cp $L/libluminoth_hip_base.so $L/libluminoth_hip.so
run base
cp $L/libluminoth_hip_new.so $L/libluminoth_hip.so
run new

cp $L/libluminoth_hip_base.so $L/libluminoth_hip.so
run base
cp $L/libluminoth_hip_new.so $L/libluminoth_hip.so
