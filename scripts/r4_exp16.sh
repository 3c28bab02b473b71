# Round-4 A/B (6): backward set of the Winograd weight transforms behind the RPN heads (LUMINOTH_AMD_WINO_BWD_LATE=1) instead of
# at the start of the step beside the first trunk convolution ("late": the whole set on the weight-gradient stream in front of the RPN backward;
# "split" = 2: the RPN layer there, the trunk layers on the main stream behind the RPN loss); same library.
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
LUMINOTH_AMD_WINO_BWD_LATE=2 python -m pytest tests/test_gpu_plan.py -x -q -m gpu 2>&1 | tail -n 2
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline --phases 20"
run() {
  $B > /tmp/o.json 2>/tmp/o.err || tail -n 5 /tmp/o.err
  python - /tmp/o.json "$1" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phases_ms']
print(sys.argv[2], '%.3f ms (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']), 'fwd %.3f heads %.3f' % (p['trunk_fwd_done'], p['rpn_heads_done']), 'rpn_bwd %.3f' % p.get('side:rpn_bwd_done', 0), 'join %.3f' % p['joined'], 'bwd_done %.3f' % p['trunk_bwd_data_done'])
P
}
LUMINOTH_AMD_WINO_BWD_LATE=2 run "split"
run "early"
LUMINOTH_AMD_WINO_BWD_LATE=2 run "split"
run "early"
