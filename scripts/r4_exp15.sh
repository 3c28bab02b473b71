# Round-4 sweep: how many of the LAST trunk layers (backward order) take their weight gradient on the main stream instead of
# the weight-gradient stream (LUMINOTH_AMD_INLINE_LAYERS, default 4), final code, one box.
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline --phases 20"
B5="python bench.py --workload frcnn_r50_coco --dtype f16 --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --phases 20"
run() {
  $2 > /tmp/o.json 2>/tmp/o.err || tail -n 5 /tmp/o.err
  python - /tmp/o.json "$1" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phases_ms']
print(sys.argv[2], '%.3f ms (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']), 'join %.3f' % p['joined'], 'bwd_done %.3f wgrad_joined %.3f tails %.3f' % (p['trunk_bwd_data_done'], p['wgrad_stream_joined'], p['tails_done']))
P
}
run "f32 inline4" "$B"
LUMINOTH_AMD_INLINE_LAYERS=10 run "f32 inline10" "$B"
LUMINOTH_AMD_INLINE_LAYERS=0 run "f32 inline0" "$B"
LUMINOTH_AMD_INLINE_LAYERS=10 run "f16 inline10" "$B5"
run "f16 inline4" "$B5"
