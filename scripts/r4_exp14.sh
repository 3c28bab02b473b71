# Round-4 A/B (5): fused ROI pooling with 4-channel slabs (64 KB of LDS per block, lmh_set_option("roi_mean_cs", 4)) in the
# half-storage step, where its 128 KB blocks wait longest for a CU (396 us inside the step against 116 alone).
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
LMH_OPT_ROI_MEAN_CS=4 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "roi" 2>&1 | tail -n 2
B5="python bench.py --workload frcnn_r50_coco --dtype f16 --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --phases 20"
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline --phases 20"
run() {
  $2 > /tmp/o.json 2>/tmp/o.err || tail -n 5 /tmp/o.err
  python - /tmp/o.json "$1" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); p=d['phases_ms']
print(sys.argv[2], '%.3f ms (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']), 'proposals %.3f' % (p['aux:proposals_done']-p['rpn_heads_done']), 'rcnn %.3f' % (p['aux:rcnn_bwd_done']-p['aux:proposals_done']), 'join %.3f' % p['joined'], 'prefix %.3f' % p['next_prefix_done'], 'bwd_done %.3f tails %.3f' % (p['trunk_bwd_data_done'], p['tails_done']))
P
}
run "f16 cs8" "$B5"
LMH_OPT_ROI_MEAN_CS=4 run "f16 cs4" "$B5"
run "f16 cs8" "$B5"
LMH_OPT_ROI_MEAN_CS=4 run "f16 cs4" "$B5"
LMH_OPT_ROI_MEAN_CS=4 run "f32 cs4" "$B"
run "f32 cs8" "$B"
