# Round 2, GPU call W: LDS footprint of the fused ROI pool + mean inside the step; NMS row prefetch
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "nms or roi or proposal" 2>&1 | grep -E "passed|failed|^E  " | head -5
timeout 120 python scripts/bench_nms.py 2>&1 | tail -1
for cs in 8 4 0; do
  LMH_ROI_MEAN_CS=$cs timeout 200 python bench.py --no-cpu-baseline --no-roofline --phases 10 > gpurun_out/r2w_bench.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r2w_bench.json"))
p = d["phases_ms"]
print("roi mean cs $cs:", round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms; rpn_heads", p["rpn_heads_done"], "proposals", p["aux:proposals_done"], "rcnn_enq", p["aux:rcnn_enqueue"], "rcnn_loss", p["aux:rcnn_loss_done"], "rcnn_bwd", p["aux:rcnn_bwd_done"], "prefix", p["next_prefix_done"], "joined", p["joined"])
PY
done
