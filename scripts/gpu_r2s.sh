# Round 2, GPU call S: stream priority experiment; early-tail batch size; suite re-check
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^E  " | head
for cfg in "0 8" "1 8" "0 5" "1 5"; do
  set -- $cfg
  LUMINOTH_AMD_MAIN_PRIORITY=$1 LUMINOTH_AMD_EARLY_TAILS=$2 timeout 200 python bench.py --no-cpu-baseline --no-roofline --phases 10 > gpurun_out/r2s_bench.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r2s_bench.json"))
p = d["phases_ms"]
print("priority $1 early $2:", round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms; fwd", p["trunk_fwd_done"], "rpn_bwd", p["rpn_bwd_done"], "aux_bwd", p["aux:rcnn_bwd_done"], "joined", p["joined"], "bwd_data_done", p["trunk_bwd_data_done"], "wgrad_joined", p["wgrad_stream_joined"], "tails_done", p["tails_done"], "next", p["next_step_start"])
PY
done
