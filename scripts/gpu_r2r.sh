# Round 2, GPU call R: early tail flush on the idle aux stream
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -3
for et in 0 12 8; do
  LUMINOTH_AMD_EARLY_TAILS=$et timeout 200 python bench.py --no-cpu-baseline --no-roofline --phases 10 > gpurun_out/r2r_bench_$et.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r2r_bench_$et.json"))
p = d["phases_ms"]
print("early tails $et:", round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms; joined", p["joined"], "bwd_data_done", p["trunk_bwd_data_done"], "wgrad_joined", p["wgrad_stream_joined"], "tails_done", p["tails_done"], "next", p["next_step_start"])
PY
done
