# Round 2, GPU call T: next-batch anchor targets on the idle aux stream
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^E  " | head
for la in "" "--no-lookahead"; do
  timeout 200 python bench.py --no-cpu-baseline --no-roofline --phases 10 $la > gpurun_out/r2t_bench.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r2t_bench.json"))
print("'$la':", round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms;", d["phases_ms"])
PY
done
