# Round 2, GPU call J: model-level suite without the graph path; cross-step prefix prefetch A/B with event timelines
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ssd.py tests/test_gpu_predict.py tests/test_gpu_eval.py tests/test_gpu_dataset.py -m gpu -q 2>&1 | tail -25
for la in "" "--no-lookahead"; do
  timeout 200 python bench.py --no-cpu-baseline --no-roofline --phases 10 $la > gpurun_out/r2j_bench$la.json 2> gpurun_out/r2j_bench$la.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r2j_bench$la.json"))
print("lookahead '$la':", d["value"], "img/s", d["ms_per_step"], "ms")
for k, v in d.get("phases_ms", {}).items(): print("    %-26s %7.3f" % (k, v))
PY
  tail -2 gpurun_out/r2j_bench$la.err
done
