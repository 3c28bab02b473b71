# Round 2, GPU call O: full GPU suite + bf16x3 step with the per-pass pipelines
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
for dt in f32 bf16x3; do
  timeout 200 python bench.py --no-cpu-baseline --dtype $dt --phases 10 > gpurun_out/r2o_bench_$dt.json 2> gpurun_out/r2o_bench_$dt.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r2o_bench_$dt.json"))
print("$dt", d["value"], "img/s", d["ms_per_step"], "ms; loss", d["config"]["final_total_loss"], ";", d["roofline"]["kernel"], d["roofline"]["bound"], round(d["roofline"]["frac"], 3))
print("   serial conv sum %.3f ms" % sum(v["ms_per_step"] for v in d["roofline"]["all_conv_kernels"].values()))
print("   phases", d.get("phases_ms"))
PY
  tail -1 gpurun_out/r2o_bench_$dt.err
done
