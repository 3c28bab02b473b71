# Round 2, GPU call K: bf16x3 (fp32 arithmetic on the bf16 matrix pipe): exactness, accuracy, e2e parity, step time
R=$GRAFT_REPO_ROOT
cd $R
timeout 400 python -m pytest tests/test_gpu_x3.py -m gpu -q -s 2>&1 | grep -E "native fp32|passed|failed|Error|^E " | sort | uniq -c | sort -rn | head -40
for dt in f32 bf16x3; do
  timeout 200 python bench.py --no-cpu-baseline --dtype $dt > gpurun_out/r2k_bench_$dt.json 2> gpurun_out/r2k_bench_$dt.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r2k_bench_$dt.json"))
print("$dt", d["value"], "img/s", d["ms_per_step"], "ms; loss", d["config"]["final_total_loss"], ";", d["roofline"]["kernel"], d["roofline"]["bound"], round(d["roofline"]["frac"], 3))
tot = 0
for k, v in list(d["roofline"]["all_conv_kernels"].items())[:12]:
    print("   %-46s %5.1f launches %7.1f TF/s %7.1f GB/s %7.3f ms/step" % (k, v["launches_per_step"], v["tflops"], v["gbs"], v["ms_per_step"]))
print("   serial conv sum %.3f ms" % sum(v["ms_per_step"] for v in d["roofline"]["all_conv_kernels"].values()))
PY
  tail -2 gpurun_out/r2k_bench_$dt.err
done
LUMINOTH_AMD_X3_WINOGRAD=1 timeout 200 python bench.py --no-cpu-baseline --dtype bf16x3 > gpurun_out/r2k_bench_x3w.json 2>/dev/null
python - <<PY
import json
d = json.load(open("gpurun_out/r2k_bench_x3w.json"))
print("bf16x3 + winograd layers", d["value"], "img/s", d["ms_per_step"], "ms")
print("   serial conv sum %.3f ms" % sum(v["ms_per_step"] for v in d["roofline"]["all_conv_kernels"].values()))
PY
