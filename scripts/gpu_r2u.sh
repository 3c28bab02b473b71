# Round 2, GPU call U: join on the RCNN event; plane-major Winograd weight-gradient tiles
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | grep -E "passed|failed|^E  " | head
timeout 200 python bench.py --no-cpu-baseline --phases 10 > gpurun_out/r2u_bench.json 2>/dev/null
python - <<PY
import json
d = json.load(open("gpurun_out/r2u_bench.json"))
print(round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms;", d["phases_ms"])
for k, v in d["roofline"]["all_conv_kernels"].items():
    if "bwd_weight" in k: print("   %-46s %5.1f launches %7.1f TF/s %7.1f GB/s %7.3f ms/step" % (k, v["launches_per_step"], v["tflops"], v["gbs"], v["ms_per_step"]))
PY
echo "== Winograd on, 3x3 layers"; timeout 120 python scripts/bench_conv.py 3x3 2>&1 | grep -v amdgpu.ids | cut -c1-140
