# Round 2, GPU call D: half path with large tiles, HIP-graph replay, full GPU suite
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_half.py tests/test_gpu_model.py -m gpu -q -x -s -k "half or hip_graph" 2>&1 | grep -E "passed|failed|half-precision|gradient cosine|Error|assert" | head -20
python bench.py --no-cpu-baseline > gpurun_out/r2d_bench.json 2> gpurun_out/r2d_bench.err; head -c 900 gpurun_out/r2d_bench.json | cut -c1-900; echo; tail -2 gpurun_out/r2d_bench.err
python bench.py --no-cpu-baseline --no-graph --no-roofline > gpurun_out/r2d_bench_eager.json 2>/dev/null; head -c 330 gpurun_out/r2d_bench_eager.json; echo
for dt in f16 bf16; do
  python bench.py --workload frcnn_r50_coco --dtype $dt --no-cpu-baseline > gpurun_out/r2d_bench_coco_$dt.json 2> gpurun_out/r2d_bench_coco_$dt.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r2d_bench_coco_$dt.json"))
    print("$dt", d["value"], "img/s", d["ms_per_step"], "ms loss", d["config"]["final_total_loss"], d["config"]["schedule"], d["roofline"]["kernel"], d["roofline"]["achieved"])
    for k, v in list(d["roofline"]["all_conv_kernels"].items())[:8]: print("   %-44s %5.1f launches %7.1f TF/s %7.3f ms/step" % (k, v["launches_per_step"], v["tflops"], v["ms_per_step"]))
except Exception as e:
    print("$dt FAILED", e); print(open("gpurun_out/r2d_bench_coco_$dt.err").read()[-1500:])
PY
done
