# Round 2, GPU call C: half-precision conv kernels — unit tests, e2e test, bench lines of config 5 in f32 / f16 / bf16
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_half.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -25
for dt in f32 f16 bf16; do
  python bench.py --workload frcnn_r50_coco --dtype $dt --no-cpu-baseline > gpurun_out/r2c_bench_coco_$dt.json 2> gpurun_out/r2c_bench_coco_$dt.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r2c_bench_coco_$dt.json"))
    print("$dt", d["value"], "img/s", d["ms_per_step"], "ms loss", d["config"]["final_total_loss"], d["roofline"]["kernel"], d["roofline"]["achieved"])
    for k, v in list(d["roofline"]["all_conv_kernels"].items())[:8]: print("   %-44s %5.1f launches %7.1f TF/s %7.3f ms/step" % (k, v["launches_per_step"], v["tflops"], v["ms_per_step"]))
except Exception as e:
    print("$dt FAILED", e); print(open("gpurun_out/r2c_bench_coco_$dt.err").read()[-1500:])
PY
done
python bench.py --dtype f16 --no-cpu-baseline > gpurun_out/r2c_bench_r50_f16.json 2> gpurun_out/r2c_bench_r50_f16.err; head -c 300 gpurun_out/r2c_bench_r50_f16.json; echo
