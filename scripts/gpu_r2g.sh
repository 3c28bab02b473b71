# Round 2, GPU call G: fused colsum in k_wgrad_1x1, tails, NMS ablations, half-kernel prefetch A/B
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_half.py -m gpu -q -x 2>&1 | tail -4
python -m pytest tests/test_gpu_model.py tests/test_gpu_ssd.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed|Error" | head -30
python bench.py --no-cpu-baseline > gpurun_out/r2g_bench.json 2> gpurun_out/r2g_bench.err; head -c 420 gpurun_out/r2g_bench.json; echo; tail -2 gpurun_out/r2g_bench.err
for pf in 1 2; do
  LMH_HALF_PF=$pf python bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline > gpurun_out/r2g_coco_f16_pf$pf.json 2> /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r2g_coco_f16_pf$pf.json"))
print("PF$pf", d["value"], "img/s", d["ms_per_step"], "ms;", d["roofline"]["kernel"], d["roofline"]["bound"], d["roofline"]["frac"])
for k, v in list(d["roofline"]["all_conv_kernels"].items())[:5]: print("   %-44s %5.1f launches %7.1f TF/s %7.1f GB/s %7.3f ms/step" % (k, v["launches_per_step"], v["tflops"], v["gbs"], v["ms_per_step"]))
PY
done
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2 4 7; do
  LMH_NMS_DBG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_nms$dbg -o r -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  python - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/prof_nms$dbg/r_kernel_stats.csv")))
for r in rows:
    if 'nms' in r['Name'] or 'rcnn_target' in r['Name']: print("dbg $dbg", r['Name'].split('(')[0][-30:], float(r['TotalDurationNs']) / float(r['Calls']) / 1e3, "us")
PY
done
