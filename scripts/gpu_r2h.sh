# Round 2, GPU call H: every sub-command under its own timeout
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_half.py -m gpu -q -x 2>&1 | tail -5
timeout 400 python -m pytest tests/test_gpu_model.py tests/test_gpu_ssd.py tests/test_gpu_predict.py tests/test_gpu_eval.py tests/test_gpu_dataset.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed|Error" | head -30
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r2h_bench.json 2> gpurun_out/r2h_bench.err; head -c 420 gpurun_out/r2h_bench.json; echo; tail -2 gpurun_out/r2h_bench.err
LUMINOTH_AMD_FUSE_MASK=1 timeout 200 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r2h_bench_fusemask.json 2>/dev/null; head -c 330 gpurun_out/r2h_bench_fusemask.json | tail -c 130; echo
for pf in 1 2; do
  LMH_HALF_PF=$pf timeout 200 python bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline > gpurun_out/r2h_coco_f16_pf$pf.json 2> /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r2h_coco_f16_pf$pf.json"))
print("PF$pf", d["value"], "img/s", d["ms_per_step"], "ms;", d["roofline"]["kernel"], d["roofline"]["bound"], d["roofline"]["frac"])
for k, v in list(d["roofline"]["all_conv_kernels"].items())[:4]: print("   %-44s %5.1f launches %7.1f TF/s %7.1f GB/s %7.3f ms/step" % (k, v["launches_per_step"], v["tflops"], v["gbs"], v["ms_per_step"]))
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2h -o r02 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_r2h.log 2>&1
cd $R; python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_r2h/r02_kernel_stats.csv')))
for r in rows:
    n=r['Name'].replace('void ','').split('(')[0]
    if any(k in n for k in ('nms','rcnn_target','roi_','rpn_target','tail','splitk','act_bwd')):
        print('  %-50s calls/step %5.1f us/step %8.1f'%(n[:50],float(r['Calls'])/7,float(r['TotalDurationNs'])/7/1e3))
PY
python scripts/timeline.py gpurun_out/prof_r2h/r02_kernel_trace.csv | grep -E "queue|gap|step" | head -20
