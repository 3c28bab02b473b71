# Round 2, GPU call I: fused ROI pool+mean; NMS ablation; full GPU suite
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_half.py -m gpu -q 2>&1 | tail -8
timeout 500 python -m pytest tests/test_gpu_model.py tests/test_gpu_ssd.py tests/test_gpu_predict.py tests/test_gpu_eval.py tests/test_gpu_dataset.py -m gpu -q 2>&1 | tail -40
timeout 120 python scripts/bench_roi.py 2>&1 | tail -8
for d in 0 1 2 3; do LMH_NMS_DBG=$d timeout 120 python scripts/bench_nms.py 2>&1 | tail -1; done
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r2i_bench.json 2> gpurun_out/r2i_bench.err; head -c 420 gpurun_out/r2i_bench.json; echo; tail -2 gpurun_out/r2i_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2i -o r02 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_r2i.log 2>&1
cd $R; python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_r2i/r02_kernel_stats.csv')))
for r in rows:
    n=r['Name'].replace('void ','').split('(')[0]
    if any(k in n for k in ('nms','rcnn_target','roi_','rpn_target','tail','splitk','act_bwd','sort','spatial')):
        print('  %-50s calls/step %5.1f us/step %8.1f'%(n[:50],float(r['Calls'])/7,float(r['TotalDurationNs'])/7/1e3))
PY
python scripts/timeline.py gpurun_out/prof_r2i/r02_kernel_trace.csv | grep -E "queue|gap|step" | head -20
