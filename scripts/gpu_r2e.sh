# Round 2, GPU call E: new NMS reduce, optimizer/dropout kernels, half-precision e2e thresholds; full suite + bench
R=$GRAFT_REPO_ROOT
cd $R
( time python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) 2>&1 | tail -16
python bench.py --no-cpu-baseline > gpurun_out/r2e_bench.json 2> gpurun_out/r2e_bench.err; head -c 420 gpurun_out/r2e_bench.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2e -o r02 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_r2e.log 2>&1
cd $R; python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_r2e/r02_kernel_stats.csv')))
for r in rows:
    n=r['Name'].replace('void ','').split('(')[0]
    if any(k in n for k in ('nms','rcnn_target','roi_','rpn_target')):
        print('  %-50s us/step %8.1f'%(n[:50],float(r['TotalDurationNs'])/7/1e3))
PY
python scripts/timeline.py gpurun_out/prof_r2e/r02_kernel_trace.csv | grep -E "queue|gap|step" | head -24
