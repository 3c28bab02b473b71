# Round 2, GPU call F: deferred tails + scalar NMS: full suite, bench, timeline
R=$GRAFT_REPO_ROOT
cd $R
( time python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) 2>&1 | tail -16
python bench.py --no-cpu-baseline > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err; head -c 420 gpurun_out/r2f_bench.json; echo; tail -2 gpurun_out/r2f_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2f -o r02 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_r2f.log 2>&1
cd $R; python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_r2f/r02_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms/step', tot/7/1e6, 'launches/step', sum(float(r['Calls']) for r in rows)/7)
for r in rows:
    n=r['Name'].replace('void ','').split('(')[0]
    if any(k in n for k in ('nms','rcnn_target','roi_','rpn_target','tail','splitk','bn_','colsum','act_bwd')):
        print('  %-50s calls/step %5.1f us/step %8.1f'%(n[:50],float(r['Calls'])/7,float(r['TotalDurationNs'])/7/1e3))
PY
python scripts/timeline.py gpurun_out/prof_r2f/r02_kernel_trace.csv | grep -E "queue|gap|step" | head -24
