python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/tq.log; tail -2 gpurun_out/tq.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/benchq.json 2> gpurun_out/benchq.err; cut -c1-330 gpurun_out/benchq.json; tail -2 gpurun_out/benchq.err
bash scripts/gpu_prof.sh q > /dev/null 2>&1
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof_q/r01_kernel_stats.csv')))
for r in rows:
    n = r['Name'].split('(')[0].replace('void ', '')[:50]
    if 'conv' in n: continue
    t = float(r['TotalDurationNs']) / 7e3
    if t > 40: print('%-52s calls/step %5.1f us/step %7.1f avg %7.1f' % (n, int(r['Calls']) / 7, t, float(r['AverageNs']) / 1e3))
print('total us/step', sum(float(r['TotalDurationNs']) for r in rows) / 7e3)
PY
