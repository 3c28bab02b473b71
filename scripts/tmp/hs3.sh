R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/f16hs_line.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/f16hs_line.json'))
print(d['ms_per_step'], d['value'])
r=d['roofline']
print({k:r[k] for k in ('bound','achieved','peak','frac','kernel','ms_per_launch','launches_per_step')})
print(json.dumps(r['whole_step']))
for k,v in r['all_conv_kernels'].items(): print(k, v)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_f16hs -o r03 -- python $R/bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_r03_f16hs_line.json 2> $R/gpurun_out/prof_r03_f16hs.err
cd $R
python scripts/make_profile_summary.py gpurun_out/prof_r03_f16hs r03_f16hs_bench "python bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline --no-roofline" 60 4 | head -90
