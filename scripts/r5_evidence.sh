# Round 5 evidence (one gpurun call = one box).  Bench lines (cold driver command first), timelines, the other workloads,
# rocprofv3 kernel traces of the production steps, the kernel traces bench.py's own rocprofv3 children took for `roofline.frac`
# (kept with --rocprof-keep), and PMC passes (MFMA busy, FETCH_SIZE, WRITE_SIZE) over the serialised roofline steps of both legs.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_evidence; rm -rf $O; mkdir -p $O
cd $R
# 1. the driver's command as the FIRST process on the box, then the warm default run (both with their rocprofv3 children)
python bench.py --gpus 1 --steps 20 --warmup 5 --rocprof-keep $O/rp_cold > $O/r05_bench_line_cold_20_5.json 2> $O/cold.err
python bench.py --rocprof-keep $O/rp > $O/r05_bench_line.json 2> $O/warm.err
LUMINOTH_AMD_PLAN=0 python bench.py --no-cpu-baseline --no-other-configs --no-roofline > $O/r05_bench_line_eager_launches.json 2>/dev/null
LMH_OPT_CONV_PP=0 python bench.py --no-cpu-baseline --no-other-configs --no-rocprof > $O/r05_bench_line_tiled_1x1.json 2>/dev/null
# 2. timelines
python bench.py --steps 40 --warmup 10 --phases 30 --no-cpu-baseline --no-other-configs --no-roofline > $O/r05_bench_phases.json 2>/dev/null
python bench.py --workload frcnn_r50_coco --dtype f16 --steps 40 --warmup 10 --phases 30 --no-cpu-baseline --no-roofline > $O/r05_bench_frcnn_r50_coco_f16_phases.json 2>/dev/null
# 3. the other workloads (roofline from HIP events only: one rocprofv3 child per line would double the call)
B="python bench.py --no-cpu-baseline --no-rocprof"
$B --workload frcnn_vgg16 > $O/r05_bench_frcnn_vgg16_f32.json 2>/dev/null
$B --workload ssd300_b32 --steps 20 --warmup 5 > $O/r05_bench_ssd300_b32_f32.json 2>/dev/null
$B --workload frcnn_r101 --steps 30 --warmup 8 > $O/r05_bench_frcnn_r101_f32.json 2>/dev/null
$B --workload frcnn_r101 --dtype f16 --steps 30 --warmup 8 > $O/r05_bench_frcnn_r101_f16.json 2>/dev/null
$B --workload frcnn_r50_coco > $O/r05_bench_frcnn_r50_coco_f32.json 2>/dev/null
python bench.py --no-cpu-baseline --workload frcnn_r50_coco --dtype f16 --rocprof-keep $O/rp_f16 > $O/r05_bench_frcnn_r50_coco_f16.json 2>/dev/null
$B --workload frcnn_r50_coco --dtype bf16 > $O/r05_bench_frcnn_r50_coco_bf16.json 2>/dev/null
$B --workload frcnn_r50_coco --dtype f16 --batch 8 --steps 30 --warmup 8 > $O/r05_bench_frcnn_r50_coco_f16_batch8.json 2>/dev/null
$B --workload frcnn_r50 --batch 8 --steps 30 --warmup 8 --no-other-configs > $O/r05_bench_frcnn_r50_f32_batch8.json 2>/dev/null
# 4. kernel traces of the production steps
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r05 -- python $R/bench.py --no-cpu-baseline --no-other-configs --no-rocprof > $O/r05_bench_profiled_line.json 2> $O/prof.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f16 -o r05 -- python $R/bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline --no-rocprof > $O/r05_f16hs_bench_profiled_line.json 2> $O/prof_f16.err
# 5. PMC passes over the serialised roofline steps (the command bench.py's rocprofv3 child runs), both legs; counters in runs of
#    their own with --kernel-trace only
for leg in "f32 --workload frcnn_r50 --dtype f32" "f16 --workload frcnn_r50_coco --dtype f16"; do
  tag=${leg%% *}; args=${leg#* }
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmc_${tag}_mfma -o p -- python $R/bench.py --roofline-child $args > $O/pmc_${tag}_mfma.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${tag}_fetch -o p -- python $R/bench.py --roofline-child $args > $O/pmc_${tag}_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${tag}_write -o p -- python $R/bench.py --roofline-child $args > $O/pmc_${tag}_write.log 2>&1
done
cd $R
# 6. summaries (written into profiles/ of this copy, then copied to the output directory)
D=$(dirname $(find $O/prof -name '*kernel_trace.csv' | head -n 1))
python scripts/make_profile_summary.py $D r05_bench "python bench.py --no-cpu-baseline --no-other-configs --no-rocprof (60 timed production steps, launch-plan replay)" 60 4 > $O/summary.txt 2>&1
D2=$(dirname $(find $O/prof_f16 -name '*kernel_trace.csv' | head -n 1))
python scripts/make_profile_summary.py $D2 r05_f16hs_bench "python bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline --no-rocprof (60 timed production steps, launch-plan replay)" 60 4 > $O/summary_f16.txt 2>&1
python scripts/make_profile_summary.py $O/rp/frcnn_r50_f32 r05_bench_roofline_steps "rocprofv3 child of python bench.py: bench.py --roofline-child --workload frcnn_r50 --dtype f32 (the 3 serialised roofline steps roofline.frac is computed from)" 3 0 > $O/summary_roofline.txt 2>&1
python scripts/make_profile_summary.py $O/rp/frcnn_r50_coco_f16 r05_f16hs_roofline_steps "rocprofv3 child of python bench.py: bench.py --roofline-child --workload frcnn_r50_coco --dtype f16 (other_configs.frcnn_r50_coco_f16.roofline)" 3 0 > $O/summary_roofline_f16.txt 2>&1
python scripts/r5_pmc_reduce.py $O/pmc_f32_mfma $O/pmc_f32_fetch $O/pmc_f32_write $O/r05_f32_pmc_traffic.json > $O/pmc_f32.txt 2>&1
python scripts/r5_pmc_reduce.py $O/pmc_f16_mfma $O/pmc_f16_fetch $O/pmc_f16_write $O/r05_f16_pmc_traffic.json > $O/pmc_f16.txt 2>&1
cp profiles/r05_bench_summary.md profiles/r05_bench_kernel_stats.csv profiles/r05_bench_roofline_steps_summary.md profiles/r05_bench_roofline_steps_kernel_stats.csv profiles/r05_f16hs_bench_summary.md profiles/r05_f16hs_bench_kernel_stats.csv profiles/r05_f16hs_roofline_steps_summary.md profiles/r05_f16hs_roofline_steps_kernel_stats.csv $O/ 2>/dev/null
rm -rf $O/prof $O/prof_f16 $O/pmc_*_mfma $O/pmc_*_fetch $O/pmc_*_write $O/rp $O/rp_cold $O/rp_f16
cat $O/pmc_f32.txt $O/pmc_f16.txt
for f in $O/r05_bench*.json $O/r05_f16hs*.json; do python - $f <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print('%-48s %.3f ms  median %.3f  %.1f img/s  %s frac %s (%s)'%(sys.argv[1].split('/')[-1], d['ms_per_step'], d.get('ms_per_step_median',0), d['value'], r.get('kernel'), ('%.3f'%r['frac']) if r.get('frac') else None, (r.get('frac_source') or '')[:9]))
    o=(d.get('other_configs') or {}).get('frcnn_r50_coco_f16')
    if o: ro=o.get('roofline') or {}; print('   f16 leg %.3f ms %s frac %s'%(o['ms_per_step'], ro.get('kernel'), ro.get('frac')))
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
