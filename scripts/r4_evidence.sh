# Round 4 evidence (one gpurun call, one box): cold and warm bench lines, phases, the other workloads, rocprofv3 kernel traces
# (fp32 line incl. its serialised roofline steps; f16 half-storage step) and the FETCH_SIZE / WRITE_SIZE passes.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_evidence; mkdir -p $O
cd $R
# 1. the driver's command, as the FIRST process on the box, then the warm default run
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_line_cold_20_5.json 2> $O/cold.err
python bench.py > $O/r04_bench_line.json 2> $O/warm.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_bench_line_20_5_again.json 2>/dev/null
LUMINOTH_AMD_PLAN=0 python bench.py --no-cpu-baseline --no-other-configs --no-roofline > $O/r04_bench_line_eager_launches.json 2>/dev/null
# 2. timelines
python bench.py --steps 40 --warmup 10 --phases 30 --no-cpu-baseline --no-other-configs --no-roofline > $O/r04_bench_phases.json 2>/dev/null
python bench.py --workload frcnn_r50_coco --dtype f16 --steps 40 --warmup 10 --phases 30 --no-cpu-baseline --no-roofline > $O/r04_bench_frcnn_r50_coco_f16_phases.json 2>/dev/null
# 3. the other workloads
B="python bench.py --no-cpu-baseline"
$B --workload frcnn_vgg16 > $O/r04_bench_frcnn_vgg16_f32.json 2>/dev/null
$B --workload ssd300_b32 --steps 20 --warmup 5 > $O/r04_bench_ssd300_b32_f32.json 2>/dev/null
$B --workload frcnn_r101 --steps 30 --warmup 8 > $O/r04_bench_frcnn_r101_f32.json 2>/dev/null
$B --workload frcnn_r101 --dtype f16 --steps 30 --warmup 8 > $O/r04_bench_frcnn_r101_f16.json 2>/dev/null
$B --workload frcnn_r50_coco > $O/r04_bench_frcnn_r50_coco_f32.json 2>/dev/null
$B --workload frcnn_r50_coco --dtype f16 > $O/r04_bench_frcnn_r50_coco_f16.json 2>/dev/null
$B --workload frcnn_r50_coco --dtype bf16 > $O/r04_bench_frcnn_r50_coco_bf16.json 2>/dev/null
LUMINOTH_AMD_RPN_BWD_SIDE=1 $B --workload frcnn_r50_coco --dtype f16 --no-roofline > $O/r04_bench_frcnn_r50_coco_f16_rpn_bwd_side.json 2>/dev/null
$B --workload frcnn_r50_coco --dtype f16 --batch 8 --steps 30 --warmup 8 > $O/r04_bench_frcnn_r50_coco_f16_batch8.json 2>/dev/null
$B --workload frcnn_r50 --batch 8 --steps 30 --warmup 8 --no-other-configs > $O/r04_bench_frcnn_r50_f32_batch8.json 2>/dev/null
$B --alt --no-other-configs > $O/r04_bench_with_alt.json 2>/dev/null
# 4. kernel traces
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r04 -- python $R/bench.py --no-cpu-baseline --no-other-configs > $O/r04_bench_profiled_line.json 2> $O/prof.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f16 -o r04 -- python $R/bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline > $O/r04_f16hs_bench_profiled_line.json 2> $O/prof_f16.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs > $O/pmc_$c.log 2>&1
done
cd $R
D=$(dirname $(find $O/prof -name '*kernel_trace.csv' | head -n 1))
python scripts/make_profile_summary.py $D r04_bench "python bench.py --no-cpu-baseline --no-other-configs (60 timed production steps, launch-plan replay)" 60 4 > $O/summary.txt 2>&1
python scripts/make_profile_summary.py $D r04_bench_roofline_steps "python bench.py --no-cpu-baseline --no-other-configs (the 3 serialised roofline steps at its end)" 3 0 > $O/summary_roofline.txt 2>&1
D2=$(dirname $(find $O/prof_f16 -name '*kernel_trace.csv' | head -n 1))
python scripts/make_profile_summary.py $D2 r04_f16hs_bench "python bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline (60 timed production steps, launch-plan replay)" 60 4 > $O/summary_f16.txt 2>&1
python scripts/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -n 1) $(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -n 1) $O/r04_pmc_traffic.json > $O/pmc_traffic.txt 2>&1
cp profiles/r04_bench_summary.md profiles/r04_bench_kernel_stats.csv profiles/r04_bench_roofline_steps_summary.md profiles/r04_bench_roofline_steps_kernel_stats.csv profiles/r04_f16hs_bench_summary.md profiles/r04_f16hs_bench_kernel_stats.csv $O/ 2>/dev/null
rm -rf $O/prof $O/prof_f16 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
for f in $O/r04_bench*.json $O/r04_f16hs*.json; do python - $f <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%-48s %.3f ms  median %.3f  %.1f img/s'%(sys.argv[1].split('/')[-1], d['ms_per_step'], d.get('ms_per_step_median',0), d['value']))
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
