#!/bin/bash
# Round 6 evidence: ONE gpurun call = one box.  The driver's command first (cold box), the warm default line, the other workloads,
# kernel traces (production + serialised) and timelines of both legs, the traces bench.py's own rocprofv3 children took for
# `roofline.frac`, and PMC passes (MFMA busy, FETCH_SIZE, WRITE_SIZE) over the serialised roofline steps of the three arithmetics.
#   gpurun --timeout 2400 -- bash scripts/r6_evidence.sh        then copy gpurun_out/r06_evidence/* into profiles/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_evidence; rm -rf $O; mkdir -p $O
cd $R
# 1. the driver's command as the FIRST process on the box, then the default run (both with their rocprofv3 children)
python bench.py --gpus 1 --steps 20 --warmup 5 --rocprof-keep $O/rp_cold > $O/r06_bench_line_cold_20_5.json 2> $O/cold.err
python bench.py --rocprof-keep $O/rp > $O/r06_bench_line.json 2> $O/warm.err
LUMINOTH_AMD_PLAN=0 python bench.py --no-cpu-baseline --no-other-configs --no-native --no-roofline > $O/r06_bench_line_eager_launches.json 2>/dev/null
# 2. the other workloads (roofline from HIP events only)
B="python bench.py --no-cpu-baseline --no-rocprof --no-native"
$B --workload frcnn_r50 --dtype f32 --no-other-configs > $O/r06_bench_frcnn_r50_native_f32.json 2>/dev/null
$B --workload frcnn_vgg16 > $O/r06_bench_frcnn_vgg16.json 2>/dev/null
$B --workload ssd300_b32 --steps 20 --warmup 5 > $O/r06_bench_ssd300_b32_f32.json 2>/dev/null
$B --workload frcnn_r101 --steps 30 --warmup 8 > $O/r06_bench_frcnn_r101.json 2>/dev/null
$B --workload frcnn_r101 --dtype f16 --steps 30 --warmup 8 > $O/r06_bench_frcnn_r101_f16.json 2>/dev/null
$B --workload frcnn_r50_coco > $O/r06_bench_frcnn_r50_coco.json 2>/dev/null
python bench.py --no-cpu-baseline --workload frcnn_r50_coco --dtype f16 --rocprof-keep $O/rp_f16 > $O/r06_bench_frcnn_r50_coco_f16.json 2>/dev/null
$B --workload frcnn_r50_coco --dtype bf16 > $O/r06_bench_frcnn_r50_coco_bf16.json 2>/dev/null
$B --workload frcnn_r50_coco --dtype f16 --batch 8 --steps 30 --warmup 8 > $O/r06_bench_frcnn_r50_coco_f16_batch8.json 2>/dev/null
$B --workload frcnn_r50 --batch 8 --steps 30 --warmup 8 --no-other-configs > $O/r06_bench_frcnn_r50_batch8.json 2>/dev/null
# 3. kernel traces + timelines of both legs (scripts/r6_trace.sh writes profiles/<tag>_{bench,serial}_* in this copy)
bash scripts/r6_trace.sh r06_x3 "--dtype bf16x3" > $O/trace_x3.txt 2>&1
bash scripts/r6_trace.sh r06_f16hs "--workload frcnn_r50_coco --dtype f16" > $O/trace_f16hs.txt 2>&1
cp gpurun_out/r6_trace_r06_x3/r06_x3_* gpurun_out/r6_trace_r06_f16hs/r06_f16hs_* $O/ 2>/dev/null
# 4. the rocprofv3 children of the default run: the traces `roofline.frac` was computed from
python scripts/make_profile_summary.py $O/rp/frcnn_r50_bf16x3 r06_bench_roofline_steps "rocprofv3 child of python bench.py: bench.py --roofline-child --workload frcnn_r50 --dtype bf16x3 (the 3 serialised roofline steps roofline.frac is computed from)" 3 0 > $O/summary_roofline.txt 2>&1
python scripts/make_profile_summary.py $O/rp/frcnn_r50_coco_f16 r06_f16hs_roofline_steps "rocprofv3 child of python bench.py: bench.py --roofline-child --workload frcnn_r50_coco --dtype f16 (other_configs.frcnn_r50_coco_f16.roofline)" 3 0 > $O/summary_roofline_f16.txt 2>&1
cp profiles/r06_bench_roofline_steps_* profiles/r06_f16hs_roofline_steps_* $O/ 2>/dev/null
rm -rf $O/rp $O/rp_cold $O/rp_f16
# 5. PMC passes, counters in runs of their own (scripts/r6_pmc.sh): bf16x3 headline, native fp32, f16 half-storage
bash scripts/r6_pmc.sh x3 frcnn_r50 bf16x3 > $O/pmc_x3.txt 2>&1
bash scripts/r6_pmc.sh f32 frcnn_r50 f32 > $O/pmc_f32.txt 2>&1
bash scripts/r6_pmc.sh f16 frcnn_r50_coco f16 > $O/pmc_f16.txt 2>&1
cp gpurun_out/r6_pmc_x3/r06_* gpurun_out/r6_pmc_f32/r06_* gpurun_out/r6_pmc_f16/r06_* $O/ 2>/dev/null
head -n 12 $O/pmc_x3.txt $O/pmc_f16.txt
for f in $O/r06_bench*.json; do python scripts/r6_line.py $(basename $f .json) < $f; done
