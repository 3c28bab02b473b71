#!/bin/bash
# kernel traces of the bf16x3 step: production (3 streams) and serialised
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_x3; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o x3 -- python $R/bench.py --dtype bf16x3 --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs --no-rocprof > $O/line.json 2> $O/prof.err
cd $R
D=$(dirname $(find $O/prof -name '*kernel_trace.csv' | head -n 1))
python scripts/make_profile_summary.py $D r06_x3_s1_bench "python bench.py --dtype bf16x3 --steps 30 --warmup 10 (production steps, launch-plan replay; round 6 step 1: conv_x3.h kernels, phase-by-phase schedule)" 30 4 > $O/summary.txt 2>&1
python scripts/make_profile_summary.py $D r06_x3_s1_serial "python bench.py --dtype bf16x3 (the 3 serialised roofline steps at the end of the run; round 6 step 1: conv_x3.h kernels, phase-by-phase schedule)" 3 0 > $O/summary_serial.txt 2>&1
cp profiles/r06_x3_s1_*.md $O/
rm -rf $O/prof
