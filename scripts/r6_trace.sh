#!/bin/bash
# Kernel traces (production = 3 streams + launch-plan replay, and the serialised roofline steps) and the phase timeline of
# one leg:  bash scripts/r6_trace.sh <tag> "<bench args>"     e.g.  r6_trace.sh r06_x3 "--dtype bf16x3"
#                                                                   r6_trace.sh r06_f16hs "--workload frcnn_r50_coco --dtype f16"
# -> profiles/<tag>_{bench,serial}_summary.md + kernel_stats.csv in this copy, mirrored to gpurun_out/r6_trace_<tag>/
R=$GRAFT_REPO_ROOT; TAG=$1; ARGS="$2"
O=$R/gpurun_out/r6_trace_$TAG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python $R/bench.py $ARGS --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs --no-native --no-rocprof > $O/profiled_line.json 2> $O/prof.err
cd $R
D=$(dirname $(find $O/prof -name '*kernel_trace.csv' | head -n 1))
python scripts/make_profile_summary.py $D ${TAG}_bench "python bench.py $ARGS --steps 30 --warmup 10 --no-native --no-rocprof (production steps: three streams, launch-plan replay)" 30 4 > $O/summary.txt 2>&1
python scripts/make_profile_summary.py $D ${TAG}_serial "python bench.py $ARGS (the 3 serialised roofline steps at the end of the same run)" 3 0 > $O/summary_serial.txt 2>&1
cp profiles/${TAG}_bench_summary.md profiles/${TAG}_bench_kernel_stats.csv profiles/${TAG}_serial_summary.md profiles/${TAG}_serial_kernel_stats.csv $O/ 2>/dev/null
rm -rf $O/prof
python bench.py $ARGS --steps 40 --warmup 10 --phases 30 --no-cpu-baseline --no-other-configs --no-roofline > $O/${TAG}_phases.json 2>/dev/null
python - $O/${TAG}_phases.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%.3f ms/step' % d['ms_per_step'])
for k,v in sorted((d.get('phases_ms') or d.get('phases') or {}).items(), key=lambda kv: kv[1]): print('  %-28s %.3f' % (k, v))
P
head -c 1500 $O/summary.txt | head -12
