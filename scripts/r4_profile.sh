# Round 4 evidence: plan tests, phases timeline (un-profiled), rocprofv3 kernel trace of the default bench line (timed production
# steps + the serialised roofline steps + the configs[4] leg), and a dump of the last step's per-queue timeline.
R=$GRAFT_REPO_ROOT
TAG=${1:-r04a}
mkdir -p $R/gpurun_out/$TAG
cd $R
python -m pytest tests/test_gpu_plan.py -x -q -m gpu 2>&1 | tail -n 15 > gpurun_out/$TAG/plan_tests.log
python bench.py --steps 40 --warmup 10 --phases 30 --no-cpu-baseline --no-other-configs --no-roofline > gpurun_out/$TAG/phases.json 2> gpurun_out/$TAG/phases.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -o r04 -- python $R/bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-other-configs > $R/gpurun_out/$TAG/prof_line.json 2> $R/gpurun_out/$TAG/prof.err
cd $R
D=$(dirname $(find gpurun_out/$TAG/prof -name '*kernel_trace.csv' | head -n 1))
python scripts/make_profile_summary.py $D ${TAG}_bench "python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-other-configs (timed production steps, launch plan replay)" 16 4 > gpurun_out/$TAG/summary.txt 2>&1
python scripts/make_profile_summary.py $D ${TAG}_bench_roofline_steps "python bench.py (the 3 serialised roofline steps at its end)" 3 0 > gpurun_out/$TAG/summary_roofline.txt 2>&1
python scripts/timeline.py $D/*kernel_trace.csv --skip 4 --dump > gpurun_out/$TAG/timeline_dump.txt 2>&1
cp profiles/${TAG}_bench* gpurun_out/$TAG/ 2>/dev/null
rm -rf gpurun_out/$TAG/prof
