R=$GRAFT_REPO_ROOT
TAG=${1:-r04f}
mkdir -p $R/gpurun_out/$TAG
cd $R
python bench.py --workload frcnn_r50_coco --dtype f16 --steps 40 --warmup 10 --phases 30 --no-cpu-baseline --no-roofline > gpurun_out/$TAG/phases.json 2> gpurun_out/$TAG/phases.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -o r04 -- python $R/bench.py --workload frcnn_r50_coco --dtype f16 --steps 20 --warmup 8 --no-cpu-baseline --no-roofline > $R/gpurun_out/$TAG/prof_line.json 2> $R/gpurun_out/$TAG/prof.err
cd $R
D=$(dirname $(find gpurun_out/$TAG/prof -name '*kernel_trace.csv' | head -n 1))
python scripts/make_profile_summary.py $D ${TAG}_f16hs_bench "python bench.py --workload frcnn_r50_coco --dtype f16 --steps 20 --warmup 8 (launch plan replay)" 16 0 > gpurun_out/$TAG/summary.txt 2>&1
python scripts/timeline.py $D/*kernel_trace.csv --dump > gpurun_out/$TAG/timeline_dump.txt 2>&1
cp profiles/${TAG}_f16hs_bench* gpurun_out/$TAG/ 2>/dev/null
rm -rf gpurun_out/$TAG/prof
