# usage: bash scripts/gpu_prof.sh <tag>   -> gpurun_out/prof_<tag>/ (kernel trace + stats csv)
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_$TAG/*; head -40 gpurun_out/prof_$TAG/*kernel_stats.csv
