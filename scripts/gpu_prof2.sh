# usage: bash scripts/gpu_prof2.sh <tag>   -> kernel + memory-copy trace of 5 bench steps
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/
