# HBM traffic per kernel from PMC counters (separate passes, kernel-trace only), as MI355X_MICROARCH.md §HBM says.
# usage: bash scripts/gpu_pmc_bench.sh <tag>  -> gpurun_out/pmc_<tag>_{fetch,write}/
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${TAG}_$c -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_${TAG}_$c.log 2>&1
done
cd $R; python scripts/pmc_traffic.py gpurun_out/pmc_${TAG}_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc_${TAG}_WRITE_SIZE/p_counter_collection.csv gpurun_out/pmc_${TAG}_traffic.json
