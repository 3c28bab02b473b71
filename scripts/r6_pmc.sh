#!/bin/bash
# PMC passes (MFMA busy, FETCH_SIZE, WRITE_SIZE; counters in runs of their own with --kernel-trace only) over the serialised
# roofline steps of bench.py for one leg:  bash scripts/r6_pmc.sh <tag> <workload> <dtype>
# -> gpurun_out/r6_pmc_<tag>/<round>_<tag>_pmc_traffic.json (+ .txt): copy into profiles/.
R=$GRAFT_REPO_ROOT; TAG=$1; WL=$2; DT=$3
O=$R/gpurun_out/r6_pmc_$TAG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/mfma -o p -- python $R/bench.py --roofline-child --workload $WL --dtype $DT > $O/mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python $R/bench.py --roofline-child --workload $WL --dtype $DT > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python $R/bench.py --roofline-child --workload $WL --dtype $DT > $O/write.log 2>&1
cd $R
python scripts/pmc_reduce.py $O/mfma $O/fetch $O/write $O/r06_${TAG}_pmc_traffic.json 3 $WL $DT > $O/r06_${TAG}_pmc_top_kernels.txt 2>&1
cat $O/r06_${TAG}_pmc_top_kernels.txt
rm -rf $O/mfma $O/fetch $O/write
