# kernel-trace stats of the SSD train step and of the predict path
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ssd -o ssd -- python $R/scripts/bench_ssd.py 32 > $R/gpurun_out/prof_ssd.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pred -o pred -- python $R/scripts/bench_predict.py 20 > $R/gpurun_out/prof_pred.log 2>&1
tail -2 $R/gpurun_out/prof_ssd.log | head -1; grep workload $R/gpurun_out/prof_pred.log
