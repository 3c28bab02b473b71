# Kernel trace of a few production steps (fp32 headline and the configs[4] f16 step) and the per-queue dump of one step each.
R=$GRAFT_REPO_ROOT
TAG=${1:-r04t}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/p32 -o t -- python $R/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-other-configs --no-roofline > $R/gpurun_out/$TAG/line32.json 2> $R/gpurun_out/$TAG/err32
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/p16 -o t -- python $R/bench.py --workload frcnn_r50_coco --dtype f16 --steps 30 --warmup 6 --no-cpu-baseline --no-roofline > $R/gpurun_out/$TAG/line16.json 2> $R/gpurun_out/$TAG/err16
cd $R
python scripts/timeline.py $(find gpurun_out/$TAG/p32 -name '*kernel_trace.csv' | head -n 1) --skip 2 --dump > gpurun_out/$TAG/timeline32.txt 2>&1
python scripts/timeline.py $(find gpurun_out/$TAG/p16 -name '*kernel_trace.csv' | head -n 1) --skip 2 --dump > gpurun_out/$TAG/timeline16.txt 2>&1
python scripts/chain_stats.py $(find gpurun_out/$TAG/p32 -name '*kernel_trace.csv' | head -n 1) > gpurun_out/$TAG/chain32.txt 2>&1
python scripts/chain_stats.py $(find gpurun_out/$TAG/p16 -name '*kernel_trace.csv' | head -n 1) > gpurun_out/$TAG/chain16.txt 2>&1
rm -rf gpurun_out/$TAG/p32 gpurun_out/$TAG/p16
