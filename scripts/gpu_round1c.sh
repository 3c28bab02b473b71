python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/t10.log
python scripts/bench_conv.py > gpurun_out/conv2.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench3.json 2> gpurun_out/bench3.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof2.log 2>&1
cd $GRAFT_REPO_ROOT; tail -4 gpurun_out/t10.log; cat gpurun_out/conv2.log; cat gpurun_out/bench3.json; tail -3 gpurun_out/bench3.err; ls gpurun_out/prof2/*
