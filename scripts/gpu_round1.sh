python -m pytest tests/test_gpu_model.py -m gpu -q 2>&1 | tail -30 > gpurun_out/t8m.log
python bench.py --steps 5 --warmup 2 --cpu-images 1 > gpurun_out/bench1.json 2> gpurun_out/bench1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1
cd $GRAFT_REPO_ROOT; tail -5 gpurun_out/t8m.log; cat gpurun_out/bench1.json; tail -5 gpurun_out/bench1.err; ls gpurun_out/prof1 | head
