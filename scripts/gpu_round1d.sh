python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k conv 2>&1 | tail -15 > gpurun_out/t11.log
python scripts/bench_conv.py > gpurun_out/conv3.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench4.json 2> gpurun_out/bench4.err
tail -4 gpurun_out/t11.log; cat gpurun_out/conv3.log; cat gpurun_out/bench4.json; tail -3 gpurun_out/bench4.err
