python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv" 2>&1 | tail -5 > gpurun_out/t13.log
tail -3 gpurun_out/t13.log
python scripts/sweep_conv.py > gpurun_out/sweep2.log 2>&1; cat gpurun_out/sweep2.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench6.json 2> gpurun_out/bench6.err; cut -c1-330 gpurun_out/bench6.json; tail -2 gpurun_out/bench6.err
