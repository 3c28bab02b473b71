python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q 2>&1 | tail -15 > gpurun_out/t9.log
python scripts/bench_conv.py > gpurun_out/conv1.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench2.json 2> gpurun_out/bench2.err
tail -4 gpurun_out/t9.log; cat gpurun_out/conv1.log; cat gpurun_out/bench2.json
