python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/t14.log; tail -2 gpurun_out/t14.log
python scripts/bench_conv.py > gpurun_out/conv4.log 2>&1; cat gpurun_out/conv4.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench7.json 2> gpurun_out/bench7.err; cut -c1-330 gpurun_out/bench7.json; tail -2 gpurun_out/bench7.err
bash scripts/gpu_prof.sh d | head -34 | cut -c1-160
