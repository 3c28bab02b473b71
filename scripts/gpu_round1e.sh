python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/t12.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench5.json 2> gpurun_out/bench5.err
tail -4 gpurun_out/t12.log; cut -c1-600 gpurun_out/bench5.json; tail -3 gpurun_out/bench5.err
bash scripts/gpu_prof.sh c | head -36 | cut -c1-150
