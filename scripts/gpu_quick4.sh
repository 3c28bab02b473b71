python -m pytest tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -3 > gpurun_out/tq.log; tail -1 gpurun_out/tq.log
python scripts/cpu_overhead.py 2>&1 | grep "host enqueue"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-250
LUMINOTH_AMD_FUSED_STEP=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-250
