python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/tq.log; tail -5 gpurun_out/tq.log
LUMINOTH_AMD_FUSED_STEP=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-250
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -3 | cut -c1-250
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-250
