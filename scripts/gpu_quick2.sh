python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/tq.log; tail -2 gpurun_out/tq.log
LUMINOTH_AMD_SIDE_STREAM=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-250
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-250
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-250
