#!/bin/bash
# round 6: smoke + GPU tests + the driver's command on one box
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r6_check_tests.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_check_line.json 2> gpurun_out/r6_check_line.err
python scripts/r6_line.py check < gpurun_out/r6_check_line.json
tail -c 400 gpurun_out/r6_check_line.err
