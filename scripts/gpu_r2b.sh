# Round 2, GPU call B: new 1x1 weight-gradient kernel — correctness + sweep
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad_1x1 or conv_fwd_bwd" 2>&1 | tail -8
python scripts/sweep_wgrad1x1.py > gpurun_out/r2b_sweep.log 2>&1; cat gpurun_out/r2b_sweep.log | grep -v amdgpu.ids
python bench.py --no-cpu-baseline > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err; head -c 600 gpurun_out/r2b_bench.json; echo
