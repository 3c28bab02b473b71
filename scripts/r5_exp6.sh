R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5e; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "pp_equals" > $O/t.log 2>&1; echo "rc $?" >> $O/t.log; tail -n 15 $O/t.log
timeout 300 python scripts/r5_epilogue_decomp.py > $O/decomp.log 2>&1; tail -n 8 $O/decomp.log
