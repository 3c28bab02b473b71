# Round 2, GPU call N: counters of the bf16x3 forward kernel on the RPN convolution
export BENCH_COMPUTE=bf16x3 LMH_X3_PF=3
bash scripts/gpu_pmc.sh "rpn 3x3" fwd 2>&1 | grep -v amdgpu.ids | tail -40
tail -3 gpurun_out/pmcx_3.log
