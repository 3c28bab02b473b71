# Round 2, GPU call M: warp-specialised bf16x3 pipeline
R=$GRAFT_REPO_ROOT
cd $R
LMH_X3_PF=3 timeout 300 python -m pytest tests/test_gpu_x3.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed" | head -12
echo "== bf16x3 PF=3 (warp specialised)"; BENCH_COMPUTE=bf16x3 LMH_X3_PF=3 timeout 120 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids
