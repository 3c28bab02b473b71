# Round 2, GPU call L: per-layer table native fp32 vs bf16x3 pipelines
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python -m pytest tests/test_gpu_x3.py -m gpu -q 2>&1 | tail -3
echo "== native fp32, direct kernels (Winograd off)"; LUMINOTH_AMD_WINOGRAD=0 timeout 120 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids
echo "== native fp32, Winograd on"; timeout 120 python scripts/bench_conv.py 3x3 2>&1 | grep -v amdgpu.ids
for pf in 0 1 2; do
  echo "== bf16x3 PF=$pf"; BENCH_COMPUTE=bf16x3 LMH_X3_PF=$pf timeout 120 python scripts/bench_conv.py 2>&1 | grep -v amdgpu.ids
done
