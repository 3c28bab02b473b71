# Round 2, GPU call Q: warp-specialised pipelines for the half-precision kernels
R=$GRAFT_REPO_ROOT
cd $R
for pf in 3 4; do LMH_HALF_PF=$pf LMH_X3_PF=$pf timeout 300 python -m pytest tests/test_gpu_half.py tests/test_gpu_x3.py -m gpu -q -k "exact or accuracy or rounding" 2>&1 | tail -2; done
for pf in 1 3 4; do echo "== f16 PF=$pf"; BENCH_COMPUTE=f16 LMH_HALF_PF=$pf timeout 120 python scripts/bench_conv.py 2>&1 | grep -E "b2 1x1 256->512|b3 1x1 512->1024|b3 1x1 1024->256|b3 3x3|rpn 3x3|b2 3x3 128|sum ms"; done
echo "== bf16x3 PF=4"; BENCH_COMPUTE=bf16x3 LMH_X3_PF=4 timeout 120 python scripts/bench_conv.py 2>&1 | grep -E "b2 1x1 256->512|b3 1x1 512->1024|b3 1x1 1024->256|b3 3x3|rpn 3x3|b2 3x3 128|sum ms"
