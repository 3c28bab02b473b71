"""Fixed vs marginal cost of the forward implicit-GEMM kernel: 1x1 convolutions on 2 x 64 x 64 pixels with a growing
reduction length C (K-loop stages = C / 32) at fixed output width — the intercept is launch + prologue + epilogue, the
slope the steady-state stage time.  python scripts/bench_fixed_cost.py [Kout]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from luminoth_amd import kernels as K
from scripts.bench_conv import timeit

dev = torch.device('cuda:0')
Kout = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B, H = 2, 64
print('1x1 conv, %d x %d x %d pixels, K = %d (tile from pick_tile); us per launch' % (B, H, H, Kout))
for C in (32, 64, 128, 256, 512, 1024, 2048):
    x = torch.randn(B, H, H, C, device=dev)
    w = torch.randn(1, 1, C, Kout, device=dev) * 0.05
    d = K.conv_desc(x.shape, w.shape, 1, 1, 'SAME', 'relu', os.environ.get('BENCH_COMPUTE') or None)
    scale, shift = torch.ones(Kout, device=dev), torch.zeros(Kout, device=dev)
    y = K.conv2d_fwd(d, x, w, scale, shift)
    gy = torch.randn_like(y)
    dx = torch.empty_like(x)
    t_f = timeit(lambda: K.conv2d_fwd(d, x, w, scale, shift, out=y)) * 1e3
    t_d = timeit(lambda: K.conv2d_bwd_data(d, gy, w, scale, out=dx)) * 1e3
    fl = 2.0 * B * H * H * C * Kout
    print('C %5d (%2d stages): fwd %7.1f us %6.1f TF/s (plan %s) | bwd_data %7.1f us %6.1f TF/s'
          % (C, C // 32, t_f, fl / t_f / 1e6, K._lib.load().lmh_conv2d_kernel_id(d, 0) % 1000000, t_d, fl / t_d / 1e6))
