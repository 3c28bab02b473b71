"""Direct vs Winograd F(2x2,3x3) and F(4x4,3x3) on the wide 3x3 layers (HIP-event timed, isolated)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luminoth_amd import kernels as K
from scripts.bench_conv import timeit

dev = torch.device('cuda:0')
for name, N, H, C, Kc in [('rpn 3x3 1024->512 @64', 2, 64, 1024, 512), ('b3 3x3 256->256 @64', 2, 64, 256, 256),
                          ('b2 3x3 128->128 @128', 2, 128, 128, 128), ('vgg conv4 512->512 @38 B=32', 32, 38, 512, 512),
                          ('vgg conv3 256->256 @76 B=32', 32, 76, 256, 256), ('vgg conv5 512->512 @64 B=2', 2, 64, 512, 512)]:
    x = torch.randn(N, H, H, C, device=dev)
    w = torch.randn(3, 3, C, Kc, device=dev) * 0.02
    sc, sh = torch.ones(Kc, device=dev), torch.zeros(Kc, device=dev)
    d = K.conv_desc(x.shape, w.shape, 1, 1, 'SAME', 'relu')
    g = torch.randn(N, H, H, Kc, device=dev)
    y, dx = torch.empty_like(g), torch.empty_like(x)
    gf = 2.0 * N * H * H * 9 * C * Kc / 1e9
    K.WINOGRAD = False
    dw = torch.empty_like(w)
    t = [timeit(lambda: K.conv2d_fwd(d, x, w, sc, sh, out=y)) * 1e3, timeit(lambda: K.conv2d_bwd_data(d, g, w, sc, out=dx)) * 1e3,
         timeit(lambda: K.conv2d_bwd_weight(d, x, g, out=dw)) * 1e3]
    K.WINOGRAD = True
    line = '%-30s %6.1f GF | direct fwd %6.1f bwd_data %6.1f bwd_weight %6.1f us' % (name, gf, t[0], t[1], t[2])
    for m in (2, 4):
        K.set_option('wino_m', m)
        tw = [timeit(lambda: K.conv2d_fwd_winograd(d, x, w, sc, sh, out=y)) * 1e3,
              timeit(lambda: K.conv2d_bwd_data_winograd(d, g, w, sc, out=dx)) * 1e3,
              timeit(lambda: K.conv2d_bwd_weight_winograd(d, x, g, out=dw)) * 1e3]
        line += ' | F%dx%d %6.1f %6.1f %6.1f us (%5.1f TF-eq fwd)' % (m, m, tw[0], tw[1], tw[2], gf / tw[0] * 1e3)
    K.set_option('wino_m', 4)
    print(line)
