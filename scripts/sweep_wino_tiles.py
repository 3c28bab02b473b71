"""Tile sweep of the stacked Winograd F(4x4,3x3) GEMMs (forward / backward-data: k_conv_fwd<..., GB>; weight gradient:
k_conv_bwd_weight<..., GB>) on the step's layer shapes, with the tile forced through lmh_conv2d_force_config."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luminoth_amd import kernels as K, _lib
from scripts.bench_conv import timeit

lib = _lib.load()
dev = torch.device('cuda:0')
for name, N, H, C, Kc in [('rpn 3x3 1024->512 @64', 2, 64, 1024, 512), ('b3 3x3 256->256 @64', 2, 64, 256, 256),
                          ('b2 3x3 128->128 @128', 2, 128, 128, 128), ('b3 @50x84 (coco)', 2, 64, 256, 256)]:
    W_ = 84 if 'coco' in name else H
    Hh = 50 if 'coco' in name else H
    x = torch.randn(N, Hh, W_, C, device=dev)
    w = torch.randn(3, 3, C, Kc, device=dev) * 0.02
    sc, sh = torch.ones(Kc, device=dev), torch.zeros(Kc, device=dev)
    d = K.conv_desc(x.shape, w.shape, 1, 1, 'SAME', 'relu')
    g = torch.randn(N, Hh, W_, Kc, device=dev)
    y, dx, dw = torch.empty_like(g), torch.empty_like(x), torch.empty_like(w)
    line = '%-26s' % name
    for bm, bn in ((0, 0), (128, 128), (128, 64), (64, 64)):
        lib.lmh_conv2d_force_config(bm, bn, 0)
        t = [timeit(lambda: K.conv2d_fwd_winograd(d, x, w, sc, sh, out=y)) * 1e3,
             timeit(lambda: K.conv2d_bwd_data_winograd(d, g, w, sc, out=dx)) * 1e3,
             timeit(lambda: K.conv2d_bwd_weight_winograd(d, x, g, out=dw)) * 1e3]
        line += ' | %s %5.1f %5.1f %5.1f' % ('auto   ' if bm == 0 else '%3dx%-3d' % (bm, bn), t[0], t[1], t[2])
    lib.lmh_conv2d_force_config(0, 0, 0)
    print(line)
