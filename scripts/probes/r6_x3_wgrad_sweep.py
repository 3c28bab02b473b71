"""Tile / split sweep of the bf16x3 weight gradient (k_x3_bwd_weight + its split-K reduction) on the trainable ResNet-50 layers at
2 x 1024^2.  python scripts/probes/r6_x3_wgrad_sweep.py [filter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from luminoth_amd import kernels as K
from scripts.bench_conv import LAYERS, B, timeit   # noqa
lib = K._lib.load()
flt = sys.argv[1] if len(sys.argv) > 1 else ''
dev = torch.device('cuda:0')
K.WINOGRAD = False
for name, H, C, Kc, R, stride, pad in LAYERS:
    if (flt and flt not in name) or C % 128 or Kc % 128 or R != 1:
        continue
    x = torch.randn(B, H, H, C, device=dev)
    d = K.conv_desc(x.shape, (R, R, C, Kc), stride, 1, pad, 'relu', 'bf16x3')
    gy = torch.randn(B, d.OH, d.OW, Kc, device=dev)
    dw = torch.empty(R, R, C, Kc, device=dev)
    fl = 2.0 * B * d.OH * d.OW * Kc * R * R * C
    lib.lmh_conv2d_force_config(0, 0, 0)
    t0 = timeit(lambda: K.conv2d_bwd_weight(d, x, gy, out=dw), 20)
    row = '%-20s %5.2f GF | auto %5.1f us %4.0f TF |' % (name, fl / 1e9, t0 * 1e3, fl / t0 / 1e9)
    for bm, bn in ((128, 128), (128, 64), (64, 128), (64, 64)):
        best = None
        for sp in (2, 4, 8, 16, 32, 64):
            lib.lmh_conv2d_force_config(bm, bn, sp)
            try:
                t = timeit(lambda: K.conv2d_bwd_weight(d, x, gy, out=dw), 12)
            except Exception:
                continue
            if best is None or t < best[0]:
                best = (t, sp)
        if best:
            row += ' %dx%d/s%d %5.1f |' % (bm, bn, best[1], best[0] * 1e3)
    lib.lmh_conv2d_force_config(0, 0, 0)
    print(row, flush=True)
