"""Tile / split sweep of the conv kernels on the trainable ResNet-50 @1024^2 B=2 shapes.
python scripts/sweep_conv.py [filter]   (diagnostics; uses lmh_conv2d_force_config)"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luminoth_amd import kernels as K
from scripts.bench_conv import LAYERS, B, timeit   # noqa

lib = K._lib.load()
flt = sys.argv[1] if len(sys.argv) > 1 else ''
dev = torch.device('cuda:0')
TILES = [(128, 128), (128, 64), (64, 64)]
WTILES = [(128, 128), (128, 64), (64, 128), (64, 64)]
for name, H, C, Kc, R, stride, pad in LAYERS:
    if (flt and flt not in name) or C % 32 or Kc % 32:
        continue
    x = torch.randn(B, H, H, C, device=dev)
    w = torch.randn(R, R, C, Kc, device=dev) * 0.05
    d = K.conv_desc(x.shape, w.shape, stride, 1, pad, 'relu')
    scale = torch.ones(Kc, device=dev); shift = torch.zeros(Kc, device=dev)
    y = K.conv2d_fwd(d, x, w, scale, shift)
    gy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w)
    fl = 2.0 * B * d.OH * d.OW * Kc * R * R * C
    out = ['%-20s %6.2fGF' % (name, fl / 1e9)]
    for op, fn in (('fwd', lambda: K.conv2d_fwd(d, x, w, scale, shift, out=y)),
                   ('bwd_d', lambda: K.conv2d_bwd_data(d, gy, w, scale, out=dx))):
        res = []
        for bm, bn in TILES:
            lib.lmh_conv2d_force_config(bm, bn, 0)
            t = timeit(fn, 20)
            res.append('%dx%d:%.0fus/%.0fTF' % (bm, bn, t * 1e3, fl / t / 1e9))
        lib.lmh_conv2d_force_config(0, 0, 0)
        out.append(op + ' ' + ' '.join(res))
    print(' | '.join(out), flush=True)
    res = []
    for bm, bn in WTILES:
        if bm > C or bn > Kc:
            continue
        best = None
        for sp in (1, 2, 4, 8, 16, 32, 64, 128):
            lib.lmh_conv2d_force_config(bm, bn, sp)
            try:
                t = timeit(lambda: K.conv2d_bwd_weight(d, x, gy, out=dw), 20)
            except Exception as e:      # workspace too small etc.
                continue
            res.append((t, bm, bn, sp))
    lib.lmh_conv2d_force_config(0, 0, 0)
    t0 = timeit(lambda: K.conv2d_bwd_weight(d, x, gy, out=dw), 20)
    res.sort()
    print('    bwd_w auto %.0fus/%.0fTF | best: ' % (t0 * 1e3, fl / t0 / 1e9) +
          '  '.join('%dx%d/s%d:%.0fus/%.0fTF' % (bm, bn, sp, t * 1e3, fl / t / 1e9) for t, bm, bn, sp in res[:5]), flush=True)
