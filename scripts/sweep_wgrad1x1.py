"""Sweep of the 1x1 weight-gradient kernels on the trainable ResNet-50 @1024^2 B=2 shapes: register-staged
k_conv_bwd_weight (variant -1, automatic plan) vs the direct-to-LDS k_wgrad_1x1 over tile shape x ring depth x
resident-block target.  Times are kernel + split-K reduce (HIP events, 30 iterations).
python scripts/sweep_wgrad1x1.py [filter]"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luminoth_amd import kernels as K
from scripts.bench_conv import LAYERS, B, timeit   # noqa

lib = K._lib.load()
flt = sys.argv[1] if len(sys.argv) > 1 else ''
dev = torch.device('cuda:0')
tot_old = tot_new = 0.0
for name, H, C, Kc, R, stride, pad in LAYERS:
    if R != 1 or C % 32 or Kc % 32 or (flt and flt not in name) or name.startswith('b1'):
        continue
    x = torch.randn(B, H, H, C, device=dev)
    gy = torch.randn(B, H, H, Kc, device=dev)
    d = K.conv_desc(x.shape, (1, 1, C, Kc), 1, 1, 'SAME', None)
    dw = torch.empty(1, 1, C, Kc, device=dev)
    fl = 2.0 * B * H * H * Kc * C
    P = B * H * H
    lib.lmh_conv2d_force_config(0, 0, 0)
    lib.lmh_conv2d_force_wgrad_variant(-1)
    t_old = timeit(lambda: K.conv2d_bwd_weight(d, x, gy, out=dw), 30)
    res = []
    for bm, bn in ((64, 64), (128, 64), (64, 128), (128, 128)):
        if bm > C or bn > Kc:
            continue
        tiles = (C // bm) * (Kc // bn)
        for slots in (256, 512, 1024):
            sp = max(1, min(slots // tiles, P // 32 // 4))
            for nbuf in (2, 3, 4):
                lib.lmh_conv2d_force_config(bm, bn, sp)
                lib.lmh_conv2d_force_wgrad_variant(nbuf)
                try:
                    t = timeit(lambda: K.conv2d_bwd_weight(d, x, gy, out=dw), 30)
                except Exception as e:
                    continue
                res.append((t, bm, bn, sp, nbuf))
    lib.lmh_conv2d_force_config(0, 0, 0)
    lib.lmh_conv2d_force_wgrad_variant(0)
    t_auto = timeit(lambda: K.conv2d_bwd_weight(d, x, gy, out=dw), 30)
    res.sort()
    tot_old += t_old
    tot_new += t_auto
    print('%-18s %5.2fGF  old %.0fus/%.0fTF  new-auto %.0fus/%.0fTF | best: ' % (name, fl / 1e9, t_old * 1e3, fl / t_old / 1e9, t_auto * 1e3, fl / t_auto / 1e9) +
          '  '.join('%dx%d/s%d/n%d:%.0fus/%.0fTF' % (bm, bn, sp, nb, t * 1e3, fl / t / 1e9) for t, bm, bn, sp, nb in res[:6]), flush=True)
print('sum: old %.1f us, new-auto %.1f us' % (tot_old * 1e3, tot_new * 1e3))
