"""Timing decomposition of k_conv_hs (f16 half-storage forward; probe build: LMH_PROBES=1): per layer at the BASELINE configs[4]
geometry, the launch time with parts of the kernel switched off through bits 8.. of the probe word (csrc/conv_hs.h).
    LMH_PROBES=1 python scripts/r6_hs_decomp.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luminoth_amd import kernels as K
from luminoth_amd import _lib
from luminoth_amd import plan as P

dev = torch.device('cuda:0')
N = 20
# (name, H, W, C, K, R, residual)
LAYERS = [('b3 1x1 1024->256', 50, 84, 1024, 256, 1, False), ('b3 3x3 256->256', 50, 84, 256, 256, 3, False),
          ('b3 1x1 256->1024 (+res)', 50, 84, 256, 1024, 1, True), ('b2 1x1 512->128', 100, 167, 512, 128, 1, False),
          ('b2 3x3 128->128', 100, 167, 128, 128, 3, False), ('b2 1x1 128->512 (+res)', 100, 167, 128, 512, 1, True),
          ('rpn 3x3 1024->512', 50, 84, 1024, 512, 3, False)]
PARTS = [('full', 0), ('-residual read', 1), ('-stores', 2), ('-epilogue mem', 3), ('-A loads', 8), ('-B loads', 32), ('-A -B loads', 40),
         ('-MFMA', 16), ('-loads -MFMA', 56), ('-main loop', 4), ('nothing', 7)]


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    with P.StepPlan() as pl:
        for _ in range(N):
            fn()
    torch.cuda.synchronize()
    pl.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); pl.run(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / N)
    pl.destroy()
    return best * 1e3


lib = _lib.load()
for name, H, W, C, Kc, R, res in LAYERS:
    x = torch.randn(2, H, W, C, device=dev).half()
    w = torch.randn(R, R, C, Kc, device=dev) * 0.05
    sc, sh = torch.ones(Kc, device=dev), torch.zeros(Kc, device=dev)
    r = torch.randn(2, H, W, Kc, device=dev).half() if res else None
    d = K.conv_desc(x.shape, w.shape, 1, 1, 'SAME', 'relu', 'f16')
    wf = torch.empty((Kc, R, R, C), dtype=torch.float16, device=dev)
    wb = torch.empty((R, R, C, Kc), dtype=torch.float16, device=dev)
    K.half_weights_batch([(w, sc, wf, wb)], 'f16')
    bits = K.new_act_bits(2 * H * W, Kc, dev)
    y = torch.empty(2, H, W, Kc, dtype=torch.float16, device=dev)
    row = []
    for pname, b in PARTS:
        lib.lmh_conv_set_stagger(b << 8)
        row.append((pname, timeit(lambda: K.conv2d_fwd_hs(d, x, wf, sc, sh, r, act_bits=bits, out=y))))
    lib.lmh_conv_set_stagger(0)
    by = 2.0 * (2 * H * W * (C + Kc * (2 if res else 1)) + R * R * C * Kc)
    print('%-24s %5.1f MB %5.2f GF | ' % (name, by / 1e6, 2.0 * 2 * H * W * R * R * C * Kc / 1e9) + '  '.join('%s %.1f' % (p, t) for p, t in row))
