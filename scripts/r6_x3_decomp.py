"""Timing decomposition of k_x3_fwd (probe build: LMH_PROBES=1): per layer, the launch time with parts of the kernel switched
off through bits 8.. of the x3_stagger option (csrc/conv_x3.h).  Launches replayed from a plan (scripts/bench_conv_plan.py).
    LMH_PROBES=1 python scripts/r6_x3_decomp.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luminoth_amd import kernels as K
from luminoth_amd import plan as P

dev = torch.device('cuda:0')
N = 20
LAYERS = [('b3 1x1 256->1024 (+res)', 64, 256, 1024, True), ('b3 1x1 1024->256', 64, 1024, 256, False),
          ('b3 1x1 512->256', 64, 512, 256, False), ('b2 1x1 128->512 (+res)', 128, 128, 512, True),
          ('b2 1x1 512->128', 128, 512, 128, False)]
PARTS = [('full', 0), ('-B split', 1), ('-B split -B load', 3), ('-A split', 4), ('-A split -A load', 12), ('-all staging', 15),
         ('-MFMA', 16), ('-epilogue mem', 32), ('-staging -MFMA', 31), ('only epilogue (-staging -MFMA)', 31),
         ('nothing (-all)', 63)]


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


K.WINOGRAD = False
for name, H, C, Kc, res in LAYERS:
    x = torch.randn(2, H, H, C, device=dev)
    w = torch.randn(1, 1, C, Kc, device=dev) * 0.05
    sc, sh = torch.ones(Kc, device=dev), torch.zeros(Kc, device=dev)
    r = torch.randn(2, H, H, Kc, device=dev) if res else None
    d = K.conv_desc(x.shape, w.shape, 1, 1, 'SAME', 'relu', 'bf16x3')
    bits = K.new_act_bits(2 * H * H, Kc, dev)
    y = torch.empty(2, H, H, Kc, device=dev)
    row = []
    for pname, bitsv in PARTS:
        K.set_option('x3_stagger', bitsv << 8)
        row.append((pname, timeit(lambda: K.conv2d_fwd(d, x, w, sc, sh, r, out=y, act_bits=bits))))
    K.set_option('x3_stagger', 0)
    print('%-26s' % name + '  '.join('%s %.1f' % (p, t) for p, t in row))


# ---- weight gradient (k_x3_bwd_weight): 1 split of g, 2 split of x, 4 loads, 16 MFMA phase, 32 slab store
WPARTS = [('full', 0), ('-g split', 1), ('-x split', 2), ('-both splits', 3), ('-splits -loads', 7), ('-MFMA', 16), ('-slab store', 32),
          ('-staging -MFMA', 23), ('nothing', 55)]
print()
for name, H, C, Kc, res in LAYERS:
    x = torch.randn(2, H, H, C, device=dev)
    g = torch.randn(2, H, H, Kc, device=dev)
    d = K.conv_desc(x.shape, (1, 1, C, Kc), 1, 1, 'SAME', 'relu', 'bf16x3')
    dw = torch.empty(1, 1, C, Kc, device=dev)
    row = []
    for pname, bitsv in WPARTS:
        K.set_option('x3_stagger', bitsv << 8)
        row.append((pname, timeit(lambda: K.conv2d_bwd_weight(d, x, g, out=dw))))
    K.set_option('x3_stagger', 0)
    print('wgrad %-20s' % name.replace(' (+res)', '') + '  '.join('%s %.1f' % (p, t) for p, t in row))
