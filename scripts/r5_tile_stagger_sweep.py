"""(needs the probe build: `LMH_PROBES=1 bash luminoth_amd/csrc/build.sh` after touching conv.hip)  Round 5 probe: do SMALLER tiles (more tiles than resident-block slots) with start-time stagger of the co-resident blocks
beat one lock-step wave of 128x128 tiles on the 1x1 layers of block2 / block3?  (a 512-tile launch is 2 blocks
per CU that do prologue / main loop / epilogue at the same time; the round-4 stagger probe gained 8 % only where a CU ran
>= 4 tiles.)  One layer at a time, 40 launches back to back on one stream (kernels of a stream run one after the other),
rotating over 4 sets of tensors (~300 MB: past the 256 MB Infinity Cache), HIP events."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luminoth_amd import kernels as K

lib = K._lib.load()
dev = torch.device('cuda:0')
B = 2
LAYERS = [  # name, H, C, K, residual
    ('b3 256->1024+res', 64, 256, 1024, True),
    ('b3 1024->256', 64, 1024, 256, False),
    ('b3 512->256', 64, 512, 256, False),
    ('b2 128->512+res', 128, 128, 512, True),
    ('b2 512->128', 128, 512, 128, False),
    ('b2 256->128', 128, 256, 128, False),
]
TILES = [(0, 0), (128, 128), (128, 64), (64, 64)]
STAG = [0, 2, 4, 8]
NSET = 4


def timeit(fn, iters=40):
    for i in range(4):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for op in ('fwd', 'bwd_data'):
    print('==== %s: us per launch (TF/s) by tile and stagger units %s' % (op, STAG))
    for name, H, C, Kc, res in LAYERS:
        xs = [torch.randn(B, H, H, C, device=dev) for _ in range(NSET)]
        ws = [torch.randn(1, 1, C, Kc, device=dev) * 0.05 for _ in range(NSET)]
        rs = [torch.randn(B, H, H, Kc, device=dev) for _ in range(NSET)]
        ys = [torch.empty(B, H, H, Kc, device=dev) for _ in range(NSET)]
        dxs = [torch.empty(B, H, H, C, device=dev) for _ in range(NSET)]
        scale, shift = torch.ones(Kc, device=dev), torch.zeros(Kc, device=dev)
        d = K.conv_desc(xs[0].shape, ws[0].shape, 1, 1, 'SAME', 'relu')
        fl = 2.0 * B * H * H * C * Kc
        ref = None
        for bm, bn in TILES:
            lib.lmh_conv2d_force_config(bm, bn, 0)
            row = []
            for st in STAG:
                lib.lmh_conv_set_stagger(st)
                if op == 'fwd':
                    t = timeit(lambda i: K.conv2d_fwd(d, xs[i % NSET], ws[i % NSET], scale, shift,
                                                      residual=rs[i % NSET] if res else None, out=ys[i % NSET]))
                    out = ys[0]
                else:
                    t = timeit(lambda i: K.conv2d_bwd_data(d, rs[i % NSET], ws[i % NSET], scale,
                                                           addend=xs[i % NSET] if res else None, out=dxs[i % NSET]))
                    out = dxs[0]
                row.append(t)
            if ref is None:
                ref = out.clone()
            else:
                if not torch.equal(out, ref):
                    print('   !! result differs from the first tile config: max |d| %g' % float((out - ref).abs().max()))
            kid = lib.lmh_conv2d_kernel_id(d, 0 if op == 'fwd' else 1) % 1000000
            print('%-18s tile %-8s (%6d): %s' % (name, 'auto' if not bm else '%dx%d' % (bm, bn), kid,
                                                 '  '.join('%6.1f (%5.1f)' % (t, fl / t / 1e6) for t in row)))
        lib.lmh_conv2d_force_config(0, 0, 0)
        lib.lmh_conv_set_stagger(0)
