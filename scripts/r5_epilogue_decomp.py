"""(needs the probe build: `LMH_PROBES=1 bash luminoth_amd/csrc/build.sh` after touching conv.hip)  Round 5 probe: what does the epilogue of the dominant 1x1 forward class cost?  Times k_conv_fwd on the two expand layers
with (a) everything, (b) no residual read, (c) no output stores, (d) neither, (e) no main loop (prologue + epilogue only),
(f) no main loop and no epilogue traffic (launch + index math).  40 launches back to back on one stream, 4 rotating tensor
sets (past the Infinity Cache).  The variants with bits set compute WRONG results: timing only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luminoth_amd import kernels as K

lib = K._lib.load()
dev = torch.device('cuda:0')
B, NSET = 2, 4
LAYERS = [('b3 256->1024+res', 64, 256, 1024), ('b2 128->512+res', 128, 128, 512), ('b3 512->1024', 64, 512, 1024),
          ('b2 256->512', 128, 256, 512), ('b3 1024->256', 64, 1024, 256), ('b2 512->128', 128, 512, 128)]
VAR = [('pp', -1), ('all', 0), ('no residual', 256), ('no stores', 512), ('no res, no stores', 768), ('no main loop', 1024),
       ('no loop, no res/stores', 1792)]


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


for name, H, C, Kc in LAYERS:
    res = '+res' in name
    xs = [torch.randn(B, H, H, C, device=dev) for _ in range(NSET)]
    ws = [torch.randn(1, 1, C, Kc, device=dev) * 0.05 for _ in range(NSET)]
    rs = [torch.randn(B, H, H, Kc, device=dev) for _ in range(NSET)]
    ys = [torch.empty(B, H, H, Kc, device=dev) for _ in range(NSET)]
    bits = [K.new_act_bits(B * H * H, Kc, dev) for _ in range(NSET)]
    scale, shift = torch.ones(Kc, device=dev), torch.zeros(Kc, device=dev)
    d = K.conv_desc(xs[0].shape, ws[0].shape, 1, 1, 'SAME', 'relu')
    row = []
    for vn, code in VAR:
        K.set_option('conv_pp', 1 if code < 0 else 0)
        lib.lmh_conv_set_stagger(max(code, 0))
        t = timeit(lambda i: K.conv2d_fwd(d, xs[i % NSET], ws[i % NSET], scale, shift, residual=rs[i % NSET] if res else None,
                                          out=ys[i % NSET], act_bits=bits[i % NSET]))
        row.append('%s %.1f' % (vn, t))
    lib.lmh_conv_set_stagger(0)
    K.set_option('conv_pp', 1)
    print('%-18s (kernel %d): %s' % (name, lib.lmh_conv2d_kernel_id(d, 0) % 1000000, ' | '.join(row)))
