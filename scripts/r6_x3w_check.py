"""bf16x3 forward with pre-split weights against the in-kernel split: bit identity and launch time per layer (plan replay)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luminoth_amd import kernels as K
from luminoth_amd import plan as P

dev = torch.device('cuda:0')
N = 20
LAYERS = [('b3 1x1 256->1024 (+res)', 64, 256, 1024, 1, True), ('b3 1x1 1024->256', 64, 1024, 256, 1, False),
          ('b3 1x1 512->256', 64, 512, 256, 1, False), ('b2 1x1 128->512 (+res)', 128, 128, 512, 1, True),
          ('b2 1x1 512->128', 128, 512, 128, 1, False), ('b2 1x1 256->128', 128, 256, 128, 1, False),
          ('b1 1x1 64->256 (+res)', 256, 64, 256, 1, True), ('b1 1x1 256->64', 256, 256, 64, 1, False),
          ('b1 3x3 64->64', 256, 64, 64, 3, False), ('odd 19x23 96->96 3x3', 19, 96, 96, 3, True)]


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
for name, H, C, Kc, R, res in LAYERS:
    W = H if H != 19 else 23
    x = torch.randn(2, H, W, C, device=dev)
    w = torch.randn(R, R, C, Kc, device=dev) * 0.05
    sc, sh = torch.rand(Kc, device=dev) + 0.5, torch.randn(Kc, device=dev)
    r = torch.randn(2, H, W, Kc, device=dev) if res else None
    d = K.conv_desc(x.shape, w.shape, 1, 1, 'SAME', 'relu', 'bf16x3')
    assert K.conv2d_fwd_x3w_ok(d)
    w3 = K.new_x3_weights(R * R, C, Kc, dev)
    K.x3_split_weights_batch([(w, R * R, w3)])
    b0, b1 = K.new_act_bits(2 * H * W, Kc, dev), K.new_act_bits(2 * H * W, Kc, dev)
    y0 = K.conv2d_fwd(d, x, w, sc, sh, r, act_bits=b0)
    y1 = K.conv2d_fwd_x3w(d, x, w3, sc, sh, r, act_bits=b1)
    torch.cuda.synchronize()
    same = bool(torch.equal(y0, y1)) and bool(torch.equal(b0, b1))
    t0 = timeit(lambda: K.conv2d_fwd(d, x, w, sc, sh, r, out=y0, act_bits=b0))
    t1 = timeit(lambda: K.conv2d_fwd_x3w(d, x, w3, sc, sh, r, out=y1, act_bits=b1))
    ts = timeit(lambda: K.x3_split_weights_batch([(w, R * R, w3)]))
    print('%-26s fwd identical %s   in-kernel split %.1f us   pre-split %.1f us   (split pass %.1f us)'
          % (name, same, t0, t1, ts))
    # backward data: dy (2,H,W,Kc) -> dx (2,H,W,C), kscale + addend + input mask
    assert K.conv2d_bwd_data_x3w_ok(d)
    w3b = K.new_x3_weights(R * R, C, Kc, dev, backward=True)
    K.x3_split_weights_batch([(w, R * R, w3b)], backward=True)
    dy = torch.randn(2, H, W, Kc, device=dev)
    add = torch.randn(2, H, W, C, device=dev)
    xb = torch.randint(-2 ** 31, 2 ** 31 - 1, (2 * H * W, C // 32), dtype=torch.int32, device=dev)
    d0 = K.conv2d_bwd_data(d, dy, w, kscale=sc, addend=add, xbits=xb)
    d1 = K.conv2d_bwd_data_x3w(d, dy, w3b, kscale=sc, addend=add, xbits=xb)
    torch.cuda.synchronize()
    sameb = bool(torch.equal(d0, d1))
    t0 = timeit(lambda: K.conv2d_bwd_data(d, dy, w, kscale=sc, addend=add, xbits=xb, out=d0))
    t1 = timeit(lambda: K.conv2d_bwd_data_x3w(d, dy, w3b, kscale=sc, addend=add, xbits=xb, out=d1))
    print('%-26s bwd identical %s   in-kernel split %.1f us   pre-split %.1f us' % ('', sameb, t0, t1))
