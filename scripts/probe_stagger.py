"""Does running the two images of a batch as two independent chains on two streams (each kernel half the tiles, the chains
drifting out of phase so that one's HBM-bound epilogue overlaps the other's MFMA-bound main loop) beat one batched chain?
Chain = 6 block3 bottlenecks of ResNet-50 at 64 x 64 (1x1 1024->256, 3x3 256->256, 1x1 256->1024 + residual), forward only,
recorded in a launch plan and replayed (no host gaps)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luminoth_amd import kernels as K, plan as P

dev = torch.device('cuda')
torch.manual_seed(0)
B, H, W = 2, 64, 64
units = 6
w1 = [torch.randn(1, 1, 1024, 256, device=dev) * 0.03 for _ in range(units)]
w2 = [torch.randn(3, 3, 256, 256, device=dev) * 0.02 for _ in range(units)]
w3 = [torch.randn(1, 1, 256, 1024, device=dev) * 0.06 for _ in range(units)]
x0 = torch.randn(B, H, W, 1024, device=dev).relu_()


def chain(x, n):
    d1 = K.conv_desc((n, H, W, 1024), (1, 1, 1024, 256), 1, 1, 'SAME', 'relu')
    d2 = K.conv_desc((n, H, W, 256), (3, 3, 256, 256), 1, 1, 'SAME', 'relu')
    d3 = K.conv_desc((n, H, W, 256), (1, 1, 256, 1024), 1, 1, 'SAME', 'relu')
    for u in range(units):
        a = K.conv2d_fwd(d1, x, w1[u])
        b = K.conv2d_fwd(d2, a, w2[u])
        x = K.conv2d_fwd(d3, b, w3[u], residual=x)
    return x


def timed(plan, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        plan.run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


main = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
chain(x0, B)
torch.cuda.synchronize()
p1 = P.StepPlan()
with p1:
    y_batched = chain(x0, B)
xa, xb = x0[0:1].contiguous(), x0[1:2].contiguous()
with torch.cuda.stream(s1):
    chain(xa, 1)
with torch.cuda.stream(s2):
    chain(xb, 1)
torch.cuda.synchronize()
p2 = P.StepPlan()
with p2:
    K.stream_wait(s1, main)
    K.stream_wait(s2, main)
    with torch.cuda.stream(s1):
        ya = chain(xa, 1)
    with torch.cuda.stream(s2):
        yb = chain(xb, 1)
    K.stream_wait(main, s1)
    K.stream_wait(main, s2)
torch.cuda.synchronize()
assert torch.equal(torch.cat([ya, yb]), y_batched)
t1, t2 = timed(p1), timed(p2)
print('batched chain, one stream : %.1f us (%d launches)' % (t1, p1.n_kernels))
print('two per-image chains      : %.1f us (%d launches)' % (t2, p2.n_kernels))
