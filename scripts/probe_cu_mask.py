"""Does hipExtStreamCreateWithCUMask confine a kernel?  Times one MFMA-bound convolution (RPN-size 1x1, 8.6 GFLOP) on an
unmasked stream and on streams masked to keep/period of the CUs (lmh_stream_create_cu_mask), alone on the GPU."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luminoth_amd import kernels as K, _lib

lib = _lib.load()
dev = torch.device('cuda')
x = torch.randn(2, 64, 64, 1024, device=dev)
w = torch.randn(1, 1, 1024, 512, device=dev) * 0.03
d = K.conv_desc(x.shape, w.shape, 1, 1, 'SAME', 'relu')
y = torch.empty(2, 64, 64, 512, device=dev)


def run(stream, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        for _ in range(3):
            K.conv2d_fwd(d, x, w, out=y)
        e0.record()
        for _ in range(reps):
            K.conv2d_fwd(d, x, w, out=y)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print('unmasked: %.1f us' % run(torch.cuda.Stream()))
for period, keep in ((2, 1), (4, 1), (4, 3), (3, 1), (8, 1), (32, 8)):
    h = lib.lmh_stream_create_cu_mask(period, keep)
    if not h:
        print(period, keep, 'create failed', lib.lmh_last_error())
        continue
    st = torch.cuda.ExternalStream(h)
    print('per-XCD CUs c %% %d < %d: %.1f us' % (period, keep, run(st)))
