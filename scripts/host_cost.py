"""Host cost (Python + ctypes + hipLaunchKernel) of one call of each hot wrapper, on shapes small enough that the GPU
is never the bottleneck: what a train step pays per launch on the enqueue side."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from luminoth_amd import kernels as K
from luminoth_amd.models.base import layers as L

dev = torch.device('cuda:0')


def cost(fn, n=400):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6


x = torch.randn(1, 16, 16, 64, device=dev)
w1 = torch.randn(1, 1, 64, 64, device=dev)
w3 = torch.randn(3, 3, 64, 64, device=dev)
sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
d1 = K.conv_desc(x.shape, w1.shape, 1, 1, 'SAME', 'relu')
d3 = K.conv_desc(x.shape, w3.shape, 1, 1, 'SAME', 'relu')
y = torch.empty_like(x)
g = torch.randn_like(x)
dw1, dw3 = torch.empty_like(w1), torch.empty_like(w3)
bits = K.new_act_bits(256, 64, dev)
print('torch.empty                 %6.1f us' % cost(lambda: torch.empty((1, 16, 16, 64), dtype=torch.float32, device=dev)))
print('conv2d_fwd 1x1 (out given)  %6.1f us' % cost(lambda: K.conv2d_fwd(d1, x, w1, sc, sh, out=y)))
print('conv2d_fwd 1x1 (+alloc,bits)%6.1f us' % cost(lambda: K.conv2d_fwd(d1, x, w1, sc, sh, act_bits=bits)))
K.WINOGRAD_MIN_CK = 1
print('conv2d_fwd 3x3 winograd     %6.1f us' % cost(lambda: K.conv2d_fwd(d3, x, w3, sc, sh, out=y)))
print('conv2d_bwd_data 1x1         %6.1f us' % cost(lambda: K.conv2d_bwd_data(d1, g, w1, sc, out=y, xbits=bits)))
print('conv2d_bwd_data 3x3 wino    %6.1f us' % cost(lambda: K.conv2d_bwd_data(d3, g, w3, sc, out=y)))
print('conv2d_bwd_weight 1x1       %6.1f us' % cost(lambda: K.conv2d_bwd_weight(d1, x, g, out=dw1)))
K.WINOGRAD_WGRAD_MIN_CK = 1
print('conv2d_bwd_weight 3x3 wino  %6.1f us' % cost(lambda: K.conv2d_bwd_weight(d3, x, g, out=dw3)))
print('act_bwd                     %6.1f us' % cost(lambda: K.act_bwd(g, x, 'relu')))
lay = L.ConvLayer('s', 64, 64, 1)
lay.w, lay.scale, lay.shift, lay.norm = w1, sc, sh, None
print('ConvLayer.forward 1x1       %6.1f us' % cost(lambda: lay.forward(x, want_bits=True)))
st = torch.cuda.Stream()
ev = torch.cuda.Event()
print('event record + stream wait  %6.1f us' % cost(lambda: (ev.record(), st.wait_event(ev))))
print('torch add (33 MB eltwise op host cost on small) %6.1f us' % cost(lambda: torch.add(x, g, out=y)))
