"""Runs one conv shape a few times (for rocprofv3 --pmc).  python scripts/pmc_conv.py <layer-substr> <op> [bm bn splits]"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luminoth_amd import kernels as K
from scripts.bench_conv import LAYERS, B
name_f, op = sys.argv[1], sys.argv[2]
force = [int(v) for v in sys.argv[3:6]] if len(sys.argv) > 5 else None
lib = K._lib.load()
dev = torch.device('cuda:0')
for name, H, C, Kc, R, stride, pad in LAYERS:
    if name_f not in name:
        continue
    x = torch.randn(B, H, H, C, device=dev)
    w = torch.randn(R, R, C, Kc, device=dev) * 0.05
    d = K.conv_desc(x.shape, w.shape, stride, 1, pad, 'relu', os.environ.get('BENCH_COMPUTE') or None)
    scale = torch.ones(Kc, device=dev); shift = torch.zeros(Kc, device=dev)
    y = K.conv2d_fwd(d, x, w, scale, shift)
    gy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w)
    if force:
        lib.lmh_conv2d_force_config(*force)
    for _ in range(5):
        if op == 'fwd':
            K.conv2d_fwd(d, x, w, scale, shift, out=y)
        elif op == 'bwd_data':
            K.conv2d_bwd_data(d, gy, w, scale, out=dx)
        else:
            K.conv2d_bwd_weight(d, x, gy, out=dw)
    torch.cuda.synchronize()
