import sys
sys.path.insert(0, '.')
import numpy as np, torch
from luminoth_amd import kernels as K
from oracle import torch_ops as ot
F = np.float32
T = lambda a: torch.tensor(a).cuda()
for case in [(1, 64, 64, 1024, 512, 3, 1, 1, 'SAME', 'relu6'), (1, 64, 64, 1024, 512, 1, 1, 1, 'SAME', 'relu6'),
             (1, 64, 64, 512, 512, 3, 1, 1, 'SAME', 'relu')]:
    N, H, W, C, Kc, R, stride, dil, padding, act = case
    rs = np.random.RandomState(1)
    x = rs.randn(N, H, W, C).astype(F)
    w = (rs.randn(R, R, C, Kc) * np.sqrt(2.0 / (R * R * C))).astype(F)
    scale = (1 + 0.1 * rs.randn(Kc)).astype(F)
    d = K.conv_desc(x.shape, w.shape, stride, dil, padding, act)
    gy = rs.randn(N, d.OH, d.OW, Kc).astype(F)
    xt = torch.tensor(x, requires_grad=True)
    conv = ot.conv2d_nhwc(xt, torch.tensor(w), stride, dil, padding) * torch.tensor(scale)
    conv.backward(torch.tensor(gy))
    ref = xt.grad.numpy()
    print(case, 'kernel id', K._lib.load().lmh_conv2d_kernel_id(d, 1))
    for trial in range(3):
        dx = K.conv2d_bwd_data(d, T(gy), T(w), T(scale)).cpu().numpy()
        err = np.abs(dx - ref)
        bad = err > 1e-3 * max(1, np.abs(ref).max())
        print(' trial', trial, 'max err', err.max(), 'bad frac', bad.mean())
        if bad.any():
            b = bad.reshape(H * W, C)
            rows = np.where(b.any(1))[0]; cols = np.where(b.any(0))[0]
            print('  bad rows', len(rows), rows[:20], 'bad cols', len(cols), cols[:20])
            print('  per row-tile(128):', b.reshape(-1, 128, C).any(axis=(1, 2)).astype(int))
            print('  per col-tile(64):', b.reshape(H * W, -1, 64).any(axis=(0, 2)).astype(int))
            print('  sample', dx.reshape(-1, C)[rows[0], cols[:4]], ref.reshape(-1, C)[rows[0], cols[:4]])
