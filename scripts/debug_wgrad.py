import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luminoth_amd import kernels as K
lib = K._lib.load()
dev = torch.device('cuda:0')
N, H, W, C, Kc = 2, 16, 16, 64, 64
x = torch.randn(N, H, W, C, device=dev); g = torch.randn(N, H, W, Kc, device=dev)
d = K.conv_desc(x.shape, (1, 1, C, Kc), 1, 1, 'VALID', None)
ref = x.reshape(-1, C).t() @ g.reshape(-1, Kc)
for var in (-1, 2, 3, 4):
    for bm, bn in ((64, 64), (128, 128)):
        for sp in (1, 2):
            lib.lmh_conv2d_force_config(bm, bn, sp); lib.lmh_conv2d_force_wgrad_variant(var)
            print('variant', var, 'tile', bm, bn, 'splits', sp, 'ws', lib.lmh_conv2d_bwd_weight_workspace_bytes(d), flush=True)
            dw = K.conv2d_bwd_weight(d, x, g)
            torch.cuda.synchronize()
            print('   max err', float((dw.reshape(C, Kc) - ref).abs().max()), flush=True)
