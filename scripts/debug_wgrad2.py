import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luminoth_amd import kernels as K
lib = K._lib.load()
dev = torch.device('cuda:0')
shapes = [(2, 16, 16, 64, 64), (2, 32, 32, 64, 256), (1, 16, 16, 512, 72), (1, 1, 500, 1024, 320), (2, 128, 128, 256, 128), (2, 64, 64, 1024, 256)]
for (N, H, W, C, Kc) in shapes:
    x = torch.randn(N, H, W, C, device=dev); g = torch.randn(N, H, W, Kc, device=dev)
    d = K.conv_desc(x.shape, (1, 1, C, Kc), 1, 1, 'VALID', None)
    ref = x.reshape(-1, C).t() @ g.reshape(-1, Kc)
    torch.cuda.synchronize()
    for var in (-1, 2, 4, 0):
        lib.lmh_conv2d_force_wgrad_variant(var)
        print('shape', (N, H, W, C, Kc), 'variant', var, 'ws', lib.lmh_conv2d_bwd_weight_workspace_bytes(d), flush=True)
        for rep in range(3):
            dw = K.conv2d_bwd_weight(d, x, g)
        torch.cuda.synchronize()
        print('   max err', float((dw.reshape(C, Kc) - ref).abs().max()), 'ref scale', float(ref.abs().max()), flush=True)
