"""The stacked GEMM of the bf16x3 Winograd WEIGHT GRADIENT (k_x3_bwd_weight<..., GB>) alone, per forced tile and split count: the
library's own HIP events around that launch (K._Profile), ResNet-50 shapes at 2 x 1024^2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from luminoth_amd import kernels as K
lib = K._lib.load()
dev = torch.device('cuda:0')
for name, N, H, C, Kc in [('b3 3x3 256->256 @64', 2, 64, 256, 256), ('b2 3x3 128->128 @128', 2, 128, 128, 128),
                          ('rpn 3x3 1024->512 @64', 2, 64, 1024, 512)]:
    x = torch.randn(N, H, H, C, device=dev)
    g = torch.randn(N, H, H, Kc, device=dev)
    d = K.conv_desc(x.shape, (3, 3, C, Kc), 1, 1, 'SAME', 'relu', 'bf16x3')
    dw = torch.empty(3, 3, C, Kc, device=dev)
    line = '%-24s' % name
    for bm, bn, sp in ((0, 0, 0), (128, 128, 1), (128, 128, 2), (128, 128, 4), (128, 128, 8), (128, 64, 2), (128, 64, 4), (64, 64, 1), (64, 64, 2), (64, 64, 4), (64, 64, 8)):
        lib.lmh_conv2d_force_config(bm, bn, sp)
        try:
            for _ in range(2):
                K.conv2d_bwd_weight_winograd(d, x, g, out=dw)
            K._Profile.start()
            for _ in range(8):
                K.conv2d_bwd_weight_winograd(d, x, g, out=dw)
            r = K._Profile.stop()
        except Exception as e:
            line += ' | %dx%d/s%d ERR' % (bm, bn, sp)
            continue
        t = [v['ms'] / v['launches'] * 1e3 for k, v in r.items() if 'bwd_weight' in k]
        line += ' | %s %.1f' % ('auto' if bm == 0 else '%dx%d/s%d' % (bm, bn, sp), t[0] if t else -1)
    lib.lmh_conv2d_force_config(0, 0, 0)
    print(line, flush=True)
