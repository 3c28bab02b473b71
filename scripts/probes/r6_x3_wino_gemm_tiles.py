"""The stacked transformed-domain GEMM of the bf16x3 Winograd layers (k_x3_fwd<..., GB>) alone, per forced tile: the library's own
HIP events around that launch (K._Profile), forward direction, ResNet-50 shapes at 2 x 1024^2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from luminoth_amd import kernels as K
lib = K._lib.load()
dev = torch.device('cuda:0')
for name, N, H, C, Kc in [('b3 3x3 256->256 @64', 2, 64, 256, 256), ('b2 3x3 128->128 @128', 2, 128, 128, 128),
                          ('rpn 3x3 1024->512 @64', 2, 64, 1024, 512)]:
    x = torch.randn(N, H, H, C, device=dev)
    w = torch.randn(3, 3, C, Kc, device=dev) * 0.02
    sc, sh = torch.ones(Kc, device=dev), torch.zeros(Kc, device=dev)
    d = K.conv_desc(x.shape, w.shape, 1, 1, 'SAME', 'relu', 'bf16x3')
    y = torch.empty(N, H, H, Kc, device=dev)
    u = K.new_winograd_u(C, Kc, dev)
    K.winograd_transform_weights(d, w, None, False, u)
    line = '%-24s' % name
    for bm, bn in ((0, 0), (128, 128), (128, 64), (64, 64)):
        lib.lmh_conv2d_force_config(bm, bn, 0)
        for _ in range(2):
            K.conv2d_fwd_winograd(d, x, w, sc, sh, out=y, u=u)
        K._Profile.start()
        for _ in range(10):
            K.conv2d_fwd_winograd(d, x, w, sc, sh, out=y, u=u)
        r = K._Profile.stop()
        line += ' | %s ' % ('auto' if bm == 0 else '%dx%d' % (bm, bn)) + ' '.join('%s %.1f' % (k.split('<')[0][2:] + '<' + k.split('<')[1][:11], v['ms'] / v['launches'] * 1e3) for k, v in r.items())
    lib.lmh_conv2d_force_config(0, 0, 0)
    print(line, flush=True)
