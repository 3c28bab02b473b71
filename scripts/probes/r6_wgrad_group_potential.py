"""What a grouped, unsplit launch of the half-storage weight gradient could reach: the existing kernel on ONE synthetic layer with as
many 128 x 128 tiles as a whole ResNet block has (no split-K, every block walks all pixels) against the real layers of that block
launched one by one with their split-K plans (+ the slab reduction each of them needs).  us per launch, HIP events of the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from luminoth_amd import kernels as K
dev = torch.device('cuda:0')
lib = K._lib.load()


def timed(fn, n=10):
    for _ in range(2):
        fn()
    K._Profile.start()
    for _ in range(n):
        fn()
    r = K._Profile.stop()
    return sum(v['ms'] for v in r.values()) / n * 1e3, list(r.keys())


def wgrad(H, W, C, Kc, R, force=None):
    x = torch.randn(2, H, W, C, device=dev).half()
    g = (torch.randn(2, H, W, Kc, device=dev) * 0.05).half()
    d = K.conv_desc(x.shape, (R, R, C, Kc), 1, 1, 'SAME', 'relu', 'f16')
    dw = torch.empty(R, R, C, Kc, device=dev)
    lib.lmh_conv2d_force_config(*(force or (0, 0, 0)))
    t, names = timed(lambda: K.conv2d_bwd_weight_hs(d, x, g, 1.0, out=dw))
    lib.lmh_conv2d_force_config(0, 0, 0)
    fl = 2.0 * 2 * H * W * R * R * C * Kc
    print('%3dx%-3d %4d->%-4d %dx%d %-12s %7.1f us %5.0f TF/s  %s' % (H, W, C, Kc, R, R, force or 'default', t, fl / t / 1e6, names), flush=True)
    return t


S16, S8 = (50, 84), (100, 167)
print('block3 unit, real layers (default plans):')
tot = wgrad(*S16, 1024, 256, 1) + wgrad(*S16, 256, 256, 3) + wgrad(*S16, 256, 1024, 1)
print('  one unit: %.1f us; six units + shortcut ~ %.0f us' % (tot, 6 * tot + 20))
print('synthetic, unsplit:')
wgrad(*S16, 2048, 2048, 1, (128, 128, 1))        # 256 tiles, 131 stages each: 70 GF
wgrad(*S16, 2048, 1024, 1, (128, 128, 1))        # 128 tiles
wgrad(*S16, 1024, 1024, 3, (128, 128, 1))        # 576 tiles (taps), gathered
wgrad(*S16, 1024, 512, 3, (128, 128, 1))         # the RPN shape unsplit: 288 tiles
print('block2 unit, real layers:')
tot2 = wgrad(*S8, 512, 128, 1) + wgrad(*S8, 128, 128, 3) + wgrad(*S8, 128, 512, 1)
print('  one unit: %.1f us' % tot2)
wgrad(*S8, 1024, 1024, 1, (128, 128, 4))         # 64 tiles x 4 splits, 130 stages each
