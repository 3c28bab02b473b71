"""Per-layer timing of the half-storage convolution kernels (csrc/conv_hs.h) on the ResNet-50 layers of BASELINE configs[4]
(2 x 800 x 1333): forward / backward data / weight gradient, each under the default tile plan and under forced plans.
Durations are the library's own HIP events around the kernel (no host time in them).
python scripts/bench_conv_hs.py [f16|bf16] [filter] [--sweep]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from luminoth_amd import kernels as K

storage = sys.argv[1] if len(sys.argv) > 1 else 'f16'
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else ''
sweep = '--sweep' in sys.argv
only = [a[7:] for a in sys.argv if a.startswith('--only=')]
bgs = [int(a[5:]) for a in sys.argv if a.startswith('--bg=')]      # hs_bg: B fragments from global memory (1) / through LDS (0)
B = 2
S4, S8, S16 = (200, 334), (100, 167), (50, 84)
LAYERS = [
    ('b1 1x1 64->64', S4, 64, 64, 1, 1, 'SAME'), ('b1 3x3 64->64', S4, 64, 64, 3, 1, 'SAME'),
    ('b1 1x1 64->256', S4, 64, 256, 1, 1, 'SAME'), ('b1 1x1 256->64', S4, 256, 64, 1, 1, 'SAME'),
    ('b1 3x3/2 64->64', S4, 64, 64, 3, 2, 'SAME_EXPLICIT'),
    ('b2 1x1 256->128', S8, 256, 128, 1, 1, 'SAME'), ('b2 3x3 128->128', S8, 128, 128, 3, 1, 'SAME'),
    ('b2 1x1 128->512', S8, 128, 512, 1, 1, 'SAME'), ('b2 1x1 256->512', S8, 256, 512, 1, 1, 'SAME'),
    ('b2 1x1 512->128', S8, 512, 128, 1, 1, 'SAME'), ('b2 3x3/2 128->128', S8, 128, 128, 3, 2, 'SAME_EXPLICIT'),
    ('b3 1x1 512->256', S16, 512, 256, 1, 1, 'SAME'), ('b3 3x3 256->256', S16, 256, 256, 3, 1, 'SAME'),
    ('b3 1x1 256->1024', S16, 256, 1024, 1, 1, 'SAME'), ('b3 1x1 512->1024', S16, 512, 1024, 1, 1, 'SAME'),
    ('b3 1x1 1024->256', S16, 1024, 256, 1, 1, 'SAME'), ('rpn 3x3 1024->512', S16, 1024, 512, 3, 1, 'SAME'),
]
dev = torch.device('cuda:0')
if bgs:
    K.set_option('hs_bg', bgs[0])
for a in sys.argv:
    if a.startswith('--rs='):      # hs_wg_rs: weight-gradient tiles through registers (1) / LDS-DMA (0)
        K.set_option('hs_wg_rs', int(a[5:]))
_, tdt = K.half_type(storage)
lib = K._lib.load()


def timed(fn, n=12):
    for _ in range(2):
        fn()
    K._Profile.start()
    for _ in range(n):
        fn()
    r = K._Profile.stop()
    (name, v), = r.items()
    return v['ms'] / v['launches'] * 1e3, name


def main():
    tot = [0.0, 0.0, 0.0]
    for name, (H, W), C, Kc, R, stride, pad in LAYERS:
        if flt and flt not in name:
            continue
        x = torch.randn(B, H, W, C, device=dev).to(tdt)
        w = torch.randn(R, R, C, Kc, device=dev) * 0.05
        d = K.conv_desc(x.shape, w.shape, stride, 1, pad, 'relu', storage)
        wf = torch.empty((Kc, R, R, C), dtype=tdt, device=dev)
        wb = torch.empty((R, R, C, Kc), dtype=tdt, device=dev)
        K.half_weights_batch([(w, None, wf, wb)], storage)
        y = K.conv2d_fwd_hs(d, x, wf)
        g = torch.randn_like(y)
        dw = torch.empty_like(w)
        fl = 2.0 * B * d.OH * d.OW * Kc * R * R * C
        by = 2.0 * (x.numel() + w.numel() + y.numel())
        ops = (('fwd', lambda: K.conv2d_fwd_hs(d, x, wf)), ('bwd', lambda: K.conv2d_bwd_data_hs(d, g, wb)),
               ('wgr', lambda: K.conv2d_bwd_weight_hs(d, x, g, 1.0, out=dw)))
        row = '%-20s %6.2f GF %5.1f MB |' % (name, fl / 1e9, by / 1e6)
        for i, (op, fn) in enumerate(ops):
            if only and op not in only:
                continue
            lib.lmh_conv2d_force_config(0, 0, 0)
            t, kn = timed(fn)
            tot[i] += t
            row += ' %s %6.1f us %5.0f TF %4.2f TB/s %-14s|' % (op, t, fl / t / 1e6, by / t / 1e6, kn[kn.index('<') + 4:-1] if '<' in kn else kn)
            if sweep:
                cfgs = ((64, 64, 0), (64, 128, 0), (128, 64, 0), (128, 128, 0), (256, 128, 0)) if op != 'wgr' else \
                    ((64, 64, 0), (64, 64, 2), (64, 64, 4), (64, 64, 8), (64, 64, 16), (128, 128, 0), (128, 128, 4), (128, 128, 8), (128, 128, 16))
                for bm, bn, sp in cfgs:
                    lib.lmh_conv2d_force_config(bm, bn, sp)
                    t2, _ = timed(fn, 8)
                    row += ' %dx%d/%d:%.1f' % (bm, bn, sp, t2)
                row += ' |'
        lib.lmh_conv2d_force_config(0, 0, 0)
        print(row, flush=True)
    print('sum us: fwd %.1f bwd %.1f wgrad %.1f' % tuple(tot))


if __name__ == '__main__':
    main()
