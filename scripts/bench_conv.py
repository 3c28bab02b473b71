"""Per-layer micro-benchmark of the conv kernels on the ResNet-50 @1024^2, B=2 shapes
(fwd / bwd_data / bwd_weight), HIP-event timed.  python scripts/bench_conv.py [filter]"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luminoth_amd import kernels as K

B = 2
LAYERS = [
    # name, H, C, K, R, stride, padding
    ('conv1 7x7/2 3->64', 1024, 3, 64, 7, 2, 'SAME_EXPLICIT'),
    ('b1 1x1 64->64', 256, 64, 64, 1, 1, 'SAME'),
    ('b1 3x3 64->64', 256, 64, 64, 3, 1, 'SAME'),
    ('b1 1x1 64->256', 256, 64, 256, 1, 1, 'SAME'),
    ('b1 1x1 256->64', 256, 256, 64, 1, 1, 'SAME'),
    ('b1 3x3/2 64->64', 256, 64, 64, 3, 2, 'SAME_EXPLICIT'),
    ('b2 1x1 256->128', 128, 256, 128, 1, 1, 'SAME'),
    ('b2 3x3 128->128', 128, 128, 128, 3, 1, 'SAME'),
    ('b2 1x1 128->512', 128, 128, 512, 1, 1, 'SAME'),
    ('b2 1x1 256->512', 128, 256, 512, 1, 1, 'SAME'),
    ('b2 1x1 512->128', 128, 512, 128, 1, 1, 'SAME'),
    ('b2 3x3/2 128->128', 128, 128, 128, 3, 2, 'SAME_EXPLICIT'),
    ('b3 1x1 512->256', 64, 512, 256, 1, 1, 'SAME'),
    ('b3 3x3 256->256', 64, 256, 256, 3, 1, 'SAME'),
    ('b3 1x1 256->1024', 64, 256, 1024, 1, 1, 'SAME'),
    ('b3 1x1 512->1024', 64, 512, 1024, 1, 1, 'SAME'),
    ('b3 1x1 1024->256', 64, 1024, 256, 1, 1, 'SAME'),
    ('rpn 3x3 1024->512', 64, 1024, 512, 3, 1, 'SAME'),
    ('rpn 1x1 512->48', 64, 512, 48, 1, 1, 'VALID'),
]
flt = sys.argv[1] if len(sys.argv) > 1 and __name__ == "__main__" else ""
dev = torch.device('cuda:0')


def timeit(fn, iters=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    tot = {'fwd': 0.0, 'bwd_data': 0.0, 'bwd_weight': 0.0}
    print('%-22s %8s | %-22s | %-22s | %-22s' % ('layer', 'GFLOP', 'fwd us / TF', 'bwd_data us / TF', 'bwd_weight us / TF'))
    for name, H, C, Kc, R, stride, pad in LAYERS:
        if flt and flt not in name:
            continue
        x = torch.randn(B, H, H, C, device=dev)
        w = torch.randn(R, R, C, Kc, device=dev) * 0.05
        d = K.conv_desc(x.shape, w.shape, stride, 1, pad, 'relu', os.environ.get('BENCH_COMPUTE') or None)
        scale = torch.ones(Kc, device=dev)
        shift = torch.zeros(Kc, device=dev)
        y = K.conv2d_fwd(d, x, w, scale, shift)
        gy = torch.randn_like(y)
        fl = 2.0 * B * d.OH * d.OW * Kc * R * R * C
        t_f = timeit(lambda: K.conv2d_fwd(d, x, w, scale, shift, out=y))
        row = '%-22s %8.2f | %8.1f %6.1f (%s)' % (name, fl / 1e9, t_f * 1e3, fl / t_f / 1e9,
                                                  K._lib.load().lmh_conv2d_kernel_id(d, 0) % 1000000)
        tot['fwd'] += t_f
        if C % 4 == 0:
            dx = torch.empty_like(x)
            t_d = timeit(lambda: K.conv2d_bwd_data(d, gy, w, scale, out=dx))
            dw = torch.empty_like(w)
            t_w = timeit(lambda: K.conv2d_bwd_weight(d, x, gy, out=dw))
            row += ' | %8.1f %6.1f (%s) | %8.1f %6.1f (%s)' % (
                t_d * 1e3, fl / t_d / 1e9, K._lib.load().lmh_conv2d_kernel_id(d, 1),
                t_w * 1e3, fl / t_w / 1e9, K._lib.load().lmh_conv2d_kernel_id(d, 2))
            tot['bwd_data'] += t_d
            tot['bwd_weight'] += t_w
        print(row)
    print('sum ms:', {k: round(v, 3) for k, v in tot.items()})


if __name__ == '__main__':
    main()
