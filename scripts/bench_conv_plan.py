"""Per-layer micro-benchmark of the conv kernels with the launches REPLAYED FROM C (csrc/plan.hip): scripts/bench_conv.py
issues every call through Python -> ctypes (~20 us each), so anything shorter than that measures the host.  Here N calls
are recorded into a launch plan once and replayed (3-4 us per node from C): the figure is the back-to-back kernel time
on the GPU.  BENCH_COMPUTE=bf16x3, LMH_OPT_X3_NEW=0/1, ... select the arithmetic / kernels.
    python scripts/bench_conv_plan.py [filter]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from luminoth_amd import kernels as K
from luminoth_amd import plan as P
from scripts.bench_conv import LAYERS, B

flt = sys.argv[1] if len(sys.argv) > 1 else ''
dev = torch.device('cuda:0')
N = int(os.environ.get('BENCH_N', '20'))
WINO = os.environ.get('BENCH_WINO', '0') == '1'


def timeit(fn):
    fn()
    fn()
    torch.cuda.synchronize()
    with P.StepPlan() as pl:
        for _ in range(N):
            fn()
    torch.cuda.synchronize()
    pl.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        pl.run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / N)
    nk = pl.n_kernels // N
    pl.destroy()
    return best, nk


def main():
    if not WINO:
        K.WINOGRAD = False
    if os.environ.get('BENCH_TILE'):      # "bm,bn[,splits]": lmh_conv2d_force_config
        t = [int(v) for v in os.environ['BENCH_TILE'].split(',')] + [0]
        K._lib.load().lmh_conv2d_force_config(t[0], t[1], t[2])
    comp = os.environ.get('BENCH_COMPUTE') or None
    tot = {'fwd': 0.0, 'bwd_data': 0.0, 'bwd_weight': 0.0}
    print('compute %s, winograd %s; us per call (kernels per call), TF/s of direct-convolution work' % (comp, WINO))
    print('%-22s %8s | %-20s | %-20s | %-20s' % ('layer', 'GFLOP', 'fwd', 'bwd_data', 'bwd_weight'))
    for name, H, C, Kc, R, stride, pad in LAYERS:
        if flt and flt not in name:
            continue
        if C % 32:
            continue
        x = torch.randn(B, H, H, C, device=dev)
        w = torch.randn(R, R, C, Kc, device=dev) * 0.05
        d = K.conv_desc(x.shape, w.shape, stride, 1, pad, 'relu', comp)
        scale = torch.ones(Kc, device=dev)
        shift = torch.zeros(Kc, device=dev)
        y = K.conv2d_fwd(d, x, w, scale, shift)
        res = torch.randn_like(y)
        bits = K.new_act_bits(y.numel() // Kc, Kc, dev) if Kc % 32 == 0 else None
        xbits = K.act_bits(x, 'relu')
        gy = torch.randn_like(y)
        fl = 2.0 * B * d.OH * d.OW * Kc * R * R * C
        t_f, k_f = timeit(lambda: K.conv2d_fwd(d, x, w, scale, shift, res, out=y, act_bits=bits))
        dx = torch.empty_like(x)
        t_d, k_d = timeit(lambda: K.conv2d_bwd_data(d, gy, w, scale, addend=x, out=dx, xbits=xbits))
        dw = torch.empty_like(w)
        cs = torch.empty(Kc, device=dev) if K.conv_fused_colsum_ok(d) else None
        t_w, k_w = timeit(lambda: K.conv2d_bwd_weight(d, x, gy, out=dw, colsum=cs))
        print('%-22s %8.2f | %7.1f (%d) %6.1f | %7.1f (%d) %6.1f | %7.1f (%d) %6.1f' % (
            name, fl / 1e9, t_f * 1e3, k_f, fl / t_f / 1e9, t_d * 1e3, k_d, fl / t_d / 1e9, t_w * 1e3, k_w, fl / t_w / 1e9))
        tot['fwd'] += t_f
        tot['bwd_data'] += t_d
        tot['bwd_weight'] += t_w
    print('sum ms:', {k: round(v, 3) for k, v in tot.items()})


if __name__ == '__main__':
    main()
