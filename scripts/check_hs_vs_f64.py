"""Accuracy of the half-storage convolution kernels (csrc/conv_hs.h) against float64 references on the SAME 16-bit operands,
beside the native fp32-MFMA kernel on those operands: the f16 / bf16 MFMA accumulates at least as accurately as the fp32 one
(forward 5e-7 vs 1.8e-6 of the output scale), and 99.9 % of the stored 16-bit results equal the rounded float64 result."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from luminoth_amd import kernels as K
import oracle.torch_ops as ot
dev = 'cuda:0'
rs = np.random.RandomState(5)
F = np.float32
def run(storage, tdt, N, H, W, C, Kc, R, gs):
    x = torch.tensor(rs.randn(N, H, W, C).astype(F)).to(tdt)
    w = torch.tensor((rs.randn(R, R, C, Kc) * np.sqrt(2.0 / (R * R * C))).astype(F))
    scale = torch.tensor((1 + 0.1 * rs.randn(Kc)).astype(F))
    d = K.conv_desc(x.shape, w.shape, 1, 1, 'SAME', None, storage)
    d32 = K.conv_desc(x.shape, w.shape, 1, 1, 'SAME', None, None)
    wf = torch.empty((Kc, R, R, C), dtype=tdt, device=dev)
    wb = torch.empty((R, R, C, Kc), dtype=tdt, device=dev)
    K.half_weights_batch([(w.to(dev), scale.to(dev), wf, wb)], storage)
    # forward fp32-out vs fp64 reference and vs the fp32 MFMA kernel on the same rounded operands
    y = K.conv2d_fwd_hs(d, x.to(dev), wf, out_f32=True).cpu().double()
    wq = w.to(tdt).to(dev)      # HWIO rounded (wf / wb themselves are in MFMA fragment order: opaque)
    wbq = (torch.tensor((w.double().numpy() * scale.double().numpy()).astype(np.float16)) if storage == 'f16' else (w * scale).to(tdt)).to(dev)
    K.WINOGRAD = False
    y32 = K.conv2d_fwd(d32, x.to(dev).float(), wq.float()).cpu().double()
    ref = ot.conv2d_nhwc(x.double(), wq.cpu().double(), 1, 1, 'SAME')
    sc = float(ref.abs().max())
    print(storage, (N, H, W, C, Kc, R), 'FWD  hs-vs-f64 max %.2e rms %.2e | fp32kernel-vs-f64 max %.2e rms %.2e  (of scale)' % (
        float((y - ref).abs().max()) / sc, float((y - ref).pow(2).mean().sqrt()) / sc,
        float((y32 - ref).abs().max()) / sc, float((y32 - ref).pow(2).mean().sqrt()) / sc))
    g = torch.tensor((rs.randn(N, H, W, Kc) * gs).astype(F)).to(tdt)
    dx = K.conv2d_bwd_data_hs(d, g.to(dev), wb).cpu()
    dx32 = K.conv2d_bwd_data(d32, g.to(dev).float(), wbq.float()).cpu().double()
    xt = x.double().clone().requires_grad_(True)
    ot.conv2d_nhwc(xt, wbq.cpu().double(), 1, 1, 'SAME').backward(g.double())
    r = xt.grad
    sc = float(r.abs().max())
    rq = r.float().to(tdt)
    print('      BWD  hs(rounded)-vs-f64 max %.2e | q(f64) exact frac %.4f | fp32kernel-vs-f64 max %.2e; q(fp32kernel)==hs frac %.4f' % (
        float((dx.double() - r).abs().max()) / sc, float((dx == rq).float().mean()),
        float((dx32 - r).abs().max()) / sc, float((dx32.float().to(tdt) == dx).float().mean())))
for storage, tdt in (('f16', torch.float16), ('bf16', torch.bfloat16)):
    run(storage, tdt, 1, 64, 64, 1024, 256, 1, 0.05)
    run(storage, tdt, 1, 64, 64, 1024, 256, 1, 1.0)
    run(storage, tdt, 2, 32, 32, 256, 256, 3, 0.05)
    run(storage, tdt, 2, 16, 16, 64, 128, 1, 0.05)
