import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from luminoth_amd import kernels as K
from scripts.bench_conv import timeit
lib = ctypes.CDLL(K._lib.LIB_PATH)
dev = 'cuda:0'
B, R, FH, FW, C = 2, 256, 64, 64, 1024
rs = np.random.RandomState(0)
# clustered ROIs around 8 gt boxes (what training produces)
rois = np.zeros((B, R, 4), np.float32)
for b in range(B):
    for r in range(R):
        cx, cy = rs.choice([200, 500, 800]), rs.choice([300, 700])
        w, h = rs.randint(60, 400), rs.randint(60, 400)
        x1 = np.clip(cx - w / 2 + rs.randn() * 20, 0, 1023); y1 = np.clip(cy - h / 2 + rs.randn() * 20, 0, 1023)
        rois[b, r] = [x1, y1, min(x1 + w, 1023), min(y1 + h, 1023)]
feat = torch.randn(B, FH, FW, C, device=dev)
roist = torch.tensor(rois, device=dev); cnt = torch.full((B,), R, dtype=torch.int32, device=dev)
out, am = K.roi_pool_fwd(feat, roist, cnt, (1024, 1024))
g = torch.randn_like(out)
for dbg in (0,):
    pass
    t = timeit(lambda: K.roi_pool_bwd(g, am, roist, cnt, (B, FH, FW, C), (1024, 1024)), 10)
    print('roi_pool_bwd dbg=%d: %.1f us' % (dbg, t * 1e3))
pass
t = timeit(lambda: K.roi_pool_fwd(feat, roist, cnt, (1024, 1024)), 10)
print('roi_pool_fwd: %.1f us' % (t * 1e3))
t = timeit(lambda: K.spatial_mean_fwd(out), 10)
print('spatial_mean_fwd: %.1f us' % (t * 1e3))
dy = torch.randn(B * R, C, device=dev)
t = timeit(lambda: K.spatial_mean_bwd(dy, tuple(out.shape)), 10)
print('spatial_mean_bwd: %.1f us' % (t * 1e3))
mean, am2 = K.roi_pool_mean_fwd(feat, roist, cnt, (1024, 1024))
t = timeit(lambda: K.roi_pool_mean_fwd(feat, roist, cnt, (1024, 1024)), 10)
print('roi_pool_mean_fwd (fused): %.1f us' % (t * 1e3))
t = timeit(lambda: K.roi_pool_mean_bwd(dy, am2, roist, cnt, (B, FH, FW, C), (1024, 1024)), 10)
print('roi_pool_mean_bwd (fused): %.1f us' % (t * 1e3))
for cs in (8, 4, 2, 1):
    K.set_option('roi_mean_cs', -1 if cs == 8 else cs)
    t = timeit(lambda: K.roi_pool_mean_fwd(feat, roist, cnt, (1024, 1024)), 10)
    print('roi_pool_mean_fwd slab of %d channels: %.1f us' % (cs, t * 1e3))
K.set_option('roi_mean_cs', -1)
