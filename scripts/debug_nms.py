import sys; sys.path.insert(0, '.')
import numpy as np, torch
from luminoth_amd import kernels as K
from oracle import tfops
F = np.float32
def rand_boxes(rs, n, lim=1024, smin=16, smax=400):
    wh = rs.randint(smin, smax, size=(n, 2))
    xy = np.stack([rs.randint(0, lim - wh[:, 0]), rs.randint(0, lim - wh[:, 1])], 1)
    return np.concatenate([xy, xy + wh - 1], 1).astype(F)
rs = np.random.RandomState(0)
Kn = 3000
base = rand_boxes(rs, 60, 600, 40, 200)
boxes = (base[rs.randint(0, 60, size=Kn)] + rs.randint(-12, 13, size=(Kn, 4))).astype(F)[None]
cnt = np.array([Kn], np.int32)
keep, kc = K.nms(torch.tensor(boxes).cuda(), torch.tensor(cnt).cuda(), 0.7, 300)
torch.cuda.synchronize()
keep = keep.cpu().numpy()[0]; kc = int(kc[0])
ref = tfops.non_max_suppression(boxes[0][:, [1, 0, 3, 2]], np.arange(Kn, 0, -1).astype(F), 300, 0.7)
print('gpu', kc, 'ref', len(ref))
g = keep[:kc]
i = 0
while i < min(len(g), len(ref)) and g[i] == ref[i]: i += 1
print('first diff at pos', i, 'gpu', g[i:i+3], 'ref', ref[i:i+3])
miss = ref[i]
W = (Kn + 63) // 64
ws = K._ws_cache[('nms', torch.device('cuda', 0))]
mask = ws[:Kn * W * 8].view(torch.int64).cpu().numpy().view(np.uint64).reshape(Kn, W)
# which kept gpu rows claim to suppress `miss`?
for r in g[:i]:
    if (mask[r, miss // 64] >> np.uint64(miss % 64)) & np.uint64(1):
        bi, bj = boxes[0][r], boxes[0][miss]
        print('row', r, 'suppresses', miss, bi, bj, 'oracle says', tfops.nms_iou_greater(bi[[1,0,3,2]], bj[[1,0,3,2]], 0.7))
        y1=max(bi[1],bj[1]); x1=max(bi[0],bj[0]); y2=min(bi[3],bj[3]); x2=min(bi[2],bj[2])
        inter=max(y2-y1,0)*max(x2-x1,0); a=(bi[3]-bi[1])*(bi[2]-bi[0]); b=(bj[3]-bj[1])*(bj[2]-bj[0])
        print(' inter', inter, 'a', a, 'b', b, 'iou', F(inter)/F(F(a)+F(b)-F(inter)))
print('chunk of miss', miss // 64, 'bit', miss % 64)
