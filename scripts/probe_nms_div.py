import numpy as np, torch, sys
sys.path.insert(0, '.')
from luminoth_amd import kernels as K
b = np.array([[[0,0,10,10],[0,0,10,7]]], np.float32)
for thr in (0.7, 0.69, 0.71):
    keep, kc = K.nms(torch.tensor(b).cuda(), torch.tensor([2], dtype=torch.int32).cuda(), thr, 2)
    print('thr', thr, 'kept', int(kc[0]), keep.cpu().numpy())
a = torch.arange(1, 2001, dtype=torch.float32).cuda()
q = (a[:, None] / a[None, :]).cpu().numpy()
an = np.arange(1, 2001, dtype=np.float32)
qr = an[:, None] / an[None, :]
print('torch div mismatches vs numpy:', (q != qr).sum())
