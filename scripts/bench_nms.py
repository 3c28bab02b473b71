"""Standalone timing of lmh_nms (k_nms_mask + k_nms_reduce) at the train-step size: 2 images x 12 000 candidates,
2 000 kept, threshold 0.7, on anchor-like boxes (a random 12 000 of the 64x64x9 anchors, jittered — what the RPN of a
freshly initialised network proposes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from luminoth_amd import kernels as K
from scripts.bench_conv import timeit

dev = 'cuda:0'
rs = np.random.RandomState(0)
B, Kn, max_out, thr = 2, 12000, 2000, 0.7
ys, xs = np.meshgrid(np.arange(64) * 16 + 8, np.arange(64) * 16 + 8, indexing='ij')
boxes = []
for s in (128, 256, 512):
    for ar in (0.5, 1.0, 2.0):
        w, h = s / np.sqrt(ar), s * np.sqrt(ar)
        boxes.append(np.stack([xs - w / 2, ys - h / 2, xs + w / 2, ys + h / 2], -1).reshape(-1, 4))
anchors = np.concatenate(boxes, 0)
batch = np.zeros((B, Kn, 4), np.float32)
for b in range(B):
    sel = rs.choice(len(anchors), Kn, replace=False)
    batch[b] = np.clip(anchors[sel] + rs.randn(Kn, 4) * 4, 0, 1023)
bt = torch.tensor(batch, device=dev)
cnt = torch.full((B,), Kn, dtype=torch.int32, device=dev)
keep, kc = K.nms(bt, cnt, thr, max_out)
torch.cuda.synchronize()
t = timeit(lambda: K.nms(bt, cnt, thr, max_out), 20)
print('nms: %.1f us, kept %s' % (t * 1e3, kc.cpu().tolist()))
