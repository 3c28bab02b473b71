"""cProfile of the host side of the train step (which Python / ctypes frames the enqueue time goes to).
usage: python scripts/host_profile.py [workload] [dtype] [steps]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from luminoth_amd.utils import training as T

wl_name = sys.argv[1] if len(sys.argv) > 1 else 'frcnn_r50_coco'
dtype = sys.argv[2] if len(sys.argv) > 2 else 'f16'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = torch.device('cuda:0')
wl = dict(bench.WORKLOADS[wl_name])
cfg, model = bench.build(wl, dev, dtype)
opt = T.get_optimizer(cfg.train, model)
batches = [bench.inputs(wl, 100, dev), bench.inputs(wl, 1000, dev)]
counter = [0]


def step():
    i = counter[0]
    counter[0] += 1
    cur, nxt = batches[i % 2], batches[(i + 1) % 2]
    return T.train_step(model, opt, cur[0], cur[1], next_image=nxt[0], next_gt=nxt[1])


for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('enqueue %.3f ms/step, with final sync %.3f ms/step' % (t_enq / steps * 1e3, t_all / steps * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(32)
