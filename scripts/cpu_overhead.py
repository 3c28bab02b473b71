"""Host-side enqueue time per train step vs GPU time (are we launch-bound?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cProfile, pstats
from bench import synth_batch
from luminoth_amd.models import get_model
from luminoth_amd.utils.config import get_config
from luminoth_amd.utils.training import get_optimizer, train_step
cfg = get_config({'model': {'type': 'fasterrcnn', 'network': {'num_classes': 80},
                            'base_network': {'architecture': 'resnet_v1_50'}}, 'train': {'seed': 0, 'debug': False}})
model = get_model('fasterrcnn')(cfg, device='cuda:0')
sd = model.state_dict()
sd['truncated_base_network/resnet_v1_50/conv1/BatchNorm/moving_variance'].fill_(73.6 ** 2 * 2)
for k in sd:
    if k.endswith('conv3/BatchNorm/moving_variance'):
        sd[k].fill_(16.0)
model.load_state_dict(sd)
opt = get_optimizer(cfg.train, model)
images, gts = synth_batch(2, 1024, 1024, 8, 80, 100, 'cuda:0')
for _ in range(5):
    train_step(model, opt, images, gts)
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    train_step(model, opt, images, gts)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue %.2f ms/step, total %.2f ms/step' % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
# host-only cost: time one step's enqueue right after a sync (GPU idle, nothing blocks)
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    a = time.perf_counter(); train_step(model, opt, images, gts); ts.append(time.perf_counter() - a)
print('host enqueue from idle: %s ms' % ['%.2f' % (t * 1e3) for t in ts])
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    train_step(model, opt, images, gts)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(22)
