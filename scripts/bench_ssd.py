"""SSD VGG-300 train step (BASELINE configs[2]: batch 32, 300x300 synthetic, 20 classes, 4 gt/image)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from luminoth_amd.models import get_model
from luminoth_amd.utils.config import get_config
from luminoth_amd.utils.training import get_optimizer, train_step
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = get_config({'model': {'type': 'ssd', 'network': {'num_classes': 20}}, 'train': {'seed': 0, 'debug': False}})
model = get_model('ssd')(cfg, device='cuda:0')
opt = get_optimizer(cfg.train, model)
g = torch.Generator().manual_seed(0)
images = (torch.rand((B, 300, 300, 3), generator=g) * 2 - 1).cuda()      # O(1) inputs keep the random-init net finite
gt = torch.zeros((B, 4, 5))
for b in range(B):
    wh = torch.randint(30, 201, (4, 2), generator=g)
    xy = (torch.rand((4, 2), generator=g) * (300 - wh).float()).floor()
    gt[b, :, :2], gt[b, :, 2:4], gt[b, :, 4] = xy, xy + wh - 1, torch.randint(0, 20, (4,), generator=g).float()
gts = (gt.cuda(), torch.full((B,), 4, dtype=torch.int32).cuda())
for _ in range(3):
    total, _ = train_step(model, opt, images, gts)
torch.cuda.synchronize(); t0 = time.time()
N = 10
for _ in range(N):
    total, _ = train_step(model, opt, images, gts)
torch.cuda.synchronize(); dt = (time.time() - t0) / N
print('SSD-300 B=%d: loss %.4f  %.2f ms/step  %.1f images/s  (fwd+bwd ~ 182 GFLOP/image -> %.1f TF/s)' %
      (B, float(total), dt * 1e3, B / dt, 182e9 * B / dt / 1e12))
