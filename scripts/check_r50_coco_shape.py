"""ResNet-50 Faster R-CNN at the COCO shape 800x1344 (BASELINE configs[4] geometry), fp32: ms/step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
images, gts = synth_batch(2, 800, 1344, 8, 80, 100, 'cuda:0')
gts = (gts[0] * torch.tensor([1344 / 1024., 800 / 1024., 1344 / 1024., 800 / 1024., 1.], device='cuda:0'), gts[1])
for i in range(5):
    total, _ = train_step(model, opt, images, gts)
torch.cuda.synchronize(); t0 = time.time()
for i in range(10):
    total, _ = train_step(model, opt, images, gts)
torch.cuda.synchronize()
print('WINOGRAD=%s loss %.4f  %.2f ms/step' % (os.environ.get('LUMINOTH_AMD_WINOGRAD', '1'), float(total), (time.time() - t0) / 10 * 1e3))
