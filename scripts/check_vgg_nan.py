"""VGG-16 Faster R-CNN smoke: per-step losses (random init, raw pixels: does it diverge with or without Winograd?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from luminoth_amd.models import get_model
from luminoth_amd.utils.config import get_config
from luminoth_amd.utils.training import get_optimizer, train_step
cfg = get_config({'model': {'type': 'fasterrcnn', 'network': {'num_classes': 20},
                            'base_network': {'architecture': 'vgg_16', 'fine_tune_from': 'conv4/conv4_1'}},
                  'train': {'seed': 0, 'debug': False}})
model = get_model('fasterrcnn')(cfg, device='cuda:0')
opt = get_optimizer(cfg.train, model)
images, gts = synth_batch(2, 512, 512, 8, 20, 100, 'cuda:0')
gts = (gts[0] * torch.tensor([.5, .5, .5, .5, 1.], device='cuda:0'), gts[1])
losses = []
for i in range(8):
    total, _ = train_step(model, opt, images, gts)
    losses.append(float(total))
torch.cuda.synchronize(); t0 = time.time()
for i in range(5):
    train_step(model, opt, images, gts)
torch.cuda.synchronize()
print('WINOGRAD=%s losses %s  %.2f ms/step' % (os.environ.get('LUMINOTH_AMD_WINOGRAD', '1'), ['%.4g' % l for l in losses], (time.time() - t0) / 5 * 1e3))
