import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from luminoth_amd.models import get_model
from luminoth_amd.utils.config import get_config
from oracle.model import OracleFasterRCNN
cfg = get_config({'model': {'type': 'fasterrcnn', 'network': {'num_classes': 80}, 'base_network': {'architecture': 'resnet_v1_50'}}, 'train': {'seed': 0}})
model = get_model('fasterrcnn')(cfg, device='cuda:0')
sd = model.state_dict()
sd['truncated_base_network/resnet_v1_50/conv1/BatchNorm/moving_variance'].fill_(73.6 ** 2 * 2)
print('cpu_count', os.cpu_count(), flush=True)
S = 512
images, (gt, _) = synth_batch(1, S, S, 8, 80, 1234, 'cpu')
gt = gt.clone(); gt[..., :4] *= S / 1024.0
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    o = OracleFasterRCNN(sd, num_classes=80, seed=0, arch='resnet_v1_50')
    t0 = time.time(); o.train_step([images[0]], [gt[0].numpy()]); print('threads %d: %.1f s' % (th, time.time() - t0), flush=True)
