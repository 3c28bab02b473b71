"""Layer-by-layer comparison of the mixed-precision trunk (HIP) with the rounded-operand oracle."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch

from e2e_util import condition_like_pretrained, make_config, synth
from luminoth_amd.models import get_model
from luminoth_amd.models.base import layers as L
from oracle.model import OracleFasterRCNN

compute = sys.argv[1] if len(sys.argv) > 1 else 'f16'
cfg = make_config('resnet_v1_50', 20, **{'model.base_network.compute_dtype': compute})
model = condition_like_pretrained(get_model('fasterrcnn')(cfg), 'resnet_v1_50')
images, gts = synth(1, 320, 384, 4, 20, 3)
L.ACT_TAP = {}
with torch.no_grad():
    pred = model(images, gts, is_training=True)
tap = {k: v.cpu() for k, v in L.ACT_TAP.items()}
L.ACT_TAP = None
oracle = OracleFasterRCNN(model.state_dict(), num_classes=20, seed=0, compute=compute)
orig = oracle._activate


def spy(z, act, scope):
    y = orig(z, act, scope)
    if scope in tap:
        t = tap[scope].reshape(y.shape)
        print('%-70s %.2e' % (scope[-70:], float((t - y).abs().max() / (y.abs().max() + 1e-30))))
    return y


oracle._activate = spy
with torch.no_grad():
    oracle.backbone(images[0][None])
