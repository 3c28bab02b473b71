"""CPU checks of the end-to-end oracle (oracle/model.py) for every backbone the GPU parity tests use: the oracle's
trainable-variable set equals the model's (reference: base_network.py:211-241, truncated_base_network.py:97-144),
the fp64 mode agrees with fp32, and pinning the ReLU branches to the oracle's OWN activations changes nothing."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from e2e_util import make_config, synth   # noqa: E402
from oracle import rng as orng   # noqa: E402
from oracle.model import OracleFasterRCNN   # noqa: E402


def _model(arch, num_classes, **over):
    from luminoth_amd.models import get_model
    return get_model('fasterrcnn')(make_config(arch, num_classes, **over), device='cpu')


@pytest.mark.parametrize('arch,over,okw', [
    ('resnet_v1_50', {}, {}),
    ('resnet_v1_101', {}, {}),
    ('vgg_16', {'model.base_network.fine_tune_from': 'conv3'}, {'fine_tune_from': 'conv3'}),
    ('vgg_16', {'model.base_network.fine_tune_from': None}, {'fine_tune_from': None}),
])
def test_oracle_trainable_set_equals_model(arch, over, okw):
    model = _model(arch, 20, **over)
    oracle = OracleFasterRCNN(model.state_dict(), arch=arch, num_classes=20, seed=0, **okw)
    assert sorted(oracle.trainable_names()) == sorted(model.get_trainable_vars().keys())


@pytest.mark.parametrize('arch,okw', [('resnet_v1_50', {}), ('vgg_16', {'fine_tune_from': 'conv3'})])
def test_oracle_fp64_and_pinned_masks(arch, okw):
    over = {'model.base_network.fine_tune_from': 'conv3'} if arch == 'vgg_16' else {}
    model = _model(arch, 20, **over)
    sd = model.state_dict()
    if arch == 'vgg_16':
        sd['truncated_base_network/vgg_16/conv1/conv1_1/weights'].mul_(1.0 / 73.6)
    else:
        sd['truncated_base_network/%s/conv1/BatchNorm/moving_variance' % arch].fill_(73.6 ** 2 * 2)
    images, gts = synth(1, 96, 128, 2, 20, 11)
    seed = orng.image_seed(0, 0, 0)
    o32 = OracleFasterRCNN(sd, arch=arch, num_classes=20, seed=0, **okw)
    a = o32.forward_image(images[0], gts[0], seed)
    o64 = OracleFasterRCNN(sd, arch=arch, num_classes=20, seed=0, dtype=torch.float64, **okw)
    ov = dict(rois=a['rois'], roi_labels=a['roi_labels'], roi_targets=a['roi_targets'])
    b = o64.forward_image(images[0], gts[0], seed, overrides=ov)
    assert b['feat'].dtype == torch.float64
    for k in ('rpn_cls_loss', 'rpn_reg_loss', 'rcnn_cls_loss', 'rcnn_reg_loss'):
        assert abs(float(a[k]) - float(b[k])) <= 1e-5 * max(1.0, abs(float(b[k]))), k
    # masks taken from the oracle's own layer outputs reproduce its forward exactly
    taps = {}
    orig = o32._activate

    def tapping(z, act, scope):
        y = orig(z, act, scope)
        taps[scope] = y.detach()
        return y
    o32._activate = tapping
    a2 = o32.forward_image(images[0], gts[0], seed, overrides=ov)
    o32._activate = orig
    assert len(taps) > 10
    o32.masks = taps
    a3 = o32.forward_image(images[0], gts[0], seed, overrides=ov)
    np.testing.assert_array_equal(a2['feat'].detach().numpy(), a3['feat'].detach().numpy())
    np.testing.assert_array_equal(a2['rcnn_cls_score'].detach().numpy(), a3['rcnn_cls_score'].detach().numpy())


@pytest.mark.parametrize('storage,loss_tol,cos_tol', [('f16', 5e-3, 0.999), ('bf16', 4e-2, 0.99)])
def test_oracle_half_storage_mode_tracks_the_fp32_oracle(storage, loss_tol, cos_tol):
    """oracle/model.py `storage=` (the restatement of the half-storage trunk, csrc/conv_hs.h): same variables and trainable
    set as the fp32 oracle, losses within the rounding of 16-bit activations, trunk gradients pointing the same way — and the
    model side of the same switch (`model.base_network.storage_dtype`) selects the same 42 trunk layers + the RPN convolution."""
    model = _model('resnet_v1_50', 20, **{'model.base_network.storage_dtype': storage})
    assert model.base_network.storage_dtype == storage and len(model.base_network._hs_layers) == 42
    assert model._rpn._rpn.storage == storage and model.base_network.extra_hs_layers == [model._rpn._rpn]
    sd = model.state_dict()
    sd['truncated_base_network/resnet_v1_50/conv1/BatchNorm/moving_variance'].fill_(73.6 ** 2 * 2)
    images, gts = synth(1, 96, 128, 2, 20, 11)
    seed = orng.image_seed(0, 0, 0)
    o32 = OracleFasterRCNN(sd, arch='resnet_v1_50', num_classes=20, seed=0)
    oh = OracleFasterRCNN(sd, arch='resnet_v1_50', num_classes=20, seed=0, storage=storage)
    assert oh.compute == storage and sorted(oh.trainable_names()) == sorted(o32.trainable_names())
    name = 'truncated_base_network/resnet_v1_50/block3/unit_2/bottleneck_v1/conv2/weights'
    for o in (o32, oh):
        o.v[name].requires_grad_(True)
    a = o32.forward_image(images[0], gts[0], seed)
    ov = dict(rois=a['rois'], roi_labels=a['roi_labels'], roi_targets=a['roi_targets'], proposals=a.get('proposals'))
    ov = {k: v for k, v in ov.items() if v is not None}
    b = oh.forward_image(images[0], gts[0], seed, overrides=ov)
    assert b['feat'].dtype == torch.float32
    for k in ('rpn_cls_loss', 'rpn_reg_loss', 'rcnn_cls_loss', 'rcnn_reg_loss'):
        assert abs(float(a[k]) - float(b[k])) <= loss_tol * max(1.0, abs(float(a[k]))), (k, float(a[k]), float(b[k]))
    ga, = torch.autograd.grad(a['rpn_cls_loss'] + a['rcnn_cls_loss'], [o32.v[name]])
    gb, = torch.autograd.grad(b['rpn_cls_loss'] + b['rcnn_cls_loss'], [oh.v[name]])
    cos = float((ga * gb).sum() / (ga.norm() * gb.norm()))
    assert cos >= cos_tol, cos
