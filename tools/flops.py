"""Algorithmic MACs / FLOPs of the train step, re-derived from the layer shapes ("the builder must
re-derive them in a checked-in tools/flops.py and assert these totals").

Counting rule of §8d: convolution / FC multiply-accumulates only (1 MAC = 2 FLOP); BatchNorm, activations, pooling,
losses and post-processing count 0.  Output sizes follow the layers' own padding rules (`conv_desc`: TF SAME /
conv2d_same / VALID).  Train step = forward + 2 x the TRAINABLE part (data and weight gradients).

    python tools/flops.py            # prints the table and asserts SURVEY.md's totals
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from luminoth_amd import kernels as K                                   # noqa: E402  (conv_desc: host arithmetic only)
from luminoth_amd.models.base import networks                           # noqa: E402


def conv_macs(layer, h, w):
    d = K.conv_desc((1, h, w, layer.cin), (layer.k, layer.k, layer.cin, layer.cout), layer.stride, layer.rate,
                    layer.padding, layer.act)
    return d.OH * d.OW * layer.cout * layer.k * layer.k * layer.cin, (d.OH, d.OW)


def trunk_macs(nodes, h, w, frozen_nodes=0):
    """-> (all MACs, MACs of the first `frozen_nodes` nodes, output (h, w)).  A bottleneck's shortcut and conv1 read the
    unit input; conv2 carries the unit's stride."""
    total = frozen = 0
    for i, n in enumerate(nodes):
        m = 0
        if hasattr(n, 'conv1'):                         # BottleneckNode
            if n.shortcut is not None:
                m += conv_macs(n.shortcut, h, w)[0]
            a, hw1 = conv_macs(n.conv1, h, w)
            b, hw2 = conv_macs(n.conv2, *hw1)
            c, _ = conv_macs(n.conv3, *hw2)
            m += a + b + c
        elif n.layers:                                  # ConvNode
            m += conv_macs(n.layers[0], h, w)[0]
        h, w = n.out_hw(h, w)
        total += m
        if i < frozen_nodes:
            frozen += m
    return total, frozen, (h, w)


def resnet_frcnn(arch, H, W, num_classes, rois, anchors_per_point=12, tail=False):
    nodes, endpoints = networks.resnet_v1_nodes(arch, 'truncated_base_network', 0.0, None, up_to_block=3)
    n_frozen = endpoints['block1'] + 1                  # fine_tune_from: block2 -> conv1, pool, block1 frozen
    trunk, frozen, (fh, fw) = trunk_macs(nodes, H, W, n_frozen)
    rpn = fh * fw * (1024 * 512 * 9 + 512 * anchors_per_point * 6)
    feat = 1024
    tail_m = 0
    if tail:                                            # ResNet-101: block4 on the 7x7 pooled ROIs (truncated_base_network.py:56-95)
        t, _, _ = trunk_macs(networks.resnet_v1_tail_nodes(arch, 'truncated_base_network', 0.0, None), 7, 7)
        tail_m, feat = rois * t, 2048
    fc = rois * feat * ((num_classes + 1) + 4 * num_classes)
    fwd = trunk + rpn + tail_m + fc
    trainable = fwd - frozen
    return dict(trunk=trunk, frozen=frozen, rpn=rpn, tail=tail_m, fc=fc, fwd=fwd, train=fwd + 2 * trainable,
                feature_hw=(fh, fw))


def vgg_frcnn(H, W):
    nodes, _ = networks.vgg16_nodes('truncated_base_network', 'vgg_16', 0.0, None, None)
    trunk, _, (fh, fw) = trunk_macs(nodes, H, W)
    rpn = fh * fw * (512 * 512 * 9 + 512 * 72)
    return dict(trunk=trunk, rpn=rpn, feature_hw=(fh, fw))


G = 1e9
EXPECT = {      # SURVEY.md §8d, GMAC
    'r50_1024': dict(trunk=60.05, frozen=13.80, rpn=19.48, fc=0.105, fwd=79.63),
    'r101_1024': dict(trunk=137.62, rpn=19.48, tail=187.44, fc=0.21, fwd=344.75),
    'r50_800x1333': dict(trunk=61.38, rpn=19.97, fc=0.105, fwd=81.46),
    'vgg_600x800': dict(trunk=146.63, rpn=4.43),
}


def table():
    return {
        'r50_1024': resnet_frcnn('resnet_v1_50', 1024, 1024, 80, 256),
        'r101_1024': resnet_frcnn('resnet_v1_101', 1024, 1024, 80, 256, tail=True),
        'r50_800x1333': resnet_frcnn('resnet_v1_50', 800, 1333, 80, 256),
        'vgg_600x800': vgg_frcnn(600, 800),
    }


def check(tab=None, rel=5e-3):
    tab = tab or table()
    for name, exp in EXPECT.items():
        for k, v in exp.items():
            got = tab[name][k] / G
            assert abs(got - v) <= rel * max(v, 1.0) + 6e-3, (name, k, got, v)
    r50 = tab['r50_1024']
    assert abs(2 * r50['train'] / G - 422.6) < 1.0, 2 * r50['train'] / G           # GFLOP per image per train step
    assert abs(2 * tab['r101_1024']['train'] / G - 2013) < 8, 2 * tab['r101_1024']['train'] / G
    return tab


if __name__ == '__main__':
    t = check()
    for name, r in t.items():
        print(name, {k: (round(v / G, 3) if not isinstance(v, tuple) else v) for k, v in r.items()})
    print('R50 @1024^2: %.1f GFLOP forward, %.1f GFLOP per image per train step' %
          (2 * t['r50_1024']['fwd'] / G, 2 * t['r50_1024']['train'] / G))
