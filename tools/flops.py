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


def trunk_elems(nodes, h, w, frozen_nodes=0):
    """Activation elements the trunk WRITES per image: every convolution / pooling output (a bottleneck: shortcut, conv1,
    conv2, conv3).  -> (all, those of the first `frozen_nodes` nodes, output (h, w))."""
    total = frozen = 0
    for i, n in enumerate(nodes):
        e = 0
        if hasattr(n, 'conv1'):
            _, hw1 = conv_macs(n.conv1, h, w)
            _, hw2 = conv_macs(n.conv2, *hw1)
            e += hw1[0] * hw1[1] * n.conv1.cout + hw2[0] * hw2[1] * (n.conv2.cout + n.conv3.cout)
            if n.shortcut is not None:
                e += hw2[0] * hw2[1] * n.shortcut.cout
        elif n.layers:
            _, hw1 = conv_macs(n.layers[0], h, w)
            e += hw1[0] * hw1[1] * n.layers[0].cout
        else:                                           # pooling node
            oh, ow = n.out_hw(h, w)
            e += oh * ow * getattr(n, 'channels', 64)
        h, w = n.out_hw(h, w)
        total += e
        if i < frozen_nodes:
            frozen += e
    return total, frozen, (h, w)


def resnet_frcnn_step_bytes(arch, H, W, batch, trainable_params, act_bytes=4):
    """ALGORITHMIC (compulsory) HBM bytes of one Faster R-CNN train step, SURVEY.md §8(d)'s rule: every backbone / RPN
    activation written once in the forward pass and read once in the backward pass, the same again for the activation
    gradients of the trainable part, the input image read once, and the optimizer's 20 B per trainable parameter (read w, g,
    v; write w, v).  act_bytes = 2 for the half-storage trunk.  Post-processing stages are < 1 % and not counted.
    (§8(d) quotes 157 M elements and ~2.2 GB per image for ResNet-50 at 1024^2; asserted in check().)"""
    nodes, endpoints = networks.resnet_v1_nodes(arch, 'truncated_base_network', 0.0, None, up_to_block=3)
    n_frozen = endpoints['block1'] + 1
    elems, frozen, (fh, fw) = trunk_elems(nodes, H, W, n_frozen)
    rpn = fh * fw * (512 + 12 * 6)
    per_image = (2 * (elems + rpn) + 2 * (elems - frozen + rpn)) * act_bytes + H * W * 3 * 4
    return dict(elements_per_image=elems + rpn, frozen_elements=frozen, per_image=per_image,
                optimizer=20 * trainable_params, step=batch * per_image + 20 * trainable_params)


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
    b = resnet_frcnn_step_bytes('resnet_v1_50', 1024, 1024, 2, 13.5e6)
    # §8(d): 157 M activation elements, ~2.2 GB per image, 270 MB of optimizer traffic -> ~4.7 GB per step at batch 2
    # (the count here includes the projection shortcuts and the stem's pooling output, 193 M elements: the survey's 157 M
    # leaves them out; its rounded per-image and per-step figures hold within 6 %)
    assert 150 < b['elements_per_image'] / 1e6 < 200, b['elements_per_image']
    assert abs(b['per_image'] / G - 2.2) < 0.2 and abs(b['step'] / G - 4.7) < 0.4, b
    return tab


if __name__ == '__main__':
    t = check()
    for name, r in t.items():
        print(name, {k: (round(v / G, 3) if not isinstance(v, tuple) else v) for k, v in r.items()})
    print('R50 @1024^2: %.1f GFLOP forward, %.1f GFLOP per image per train step' %
          (2 * t['r50_1024']['fwd'] / G, 2 * t['r50_1024']['train'] / G))
