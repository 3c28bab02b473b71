"""Architecture descriptions of the slim networks the reference wraps
(models/base/base_network.py:20-29 VALID_ARCHITECTURES), as node lists for
luminoth_amd.models.base.layers.Trunk.

slim resnet_v1 (TF 1.x tf.contrib.slim.nets.resnet_v1 / resnet_utils; third
party, not in /root/reference — structure per SURVEY.md §8a-A2):
  conv1 7x7/2 conv2d_same -> BN -> ReLU -> 3x3/2 SAME max-pool -> blocks of
  bottleneck units; the stride-2 unit is the LAST unit of a block; with
  output_stride=16 the stride of block3's last unit is absorbed into the
  atrous rate of block4 (stack_blocks_dense).
slim vgg_16: 3x3 SAME conv + bias + ReLU, 2x2/2 VALID max-pools.
"""
from .layers import BottleneckNode, ConvLayer, ConvNode, MaxPoolNode, PreactBottleneckNode, PreactLayer

RESNET_UNITS = {
    'resnet_v1_50': (3, 4, 6, 3),
    'resnet_v1_101': (3, 4, 23, 3),
    'resnet_v1_152': (3, 8, 36, 3),
    'resnet_v2_50': (3, 4, 6, 3),
    'resnet_v2_101': (3, 4, 23, 3),
    'resnet_v2_152': (3, 8, 36, 3),
}
_RGB_MEANS = (123.68, 116.78, 103.94)  # base_network.py:14-16


def resnet_v1_nodes(arch, scope, wd, init, output_stride=16, in_sub=None, up_to_block=4):
    """Returns (nodes, endpoints) where endpoints maps 'block<i>' -> index of
    the last node of that block."""
    units = RESNET_UNITS[arch]
    p = '%s/%s' % (scope, arch)
    nodes, endpoints = [], {}
    conv1 = ConvLayer(p + '/conv1', 3, 64, 7, stride=2, padding='SAME_EXPLICIT', act='relu', wd=wd, init=init)
    nodes.append(ConvNode(conv1, in_sub=in_sub))
    nodes.append(MaxPoolNode(3, 2, 'SAME'))
    current_stride, rate, cin = 4, 1, 64
    for bi, (base_depth, n_units) in enumerate(zip((64, 128, 256, 512), units)):
        if bi + 1 > up_to_block:
            break
        block_stride = 2 if bi < 3 else 1
        for u in range(n_units):
            unit_stride = block_stride if u == n_units - 1 else 1
            if output_stride is not None and current_stride == output_stride:
                stride, unit_rate = 1, rate
                rate *= unit_stride
            else:
                stride, unit_rate = unit_stride, 1
                current_stride *= unit_stride
            nodes.append(BottleneckNode('%s/block%d/unit_%d' % (p, bi + 1, u + 1), cin, base_depth * 4,
                                        base_depth, stride, unit_rate, wd, init))
            cin = base_depth * 4
        endpoints['block%d' % (bi + 1)] = len(nodes) - 1
    return nodes, endpoints


def resnet_v2_nodes(arch, scope, wd, init, bias_init, output_stride=16, in_sub=None, up_to_block=4):
    """slim resnet_v2 (pre-activation): conv1 7x7/2 conv2d_same with a BIAS and neither BatchNorm nor activation
    (arg_scope activation_fn=None, normalizer_fn=None) -> 3x3/2 SAME max-pool -> blocks of pre-activation bottlenecks (the
    stride-2 unit is the last of a block; output_stride as in v1) -> `postnorm` BatchNorm + ReLU (after block4 only: not
    on the path of the `block3` endpoint, returned as a layer to REGISTER so that its variables exist).
    Returns (nodes, endpoints, unused_layers)."""
    units = RESNET_UNITS[arch]
    p = '%s/%s' % (scope, arch)
    nodes, endpoints = [], {}
    conv1 = ConvLayer(p + '/conv1', 3, 64, 7, stride=2, padding='SAME_EXPLICIT', act=None, norm='bias', wd=wd, init=init)
    conv1.bias_init = bias_init
    nodes.append(ConvNode(conv1, in_sub=in_sub))
    nodes.append(MaxPoolNode(3, 2, 'SAME'))
    current_stride, rate, cin = 4, 1, 64
    for bi, (base_depth, n_units) in enumerate(zip((64, 128, 256, 512), units)):
        if bi + 1 > up_to_block:
            break
        block_stride = 2 if bi < 3 else 1
        for u in range(n_units):
            unit_stride = block_stride if u == n_units - 1 else 1
            if output_stride is not None and current_stride == output_stride:
                stride, unit_rate = 1, rate
                rate *= unit_stride
            else:
                stride, unit_rate = unit_stride, 1
                current_stride *= unit_stride
            nodes.append(PreactBottleneckNode('%s/block%d/unit_%d' % (p, bi + 1, u + 1), cin, base_depth * 4,
                                              base_depth, stride, unit_rate, wd, init))
            cin = base_depth * 4
        endpoints['block%d' % (bi + 1)] = len(nodes) - 1
    return nodes, endpoints, [PreactLayer(p + '/postnorm', cin)]


def resnet_v1_tail_nodes(arch, scope, wd, init):
    """block4 as `_build_tail` applies it to pooled ROIs
    (truncated_base_network.py:56-95): 3 units, depth 2048/512, stride 1, rate 1,
    REUSING the trunk's block4 variables."""
    p = '%s/%s' % (scope, arch)
    nodes, cin = [], 1024
    for u in range(3):
        nodes.append(BottleneckNode('%s/block4/unit_%d' % (p, u + 1), cin, 2048, 512, 1, 1, wd, init))
        cin = 2048
    return nodes


VGG16_CFG = [('conv1', 2, 64), ('conv2', 2, 128), ('conv3', 3, 256), ('conv4', 3, 512), ('conv5', 3, 512)]


def vgg16_unused_fc_layers(scope, arch, wd, init, num_classes=1000):
    """slim vgg_16 builds its classifier even when only conv5_3 is consumed (base_network.py:70-75 calls
    vgg.vgg_16(inputs, is_training, spatial_squeeze) with the default num_classes=1000): fc6 7x7 VALID 4096, fc7 1x1 4096,
    fc8 1x1 1000.  They are never evaluated for the endpoint, but their weights exist (slim checkpoints carry them)
    and, created under vgg_arg_scope's weights_regularizer, they are part of regularization_loss (SURVEY.md appendix
    B.12).  Returned as layers to REGISTER only."""
    p = '%s/%s' % (scope, arch)
    out = []
    for name, k, cin, cout in (('fc6', 7, 512, 4096), ('fc7', 1, 4096, 4096), ('fc8', 1, 4096, num_classes)):
        out.append(ConvLayer('%s/%s' % (p, name), cin, cout, k, padding='VALID', act='relu' if name != 'fc8' else None,
                             norm='bias', wd=wd, init=init))
    return out


def vgg16_nodes(scope, arch, wd, init, bias_init, in_sub=None, pool5=False):
    p = '%s/%s' % (scope, arch)
    nodes, endpoints, cin = [], {}, 3
    for bi, (name, reps, depth) in enumerate(VGG16_CFG):
        for r in range(reps):
            lname = '%s/%s/%s_%d' % (p, name, name, r + 1)
            layer = ConvLayer(lname, cin, depth, 3, act='relu', norm='bias', wd=wd, init=init)
            layer.bias_init = bias_init
            nodes.append(ConvNode(layer, in_sub=in_sub if cin == 3 else None))
            endpoints['%s/%s_%d' % (name, name, r + 1)] = len(nodes) - 1
            cin = depth
        if bi < 4 or pool5:
            nodes.append(MaxPoolNode(2, 2, 'VALID'))
            endpoints['pool%d' % (bi + 1)] = len(nodes) - 1
    return nodes, endpoints
