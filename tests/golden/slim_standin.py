"""Stand-in for the slice of `tf.contrib.slim` (+ `slim.nets.resnet_v1` / `resnet_v2.resnet_utils` / `resnet_utils`) that
the reference's base network reaches (luminoth/models/base/base_network.py:56-101,143-151 and
truncated_base_network.py:39-95).  Test infrastructure only: tests/golden/make_golden_ref_tf.py installs it next to
tests/golden/tf_numpy_shim.py to EXECUTE the reference's top-level composition — `FasterRCNN.__init__ / _build / loss /
get_trainable_vars` and `BaseNetwork._get_base_network_vars / get_trainable_vars / get_base_network_checkpoint_vars` — in the
build container, where TensorFlow (and with it slim, a third-party dependency that is not under /root/reference) is absent.

What it does: `resnet_v1_50 / _101 / _152(inputs, ...)` CREATES the variables slim's resnet_v1 creates — the same names,
creation order, shapes, trainability and collections — under the current variable scope, registers the weight decay the
active `arg_scope` carries on every convolution's `weights` (slim's `resnet_arg_scope`: `weights_regularizer =
l2_regularizer(weight_decay)` on `slim.conv2d`; BatchNorm parameters are not regularised), and returns `(net, end_points)`
whose entry `<scope>/block3` is the feature map handed to `set_feature_map` (no convolution is computed: the arithmetic of
the slim trunk is the oracle's `port` parity; what this pins is WHICH variables exist, train, are regularised
and are mapped to checkpoint names).

What slim builds, restated from the published `tf.contrib.slim.nets.resnet_v1` / `resnet_utils` (TF 1.x):
  * root: `conv1` = conv2d_same 7x7/2, 64 filters (`conv1/weights` + `conv1/BatchNorm/{beta,gamma,moving_mean,
    moving_variance}`), then a 3x3/2 max-pool (no variables);
  * blocks `block1..block4` of `unit_<i>/bottleneck_v1` units: (depth, bottleneck depth, units) = (256, 64, 3), (512, 128, 4),
    (1024, 256, 6 | 23 | 36), (2048, 512, 3); a unit creates, in this order, `shortcut` (a 1x1 convolution — only when the
    unit's input depth differs from its depth, i.e. in unit_1 of every block), `conv1` 1x1, `conv2` 3x3, `conv3` 1x1;
  * every convolution is `weights` (HWIO, trainable, regularised) followed by its BatchNorm: `beta`, `gamma` (trainable:
    resnet_arg_scope sets `scale=True`), `moving_mean`, `moving_variance` (not trainable); slim registers ALL of them as
    model variables (`tf.GraphKeys.MODEL_VARIABLES`);
  * `num_classes=None` (base_network.py:91): no logits layer; `global_pool=False`: no pooling.
Variable values are a fixed function of the full variable name (`tf_numpy_shim.seeded_variable`), so a replaying test
rebuilds them without storing them.
"""
import contextlib
import sys
import types

import numpy as np

import tf_numpy_shim as tf

_ARG_SCOPE = [{}]
_FEATURE_MAP = [None]
DEFAULT_WEIGHT_DECAY = 0.0001          # slim resnet_arg_scope's default


def set_feature_map(fmap):
    """The (1, fh, fw, 1024) array every later resnet_v1_* call returns as its block3 endpoint."""
    _FEATURE_MAP[0] = fmap


def resnet_arg_scope(weight_decay=DEFAULT_WEIGHT_DECAY, batch_norm_decay=0.997, batch_norm_epsilon=1e-5,
                     batch_norm_scale=True, **kwargs):
    return dict(weight_decay=weight_decay, batch_norm_decay=batch_norm_decay, batch_norm_epsilon=batch_norm_epsilon,
                batch_norm_scale=batch_norm_scale)


@contextlib.contextmanager
def arg_scope(list_ops_or_scope, **kwargs):
    """`slim.arg_scope(scope_dict)` re-enters a scope made by resnet_arg_scope; `slim.arg_scope([ops], **kw)` adds keyword
    defaults for those ops (truncated_base_network.py:79-81: `is_training` of batch_norm — no variable depends on it)."""
    new = dict(_ARG_SCOPE[-1])
    if isinstance(list_ops_or_scope, dict):
        new.update(list_ops_or_scope)
    _ARG_SCOPE.append(new)
    try:
        yield new
    finally:
        _ARG_SCOPE.pop()


def batch_norm(*a, **k):
    raise NotImplementedError('slim stand-in: batch_norm is only referenced as an arg_scope target')


def _conv(scope, kh, cin, cout):
    """slim.conv2d(..., normalizer_fn=batch_norm, scope=scope): `weights` then the BatchNorm variables."""
    wd = _ARG_SCOPE[-1].get('weight_decay', DEFAULT_WEIGHT_DECAY)
    with tf.variable_scope(scope):
        full = tf.get_variable_scope().name
        tf.create_variable('weights', lambda: tf.seeded_variable(full + '/weights', 'w', (kh, kh, cin, cout)),
                           trainable=True, model_variable=True, regularizer=tf.l2_regularizer(wd) if wd else None)
        with tf.variable_scope('BatchNorm'):
            bn = tf.get_variable_scope().name
            tf.create_variable('beta', lambda: tf.seeded_variable(bn + '/beta', 'b', (cout,)), True, True)
            tf.create_variable('gamma', lambda: 1.0 + tf.seeded_variable(bn + '/gamma', 'b', (cout,)), True, True)
            tf.create_variable('moving_mean', lambda: tf.seeded_variable(bn + '/moving_mean', 'b', (cout,)), False, True)
            tf.create_variable('moving_variance',
                               lambda: 1.0 + np.abs(tf.seeded_variable(bn + '/moving_variance', 'b', (cout,))), False, True)


def variable_value(name, shape):
    """The value the stand-in (slim variables) or the shim's Sonnet layers (`.../<module>/{w,b}`) gave the variable `name`:
    what a replaying test loads into the oracle / the product before comparing with a top-level fixture."""
    shape = tuple(int(v) for v in shape if int(v) > 0)          # (fixture shapes are padded to four entries with zeros)
    leaf = name.split('/')[-1]
    if leaf in ('w', 'b'):
        return tf.seeded_variable(name.split('/')[-2], leaf, shape)
    if leaf == 'weights':
        return tf.seeded_variable(name, 'w', shape)
    v = tf.seeded_variable(name, 'b', shape)
    if leaf == 'gamma':
        return (1.0 + v).astype(np.float32)
    if leaf == 'moving_variance':
        return (1.0 + np.abs(v)).astype(np.float32)
    return v


BLOCKS = {'resnet_v1_50': (3, 4, 6, 3), 'resnet_v1_101': (3, 4, 23, 3), 'resnet_v1_152': (3, 8, 36, 3)}


def _bottleneck_unit(depth_in, depth, depth_bottleneck):
    with tf.variable_scope('bottleneck_v1'):
        if depth != depth_in:
            _conv('shortcut', 1, depth_in, depth)
        _conv('conv1', 1, depth_in, depth_bottleneck)
        _conv('conv2', 3, depth_bottleneck, depth_bottleneck)
        _conv('conv3', 1, depth_bottleneck, depth)


def _stack_block(name, depth_in, depth, depth_bottleneck, units):
    with tf.variable_scope(name):
        for i in range(units):
            with tf.variable_scope('unit_%d' % (i + 1)):
                _bottleneck_unit(depth_in if i == 0 else depth, depth, depth_bottleneck)
    return depth


def _resnet_v1(arch):
    units = BLOCKS[arch]

    def net(inputs, num_classes=None, is_training=True, global_pool=True, output_stride=None, spatial_squeeze=True,
            reuse=None, scope=arch):
        assert num_classes is None and not global_pool, 'the reference builds the trunk without logits / pooling'
        with tf.variable_scope(scope) as sc:
            end_points = {}
            _conv('conv1', 7, 3, 64)
            end_points[sc.name + '/conv1'] = None
            d = 64
            for b, (depth, bott) in enumerate(((256, 64), (512, 128), (1024, 256), (2048, 512))):
                d = _stack_block('block%d' % (b + 1), d, depth, bott, units[b])
                end_points['%s/block%d' % (sc.name, b + 1)] = _FEATURE_MAP[0] if b == 2 else None
            return end_points[sc.name + '/block4'], end_points
    net.default_image_size = 224
    return net


class Block(tuple):
    """resnet_utils.Block(scope, unit_fn, args)."""
    def __new__(cls, scope, unit_fn, args):
        return tuple.__new__(cls, (scope, unit_fn, args))
    scope = property(lambda self: self[0])
    unit_fn = property(lambda self: self[1])
    args = property(lambda self: self[2])


def bottleneck(*a, **k):
    raise NotImplementedError('slim stand-in: resnet_v1.bottleneck is only a Block field (stack_blocks_dense reads args)')


def stack_blocks_dense(net, blocks, output_stride=None, **kwargs):
    """truncated_base_network.py:82-93: block4 on the pooled ROIs with the trunk's block4 variables (`reuse=True`): the
    variables exist already, none is created; the stand-in returns its input (identity tail: computes nothing)."""
    for block in blocks:
        depth_in = None
        for i, unit in enumerate(block.args):
            with tf.variable_scope(block.scope):
                with tf.variable_scope('unit_%d' % (i + 1)):
                    _bottleneck_unit(unit['depth'] // 2 if (i == 0 and depth_in is None) else unit['depth'],
                                     unit['depth'], unit['depth_bottleneck'])
    return net


def install():
    """Registers `tensorflow.contrib.slim`, `tensorflow.contrib.slim.nets` (+ `resnet_v1`, `resnet_v2`, `resnet_utils`,
    `vgg`) as stand-in modules and hangs them under the shim's `tf.contrib`."""
    slim = types.ModuleType('tensorflow.contrib.slim')
    slim.arg_scope, slim.batch_norm = arg_scope, batch_norm
    nets = types.ModuleType('tensorflow.contrib.slim.nets')
    resnet_utils = types.ModuleType('tensorflow.contrib.slim.nets.resnet_utils')
    resnet_utils.resnet_arg_scope, resnet_utils.Block, resnet_utils.stack_blocks_dense = resnet_arg_scope, Block, stack_blocks_dense
    resnet_v1 = types.ModuleType('tensorflow.contrib.slim.nets.resnet_v1')
    for arch in BLOCKS:
        setattr(resnet_v1, arch, _resnet_v1(arch))
    resnet_v1.resnet_v1 = _resnet_v1('resnet_v1_50')
    resnet_v1.bottleneck = bottleneck
    resnet_v2 = types.ModuleType('tensorflow.contrib.slim.nets.resnet_v2')
    resnet_v2.resnet_utils = resnet_utils
    vgg = tf._Inert('tensorflow.contrib.slim.nets.vgg')
    nets.resnet_v1, nets.resnet_v2, nets.resnet_utils, nets.vgg = resnet_v1, resnet_v2, resnet_utils, vgg
    slim.nets = nets
    tf.contrib.slim = slim
    sys.modules['tensorflow.contrib'] = tf.contrib
    sys.modules['tensorflow.contrib.slim'] = slim
    sys.modules['tensorflow.contrib.slim.nets'] = nets
    for name, mod in (('resnet_v1', resnet_v1), ('resnet_v2', resnet_v2), ('resnet_utils', resnet_utils), ('vgg', vgg)):
        sys.modules['tensorflow.contrib.slim.nets.' + name] = mod
    return slim
