"""Initializer / activation factories (reference: luminoth/utils/vars.py:56-88).

Initial values are drawn on the CPU with a seeded torch.Generator so that the
HIP model and the CPU oracle start from bit-identical weights.
"""
import math

import torch

VALID_INITIALIZERS = {
    'truncated_normal_initializer', 'variance_scaling_initializer', 'random_normal_initializer',
    'xavier_initializer', 'zeros_initializer', 'constant_initializer',
}


def truncated_standard_normal(shape, gen):
    """tf.truncated_normal (vars.py:4-5; Sonnet's default Conv2D / Linear initializer): values further than two standard
    deviations from the mean are DROPPED AND RE-DRAWN, not clamped (a clamp would put 4.6 % of the mass on the bounds)."""
    t = torch.randn(shape, generator=gen)
    bad = t.abs() > 2
    while bool(bad.any()):
        t[bad] = torch.randn(int(bad.sum()), generator=gen)
        bad = t.abs() > 2
    return t


def get_initializer(cfg, seed=None):
    """cfg: {'type': ..., **kwargs}.  Returns fn(shape, generator) -> cpu fp32 tensor.
    Unknown types raise ValueError (vars.py:66-71)."""
    cfg = dict(cfg)
    cfg.pop('_replace', None)
    kind = cfg.pop('type')
    if kind not in VALID_INITIALIZERS:
        raise ValueError('Initializer "{}" is not valid.'.format(kind))

    def fans(shape):
        rf = 1
        for s in shape[:-2]:
            rf *= s
        return shape[-2] * rf, shape[-1] * rf

    truncated = truncated_standard_normal

    def init(shape, gen):
        if kind == 'random_normal_initializer':
            return torch.randn(shape, generator=gen) * cfg.get('stddev', 1.0) + cfg.get('mean', 0.0)
        if kind == 'truncated_normal_initializer':
            return truncated(shape, gen) * cfg.get('stddev', 1.0) + cfg.get('mean', 0.0)
        if kind in ('variance_scaling_initializer', 'xavier_initializer'):
            fan_in, fan_out = fans(shape)
            mode = cfg.get('mode', 'FAN_AVG' if kind == 'xavier_initializer' else 'FAN_IN')
            n = {'FAN_IN': fan_in, 'FAN_OUT': fan_out, 'FAN_AVG': (fan_in + fan_out) / 2.0}[mode]
            factor = cfg.get('factor', 1.0 if kind == 'xavier_initializer' else 2.0)
            if cfg.get('uniform', kind == 'xavier_initializer'):
                lim = math.sqrt(3.0 * factor / n)
                return (torch.rand(shape, generator=gen) * 2 - 1) * lim
            return truncated(shape, gen) * math.sqrt(1.3 * factor / n)
        if kind == 'constant_initializer':
            return torch.full(shape, float(cfg.get('value', 0.0)))
        return torch.zeros(shape)
    return init


# tf.nn.<name> -> the activation id of luminoth_amd.kernels.ACT.  relu / relu6 ride in the convolution epilogues; the
# others run as an in-place pass behind the convolution (csrc/elementwise.hip: lmh_act_fwd / lmh_act_bwd).  What tf.nn
# holds that is NOT here either is not an activation (conv2d, dropout, ...), changes the channel count (crelu) or needs
# the pre-activation for its gradient (swish, TF >= 1.7).
VALID_ACTIVATIONS = {'relu': 'relu', 'relu6': 'relu6', 'elu': 'elu', 'selu': 'selu', 'softplus': 'softplus',
                     'softsign': 'softsign', 'sigmoid': 'sigmoid', 'tanh': 'tanh', 'leaky_relu': 'leaky_relu',
                     None: None, '': None, 'none': None}


def get_activation_function(name):
    """vars.py:80-88: `getattr(tf.nn, name)`, unknown names raise ValueError; a false value is the identity.  Returns the
    activation id string the layers understand."""
    if name not in VALID_ACTIVATIONS and not (isinstance(name, str) and name.lower() in VALID_ACTIVATIONS):
        raise ValueError('Invalid activation function "{}"'.format(name))
    return VALID_ACTIVATIONS.get(name, VALID_ACTIVATIONS.get(str(name).lower()))
