"""Optimizer / learning-rate factory + the data-parallel train step.

Reference: luminoth/utils/training.py:20-120 (`get_learning_rate`,
`get_optimizer`, `clip_gradients_by_norm`) and luminoth/train.py:66-91.

The optimizer is ONE fused kernel over the flat parameter buffer
(lmh_sgd_momentum); under torch.distributed (RCCL) the flat gradient buffer is
all-reduced with one collective before the update (replaces the reference's
asynchronous parameter-server exchange, train.py:46,282-326).
"""
import os

import torch
import torch.distributed as dist

from luminoth_amd import kernels as K

FUSED_STEP = os.environ.get('LUMINOTH_AMD_FUSED_STEP', '1') != '0'
OPTIMIZERS = {'momentum', 'gradient_descent', 'adam', 'rmsprop'}
LEARNING_RATE_DECAY_METHODS = {'piecewise_constant', 'exponential_decay'}


def get_learning_rate(train_config, global_step=0):
    """training.py:20-61: constant, piecewise_constant(boundaries, values) or
    exponential_decay(decay_steps, decay_rate, staircase) schedule; host side."""
    lr_config = dict(train_config.learning_rate)
    lr_config.pop('_replace', None)
    decay = lr_config.pop('decay_method', None)
    if not decay or decay == 'none':
        return float(lr_config.get('learning_rate', lr_config.get('value')))
    if decay not in LEARNING_RATE_DECAY_METHODS:
        raise ValueError('Invalid learning_rate method "{}"'.format(decay))
    if decay == 'piecewise_constant':
        bounds, values = lr_config['boundaries'], lr_config['values']
        for b, v in zip(bounds, values):
            if global_step <= b:
                return float(v)
        return float(values[-1])
    base = float(lr_config['learning_rate'])
    p = global_step / float(lr_config['decay_steps'])
    if lr_config.get('staircase'):
        p = int(p)
    return base * float(lr_config['decay_rate']) ** p


class MomentumOptimizer(object):
    """tf.train.MomentumOptimizer (non-Nesterov): v = m*v + g ; w -= lr*v, with
    g = grad/world + wd*w (the L2 regulariser of total_loss)."""

    def __init__(self, model, train_config, momentum=0.9):
        self.model, self.store, self.cfg = model, model.store, train_config
        self.momentum = float(momentum)
        self.global_step = 0

    def reduce_gradients(self):
        """Sum the flat gradient buffer over the data-parallel replicas (one collective);
        returns the factor that turns the sum into the mean (applied inside the update kernel)."""
        st = self.store
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(st.grad)                 # ONE bucket: the flat gradient buffer
            return 1.0 / dist.get_world_size()
        return 1.0

    def step(self):
        st = self.store
        gscale = self.reduce_gradients()
        lr = get_learning_rate(self.cfg, self.global_step)
        K.sgd_momentum(st.flat, st.grad, st.mom, st.seg_offset, st.seg_wd, lr, self.momentum, gscale)
        self.global_step += 1


def get_optimizer(train_config, model):
    """training.py:64-81."""
    opt = dict(train_config.optimizer)
    opt.pop('_replace', None)
    kind = opt.pop('type')
    if kind not in OPTIMIZERS:
        raise ValueError('Invalid optimizer type "{}"'.format(kind))
    if train_config.get('clip_by_norm'):
        raise NotImplementedError('train.clip_by_norm (per-tensor clip_by_norm, training.py:84-120) '
                                  'is not implemented in the fused optimizer yet (reference default: False)')
    if kind == 'momentum':
        return MomentumOptimizer(model, train_config, momentum=opt.get('momentum', 0.9))
    if kind == 'gradient_descent':
        return MomentumOptimizer(model, train_config, momentum=0.0)
    raise NotImplementedError('optimizer "{}" has no fused HIP kernel yet (reference default: momentum)'.format(kind))


def broadcast_parameters(model, src=0):
    """Identical replicas at start (seeded init already makes them identical; this is the belt)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(model.store.flat, src)
        dist.broadcast(model.store.frozen, src)


def train_step(model, optimizer, image, gt_boxes):
    """One step of train.py:66-91: forward, loss, backward, (all-reduce), update."""
    if FUSED_STEP and hasattr(model, 'train_step'):
        total, pred = model.train_step(image, gt_boxes)     # same arithmetic, two-stream schedule
    else:
        pred = model(image, gt_boxes, is_training=True)
        total = model.loss(pred)
        model.backward(total)
    optimizer.step()
    return total, pred
