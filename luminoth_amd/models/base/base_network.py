"""BaseNetwork — same role and config surface as the reference's
luminoth/models/base/base_network.py:30-259: architecture dispatch, channel-mean
preprocessing (only for names starting with 'vgg' / 'resnet', :153-157),
fine_tune_from slicing of the trainable variables (:211-241) and the
checkpoint-name map (:243-259) — built on HIP node executors instead of
tf.contrib.slim.
"""
import math

import torch

from luminoth_amd.models.base import layers as L
from luminoth_amd.models.base import networks

VALID_ARCHITECTURES = set([
    'resnet_v1_50', 'resnet_v1_101', 'resnet_v1_152',
    'resnet_v2_50', 'resnet_v2_101', 'resnet_v2_152',
    'vgg_16', 'truncated_vgg_16',
])
_IMPLEMENTED = set(VALID_ARCHITECTURES)


def he_normal(shape, gen):
    """Pretrained slim checkpoints cannot be downloaded here (no network): the
    backbone starts from He-normal weights."""
    fan_in = shape[0] * shape[1] * shape[2]
    return torch.randn(shape, generator=gen) * math.sqrt(2.0 / fan_in)


def zeros(shape, gen):
    return torch.zeros(shape)


def ones(shape, gen):
    return torch.ones(shape)


class BaseNetwork(object):
    def __init__(self, config, name='base_network'):
        arch = config.get('architecture')
        if arch not in VALID_ARCHITECTURES:
            raise ValueError('Invalid architecture: "{}"'.format(arch))
        if arch not in _IMPLEMENTED:
            raise NotImplementedError('architecture "{}" has no HIP executor yet'.format(arch))
        self._architecture = arch
        self._config = config
        self.module_name = name
        self.pretrained_weights_scope = None

    # -- architecture predicates (base_network.py:103-127) -----------------------
    @property
    def vgg_type(self):
        return self._architecture.startswith('vgg')

    @property
    def truncated_vgg_type(self):
        return self._architecture.startswith('truncated_vgg')

    @property
    def resnet_type(self):
        return self._architecture.startswith('resnet')

    @property
    def resnet_v1_type(self):
        return self._architecture.startswith('resnet_v1')

    @property
    def resnet_v2_type(self):
        return self._architecture.startswith('resnet_v2')

    def _weight_decay(self):
        return float((self._config.get('arg_scope') or {}).get('weight_decay', 0.0) or 0.0)

    def _in_sub(self):
        # preprocess(): means are subtracted only for 'vgg*' / 'resnet*' names
        return networks._RGB_MEANS if (self.vgg_type or self.resnet_type) else None

    def get_checkpoint_file(self):
        """base_network.py:193-194 / utils/checkpoint_downloader.py:92-104 without the download: the explicit
        `weights` path of the config, else `<LUMINOTH_HOME or ~/.luminoth>/<architecture>.ckpt` (V1 file or V2
        prefix) if it is already there; None when nothing is available (no network in this deployment)."""
        import os
        explicit = self._config.get('weights')
        if explicit:
            if os.path.exists(explicit) or os.path.exists(explicit + '.index'):
                return explicit
            raise IOError('model.base_network.weights: "{}" does not exist'.format(explicit))
        home = os.environ.get('LUMINOTH_HOME') or os.path.join(os.path.expanduser('~'), '.luminoth')
        path = os.path.join(home, '{}.ckpt'.format(self._architecture))
        if os.path.exists(path) or os.path.exists(path + '.index'):
            return path
        return None

    # -- variable bookkeeping -----------------------------------------------------
    def _ordered_var_names(self):
        """Trainable variable names in TF creation order."""
        names = []
        for layer in self._creation_order_layers():
            names += layer.var_names()
        return names

    def get_trainable_var_names(self):
        """base_network.py:211-241: everything from the first variable whose name
        contains `fine_tune_from` onwards (all if None)."""
        names = self._ordered_var_names()
        ft = self._config.get('fine_tune_from')
        if ft is None:
            return names
        for i, n in enumerate(names):
            if ft in n:
                return names[i:]
        raise ValueError('"{}" is an invalid value of fine_tune_from for this architecture.'.format(ft))

    def get_base_network_checkpoint_vars(self, store):
        """{checkpoint_name: tensor} with the module scope stripped (base_network.py:243-259)."""
        prefix = self.module_name + '/'
        return {n[len(prefix):]: t for n, t in store.params.items() if n.startswith(prefix)}
