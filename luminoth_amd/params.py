"""Flat parameter store: HBM layout of the model variables.

All TRAINABLE variables live in ONE contiguous fp32 buffer (`flat`), with twin
buffers for gradients (`grad`) and momentum (`mom`); frozen variables (conv1 /
block1 weights, BatchNorm moving statistics) live in a second buffer.  Named
tensors are views.  Consequences:
  * the optimizer is one fused kernel over `flat` (lmh_sgd_momentum) with a
    per-segment weight-decay table (the L2 regulariser of the reference);
  * data-parallel training all-reduces `grad` with ONE RCCL call;
  * variables keep their TensorFlow names (e.g.
    `truncated_base_network/resnet_v1_50/block2/unit_1/bottleneck_v1/conv1/weights`)
    so slim checkpoints map 1:1 (reference: base_network.py:243-259).
Segments are padded to 4 floats so every view is 16-byte aligned.
"""
from collections import OrderedDict

import torch


class ParamSpec(object):
    __slots__ = ('name', 'shape', 'init', 'trainable', 'wd', 'reg_in_loss')

    def __init__(self, name, shape, init, trainable, wd=0.0, reg_in_loss=None):
        self.name, self.shape, self.init = name, tuple(shape), init
        self.trainable, self.wd = bool(trainable), float(wd)
        # L2 term contributes to regularization_loss even when the variable is frozen
        self.reg_in_loss = self.wd if reg_in_loss is None else float(reg_in_loss)


class ParamStore(object):
    def __init__(self):
        self.specs = OrderedDict()
        self.built = False

    def add(self, name, shape, init, trainable=True, wd=0.0):
        if self.built:
            raise RuntimeError('ParamStore already built')
        if name in self.specs:
            raise ValueError('duplicate variable %s' % name)
        self.specs[name] = ParamSpec(name, shape, init, trainable, wd)
        return name

    @staticmethod
    def _numel(shape):
        n = 1
        for s in shape:
            n *= s
        return n

    def build(self, device, seed=0):
        """Draw initial values on CPU (seeded) and place the buffers on `device`."""
        gen = torch.Generator().manual_seed(0 if seed is None else int(seed))
        lay = {True: [], False: []}
        off = {True: 0, False: 0}
        for sp in self.specs.values():
            n = self._numel(sp.shape)
            lay[sp.trainable].append((sp, off[sp.trainable], n))
            off[sp.trainable] += (n + 3) // 4 * 4
        cpu = {t: torch.zeros(max(off[t], 4), dtype=torch.float32) for t in (True, False)}
        for t in (True, False):
            for sp, o, n in lay[t]:
                cpu[t][o:o + n] = sp.init(sp.shape, gen).reshape(-1).to(torch.float32)
        self.flat = cpu[True].to(device)
        self.frozen = cpu[False].to(device)
        self.grad = torch.zeros_like(self.flat)
        self.mom = torch.zeros_like(self.flat)
        self.params, self.grads, self.offsets = OrderedDict(), OrderedDict(), OrderedDict()
        seg_off, seg_wd = [], []
        for sp, o, n in lay[True]:
            self.params[sp.name] = self.flat[o:o + n].view(sp.shape)
            self.grads[sp.name] = self.grad[o:o + n].view(sp.shape)
            self.offsets[sp.name] = (True, o, n)
            seg_off.append(o)
            seg_wd.append(sp.wd)
        seg_off.append(int(self.flat.numel()))
        if len(seg_wd) == 0:
            seg_wd = [0.0]
            seg_off = [0, int(self.flat.numel())]
        for sp, o, n in lay[False]:
            self.params[sp.name] = self.frozen[o:o + n].view(sp.shape)
            self.offsets[sp.name] = (False, o, n)
        self.seg_offset = torch.tensor(seg_off, dtype=torch.int64, device=device)
        self.seg_wd = torch.tensor(seg_wd, dtype=torch.float32, device=device)
        # regularisation value of the FROZEN-but-regularised variables is a constant
        fo, fw = [], []
        for sp, o, n in lay[False]:
            fo.append(o)
            fw.append(sp.reg_in_loss)
        fo.append(int(self.frozen.numel()))
        if len(fw) == 0:
            fw, fo = [0.0], [0, int(self.frozen.numel())]
        self.frozen_seg_offset = torch.tensor(fo, dtype=torch.int64, device=device)
        self.frozen_seg_wd = torch.tensor(fw, dtype=torch.float32, device=device)
        self.built = True
        return self

    # -- views -----------------------------------------------------------------
    def __getitem__(self, name):
        return self.params[name]

    def grad_of(self, name):
        return self.grads[name]

    def trainable_names(self):
        return [n for n, sp in self.specs.items() if sp.trainable]

    def state_dict(self):
        """name -> CPU tensor (model variables only; optimizer slots are not
        checkpointed, as in the reference: train.py:97-112)."""
        return OrderedDict((n, p.detach().cpu().clone()) for n, p in self.params.items())

    def load_state_dict(self, sd, strict=True):
        for n, p in self.params.items():
            if n in sd:
                p.copy_(torch.as_tensor(sd[n]).to(p.device).view(p.shape))
            elif strict:
                raise KeyError('missing variable %s' % n)
