"""Oracle (test infrastructure): the Faster R-CNN train step restated end to end
on the CPU — torch fp32 for the dense differentiable parts (backbone, heads,
ROI pooling, losses; `torch.autograd` supplies the reference gradients), numpy
(oracle/frcnn.py) for proposals and targets.  One image at a time, exactly like
the reference (batch 1: luminoth/models/fasterrcnn/fasterrcnn.py:101-103);
the batch loss is the mean over images.

Variables come in as a {tf_name: tensor} dict (same names as the reference's
checkpoints).  PARITY UNPINNED: the slim ResNet / VGG arithmetic is third-party
(tf.contrib.slim, not in /root/reference); structure per SURVEY.md §8a-A2.
Citations relative to /root/reference/luminoth/.

Two test-only knobs:
  * `dtype=torch.float64` runs the dense parts in double precision (conditioning studies: is a loss
    difference between two fp32 implementations round-off of an ill-conditioned network, or a bug?);
  * `masks` = {layer scope: that layer's OUTPUT as computed by the HIP kernels}: every ReLU / ReLU6
    decision is then taken from the kernels' own activations instead of the oracle's pre-activations, so the
    reference gradient is the derivative of exactly the piecewise-linear branch the kernels were on (an
    activation within round-off of a kink otherwise moves whole dy*x terms between the two implementations).
"""
import numpy as np
import torch

from . import boxes as bx
from . import frcnn as of
from . import rng
from . import torch_ops as ot

RESNET_UNITS = {'resnet_v1_50': (3, 4, 6, 3), 'resnet_v1_101': (3, 4, 23, 3), 'resnet_v1_152': (3, 8, 36, 3),
                'resnet_v2_50': (3, 4, 6, 3), 'resnet_v2_101': (3, 4, 23, 3), 'resnet_v2_152': (3, 8, 36, 3)}
MEANS = torch.tensor([123.68, 116.78, 103.94])   # models/base/base_network.py:14-16


VGG16_CFG = (('conv1', 2, 64), ('conv2', 2, 128), ('conv3', 3, 256), ('conv4', 3, 512), ('conv5', 3, 512))


class _Softsign(torch.autograd.Function):
    """tf.nn.softsign with TF's SoftsignGrad, dy / (1 + |x|)^2 (softsign_op.h).  torch.nn.functional.softsign is a composite:
    autograd differentiates x / (1 + |x|) term by term, 1 / (1 + |x|) - |x| / (1 + |x|)^2, which cancels in fp32 for the
    |x| ~ 1e3 an RPN convolution on an unnormalised feature map reaches (1e-3 relative on the weight gradient)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x / (1 + x.abs())

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        t = 1 + x.abs()
        return g / (t * t)


def _act(x, name):
    if name == 'relu':
        return torch.relu(x)
    if name == 'relu6':
        return torch.clamp(x, 0, 6)
    # the other activations of tf.nn that luminoth/utils/vars.py:80-88 hands through (getattr(tf.nn, name)); TF 1.x
    # definitions: nn_ops.py / nn_impl.py (leaky_relu: alpha 0.2; selu: scale 1.0507009873554805, alpha 1.6732632423543772)
    if name == 'elu':
        return torch.nn.functional.elu(x)
    if name == 'selu':
        return torch.nn.functional.selu(x)
    if name == 'softplus':
        return torch.nn.functional.softplus(x)
    if name == 'softsign':
        return _Softsign.apply(x)
    if name == 'sigmoid':
        return torch.sigmoid(x)
    if name == 'tanh':
        return torch.tanh(x)
    if name == 'leaky_relu':
        return torch.nn.functional.leaky_relu(x, 0.2)
    if name:
        raise ValueError('Invalid activation function "{}"'.format(name))
    return x


class OracleFasterRCNN(object):
    def __init__(self, variables, arch='resnet_v1_50', num_classes=80, scope='fasterrcnn',
                 base_scope='truncated_base_network', anchors=None, rpn=None, rcnn=None, weight_decay=5e-4,
                 l2_rpn=5e-4, l2_rcnn=5e-4, fine_tune_from='block2', seed=None, dtype=torch.float32, compute=None,
                 storage=None, train_bn=False):
        self.dtype = dtype
        # `train_batch_norm: True` (base_network.py:82-93): every BatchNorm normalises with the statistics of the tensor it
        # sees (the reference is batch-1: one image) and reports the moving-average update slim would apply (bn_updates)
        self.train_bn = bool(train_bn)
        self.bn_updates = {}
        # 'f16' / 'bf16': the ResNet blocks keep 16-bit tensors in memory (oracle/torch_ops.py HalfStorageConvFn restates
        # luminoth_amd/csrc/conv_hs.h); implies the same `compute`
        self.storage = storage if storage in ('f16', 'bf16') else None
        if self.storage:
            compute = self.storage
        # 'f16' / 'bf16': the operands of every backbone / tail convolution with C % 32 == 0 and of the RPN 3x3
        # convolution are rounded like the mixed-precision kernels round them (oracle/torch_ops.py QuantConvFn)
        self.compute = compute if compute in ('f16', 'bf16') else None
        self.v = {k: torch.as_tensor(v).clone().to(dtype) for k, v in variables.items()}
        self.masks = None
        self.arch, self.C, self.scope, self.base = arch, num_classes, scope, '%s/%s' % (base_scope, arch)
        a = anchors or {}
        self.anchor_ref = bx.generate_anchors_reference(a.get('base_size', 256), np.array(a.get('ratios', [.5, 1, 2])),
                                                        np.array(a.get('scales', [.25, .5, 1, 2])))
        self.stride = a.get('stride', 16)
        self.rpn_cfg = dict(pre_nms_top_n=12000, post_nms_top_n=2000, nms_threshold=0.7)
        self.rpn_cfg.update(rpn or {})
        self.rcnn_cfg = dict(minibatch_size=256)
        self.rcnn_cfg.update(rcnn or {})
        self.wd, self.l2_rpn, self.l2_rcnn = weight_decay, l2_rpn, l2_rcnn
        self.fine_tune_from, self.seed = fine_tune_from, seed
        self.step = 0

    def _activate(self, z, act, scope):
        """Activation of layer `scope`; with `self.masks` the branch is the one the kernels took."""
        yk = None if self.masks is None else self.masks.get(scope)
        if yk is None or act not in ('relu', 'relu6'):      # smooth / leaky activations: no branch to pin
            return _act(z, act)
        yk = torch.as_tensor(yk).reshape(z.shape)
        if act == 'relu':
            return z * (yk > 0).to(z.dtype)
        return z * ((yk > 0) & (yk < 6)).to(z.dtype) + 6.0 * (yk >= 6).to(z.dtype)

    # ---- backbone: slim resnet_v1 up to block3, output_stride 16 ----------------
    HS_LOSS_SCALE = {'f16': 1024.0, 'bf16': 1.0}

    def _conv_bn_hs(self, x, scope, stride=1, rate=1, padding='SAME', act='relu', residual=None, round_dx=False,
                    out_f32=False):
        """conv + frozen BatchNorm (+ residual) (+ ReLU) of a half-storage layer: one fused, rounded unit."""
        v = self.v
        rstd = torch.rsqrt(v[scope + '/BatchNorm/moving_variance'] + 1e-5)
        scale = v[scope + '/BatchNorm/gamma'] * rstd                    # layers.py BNTable.refresh
        shift = v[scope + '/BatchNorm/beta'] - v[scope + '/BatchNorm/moving_mean'] * scale
        yk = None if self.masks is None else self.masks.get(scope)
        cfg = dict(quant=self.storage, stride=stride, dilation=rate, padding=padding, act=act, out_f32=out_f32,
                   round_dx=round_dx, loss_scale=self.HS_LOSS_SCALE[self.storage])
        return ot.HalfStorageConvFn.apply(x, v[scope + '/weights'], scale, shift, residual, yk, cfg)

    def _bottleneck_hs(self, x, scope, depth, stride, rate, out_f32=False):
        p = scope + '/bottleneck_v1'
        if x.shape[-1] == depth:
            sc = x if stride == 1 else x[:, ::stride, ::stride, :]
        else:
            sc = self._conv_bn_hs(x, p + '/shortcut', stride=stride, act=None, round_dx=True)
        r = self._conv_bn_hs(x, p + '/conv1')
        r = self._conv_bn_hs(r, p + '/conv2', stride=stride, rate=rate, padding='SAME' if stride == 1 else 'SAME_EXPLICIT')
        return self._conv_bn_hs(r, p + '/conv3', act='relu', residual=sc, out_f32=out_f32)

    def _conv_bn(self, x, scope, stride=1, rate=1, padding='SAME', act='relu'):
        v = self.v
        q = self.compute if x.shape[-1] % 32 == 0 else None          # conv1 (3 channels) runs the fp32 stem kernel
        y = ot.conv2d_nhwc(x, v[scope + '/weights'], stride, rate, padding, quant=q)
        if self.train_bn:
            # slim.batch_norm(is_training=True): biased batch variance normalises; tf.nn.fused_batch_norm hands the
            # UNBIASED one to the moving average (decay 0.997: resnet_arg_scope's default)
            mean = y.mean(dim=(0, 1, 2))
            var = ((y - mean) ** 2).mean(dim=(0, 1, 2))
            n = float(y.numel() // y.shape[-1])
            self.bn_updates[scope] = (mean.detach().clone(), (var * (n / max(n - 1.0, 1.0))).detach().clone())
            y = (y - mean) * torch.rsqrt(var + 1e-5) * v[scope + '/BatchNorm/gamma'] + v[scope + '/BatchNorm/beta']
            return self._activate(y, act, scope)
        y = ot.frozen_batch_norm(y, v[scope + '/BatchNorm/gamma'], v[scope + '/BatchNorm/beta'],
                                 v[scope + '/BatchNorm/moving_mean'], v[scope + '/BatchNorm/moving_variance'])
        return self._activate(y, act, scope)

    def moving_statistics_after_step(self, decay=0.997):
        """What the UPDATE_OPS of the step (train.py:87-88) leave in the moving statistics: v -= (v - batch) * (1 - decay)."""
        out = {}
        for scope, (mean, var) in self.bn_updates.items():
            pre = scope if (scope + '/moving_mean') in self.v else scope + '/BatchNorm'      # resnet_v2 keys are prefixes
            mm, mv = self.v[pre + '/moving_mean'], self.v[pre + '/moving_variance']
            out[pre + '/moving_mean'] = mm - (mm - mean) * (1.0 - decay)
            out[pre + '/moving_variance'] = mv - (mv - var) * (1.0 - decay)
        return out

    def _bottleneck(self, x, scope, depth, stride, rate):
        p = scope + '/bottleneck_v1'
        if x.shape[-1] == depth:
            sc = x if stride == 1 else x[:, ::stride, ::stride, :]      # resnet_utils.subsample
        else:
            sc = self._conv_bn(x, p + '/shortcut', stride=stride, act=None)
        r = self._conv_bn(x, p + '/conv1')
        r = self._conv_bn(r, p + '/conv2', stride=stride, rate=rate,
                          padding='SAME' if stride == 1 else 'SAME_EXPLICIT')
        r = self._conv_bn(r, p + '/conv3', act=None)
        return self._activate(sc + r, 'relu', p + '/conv3')

    # ---- slim resnet_v2 (pre-activation; base_network.py:94-101) ----------------------------------------------------
    def _batch_norm(self, y, prefix):
        """BatchNorm over variables `<prefix>/{gamma,beta,moving_mean,moving_variance}`: the statistics of the batch when
        `train_bn` (the reference hands is_training to every resnet_v2 BatchNorm), else the moving ones."""
        v = self.v
        if self.train_bn:
            mean = y.mean(dim=(0, 1, 2))
            var = ((y - mean) ** 2).mean(dim=(0, 1, 2))
            n = float(y.numel() // y.shape[-1])
            self.bn_updates[prefix] = (mean.detach().clone(), (var * (n / max(n - 1.0, 1.0))).detach().clone())
            return (y - mean) * torch.rsqrt(var + 1e-5) * v[prefix + '/gamma'] + v[prefix + '/beta']
        return ot.frozen_batch_norm(y, v[prefix + '/gamma'], v[prefix + '/beta'], v[prefix + '/moving_mean'],
                                    v[prefix + '/moving_variance'])

    def _conv_bias(self, x, scope, stride=1, rate=1, padding='SAME'):
        return ot.conv2d_nhwc(x, self.v[scope + '/weights'], stride, rate, padding) + self.v[scope + '/biases']

    def _bottleneck_v2(self, x, scope, depth, stride, rate):
        p = scope + '/bottleneck_v2'
        pre = self._activate(self._batch_norm(x, p + '/preact'), 'relu', p + '/preact')
        if x.shape[-1] == depth:
            sc = x if stride == 1 else x[:, ::stride, ::stride, :]      # resnet_utils.subsample(inputs)
        else:
            sc = self._conv_bias(pre, p + '/shortcut', stride=stride)
        r = ot.conv2d_nhwc(pre, self.v[p + '/conv1/weights'], 1, 1, 'SAME')
        r = self._activate(self._batch_norm(r, p + '/conv1/BatchNorm'), 'relu', p + '/conv1')
        r = ot.conv2d_nhwc(r, self.v[p + '/conv2/weights'], stride, rate, 'SAME' if stride == 1 else 'SAME_EXPLICIT')
        r = self._activate(self._batch_norm(r, p + '/conv2/BatchNorm'), 'relu', p + '/conv2')
        return sc + self._conv_bias(r, p + '/conv3')

    def vgg_backbone(self, image):
        """slim vgg_16 up to conv5/conv5_3 (truncated_base_network.py:8-16): 3x3 SAME conv + bias + ReLU,
        2x2/2 VALID max-pools after conv1..conv4."""
        x = image - MEANS.to(self.dtype)
        for bi, (name, reps, depth) in enumerate(VGG16_CFG):
            for r in range(reps):
                sc = '%s/%s/%s_%d' % (self.base, name, name, r + 1)
                x = ot.conv2d_nhwc(x, self.v[sc + '/weights'], 1, 1, 'SAME', bias=self.v[sc + '/biases'],
                                   quant=self.compute if x.shape[-1] % 32 == 0 else None)
                x = self._activate(x, 'relu', sc)
            if bi < 4:
                x = ot.max_pool_nhwc(x, 2, 2, 'VALID')
        return x

    def backbone(self, image, upto=3, output_stride=16):
        image = image.to(self.dtype)
        if self.arch == 'vgg_16':
            return self.vgg_backbone(image)
        x = image - MEANS.to(self.dtype)
        v2 = self.arch.startswith('resnet_v2')
        if v2:       # root: conv2d_same with a bias, no BatchNorm, no activation
            x = self._conv_bias(x, self.base + '/conv1', stride=2, padding='SAME_EXPLICIT')
        else:
            x = self._conv_bn(x, self.base + '/conv1', stride=2, padding='SAME_EXPLICIT')
        x = ot.max_pool_nhwc(x, 3, 2, 'SAME')
        if self.storage:
            x = ot._q(x, self.storage)          # the pool writes the first 16-bit tensor of the trunk
        current, rate = 4, 1
        last_block = min(upto, 4)
        for bi, (depth, n) in enumerate(zip((256, 512, 1024, 2048), RESNET_UNITS[self.arch])):
            if bi + 1 > upto:
                break
            for u in range(n):
                ustride = (2 if bi < 3 else 1) if u == n - 1 else 1
                if current == output_stride:
                    s, r = 1, rate
                    rate *= ustride
                else:
                    s, r = ustride, 1
                    current *= ustride
                if self.storage:
                    top = bi + 1 == last_block and u == n - 1          # the feature map is handed on as fp32
                    x = self._bottleneck_hs(x, '%s/block%d/unit_%d' % (self.base, bi + 1, u + 1), depth, s, r, out_f32=top)
                elif v2:
                    x = self._bottleneck_v2(x, '%s/block%d/unit_%d' % (self.base, bi + 1, u + 1), depth, s, r)
                else:
                    x = self._bottleneck(x, '%s/block%d/unit_%d' % (self.base, bi + 1, u + 1), depth, s, r)
        return x

    def tail(self, pooled):
        """truncated_base_network.py:56-95: block4 on pooled ROIs, ResNet-101 only."""
        if self.arch != 'resnet_v1_101':
            return pooled
        x = pooled
        for u in range(3):
            x = self._bottleneck(x, '%s/block4/unit_%d' % (self.base, u + 1), 2048, 1, 1)
        return x

    # ---- heads --------------------------------------------------------------------
    def rpn_head(self, feat):
        v, p = self.v, self.scope + '/rpn'
        f = self._activate(ot.conv2d_nhwc(feat, v[p + '/conv/w'], padding='SAME', bias=v[p + '/conv/b'],
                                          quant=self.compute), self.rpn_cfg.get('activation_function', 'relu6'),
                           p + '/conv')
        cls = ot.conv2d_nhwc(f, v[p + '/cls_conv/w'], padding='VALID', bias=v[p + '/cls_conv/b'])
        box = ot.conv2d_nhwc(f, v[p + '/bbox_conv/w'], padding='VALID', bias=v[p + '/bbox_conv/b'])
        return cls.reshape(-1, 2), box.reshape(-1, 4)

    def rcnn_head(self, feat, rois, im_shape):
        v, p = self.v, self.scope + '/rcnn'
        pooled = ot.roi_pool(feat, rois.to(self.dtype), torch.zeros(rois.shape[0], dtype=torch.long), im_shape)
        net = self.tail(pooled).mean(dim=(1, 2))
        cls = net @ v[p + '/fc_classifier/w'] + v[p + '/fc_classifier/b']
        box = net @ v[p + '/fc_bbox/w'] + v[p + '/fc_bbox/b']
        return cls, box, pooled

    # ---- regularisation -------------------------------------------------------------
    def regularization_loss(self):
        tot = 0.0
        for k, t in self.v.items():
            if k.endswith('/weights'):
                tot = tot + self.wd * (t.double() ** 2).sum() / 2
            elif k.endswith('/w'):
                wd = self.l2_rpn if '/rpn/' in k else self.l2_rcnn
                tot = tot + wd * (t.double() ** 2).sum() / 2
        return tot

    # ---- one image ------------------------------------------------------------------
    def forward_image(self, image, gt, seed, overrides=None):
        """image (H,W,3) tensor, gt (G,5) numpy.  Returns dict of stage outputs and the
        four losses (torch scalars with graph).  `overrides` may pin 'proposals' / 'rois' etc.
        to values produced elsewhere (identical-input comparisons)."""
        ov = overrides or {}
        H, W = image.shape[0], image.shape[1]
        # 'feat': the trunk's output given (fixtures of the reference's top-level composition over a slim stand-in,
        # tests/golden/make_golden_ref_tf.py gen_toplevel: the composition around the trunk is what is compared)
        feat = torch.as_tensor(ov['feat']).to(self.dtype) if 'feat' in ov else self.backbone(image[None])
        fh, fw = feat.shape[1], feat.shape[2]
        cls_score, bbox_pred = self.rpn_head(feat)
        anchors = bx.generate_anchors(self.anchor_ref, fh, fw, self.stride)
        out = dict(feat=feat, rpn_cls_score=cls_score, rpn_bbox_pred=bbox_pred)
        labels, targets, _ = of.rpn_target(anchors, gt, (H, W), seed=seed)
        out['rpn_labels'], out['rpn_targets'] = labels, targets
        l_cls, l_reg = ot.rpn_loss(cls_score, bbox_pred, torch.tensor(labels), torch.tensor(targets).to(self.dtype), 3.0)
        out['rpn_cls_loss'], out['rpn_reg_loss'] = l_cls, l_reg
        if 'rois' in ov:
            rois, roi_labels, roi_targets = ov['rois'], ov['roi_labels'], ov['roi_targets']
        else:
            if 'proposals' in ov:
                proposals = ov['proposals']
            else:
                prob = torch.softmax(cls_score.detach(), dim=1).float().numpy()
                proposals = of.rpn_proposal(prob, bbox_pred.detach().float().numpy(), anchors, (H, W),
                                            **self.rpn_cfg)['proposals']
            out['proposals'] = proposals
            lab, tg = of.rcnn_target(proposals, gt, seed=seed, **self.rcnn_cfg)
            keep = lab >= 0                                               # rcnn.py:156-167
            rois, roi_labels, roi_targets = proposals[keep], lab[keep], tg[keep]
        out['rois'], out['roi_labels'], out['roi_targets'] = rois, roi_labels, roi_targets
        cls, box, pooled = self.rcnn_head(feat, torch.tensor(np.asarray(rois)), (H, W))
        out['rcnn_cls_score'], out['rcnn_bbox_offsets'], out['pooled'] = cls, box, pooled
        c_cls, c_reg = ot.rcnn_loss(cls, box, torch.tensor(np.asarray(roi_labels)),
                                    torch.tensor(np.asarray(roi_targets)).to(self.dtype), self.C, 1.0)
        out['rcnn_cls_loss'], out['rcnn_reg_loss'] = c_cls, c_reg
        return out

    # ---- which variables exist, in TensorFlow's creation order (slim resnet_v1: third party, restated; pinned against the
    # reference's own variable lists by tests/test_ref_tf_golden.py::test_toplevel_* over tests/golden/slim_standin.py) ----
    def resnet_variable_order(self, trainable_only=True):
        """conv1, then block1..4 / unit_i / bottleneck_v1 / [shortcut (unit_1 only),] conv1, conv2, conv3; per convolution
        `weights`, `BatchNorm/beta`, `BatchNorm/gamma` (+ moving_mean, moving_variance: not trainable)."""
        kinds = ['weights', 'BatchNorm/beta', 'BatchNorm/gamma'] + ([] if trainable_only else
                                                                    ['BatchNorm/moving_mean', 'BatchNorm/moving_variance'])
        convs = ['conv1']
        for b, units in enumerate(RESNET_UNITS[self.arch]):
            for u in range(units):
                p = 'block%d/unit_%d/bottleneck_v1/' % (b + 1, u + 1)
                convs += ([p + 'shortcut'] if u == 0 else []) + [p + 'conv1', p + 'conv2', p + 'conv3']
        return ['%s/%s/%s' % (self.base, c, k) for c in convs for k in kinds]

    def head_variable_order(self):
        """Sonnet creation order inside FasterRCNN._build: RPN (conv, cls_conv, bbox_conv: rpn.py:67-90), then RCNN (the FC
        stack, fc_classifier, fc_bbox: rcnn.py:70-98); `w` before `b`."""
        mods = ['rpn/conv', 'rpn/cls_conv', 'rpn/bbox_conv']
        mods += sorted({k.split('/')[2] for k in self.v if k.startswith(self.scope + '/rcnn/fc_') and
                        k.split('/')[2] not in ('fc_classifier', 'fc_bbox')}, key=lambda m: int(m.split('_')[1]))
        mods = [m if m.startswith('rpn/') else 'rcnn/' + m for m in mods] + ['rcnn/fc_classifier', 'rcnn/fc_bbox']
        return ['%s/%s/%s' % (self.scope, m, v) for m in mods for v in ('w', 'b')]

    def trainable_names_in_reference_order(self, endpoint='block3', use_tail=True, freeze_tail=False, base_trainable=True):
        """FasterRCNN.get_trainable_vars (fasterrcnn.py:337-358): the module's own variables, then — when
        `base_network.trainable` — BaseNetwork.get_trainable_vars (base_network.py:211-241: everything from the first
        variable whose name contains `fine_tune_from`) cut by TruncatedBaseNetwork.get_trainable_vars
        (truncated_base_network.py:96-144: up to the LAST variable whose name contains the endpoint; for resnet_v1_101 with
        a trainable tail, plus everything from the first `block4` variable on).  ResNet architectures."""
        names = self.head_variable_order()
        if not base_trainable:
            return names
        allv = self.resnet_variable_order()
        if self.fine_tune_from is not None:
            first = next((i for i, n in enumerate(allv) if self.fine_tune_from in n), None)
            if first is None:
                raise ValueError('"%s" is an invalid value of fine_tune_from for this architecture.' % self.fine_tune_from)
            allv = allv[first:]
        last = None
        for i, n in enumerate(allv):
            if endpoint in n:
                last = i
        out = allv[:last + 1] if last is not None else []
        if use_tail and not freeze_tail and self.arch == 'resnet_v1_101':
            first4 = next((i for i, n in enumerate(allv) if 'block4' in n), None)
            if first4 is None:
                raise ValueError('"block4" not present in the trainable vars retrieved from base network.')
            out = out + allv[first4:]
        return names + out

    def regularized_names(self):
        """The variables whose L2 term enters `regularization_loss`: every slim convolution `weights` (weight_decay arg_scope,
        frozen and unused blocks included) and every Sonnet layer `w` of the heads (rpn.py:54-56, rcnn.py:60-62)."""
        return [k for k in self.v if k.endswith('/weights') or k.endswith('/w')]

    # ---- trainable set (base_network.py:211-241, truncated_base_network.py:97-144) ---
    def trainable_names(self):
        names = []
        if self.arch == 'vgg_16':
            # creation order conv1_1/{weights,biases}, ...: everything from the first name containing
            # `fine_tune_from` on (base_network.py:211-241), up to the endpoint conv5_3
            order = ['%s/%s/%s_%d/%s' % (self.base, n, n, r + 1, kind) for n, reps, _ in VGG16_CFG
                     for r in range(reps) for kind in ('weights', 'biases')]
            ft = self.fine_tune_from
            first = 0 if ft is None else next(i for i, n in enumerate(order) if ft in n)
            names += order[first:]
        for k in self.v:
            if k.startswith(self.base):
                if 'moving_' in k or self.arch == 'vgg_16':
                    continue
                blk = [b for b in ('block2', 'block3') if '/%s/' % b in k]
                if blk or (self.arch == 'resnet_v1_101' and '/block4/' in k):
                    names.append(k)
            else:
                names.append(k)
        return names

    def train_step(self, images, gts, lr=3e-4, momentum=0.9, mom_state=None):
        """One full CPU train step (forward, loss, backward, momentum-SGD with the L2 term)."""
        names = self.trainable_names()
        for n in names:
            self.v[n].requires_grad_(True)
            self.v[n].grad = None
        B = len(images)
        total = 0.0
        parts = []
        for b in range(B):
            o = self.forward_image(images[b], gts[b], rng.image_seed(self.seed, self.step, b))
            loss_b = o['rpn_cls_loss'] + o['rpn_reg_loss'] + o['rcnn_cls_loss'] + o['rcnn_reg_loss']
            total = total + loss_b / B
            parts.append(o)
        reg = self.regularization_loss()
        total_loss = total + reg.float()
        total_loss.backward()
        mom_state = mom_state if mom_state is not None else {}
        with torch.no_grad():
            for n in names:
                g = self.v[n].grad
                if g is None:
                    continue
                vbuf = mom_state.get(n)
                vbuf = g.clone() if vbuf is None else vbuf.mul_(momentum).add_(g)
                mom_state[n] = vbuf
                self.v[n].sub_(lr * vbuf)
        for n in names:
            self.v[n].requires_grad_(False)
        self.step += 1
        return float(total_loss), parts, mom_state
