"""Oracle (test infrastructure): the SSD train step restated end to end on the CPU — torch fp32 for the
dense differentiable parts (truncated VGG-16, conv4_3 L2 normalisation, extra layers, multibox heads, loss:
`torch.autograd` supplies the reference gradients), numpy (oracle/ssd.py) for anchors, targets, proposals.
One image at a time like the reference (batch 1: luminoth/models/ssd/ssd.py:62-64); the batch loss is the
mean over images.

PARITY UNPINNED: the reference has no SSD tests and the slim / Sonnet convolution arithmetic is third party;
structure per SURVEY.md §8a S1-S6.  Citations relative to /root/reference/luminoth/.
"""
import numpy as np
import torch

from . import ssd as oss
from . import torch_ops as ot

VGG16_CFG = [('conv1', 2), ('conv2', 2), ('conv3', 3), ('conv4', 3), ('conv5', 3)]
EXTRA = [('conv6', 1, 6, 'SAME'), ('conv7', 1, 1, 'SAME'), ('conv8_1', 1, 1, 'SAME'), ('conv8_2', 2, 1, 'SAME'),
         ('conv9_1', 1, 1, 'SAME'), ('conv9_2', 2, 1, 'SAME'), ('conv10_1', 1, 1, 'SAME'),
         ('conv10_2', 1, 1, 'VALID'), ('conv11_1', 1, 1, 'SAME'), ('conv11_2', 1, 1, 'VALID')]
MAP_AFTER = ('conv7', 'conv8_2', 'conv9_2', 'conv10_2', 'conv11_2')


class OracleSSD(object):
    def __init__(self, variables, num_classes=20, scope='ssd', anchors_per_point=(4, 6, 6, 6, 4, 4),
                 ratios=(1, 0.5, 2, 0.333, 3), min_scale=0.1, max_scale=0.88, variances=(0.1, 0.2),
                 weight_decay=5e-4, loc_loss_weight=1.0, target=None, proposals=None, dtype=torch.float32):
        # dtype=torch.float64 (test-only): the dense path in double precision — is a gradient difference between two fp32
        # implementations the round-off of an ill-conditioned sum, or a bug?
        self.dtype = dtype
        self.v = {k: torch.as_tensor(v).clone().to(dtype) for k, v in variables.items()}
        self.C, self.scope = num_classes, scope
        self.fe = scope + '/ssd_feature_extractor'
        self.app, self.ratios = list(anchors_per_point), np.array(ratios)
        self.min_scale, self.max_scale, self.variances = min_scale, max_scale, tuple(variances)
        self.wd, self.w_loc = weight_decay, loc_loss_weight
        self.target_cfg = dict(hard_negative_ratio=3.0, foreground_threshold=0.5, background_threshold_high=0.2)
        self.target_cfg.update(target or {})
        self.prop_cfg = dict(class_nms_threshold=0.45, class_max_detections=100, total_max_detections=100,
                             min_prob_threshold=0.5)
        self.prop_cfg.update(proposals or {})
        # test-only knob (as oracle/model.py): {layer scope: that layer's output as the HIP kernels computed it} — every
        # ReLU decision is then the kernels' own, so the reference gradient is the derivative of the branch they were on
        self.masks = None

    def _relu(self, z, scope):
        yk = None if self.masks is None else self.masks.get(scope)
        if yk is None:
            return torch.relu(z)
        return z * (torch.as_tensor(yk).reshape(z.shape) > 0).to(z.dtype)

    # ---- feature extractor (feature_extractor.py:39-132, truncated_vgg.py:79-121) ------------------------
    def feature_maps(self, image):
        v = self.v
        net = image.unsqueeze(0)                      # no mean subtraction for 'truncated_vgg_16'
        maps = []
        p = self.fe + '/vgg_16'
        for bi, (name, reps) in enumerate(VGG16_CFG):
            for r in range(reps):
                s = '%s/%s/%s_%d' % (p, name, name, r + 1)
                net = self._relu(ot.conv2d_nhwc(net, v[s + '/weights'], 1, 1, 'SAME', bias=v[s + '/biases']), s)
                if name == 'conv4' and r == 2:
                    ss = (net * net).sum(dim=3, keepdim=True)
                    norm = net * torch.rsqrt(torch.clamp(ss, min=1e-12))          # tf.nn.l2_normalize
                    maps.append(norm * v[self.fe + '/conv_4_3_norm/gamma'])
            if bi < 4:
                net = ot.max_pool_nhwc(net, 2, 2, 'VALID')
        net = ot.max_pool_nhwc(net, 3, 1, 'SAME')                                  # pool5
        e = self.fe + '/extra_feature_layers'
        for name, stride, rate, pad in EXTRA:
            s = '%s/%s' % (e, name)
            net = self._relu(ot.conv2d_nhwc(net, v[s + '/w'], stride, rate, pad, bias=v[s + '/b']), s)
            if name in MAP_AFTER:
                maps.append(net)
        return maps

    def heads(self, maps):
        v, C = self.v, self.C
        offs, scores = [], []
        for i, fm in enumerate(maps):
            so = '%s/MultiBox_%d_offsets_conv' % (self.scope, i)
            sc = '%s/MultiBox_%d_classes_conv' % (self.scope, i)
            offs.append(ot.conv2d_nhwc(fm, v[so + '/w'], 1, 1, 'SAME', bias=v[so + '/b']).reshape(-1, 4))
            scores.append(ot.conv2d_nhwc(fm, v[sc + '/w'], 1, 1, 'SAME', bias=v[sc + '/b']).reshape(-1, C + 1))
        return torch.cat(offs, 0), torch.cat(scores, 0)

    def forward_image(self, image, gt=None, overrides=None):
        """image (H,W,3) tensor, gt (G,5) numpy.  overrides: {'labels','targets'} to pin the discrete stage."""
        H, W = image.shape[0], image.shape[1]
        image = image.to(self.dtype)
        maps = self.feature_maps(image)
        loc_pred, cls_pred = self.heads(maps)
        probs = torch.softmax(cls_pred, dim=1)
        anchors = oss.all_anchors([(m.shape[1], m.shape[2]) for m in maps], (H, W, 3), self.min_scale, self.max_scale,
                                  self.ratios, self.app)
        out = {'cls_pred': cls_pred, 'loc_pred': loc_pred, 'cls_prob': probs, 'anchors': anchors}
        if gt is not None:
            if overrides and 'labels' in overrides:
                labels, targets = overrides['labels'], overrides['targets']
            else:
                labels, targets = oss.ssd_target(probs.detach().float().numpy(), anchors, np.asarray(gt, np.float32),
                                                 variances=self.variances, **self.target_cfg)
            out['labels'], out['targets'] = labels, targets
            lt, tt = torch.as_tensor(labels), torch.as_tensor(targets).to(self.dtype)
            keep = lt >= 0
            pos = lt > 0
            ce = torch.nn.functional.cross_entropy(cls_pred[keep], lt[keep].long(), reduction='sum')
            reg = ot.smooth_l1(loc_pred[pos], tt[pos], 3.0).sum() if bool(pos.any()) else cls_pred.sum() * 0.0
            npos = int(pos.sum())
            out['cls_loss'], out['bbox_loss'], out['npos'] = ce, reg, npos
            out['loss'] = (ce + reg * self.w_loc) / float(npos) if npos else cls_pred.sum() * 0.0
        return out

    def regularization_loss(self):
        reg = 0.0
        for n, t in self.v.items():
            if '/vgg_16/' in n and n.endswith('/weights'):
                reg = reg + self.wd * (t.double() ** 2).sum() / 2
        return reg

    def trainable_names(self):
        return [n for n in self.v]
