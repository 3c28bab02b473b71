"""RPN — Region Proposal Network head (reference:
luminoth/models/fasterrcnn/rpn.py:19-309): 3x3 conv + activation, 1x1 cls /
bbox convs (Sonnet Conv2D: SAME / VALID, bias), proposals, anchor targets and
the RPN loss.  Variable names follow Sonnet: `<scope>/rpn/{conv,cls_conv,bbox_conv}/{w,b}`."""
import torch

from luminoth_amd import autograd as A
from luminoth_amd import kernels as K
from luminoth_amd.models.base.layers import ConvLayer
from luminoth_amd.models.fasterrcnn.rpn_proposal import RPNProposal
from luminoth_amd.models.fasterrcnn.rpn_target import RPNTarget
from luminoth_amd.utils.vars import get_activation_function, get_initializer


class RPN(object):
    def __init__(self, num_anchors, config, in_channels, debug=False, seed=None, name='rpn', scope='fasterrcnn'):
        self._num_anchors = num_anchors
        self._num_channels = config.num_channels
        self._kernel_shape = config.kernel_shape
        self._debug, self._seed, self._config = debug, seed, config
        self._l1_sigma = config.l1_sigma
        act = get_activation_function(config.activation_function)
        wd = float(config.l2_regularization_scale or 0.0)
        p = '%s/%s' % (scope, name)
        k = self._kernel_shape[0]
        self._rpn = ConvLayer(p + '/conv', in_channels, self._num_channels, k, act=act, norm='bias', wd=wd,
                              init=get_initializer(config.rpn_initializer, seed), weight_name='w', bias_name='b')
        self._rpn_cls = ConvLayer(p + '/cls_conv', self._num_channels, num_anchors * 2, 1, padding='VALID',
                                  act=None, norm='bias', wd=wd,
                                  init=get_initializer(config.cls_initializer, seed), weight_name='w', bias_name='b')
        self._rpn_bbox = ConvLayer(p + '/bbox_conv', self._num_channels, num_anchors * 4, 1, padding='VALID',
                                   act=None, norm='bias', wd=wd,
                                   init=get_initializer(config.bbox_initializer, seed), weight_name='w', bias_name='b')
        self.layers = [self._rpn, self._rpn_cls, self._rpn_bbox]
        self._proposal = RPNProposal(num_anchors, config.proposals, debug=debug)
        self._anchor_target = RPNTarget(num_anchors, config.target, seed=seed)

    def register(self, store):
        zeros = lambda shape, gen: torch.zeros(shape)
        for l in self.layers:
            store.add(l.w_name, (l.k, l.k, l.cin, l.cout), l.init, trainable=True, wd=l.wd)
            store.add(l.b_name, (l.cout,), zeros, trainable=True)

    def bind(self, store):
        for l in self.layers:
            l.bind(store, None)
        self._anchor = torch.zeros(1, device=store.flat.device, requires_grad=True)

    def heads(self, conv_feature_map):
        """3x3 conv + activation, then the 1x1 cls / bbox convs (rpn.py:148-172)."""
        B = conv_feature_map.shape[0]
        rpn_feature = A.conv(self._rpn, conv_feature_map, self._anchor)
        cls_orig = A.conv(self._rpn_cls, rpn_feature, self._anchor)      # (B,fh,fw,2A)
        bbox_orig = A.conv(self._rpn_bbox, rpn_feature, self._anchor)    # (B,fh,fw,4A)
        pred = {'rpn_cls_score': cls_orig.reshape(B, -1, 2),             # rpn.py:160
                'rpn_bbox_pred': bbox_orig.reshape(B, -1, 4)}            # rpn.py:169
        if self._debug:
            pred['rpn_feature'] = rpn_feature
        return pred

    # ---- the fused train step drives the head without torch.autograd (FasterRCNN._step_body) -----------------------
    def heads_fwd(self, feat):
        """`heads` as plain kernel calls: -> (pred dict, ctx for heads_bwd)."""
        B = feat.shape[0]
        rf, bits = self._rpn.forward(feat, want_bits=True, keep_v=True)
        cls = self._rpn_cls.forward(rf)
        bbox = self._rpn_bbox.forward(rf)
        pred = {'rpn_cls_score': cls.reshape(B, -1, 2), 'rpn_bbox_pred': bbox.reshape(B, -1, 4)}
        return pred, (feat, rf, bits, cls, bbox)

    def heads_bwd(self, ctx, d_cls, d_bbox, addend=None):
        """Gradient of the feature map through the three convolutions (their parameter gradients go to the flat
        buffer).  The two 1x1 heads read the same tensor: the second one's data gradient takes the first one's as its
        `addend` and applies the activation bit mask of the 3x3 convolution's output in the same epilogue, so what
        reaches the 3x3 layer already is g = (dx_cls + dx_bbox) * act'(rf) — TF's AddN + ReluGrad without a pass."""
        feat, rf, bits, cls, bbox = ctx
        d1, _ = self._rpn_cls.backward(rf, cls, d_cls.view_as(cls), need_dx=True)
        g, _ = self._rpn_bbox.backward(rf, bbox, d_bbox.view_as(bbox), need_dx=True, addend=d1, mask_bits=bits)
        # `addend`: the gradient another consumer of the feature map left (the RCNN branch), added in this store
        d_feat, _ = self._rpn.backward(feat, rf, g, need_dx=True, dy_is_g=bits is not None, addend=addend)
        return d_feat

    def targets(self, pred, anchor_ref_i32, feat_hw, stride, gt_boxes, gt_count, seeds, im_shape, out=None):
        labels, targets, max_ov = self._anchor_target(anchor_ref_i32, feat_hw, stride, gt_boxes, gt_count,
                                                      seeds, im_shape, out=out)
        pred['rpn_cls_target'] = labels
        pred['rpn_bbox_target'] = targets
        if self._debug:
            pred['rpn_max_overlap'] = max_ov

    def __call__(self, conv_feature_map, im_shape, anchor_ref_i32, stride, gt_boxes=None, gt_count=None,
                 seeds=None, is_training=False):
        fh, fw = conv_feature_map.shape[1], conv_feature_map.shape[2]
        pred = self.heads(conv_feature_map)
        prop = self._proposal(pred['rpn_cls_score'].detach(), pred['rpn_bbox_pred'].detach(), anchor_ref_i32,
                              (fh, fw), stride, im_shape)
        pred['rpn_cls_prob'] = prop['rpn_cls_prob']
        pred['proposals'] = prop['proposals']
        pred['scores'] = prop['scores']
        pred['num_proposals'] = prop['num_proposals']
        if self._debug:
            pred['proposal_prediction'] = prop
        if gt_boxes is not None:
            self.targets(pred, anchor_ref_i32, (fh, fw), stride, gt_boxes, gt_count, seeds, im_shape)
        return pred

    def loss_and_grads(self, prediction_dict, w_cls=1.0, w_reg=1.0):
        """`loss()` without the autograd detour: the loss kernel writes d(cls + reg)/d(scores, offsets) in the same
        launch, so the fused train step hands those straight to `torch.autograd.backward` of the head outputs instead of
        building sum / select / ones / multiply nodes around two scalars (nine tiny launches on the critical path).
        Returns (loss dict, (d_cls_score, d_bbox_pred)); same values as loss() + backward of the sum."""
        cs, bp = prediction_dict['rpn_cls_score'], prediction_dict['rpn_bbox_pred']
        losses, _, d_cls, d_bbox = K.rpn_loss(cs.detach().contiguous(), bp.detach().contiguous(),
                                              prediction_dict['rpn_cls_target'], prediction_dict['rpn_bbox_target'],
                                              float(self._l1_sigma), float(w_cls), float(w_reg), want_grad=True)
        return {'rpn_cls_loss': losses[0], 'rpn_reg_loss': losses[1]}, (d_cls, d_bbox)

    def loss(self, prediction_dict, w_cls=1.0, w_reg=1.0):
        """rpn.py:219-309.  Returns {'rpn_cls_loss','rpn_reg_loss'} already multiplied by
        the loss weights (fasterrcnn.py:183-186), batch mean over images."""
        losses = A.RpnLossFn.apply(prediction_dict['rpn_cls_score'], prediction_dict['rpn_bbox_pred'],
                                   prediction_dict['rpn_cls_target'], prediction_dict['rpn_bbox_target'],
                                   float(self._l1_sigma), float(w_cls), float(w_reg))
        return {'rpn_cls_loss': losses[0], 'rpn_reg_loss': losses[1]}
