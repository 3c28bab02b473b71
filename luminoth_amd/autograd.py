"""torch.autograd plumbing around the HIP kernels.

Each Function's forward/backward only launches kernels from
libluminoth_hip.so.  Parameter gradients are written by the kernels directly
into the flat gradient buffer (luminoth_amd/params.py) — autograd routes
activation gradients between Functions and nothing else.
"""
import torch

from luminoth_amd import kernels as K


class ConvFn(torch.autograd.Function):
    """One ConvLayer (conv + bias/BN + act) as an autograd node."""

    @staticmethod
    def forward(ctx, x, anchor, layer):
        x = x.contiguous()
        y = layer.forward(x, keep_v=True)
        ctx.layer, ctx.x, ctx.y = layer, x, y
        ctx.need_dx = x.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        dx, _ = ctx.layer.backward(ctx.x, ctx.y, dy.contiguous(), need_dx=ctx.need_dx)
        ctx.x = ctx.y = None
        return dx, None, None


def conv(layer, x, anchor):
    if torch.is_grad_enabled() and (layer.trainable or x.requires_grad):
        return ConvFn.apply(x, anchor, layer)
    return layer.forward(x.contiguous())


class RoiPoolFn(torch.autograd.Function):
    """ROIPoolingLayer (roi_pool.py:68-95): fused crop_and_resize + 2x2 max."""

    @staticmethod
    def forward(ctx, feat, rois, roi_count, im_shape, ph, pw):
        feat = feat.contiguous()
        out, argmax = K.roi_pool_fwd(feat, rois, roi_count, im_shape, ph, pw)
        ctx.save_for_backward(argmax, rois, roi_count)
        ctx.meta = (tuple(feat.shape), im_shape, ph, pw)
        return out

    @staticmethod
    def backward(ctx, dout):
        argmax, rois, roi_count = ctx.saved_tensors
        shape, im_shape, ph, pw = ctx.meta
        dfeat = K.roi_pool_bwd(dout.contiguous(), argmax, rois, roi_count, shape, im_shape, ph, pw)
        return dfeat, None, None, None, None, None


class RoiPoolMeanFn(torch.autograd.Function):
    """ROIPoolingLayer followed directly by tf.reduce_mean(pooled, [1, 2]) (roi_pool.py:68-95 + rcnn.py:185-188)."""

    @staticmethod
    def forward(ctx, feat, rois, roi_count, im_shape, ph, pw):
        feat = feat.contiguous()
        mean, argmax = K.roi_pool_mean_fwd(feat, rois, roi_count, im_shape, ph, pw)
        ctx.save_for_backward(argmax, rois, roi_count)
        ctx.meta = (tuple(feat.shape), im_shape, ph, pw)
        return mean

    @staticmethod
    def backward(ctx, dmean):
        argmax, rois, roi_count = ctx.saved_tensors
        shape, im_shape, ph, pw = ctx.meta
        dfeat = K.roi_pool_mean_bwd(dmean.contiguous(), argmax, rois, roi_count, shape, im_shape, ph, pw)
        return dfeat, None, None, None, None, None


class SpatialMeanFn(torch.autograd.Function):
    """tf.reduce_mean(features, [1, 2]) (rcnn.py:185-188)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return K.spatial_mean_fwd(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return K.spatial_mean_bwd(dy.contiguous(), ctx.shape)


class DropoutFn(torch.autograd.Function):
    """tf.nn.dropout(net, keep_prob) of the RCNN head (rcnn.py:196,218); the mask is regenerated from the seed."""

    @staticmethod
    def forward(ctx, x, keep_prob, seed):
        ctx.meta = (keep_prob, seed)
        return K.dropout(x.contiguous(), keep_prob, seed)

    @staticmethod
    def backward(ctx, dy):
        keep_prob, seed = ctx.meta
        return K.dropout(dy.contiguous(), keep_prob, seed), None, None


class RpnLossFn(torch.autograd.Function):
    """RPN.loss (rpn.py:219-309): returns (2,) = [w_cls*cls, w_reg*reg]."""

    @staticmethod
    def forward(ctx, cls_score, bbox_pred, labels, targets, sigma, w_cls, w_reg):
        losses, per_image, d_cls, d_bbox = K.rpn_loss(cls_score.contiguous(), bbox_pred.contiguous(), labels,
                                                      targets, sigma, w_cls, w_reg, want_grad=True)
        ctx.save_for_backward(d_cls, d_bbox)
        ctx.per_image = per_image
        return losses

    @staticmethod
    def backward(ctx, g):
        d_cls, d_bbox = ctx.saved_tensors
        return d_cls * g[0], d_bbox * g[1], None, None, None, None, None


class RcnnLossFn(torch.autograd.Function):
    """RCNN.loss (rcnn.py:255-411): returns (2,) = [w_cls*cls, w_reg*reg]."""

    @staticmethod
    def forward(ctx, cls_score, bbox_offsets, labels, targets, num_classes, sigma, w_cls, w_reg):
        losses, per_image, d_cls, d_off = K.rcnn_loss(cls_score.contiguous(), bbox_offsets.contiguous(), labels,
                                                      targets, num_classes, sigma, w_cls, w_reg, want_grad=True)
        ctx.save_for_backward(d_cls, d_off)
        ctx.per_image = per_image
        return losses

    @staticmethod
    def backward(ctx, g):
        d_cls, d_off = ctx.saved_tensors
        return d_cls * g[0], d_off * g[1], None, None, None, None, None, None


class MaxPoolFn(torch.autograd.Function):
    """tf.nn.max_pool NHWC (SSD: 2x2/2 VALID, 3x3/1 SAME)."""

    @staticmethod
    def forward(ctx, x, ksize, stride, padding):
        x = x.contiguous()
        y, geom = K.maxpool_fwd(x, ksize, stride, padding)
        ctx.save_for_backward(x, y)
        ctx.meta = (ksize, stride, geom)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        ksize, stride, geom = ctx.meta
        return K.maxpool_bwd(x, y, dy.contiguous(), ksize, stride, geom), None, None, None


class L2NormScaleFn(torch.autograd.Function):
    """tf.nn.l2_normalize(x, 3, eps) * gamma (ssd/feature_extractor.py:75-89); dgamma goes straight into
    the flat gradient buffer view `ggamma`."""

    @staticmethod
    def forward(ctx, x, anchor, gamma, ggamma, eps):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.gamma, ctx.ggamma, ctx.eps = gamma, ggamma, eps
        return K.l2norm_scale_fwd(x, gamma, eps)

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        dx, _ = K.l2norm_scale_bwd(x, dy.contiguous(), ctx.gamma, ctx.eps, dgamma=ctx.ggamma)
        return dx, None, None, None, None


class SsdLossFn(torch.autograd.Function):
    """SSD.loss (ssd/ssd.py:197-300): returns (3,) = [final, cls_sum, bbox_sum] (batch means);
    only losses[0] carries gradient."""

    @staticmethod
    def forward(ctx, cls_pred, loc_pred, labels, targets, num_classes, sigma, w_loc):
        losses, per_image, d_cls, d_loc = K.ssd_loss(cls_pred.contiguous(), loc_pred.contiguous(), labels, targets,
                                                     num_classes, sigma, w_loc, want_grad=True)
        ctx.save_for_backward(d_cls, d_loc)
        ctx.per_image = per_image
        return losses

    @staticmethod
    def backward(ctx, g):
        d_cls, d_loc = ctx.saved_tensors
        return d_cls * g[0], d_loc * g[0], None, None, None, None, None
