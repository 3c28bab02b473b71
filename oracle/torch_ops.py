"""Oracle (test infrastructure): differentiable torch-CPU fp32 restatements of
the dense / gather stages, so `torch.autograd` provides reference gradients for
the HIP backward kernels and the CPU-baseline train step.

Citations relative to /root/reference/luminoth/.
"""
import torch
import torch.nn.functional as F


def tf_same_pad(in_size, k, stride, dilation=1):
    out = -(-in_size // stride)
    eff = (k - 1) * dilation + 1
    total = max((out - 1) * stride + eff - in_size, 0)
    return total // 2, total - total // 2


def _q(t, quant):
    """Round to the half-precision operand format and back (RNE, like v_cvt_pk_* / the compiler's fp32 -> f16 / bf16)."""
    if quant == 'f16':      # the kernels saturate at the largest finite f16 instead of producing inf (conv_hs.h, halfstore.hip)
        return t.clamp(-65504.0, 65504.0).to(torch.float16).to(t.dtype)
    return t.to(torch.bfloat16).to(t.dtype)


class QuantConvFn(torch.autograd.Function):
    """Convolution whose MFMA OPERANDS are rounded to f16 / bf16 while tensors, accumulation and everything around it
    stay fp32 — the arithmetic of luminoth_amd/csrc/conv_half.h (BASELINE configs[4]), restated so that the
    half-precision path is compared against an oracle that rounds the SAME operands:
        forward   y  = conv(q(x), q(w))
        backward  dx = conv^T(q(dy * s), q(w)) / s        dw = corr(q(x), q(dy * s)) / s
    with the static loss scale s = 2^10 for f16 (the kernels multiply the gradient operand by it before rounding and the
    accumulators by 1/s afterwards; exact in fp32) and 1 for bf16."""

    @staticmethod
    def forward(ctx, xt, wt, stride, dilation, quant):
        xq, wq = _q(xt, quant), _q(wt, quant)
        ctx.save_for_backward(xq, wq)
        ctx.meta = (stride, dilation, quant)
        return F.conv2d(xq, wq, stride=stride, dilation=dilation)

    @staticmethod
    def backward(ctx, dy):
        xq, wq = ctx.saved_tensors
        stride, dilation, quant = ctx.meta
        s = 1024.0 if quant == 'f16' else 1.0
        gq = _q(dy * s, quant)
        dx, dw, _ = torch.ops.aten.convolution_backward(gq, xq, wq, None, [stride, stride], [0, 0], [dilation, dilation],
                                                        False, [0, 0], 1, [True, True, False])
        return dx / s, dw / s, None, None, None


def _conv_pads(H, W, R, S, stride, dilation, padding):
    if padding == 'SAME':
        pt, pb = tf_same_pad(H, R, stride, dilation)
        pl, pr = tf_same_pad(W, S, stride, dilation)
    elif padding == 'SAME_EXPLICIT':  # slim resnet_utils.conv2d_same
        ke_h, ke_w = (R - 1) * dilation + 1, (S - 1) * dilation + 1
        pt, pl = (ke_h - 1) // 2, (ke_w - 1) // 2
        pb, pr = ke_h - 1 - pt, ke_w - 1 - pl
    else:
        pt = pb = pl = pr = 0
    return pt, pb, pl, pr


class HalfStorageConvFn(torch.autograd.Function):
    """One layer of the half-STORAGE trunk (luminoth_amd/csrc/conv_hs.h; BASELINE configs[4]) restated: 16-bit tensors in
    memory, fp32 accumulation, conv + frozen-BatchNorm affine (+ residual) (+ ReLU) fused, the result rounded when stored:
        forward    y  = q( act( conv(x, q(w)) * scale + shift + residual ) )           (unrounded when out_f32)
        backward   G  = q( S * dy * act'(y) )                   the 16-bit gradient tensor the kernels hold (S = loss scale)
                   dx = conv^T(G, q(w * scale)) / S             (rounded like G when `round_dx`: a shortcut convolution's
                                                                 data gradient is stored on its own before it is added)
                   dw = corr(x, G) / S * scale,  dscale = sum G * conv / S,  dshift = sum G / S,  dresidual = G / S
    x and residual are expected to hold half-representable values already (outputs of other half-storage layers).
    `yk`: that layer's output as the kernels computed it (ReLU branch decisions, oracle/model.py `masks`), or None."""

    @staticmethod
    def forward(ctx, x, w_hwio, scale, shift, residual, yk, cfg):
        quant, stride, dil = cfg['quant'], cfg['stride'], cfg['dilation']
        R, S_ = w_hwio.shape[0], w_hwio.shape[1]
        pads = _conv_pads(x.shape[1], x.shape[2], R, S_, stride, dil, cfg['padding'])
        pt, pb, pl, pr = pads
        xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
        z = F.conv2d(xp, _q(w_hwio, quant).permute(3, 2, 0, 1), stride=stride, dilation=dil).permute(0, 2, 3, 1)
        pre = z * scale + shift
        if residual is not None:
            pre = pre + residual
        if cfg['act'] == 'relu':
            stored = torch.relu(pre) if cfg['out_f32'] else _q(torch.relu(pre), quant)
            mask = ((torch.as_tensor(yk).reshape(pre.shape) if yk is not None else stored) > 0).to(pre.dtype)
            y = pre * mask
        else:
            assert not cfg['act'], cfg['act']
            mask = torch.ones_like(pre)
            y = pre
        if not cfg['out_f32']:
            y = _q(y, quant)
        ctx.save_for_backward(xp, w_hwio, scale, z, mask)
        ctx.cfg, ctx.pads, ctx.has_res = cfg, pads, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, w_hwio, scale, z, mask = ctx.saved_tensors
        cfg = ctx.cfg
        quant, stride, dil, S = cfg['quant'], cfg['stride'], cfg['dilation'], cfg['loss_scale']
        pt, pb, pl, pr = ctx.pads
        G = _q(dy * mask * S, quant)
        Gt = G.permute(0, 3, 1, 2).contiguous()
        wb = _q(w_hwio * scale, quant).permute(3, 2, 0, 1).contiguous()
        dxp, dwt, _ = torch.ops.aten.convolution_backward(Gt, xp, wb, None, [stride, stride], [0, 0], [dil, dil], False,
                                                          [0, 0], 1, [True, True, False])
        dx = dxp[:, :, pt:dxp.shape[2] - pb, pl:dxp.shape[3] - pr].permute(0, 2, 3, 1)
        dx = _q(dx, quant) / S if cfg.get('round_dx') else dx / S
        dw = dwt.permute(2, 3, 1, 0) / S * scale
        Gf = G / S
        return (dx, dw, (Gf * z).sum(dim=(0, 1, 2)), Gf.sum(dim=(0, 1, 2)), Gf if ctx.has_res else None, None, None)


def conv2d_nhwc(x, w_hwio, stride=1, dilation=1, padding='SAME', bias=None, quant=None):
    """tf.nn.conv2d / slim conv2d / conv2d_same on NHWC input with HWIO weights.  quant: 'f16' / 'bf16' rounds the two
    operands like the mixed-precision kernels do (QuantConvFn); None = plain fp32."""
    R, S = w_hwio.shape[0], w_hwio.shape[1]
    xt = x.permute(0, 3, 1, 2)
    if padding == 'SAME':
        pt, pb = tf_same_pad(x.shape[1], R, stride, dilation)
        pl, pr = tf_same_pad(x.shape[2], S, stride, dilation)
    elif padding == 'SAME_EXPLICIT':  # slim resnet_utils.conv2d_same
        ke_h, ke_w = (R - 1) * dilation + 1, (S - 1) * dilation + 1
        pt, pl = (ke_h - 1) // 2, (ke_w - 1) // 2
        pb, pr = ke_h - 1 - pt, ke_w - 1 - pl
    else:
        pt = pb = pl = pr = 0
    if pt or pb or pl or pr:
        xt = F.pad(xt, (pl, pr, pt, pb))
    if quant:
        y = QuantConvFn.apply(xt, w_hwio.permute(3, 2, 0, 1), stride, dilation, quant)
        if bias is not None:
            y = y + bias.reshape(1, -1, 1, 1)
    else:
        y = F.conv2d(xt, w_hwio.permute(3, 2, 0, 1), bias=bias, stride=stride, dilation=dilation)
    return y.permute(0, 2, 3, 1)


def max_pool_nhwc(x, k, stride, padding='SAME'):
    xt = x.permute(0, 3, 1, 2)
    if padding == 'SAME':
        pt, pb = tf_same_pad(x.shape[1], k, stride)
        pl, pr = tf_same_pad(x.shape[2], k, stride)
        xt = F.pad(xt, (pl, pr, pt, pb), value=float('-inf'))
    return F.max_pool2d(xt, k, stride).permute(0, 2, 3, 1)


def frozen_batch_norm(x, gamma, beta, mean, var, eps=1e-5):
    """slim batch_norm in inference mode (base_network.py:84-89: is_training False)."""
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def crop_and_resize(feat, boxes_norm, box_ind, crop_h, crop_w):
    """tf.image.crop_and_resize, bilinear, extrapolation 0 (vectorised; same
    arithmetic as oracle/tfops.py::crop_and_resize).  feat (B,H,W,C)."""
    B, H, W, C = feat.shape
    R = boxes_norm.shape[0]
    y1, x1, y2, x2 = boxes_norm[:, 0], boxes_norm[:, 1], boxes_norm[:, 2], boxes_norm[:, 3]
    hs = (y2 - y1) * float(H - 1) / float(crop_h - 1)
    ws = (x2 - x1) * float(W - 1) / float(crop_w - 1)
    ys = torch.arange(crop_h, dtype=torch.float32)
    xs = torch.arange(crop_w, dtype=torch.float32)
    in_y = (y1 * float(H - 1))[:, None] + ys[None, :] * hs[:, None]      # (R, ch)
    in_x = (x1 * float(W - 1))[:, None] + xs[None, :] * ws[:, None]      # (R, cw)
    vy = ~((in_y < 0) | (in_y > H - 1))
    vx = ~((in_x < 0) | (in_x > W - 1))
    top = torch.floor(in_y).clamp(0, H - 1).long()
    bot = torch.ceil(in_y).clamp(0, H - 1).long()
    left = torch.floor(in_x).clamp(0, W - 1).long()
    right = torch.ceil(in_x).clamp(0, W - 1).long()
    yl = (in_y - torch.floor(in_y))[:, :, None, None]
    xl = (in_x - torch.floor(in_x))[:, None, :, None]
    bi = box_ind.long()[:, None, None]

    def g(yy, xx):
        return feat[bi, yy[:, :, None], xx[:, None, :]]                  # (R, ch, cw, C)
    tl, tr, bl, br = g(top, left), g(top, right), g(bot, left), g(bot, right)
    t = tl + (tr - tl) * xl
    b = bl + (br - bl) * xl
    out = t + (b - t) * yl
    valid = (vy[:, :, None] & vx[:, None, :])[..., None]
    return torch.where(valid, out, torch.zeros_like(out))


def roi_pool(feat, rois, box_ind, im_shape, pooled_h=7, pooled_w=7):
    """models/fasterrcnn/roi_pool.py:37-95: normalise by (H, W), crop 2x, 2x2 max pool."""
    H, W = float(im_shape[0]), float(im_shape[1])
    bn = torch.stack([rois[:, 1] / H, rois[:, 0] / W, rois[:, 3] / H, rois[:, 2] / W], dim=1)
    crops = crop_and_resize(feat, bn, box_ind, pooled_h * 2, pooled_w * 2)
    p = F.max_pool2d(crops.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    return p


def smooth_l1(pred, target, sigma):
    s2 = sigma ** 2
    a = (pred - target).abs()
    return torch.where(a < 1.0 / s2, 0.5 * s2 * a * a, a - 0.5 / s2).sum(dim=1)


def rpn_loss(cls_score, bbox_pred, labels, bbox_targets, sigma=3.0):
    """models/fasterrcnn/rpn.py:219-309 for one image (tensors (N,2),(N,4),(N,),(N,4))."""
    ni = labels != -1
    ce = F.cross_entropy(cls_score[ni], labels[ni].long(), reduction='none')
    pos = labels == 1
    reg = smooth_l1(bbox_pred[pos], bbox_targets[pos], sigma)
    return ce.mean(), reg.mean()


def rcnn_loss(cls_score, bbox_offsets, labels, targets, num_classes, sigma=1.0):
    """models/fasterrcnn/rcnn.py:255-411 for one image."""
    ni = labels >= 0
    ce = F.cross_entropy(cls_score[ni], labels[ni].long(), reduction='none')
    fg = labels > 0
    cls = labels[fg].long() - 1
    bo = bbox_offsets[fg].reshape(-1, num_classes, 4)
    cleaned = bo[torch.arange(bo.shape[0]), cls]
    reg = smooth_l1(cleaned, targets[fg], sigma)
    return ce.mean(), reg.mean()
