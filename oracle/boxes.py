"""Oracle (test infrastructure): anchors, IoU and box transforms, float32 numpy.

Every function restates a reference function op-for-op in IEEE float32 (one
numpy op per TF op, so there is no FMA contraction and the operation order is
the reference's).  Citations are relative to /root/reference/.
"""
import numpy as np

F = np.float32


def generate_anchors_reference(base_size, aspect_ratios, scales):
    """luminoth/utils/anchors.py:4-52 (float64; ratio-major, scale-minor)."""
    scales_grid, aspect_ratios_grid = np.meshgrid(scales, aspect_ratios)
    base_scales = scales_grid.reshape(-1)
    base_aspect_ratios = aspect_ratios_grid.reshape(-1)
    aspect_ratio_sqrts = np.sqrt(base_aspect_ratios)
    heights = base_scales * aspect_ratio_sqrts * base_size
    widths = base_scales / aspect_ratio_sqrts * base_size
    anchors = np.column_stack([
        -(widths - 1) / 2, -(heights - 1) / 2,
        (widths - 1) / 2, (heights - 1) / 2,
    ])
    # anchors.py:42-43 uses np.int (== int64 truncation).
    real_heights = (anchors[:, 3] - anchors[:, 1]).astype(np.int64)
    real_widths = (anchors[:, 2] - anchors[:, 0]).astype(np.int64)
    if (real_widths == 0).any() or (real_heights == 0).any():
        raise ValueError(
            'base_size {} is too small for aspect_ratios and scales.'.format(
                base_size))
    return anchors


def generate_anchors(anchor_reference, feat_h, feat_w, stride):
    """luminoth/models/fasterrcnn/fasterrcnn.py:261-308.

    The float64 numpy reference is added to an int32 TF tensor, so TF converts
    the reference to int32 (truncation toward zero) and the result is INT32
    (pinned by fasterrcnn_test.py:285-295).  Row-major over (y, x), then anchor.
    """
    ref_i32 = np.trunc(anchor_reference).astype(np.int32)
    shift_x = np.arange(feat_w, dtype=np.int32) * np.int32(stride)
    shift_y = np.arange(feat_h, dtype=np.int32) * np.int32(stride)
    sx, sy = np.meshgrid(shift_x, shift_y)
    sx = sx.reshape(-1)
    sy = sy.reshape(-1)
    shifts = np.stack([sx, sy, sx, sy], axis=0).T  # (H*W, 4)
    all_anchors = ref_i32[None, :, :] + shifts[:, None, :]
    return all_anchors.reshape(-1, 4).astype(np.int32)


def bbox_overlap(b1, b2):
    """luminoth/utils/bbox_overlap.py:7-48 (`bbox_overlap_tf`), +1 convention.

    iou = max(inter / union, 0); 0/0 does not occur with valid boxes, and where
    inter == 0 the quotient is +-0 which the outer max maps to 0 (the numpy
    twin at :51-94 skips the division there: identical values).
    """
    b1 = np.asarray(b1, dtype=F)
    b2 = np.asarray(b2, dtype=F)
    x11, y11, x12, y12 = [b1[:, i:i + 1] for i in range(4)]
    x21, y21, x22, y22 = [b2[:, i:i + 1] for i in range(4)]
    xI1 = np.maximum(x11, x21.T)
    yI1 = np.maximum(y11, y21.T)
    xI2 = np.minimum(x12, x22.T)
    yI2 = np.minimum(y12, y22.T)
    inter = (np.maximum(xI2 - xI1 + F(1), F(0)) *
             np.maximum(yI2 - yI1 + F(1), F(0)))
    a1 = (x12 - x11 + F(1)) * (y12 - y11 + F(1))
    a2 = (x22 - x21 + F(1)) * (y22 - y21 + F(1))
    union = (a1 + a2.T) - inter
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = np.maximum(inter / union, F(0))
    return iou.astype(F)


def bbox_overlap_np(b1, b2):
    """luminoth/utils/bbox_overlap.py:51-94 — the numpy twin `lumi eval` uses (eval.py:577).  No dtype is forced:
    float32 detections against int32 ground truth promote to float64 where they meet, while each set's own area
    stays in its input dtype, exactly as numpy evaluates the reference expression."""
    b1, b2 = np.asarray(b1), np.asarray(b2)
    ix1 = np.maximum(b1[:, 0:1], b2[:, 0:1].T)
    iy1 = np.maximum(b1[:, 1:2], b2[:, 1:2].T)
    ix2 = np.minimum(b1[:, 2:3], b2[:, 2:3].T)
    iy2 = np.minimum(b1[:, 3:4], b2[:, 3:4].T)
    inter = np.maximum(ix2 - ix1 + 1, 0.) * np.maximum(iy2 - iy1 + 1, 0.)
    area1 = (b1[:, 2:3] - b1[:, 0:1] + 1) * (b1[:, 3:4] - b1[:, 1:2] + 1)
    area2 = (b2[:, 2:3] - b2[:, 0:1] + 1) * (b2[:, 3:4] - b2[:, 1:2] + 1)
    union = (area1 + area2.T) - inter
    iou = np.zeros((b1.shape[0], b2.shape[0]))
    np.divide(inter, union, out=iou, where=inter > 0.)
    return iou


def get_width_upright(b):
    """luminoth/utils/bbox_transform_tf.py:4-15."""
    b = np.asarray(b).astype(F)
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    w = x2 - x1 + F(1)
    h = y2 - y1 + F(1)
    return w, h, x1 + F(.5) * w, y1 + F(.5) * h


def encode(bboxes, gt_boxes, variances=None):
    """luminoth/utils/bbox_transform_tf.py:18-38."""
    w, h, cx, cy = get_width_upright(bboxes)
    gw, gh, gcx, gcy = get_width_upright(gt_boxes)
    v0, v1 = (F(1), F(1)) if variances is None else (F(variances[0]), F(variances[1]))
    with np.errstate(divide='ignore', invalid='ignore'):
        dx = (gcx - cx) / (w * v0)
        dy = (gcy - cy) / (h * v0)
        dw = np.log(gw / w) / v1
        dh = np.log(gh / h) / v1
    return np.stack([dx, dy, dw, dh], axis=1).astype(F)


def decode(roi, deltas, variances=None):
    """luminoth/utils/bbox_transform_tf.py:41-66 (note the extra -1 on x2,y2)."""
    w, h, cx, cy = get_width_upright(roi)
    d = np.asarray(deltas, dtype=F)
    v0, v1 = (F(1), F(1)) if variances is None else (F(variances[0]), F(variances[1]))
    px = d[:, 0] * w * v0 + cx
    py = d[:, 1] * h * v0 + cy
    pw = np.exp(d[:, 2] * v1) * w
    ph = np.exp(d[:, 3] * v1) * h
    x1 = px - F(.5) * pw
    y1 = py - F(.5) * ph
    x2 = px + F(.5) * pw - F(1)
    y2 = py + F(.5) * ph - F(1)
    return np.stack([x1, y1, x2, y2], axis=1).astype(F)


def clip_boxes(bboxes, imshape):
    """luminoth/utils/bbox_transform_tf.py:69-99; imshape = (H, W)."""
    b = np.asarray(bboxes).astype(F)
    height, width = F(imshape[0]), F(imshape[1])
    x1 = np.maximum(np.minimum(b[:, 0], width - F(1)), F(0))
    x2 = np.maximum(np.minimum(b[:, 2], width - F(1)), F(0))
    y1 = np.maximum(np.minimum(b[:, 1], height - F(1)), F(0))
    y2 = np.maximum(np.minimum(b[:, 3], height - F(1)), F(0))
    return np.stack([x1, y1, x2, y2], axis=1).astype(F)


def area_positive(b):
    """`max(x2-x1,0)*max(y2-y1,0) > 0` (NO +1):
    rpn_proposal.py:101-105, rcnn_proposal.py:97-102, ssd/proposal.py:88-93."""
    b = np.asarray(b, dtype=F)
    return (np.maximum(b[:, 2] - b[:, 0], F(0)) *
            np.maximum(b[:, 3] - b[:, 1], F(0))) > F(0)
