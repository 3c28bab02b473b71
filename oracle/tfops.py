"""Oracle (test infrastructure): restatement of the TensorFlow-1.x op semantics
the hot path relies on.  TensorFlow itself is a third-party dependency that is
NOT under /root/reference (setup.py:107-108 `tensorflow>=1.5`; not installed
here), so these follow the published TF 1.x CPU kernel algorithms and are
anchored on the reference's call sites and tests.
"""
import numpy as np

F = np.float32


def top_k(values, k):
    """tf.nn.top_k: descending values, ties -> lower index first.
    Call sites: rpn_proposal.py:139-143, rcnn_proposal.py:152, ssd/target.py:143."""
    values = np.asarray(values)
    order = np.argsort(-values.astype(np.float64), kind='stable')[:k]
    return values[order], order.astype(np.int64)


def nms_iou_greater(box_i, box_j, thr):
    """TF 1.x non_max_suppression_op.cc `IOUGreaterThanThreshold`
    (boxes [y1,x1,y2,x2], min/max normalised; continuous areas, no +1)."""
    ymin_i, ymax_i = min(box_i[0], box_i[2]), max(box_i[0], box_i[2])
    xmin_i, xmax_i = min(box_i[1], box_i[3]), max(box_i[1], box_i[3])
    ymin_j, ymax_j = min(box_j[0], box_j[2]), max(box_j[0], box_j[2])
    xmin_j, xmax_j = min(box_j[1], box_j[3]), max(box_j[1], box_j[3])
    area_i = F(F(ymax_i - ymin_i) * F(xmax_i - xmin_i))
    area_j = F(F(ymax_j - ymin_j) * F(xmax_j - xmin_j))
    if area_i <= 0 or area_j <= 0:
        return False
    iy1, ix1 = max(ymin_i, ymin_j), max(xmin_i, xmin_j)
    iy2, ix2 = min(ymax_i, ymax_j), min(xmax_i, xmax_j)
    inter = F(max(F(iy2 - iy1), F(0)) * max(F(ix2 - ix1), F(0)))
    iou = F(inter / F(F(area_i + area_j) - inter))
    return bool(iou > F(thr))


def non_max_suppression(boxes_tf, scores, max_output_size, iou_threshold):
    """tf.image.non_max_suppression (greedy; descending score, ties -> lower
    index [unpinned in TF, oracle convention]; suppress iff IoU > thr STRICT).

    Vectorised over the kept set; same arithmetic as `nms_iou_greater`.
    Call sites: rpn_proposal.py:152-157, rcnn_proposal.py:114-117,
    ssd/proposal.py:123-126.  Returns selected indices (int64).
    """
    boxes_tf = np.asarray(boxes_tf, dtype=F).reshape(-1, 4)
    scores = np.asarray(scores, dtype=F).reshape(-1)
    n = boxes_tf.shape[0]
    if n == 0 or max_output_size <= 0:
        return np.zeros((0,), dtype=np.int64)
    thr = F(iou_threshold)
    ymin = np.minimum(boxes_tf[:, 0], boxes_tf[:, 2])
    ymax = np.maximum(boxes_tf[:, 0], boxes_tf[:, 2])
    xmin = np.minimum(boxes_tf[:, 1], boxes_tf[:, 3])
    xmax = np.maximum(boxes_tf[:, 1], boxes_tf[:, 3])
    area = (ymax - ymin) * (xmax - xmin)
    order = np.argsort(-scores.astype(np.float64), kind='stable')
    kept = np.empty((min(max_output_size, n),), dtype=np.int64)
    nk = 0
    for c in order:
        if nk >= max_output_size:
            break
        if nk > 0:
            k = kept[:nk]
            iy1 = np.maximum(ymin[c], ymin[k])
            ix1 = np.maximum(xmin[c], xmin[k])
            iy2 = np.minimum(ymax[c], ymax[k])
            ix2 = np.minimum(xmax[c], xmax[k])
            inter = np.maximum(iy2 - iy1, F(0)) * np.maximum(ix2 - ix1, F(0))
            with np.errstate(divide='ignore', invalid='ignore'):
                iou = inter / ((area[c] + area[k]) - inter)
            valid = (area[c] > 0) & (area[k] > 0)
            if np.any(valid & (iou > thr)):
                continue
        kept[nk] = c
        nk += 1
    return kept[:nk]


def crop_and_resize(image, boxes, box_ind, crop_size):
    """tf.image.crop_and_resize (bilinear, extrapolation_value 0), numpy fp32.

    image (B,H,W,C); boxes (R,4) normalised [y1,x1,y2,x2]; crop_size (ch, cw).
    TF 1.x crop_and_resize_op.cc: height_scale = (y2-y1)*(H-1)/(ch-1);
    in_y = y1*(H-1) + y*height_scale; out of [0,H-1] -> 0; lerp between
    floor/ceil.  Call site: luminoth/models/fasterrcnn/roi_pool.py:75-78.
    """
    image = np.asarray(image, dtype=F)
    boxes = np.asarray(boxes, dtype=F)
    B, H, W, C = image.shape
    ch, cw = crop_size
    R = boxes.shape[0]
    out = np.zeros((R, ch, cw, C), dtype=F)
    for r in range(R):
        y1, x1, y2, x2 = boxes[r]
        b = int(box_ind[r])
        hs = F((y2 - y1) * F(H - 1) / F(ch - 1)) if ch > 1 else F(0)
        ws = F((x2 - x1) * F(W - 1) / F(cw - 1)) if cw > 1 else F(0)
        for y in range(ch):
            in_y = F(y1 * F(H - 1) + F(y) * hs) if ch > 1 else F(F(.5) * (y1 + y2) * F(H - 1))
            if in_y < 0 or in_y > H - 1:
                continue
            top = int(np.floor(in_y))
            bot = int(np.ceil(in_y))
            ylerp = F(in_y - F(top))
            for x in range(cw):
                in_x = F(x1 * F(W - 1) + F(x) * ws) if cw > 1 else F(F(.5) * (x1 + x2) * F(W - 1))
                if in_x < 0 or in_x > W - 1:
                    continue
                left = int(np.floor(in_x))
                right = int(np.ceil(in_x))
                xlerp = F(in_x - F(left))
                tl = image[b, top, left]
                tr = image[b, top, right]
                bl = image[b, bot, left]
                br = image[b, bot, right]
                t = tl + (tr - tl) * xlerp
                bt = bl + (br - bl) * xlerp
                out[r, y, x] = t + (bt - t) * ylerp
    return out


def max_pool_2x2_valid(x):
    """tf.nn.max_pool ksize 2, stride 2, VALID (roi_pool.py:83-87)."""
    R, H, W, C = x.shape
    H2, W2 = H // 2, W // 2
    x = x[:, :H2 * 2, :W2 * 2].reshape(R, H2, 2, W2, 2, C)
    return x.max(axis=(2, 4))


def softmax(x):
    """tf.nn.softmax (last axis), max-subtracted, fp32."""
    x = np.asarray(x, dtype=F)
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=-1, keepdims=True)).astype(F)
