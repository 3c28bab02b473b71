"""Oracle (test infrastructure): numpy restatement of the SSD box stages of the
reference — anchors, targets with hard-negative mining, proposals, loss.

Pinned (round 3) by RUNNING the reference's own code in the build container:
anchors by tests/golden/make_golden_ref_numpy.py (numpy code), targets /
proposals / loss by tests/golden/make_golden_ref_tf.py (the TF-graph code of
target.py / proposal.py / ssd.py executed on an eager numpy `tf` stand-in) —
bit-exact labels, indices and fp32 values (tests/test_ref_tf_golden.py).  The
reference itself ships no SSD tests.
Citations relative to /root/reference/luminoth/.
"""
import numpy as np

from . import boxes as bx
from . import tfops

F = np.float32


# ---------------------------------------------------------------- anchors ----
def adjust_bboxes(bboxes, old_height, old_width, new_height, new_width):
    """models/ssd/utils.py:5-28 (numpy float64)."""
    x_min = bboxes[:, 0] / old_width
    y_min = bboxes[:, 1] / old_height
    x_max = bboxes[:, 2] / old_width
    y_max = bboxes[:, 3] / old_height
    return np.stack([x_min * new_width, y_min * new_height, x_max * new_width, y_max * new_height], axis=1)


def generate_anchors_reference(ratios, scales, num_anchors, feature_map_shape):
    """models/ssd/utils.py:31-61: first anchor sqrt(s_i*s_{i+1}) (last map: s*0.99), the others
    h = s/sqrt(r), w = s*sqrt(r) for ratios[:num_anchors-1]; centre 0.5, feature-map units."""
    heights = np.zeros(num_anchors)
    widths = np.zeros(num_anchors)
    if len(scales) > 1:
        widths[0] = heights[0] = np.sqrt(scales[0] * scales[1]) * feature_map_shape[0]
    else:
        heights[0] = scales[0] * feature_map_shape[0] * 0.99
        widths[0] = scales[0] * feature_map_shape[1] * 0.99
    ratios = ratios[:num_anchors - 1]
    heights[1:] = scales[0] / np.sqrt(ratios) * feature_map_shape[0]
    widths[1:] = scales[0] * np.sqrt(ratios) * feature_map_shape[1]
    x_center = y_center = 0.5
    return np.column_stack([x_center - widths / 2, y_center - heights / 2,
                            x_center + widths / 2, y_center + heights / 2])


def generate_anchors_per_feat_map(feature_map_shape, anchor_reference):
    """models/ssd/utils.py:95-145: integer grid shifts (x fastest), row-major over (y, x), anchors inner."""
    shift_x, shift_y = np.meshgrid(np.arange(feature_map_shape[1]), np.arange(feature_map_shape[0]))
    shift_x, shift_y = shift_x.reshape(-1), shift_y.reshape(-1)
    shifts = np.transpose(np.stack([shift_x, shift_y, shift_x, shift_y], axis=0))
    all_anchors = np.expand_dims(anchor_reference, axis=0) + np.expand_dims(shifts, axis=1)
    return np.reshape(all_anchors, (-1, 4))


def clip_boxes_np(bboxes, imshape):
    """utils/bbox_transform.py:105-122 (numpy twin used by ssd.py:125): clip to [0, dim-1]."""
    bboxes = bboxes.astype(np.float32)
    imshape = np.asarray(imshape, np.float32)
    max_width, max_height = imshape[1] - 1., imshape[0] - 1.
    min_width = min_height = 0.
    out = bboxes.copy()
    out[:, 0] = np.maximum(np.minimum(bboxes[:, 0], max_width), min_width)
    out[:, 1] = np.maximum(np.minimum(bboxes[:, 1], max_height), min_height)
    out[:, 2] = np.maximum(np.minimum(bboxes[:, 2], max_width), min_width)
    out[:, 3] = np.maximum(np.minimum(bboxes[:, 3], max_height), min_height)
    return out


def all_anchors(feat_shapes, image_shape, min_scale=0.1, max_scale=0.88, ratios=(1, 0.5, 2, 0.333, 3),
                anchors_per_point=(4, 6, 6, 6, 4, 4)):
    """models/ssd/ssd.py:111-129 + utils.py:64-92: (sum_i A_i*H_i*W_i, 4) float32, image coords, clipped."""
    ratios = np.array(ratios)
    scales = np.linspace(min_scale, max_scale, len(feat_shapes))
    out = []
    for i, shp in enumerate(feat_shapes):
        ref = generate_anchors_reference(ratios, scales[i:i + 2], anchors_per_point[i], shp)
        raw = generate_anchors_per_feat_map(shp, ref)
        scaled = adjust_bboxes(raw, shp[0], shp[1], image_shape[0], image_shape[1])
        out.append(clip_boxes_np(scaled, image_shape[:2]))
    return np.concatenate(out, axis=0).astype(np.float32)


# ----------------------------------------------------------------- target ----
def ssd_target(probs, anchors, gt_boxes, hard_negative_ratio=3.0, foreground_threshold=0.5,
               background_threshold_high=0.2, variances=(0.1, 0.2)):
    """models/ssd/target.py:35-200.  probs (N,C+1) softmax, anchors (N,4), gt (G,5).
    Returns labels (N,) in {-1, 0, 1..C} and bbox_targets (N,4)."""
    anchors = anchors.astype(F)
    gt_boxes = gt_boxes.astype(F)
    N = anchors.shape[0]
    labels = np.full((N,), -1., F)                                            # :70-74
    overlaps = bx.bbox_overlap(anchors, gt_boxes[:, :4])                       # :77 (N,G) fp32, +1 convention
    max_overlaps = overlaps.max(axis=1)                                        # :78
    best_gt = overlaps.argmax(axis=1)                                          # :81 first occurrence
    best_fg_labels = gt_boxes[:, 4][best_gt] + F(1.)                           # :85-88
    labels = np.where(max_overlaps >= F(foreground_threshold), best_fg_labels, labels)   # :89-96
    best_anchor_idxs = overlaps.argmax(axis=0)                                 # :99 per gt, first occurrence
    is_best = np.zeros((N,), bool)
    best_labels = np.full((N,), -1., F)
    for g, a in enumerate(best_anchor_idxs):                                   # sparse_to_dense, last write wins
        is_best[a] = True
        best_labels[a] = gt_boxes[g, 4] + F(1.)
    labels = np.where(is_best, best_labels, labels)                            # :117-122
    max_cls_probs = probs[:, 1:].max(axis=1).astype(F)                         # :125-126
    candidates = (max_overlaps <= F(background_threshold_high)) & (labels <= 0)   # :129-135
    max_cls_probs = np.where(candidates, max_cls_probs, F(-1.))                # :137-141
    num_fg = F(np.count_nonzero(labels > 0))                                   # :143-144
    num_bg = int(np.int32(num_fg * F(hard_negative_ratio)))                    # :146 float32 product, truncated
    _, idx = tfops.top_k(max_cls_probs, num_bg)                                # :147 ties -> lower index
    set_bg = np.zeros((N,), bool)
    set_bg[idx] = True
    labels = np.where(set_bg, F(0.), labels)                                   # :149-160 (may clear a fg row)
    with_target = labels > 0                                                   # :167-169
    targets = np.zeros((N, 4), F)
    if with_target.any():
        targets[with_target] = bx.encode(anchors[with_target], gt_boxes[best_gt[with_target], :4],
                                         variances=variances)                  # :184-196
    return labels.astype(F), targets


# --------------------------------------------------------------- proposal ----
def ssd_proposal(cls_prob, loc_pred, anchors, im_shape, num_classes, class_nms_threshold=0.45,
                 class_max_detections=100, total_max_detections=100, min_prob_threshold=0.5,
                 variances=(0.1, 0.2)):
    """models/ssd/proposal.py:41-171."""
    sel_boxes, sel_probs, sel_labels, sel_anchors = [], [], [], []
    raw = np.zeros((0, 4), F)
    for class_id in range(num_classes):
        p = cls_prob[:, class_id + 1]
        f = p >= F(min_prob_threshold)                                         # :74-79
        p, lp, an = p[f], loc_pred[f], anchors[f]
        raw = bx.decode(an, lp, variances)                                     # :83
        clipped = bx.clip_boxes(raw, im_shape)                                 # :85
        pf = bx.area_positive(clipped)                                         # :88-93 (no +1)
        boxes, p = clipped[pf], p[pf]
        keep = tfops.non_max_suppression(boxes[:, [1, 0, 3, 2]], p, class_max_detections, class_nms_threshold)
        sel_boxes.append(boxes[keep])
        sel_probs.append(p[keep])
        sel_labels.append(np.full((len(keep),), class_id, np.int32))
        sel_anchors.append(an[pf])                                             # :143 NOT gathered by the NMS indices
    boxes = np.concatenate(sel_boxes, 0) if sel_boxes else np.zeros((0, 4), F)
    probs = np.concatenate(sel_probs, 0) if sel_probs else np.zeros((0,), F)
    labels = np.concatenate(sel_labels, 0) if sel_labels else np.zeros((0,), np.int32)
    cat_anchors = np.concatenate(sel_anchors, 0) if sel_anchors else np.zeros((0, 4), F)
    k = min(total_max_detections, probs.shape[0])                              # :154-159
    vals, idx = tfops.top_k(probs, k)
    # 'anchors' (:162,170): the top-k indices address the concatenation of the NMS-SELECTED boxes, but are applied to
    # the concatenation of every class's FILTERED anchors (a longer list) — the reference's debug output is
    # misaligned; restated as is.  'raw_proposals' (:167) is the LAST class's unclipped decode.
    return {'objects': boxes[idx], 'labels': labels[idx], 'probs': vals, 'anchors': cat_anchors[idx],
            'raw_proposals': raw}


# ------------------------------------------------------------------- loss ----
def ssd_loss(cls_pred, loc_pred, cls_target, bbox_offsets_target, num_classes, loc_loss_weight=1.0, sigma=3.0):
    """models/ssd/ssd.py:197-300 on the ALREADY FILTERED rows (target >= 0, ssd.py:146-161); also accepts
    unfiltered rows (target -1 rows are dropped here the same way).  Returns (final, cls_sum, bbox_sum)."""
    keep = cls_target >= 0
    cls_pred, loc_pred = cls_pred[keep].astype(F), loc_pred[keep].astype(F)
    cls_target, tgt = cls_target[keep].astype(np.int32), bbox_offsets_target[keep].astype(F)
    onehot = np.eye(num_classes + 1, dtype=F)[cls_target]
    m = cls_pred.max(axis=1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(cls_pred - m).sum(axis=1))
    ce = lse - (cls_pred * onehot).sum(axis=1)                                  # softmax_cross_entropy_with_logits
    pos = cls_target > 0
    d = loc_pred[pos] - tgt[pos]
    a = np.abs(d)
    s2 = F(sigma) ** 2
    reg = np.where(a < F(1.0) / s2, F(0.5) * s2 * a * a, a - F(0.5) / s2).sum(axis=1)   # utils/losses.py:4-32
    cls_loss, bbox_loss = F(ce.sum()), F(reg.sum())
    npos = int(pos.sum())
    final = F((cls_loss + bbox_loss * F(loc_loss_weight)) / F(npos)) if npos else F(0.)   # ssd.py:252-270
    return final, cls_loss, bbox_loss
