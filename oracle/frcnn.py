"""Oracle (test infrastructure): Faster R-CNN proposal / target / ROI / loss
stages restated per image in numpy float32.  Citations relative to
/root/reference/luminoth/.
"""
import numpy as np

from . import boxes as bx
from . import tfops
from . import rng

F = np.float32


# --------------------------------------------------------------------------
# A5  RPNProposal._build   models/fasterrcnn/rpn_proposal.py:41-197
# --------------------------------------------------------------------------
def rpn_proposal(rpn_cls_prob, rpn_bbox_pred, all_anchors, im_shape,
                 pre_nms_top_n=12000, post_nms_top_n=2000, nms_threshold=0.7,
                 apply_nms=True, clip_after_nms=False,
                 filter_outside_anchors=False, min_prob_threshold=0.0):
    all_scores = np.asarray(rpn_cls_prob, dtype=F)[:, 1].reshape(-1)
    all_anchors = np.asarray(all_anchors)
    rpn_bbox_pred = np.asarray(rpn_bbox_pred, dtype=F)
    if filter_outside_anchors:  # rpn_proposal.py:69-90
        f = ((all_anchors[:, 0] >= 0) & (all_anchors[:, 1] >= 0) &
             (all_anchors[:, 2] < im_shape[1]) & (all_anchors[:, 3] < im_shape[0]))
        all_anchors, rpn_bbox_pred, all_scores = all_anchors[f], rpn_bbox_pred[f], all_scores[f]
    all_proposals = bx.decode(all_anchors, rpn_bbox_pred)          # :93
    min_prob_filter = all_scores >= F(min_prob_threshold)          # :96-98
    proposal_filter = bx.area_positive(all_proposals) & min_prob_filter  # :101-106
    unsorted_scores = all_scores[proposal_filter]
    unsorted_proposals = all_proposals[proposal_filter]
    if not clip_after_nms:
        unsorted_proposals = bx.clip_boxes(unsorted_proposals, im_shape)  # :121-123
    k = min(pre_nms_top_n, unsorted_scores.shape[0])               # :139
    sorted_top_scores, idx = tfops.top_k(unsorted_scores, k)
    sorted_top_proposals = unsorted_proposals[idx]
    if apply_nms:
        tf_order = sorted_top_proposals[:, [1, 0, 3, 2]]           # change_order
        sel = tfops.non_max_suppression(tf_order, sorted_top_scores,
                                        post_nms_top_n, nms_threshold)  # :152-157
        proposals = sorted_top_proposals[sel]
        scores = sorted_top_scores[sel]
    else:
        proposals, scores, sel = sorted_top_proposals, sorted_top_scores, None
    if clip_after_nms:
        proposals = bx.clip_boxes(proposals, im_shape)
    return {
        'proposals': proposals.astype(F), 'scores': scores.astype(F),
        'sorted_top_scores': sorted_top_scores, 'sorted_top_proposals': sorted_top_proposals,
        'unsorted_proposals': unsorted_proposals, 'unsorted_scores': unsorted_scores,
        'all_proposals': all_proposals, 'all_scores': all_scores,
        'proposal_filter': proposal_filter, 'top_k_indices': idx, 'nms_indices': sel,
    }


# --------------------------------------------------------------------------
# A6  RPNTarget._build   models/fasterrcnn/rpn_target.py:73-335
# --------------------------------------------------------------------------
def rpn_target(all_anchors, gt_boxes, im_shape, seed=0, allowed_border=0,
               clobber_positives=False, foreground_threshold=0.7,
               background_threshold_high=0.3, foreground_fraction=0.5,
               minibatch_size=256, return_pre_subsample=False):
    all_anchors = np.asarray(all_anchors)[:, :4]
    gt = np.asarray(gt_boxes, dtype=F)[:, :4]
    N = all_anchors.shape[0]
    H, W = im_shape[0], im_shape[1]
    b = allowed_border
    anchor_filter = ((all_anchors[:, 0] >= -b) & (all_anchors[:, 1] >= -b) &
                     (all_anchors[:, 2] < W + b) & (all_anchors[:, 3] < H + b))  # :111-123
    anchors = all_anchors[anchor_filter].astype(F)
    labels = np.full((anchors.shape[0],), -1, dtype=F)
    overlaps = bx.bbox_overlap(anchors, gt)                         # :137
    max_overlaps = overlaps.max(axis=1) if overlaps.size else np.zeros((anchors.shape[0],), F)
    neg = max_overlaps < F(background_threshold_high)
    if not clobber_positives:
        labels[neg] = 0                                              # :142-153
    gt_max_overlaps = overlaps.max(axis=0)                           # :155
    is_gt_argmax = (overlaps == gt_max_overlaps[None, :]).any(axis=1)  # :158-171 (where/unique)
    labels[is_gt_argmax] = 1                                         # :173-178
    labels[max_overlaps >= F(foreground_threshold)] = 1              # :183-189
    if clobber_positives:
        labels[neg] = 0                                              # :191-202
    labels_pre = labels.copy()
    # Subsample positives (:203-241) then negatives (:243-284).
    # The random choice is keyed on the anchor's index in the FULL grid (oracle/rng.py).
    inds = np.where(anchor_filter)[0]
    num_fg = int(foreground_fraction * minibatch_size)
    fg_inds = np.where(labels == 1)[0]
    if fg_inds.size > num_fg:
        keep = rng.keep_k_smallest(inds[fg_inds], num_fg, seed, rng.STREAM_RPN_FG)
        labels[fg_inds[~keep]] = -1
    num_bg = int(minibatch_size - np.sum(labels == 1))
    bg_inds = np.where(labels == 0)[0]
    if bg_inds.size > num_bg:
        keep = rng.keep_k_smallest(inds[bg_inds], num_bg, seed, rng.STREAM_RPN_BG)
        labels[bg_inds[~keep]] = -1
    if gt.shape[0] == 0:
        # no gt box at all: the reference's graph fails here (tf.argmax over an empty axis, :289); the rebuild defines
        # the image as background-only (every inside anchor IoU 0 -> label 0, no targets) — kernel and oracle alike
        argmax_overlaps = np.zeros((anchors.shape[0],), np.int64)
        bbox_targets = np.zeros((anchors.shape[0], 4), F)
    else:
        argmax_overlaps = overlaps.argmax(axis=1)                    # :289
        bbox_targets = bx.encode(anchors, gt[argmax_overlaps])       # :295-297
    bbox_targets = np.where((labels == 1)[:, None], bbox_targets, F(0)).astype(F)  # :299-304
    # Scatter back to all anchors (:311-333).
    out_t = np.zeros((N, 4), dtype=F)
    out_t[inds] = bbox_targets
    out_l = np.full((N,), -1, dtype=F)
    out_l[inds] = labels
    out_m = np.zeros((N,), dtype=F)
    out_m[inds] = max_overlaps
    if return_pre_subsample:
        pre = np.full((N,), -1, dtype=F)
        pre[inds] = labels_pre
        am = np.zeros((N,), dtype=np.int32)
        am[inds] = argmax_overlaps
        return out_l, out_t, out_m, pre, am
    return out_l, out_t, out_m


# --------------------------------------------------------------------------
# A10  RCNNTarget._build   models/fasterrcnn/rcnn_target.py:48-299
# --------------------------------------------------------------------------
def rcnn_target(proposals, gt_boxes, seed=0, foreground_fraction=0.25,
                minibatch_size=256, foreground_threshold=0.5,
                background_threshold_high=0.5, background_threshold_low=0.0,
                variances=(0.1, 0.2), return_pre_subsample=False):
    proposals = np.asarray(proposals, dtype=F)
    gt_boxes = np.asarray(gt_boxes, dtype=F)
    P = proposals.shape[0]
    overlaps = bx.bbox_overlap(proposals, gt_boxes[:, :4])           # :66
    label = np.full((P,), -1, dtype=F)
    max_overlaps = overlaps.max(axis=1)
    bg_cond = (max_overlaps >= F(background_threshold_low)) & \
              (max_overlaps < F(background_threshold_high))          # :89-102
    label[bg_cond] = 0
    best_gt = overlaps.argmax(axis=1)                                # :105
    best_fg_labels = gt_boxes[:, 4][best_gt] + F(1)                  # :109-112
    iou_is_fg = max_overlaps >= F(foreground_threshold)              # :113-115
    best_prop = overlaps.argmax(axis=0)                              # :116 (first occurrence)
    is_best_box = np.zeros((P,), dtype=bool)
    is_best_box[best_prop] = True                                    # :124-129
    label = np.where(iou_is_fg, best_fg_labels, label)               # :132-136
    best_prop_gt_labels = np.zeros((P,), dtype=F)
    for g in range(gt_boxes.shape[0]):                               # :140-147 last write wins
        best_prop_gt_labels[best_prop[g]] = gt_boxes[g, 4] + F(1)
    label = np.where(is_best_box, best_prop_gt_labels, label).astype(F)  # :148-153
    label_pre = label.copy()
    max_fg = int(foreground_fraction * minibatch_size)               # :159
    fg_inds = np.where(iou_is_fg | is_best_box)[0]                   # :160-165
    if fg_inds.size > max_fg:                                        # :196-200
        keep = rng.keep_k_smallest(fg_inds, max_fg, seed, rng.STREAM_RCNN_FG)
        dis = fg_inds[~keep]
        label[dis] = -label[dis]                                     # :189-194
    total_fg = int(np.sum(label > 0))                                # :202-206
    max_bg = minibatch_size - total_fg                               # :211
    bg_inds = np.where(label == 0)[0]                                # :216-219
    if bg_inds.size >= max_bg:                                       # :246-250
        keep = rng.keep_k_smallest(bg_inds, max_bg, seed, rng.STREAM_RCNN_BG)
        label[bg_inds[~keep]] = -1
    with_target = label > 0                                          # :259-263
    targets = np.zeros((P, 4), dtype=F)
    if with_target.any():
        targets[with_target] = bx.encode(proposals[with_target],
                                         gt_boxes[best_gt[with_target], :4],
                                         variances=variances)        # :281-294
    if return_pre_subsample:
        return label, targets, label_pre, best_gt.astype(np.int32), max_overlaps
    return label, targets


# --------------------------------------------------------------------------
# A11  ROIPoolingLayer._roi_crop   models/fasterrcnn/roi_pool.py:37-95
# --------------------------------------------------------------------------
def roi_normalised_boxes(roi_proposals, im_shape):
    """roi_pool.py:37-66: divide by H / W (not H-1 / W-1); TF order y1,x1,y2,x2."""
    r = np.asarray(roi_proposals, dtype=F)
    H, W = F(im_shape[0]), F(im_shape[1])
    return np.stack([r[:, 1] / H, r[:, 0] / W, r[:, 3] / H, r[:, 2] / W], axis=1).astype(F)


def roi_pool(roi_proposals, conv_feature_map, im_shape, pooled_width=7,
             pooled_height=7, batch_ids=None):
    bboxes = roi_normalised_boxes(roi_proposals, im_shape)
    if batch_ids is None:
        batch_ids = np.zeros((bboxes.shape[0],), dtype=np.int32)     # :73
    # crop size is passed as [pooled_width*2, pooled_height*2] (roi_pool.py:77).
    crops = tfops.crop_and_resize(conv_feature_map, bboxes, batch_ids,
                                  (pooled_width * 2, pooled_height * 2))
    return tfops.max_pool_2x2_valid(crops), crops


# --------------------------------------------------------------------------
# A14  RCNNProposal._build   models/fasterrcnn/rcnn_proposal.py:46-164
# --------------------------------------------------------------------------
def rcnn_proposal(proposals, bbox_pred, cls_prob, im_shape, num_classes,
                  variances=(0.1, 0.2), class_max_detections=100,
                  class_nms_threshold=0.5, total_max_detections=300,
                  min_prob_threshold=0.5):
    proposals = np.asarray(proposals, dtype=F)
    bbox_pred = np.asarray(bbox_pred, dtype=F)
    cls_prob = np.asarray(cls_prob, dtype=F)
    sel_boxes, sel_probs, sel_labels = [], [], []
    for c in range(num_classes):                                     # :77
        class_prob = cls_prob[:, c + 1]
        raw = bx.decode(proposals, bbox_pred[:, 4 * c:4 * c + 4], variances=variances)
        objs = bx.clip_boxes(raw, im_shape)                          # :89
        filt = bx.area_positive(objs) & (class_prob >= F(min_prob_threshold or 0.0))
        objs, class_prob = objs[filt], class_prob[filt]
        sel = tfops.non_max_suppression(objs[:, [1, 0, 3, 2]], class_prob,
                                        class_max_detections, class_nms_threshold)
        sel_boxes.append(objs[sel])
        sel_probs.append(class_prob[sel])
        sel_labels.append(np.full((sel.shape[0],), c, dtype=np.int32))
    objects = np.concatenate(sel_boxes, axis=0) if sel_boxes else np.zeros((0, 4), F)
    labels = np.concatenate(sel_labels, axis=0) if sel_labels else np.zeros((0,), np.int32)
    probs = np.concatenate(sel_probs, axis=0) if sel_probs else np.zeros((0,), F)
    k = min(total_max_detections, probs.shape[0])                    # :148-152
    top_probs, idx = tfops.top_k(probs, k)
    return {'objects': objects[idx], 'proposal_label': labels[idx],
            'proposal_label_prob': top_probs}


# --------------------------------------------------------------------------
# losses   utils/losses.py:4-32, rpn.py:219-309, rcnn.py:255-411
# --------------------------------------------------------------------------
def smooth_l1_loss(pred, target, sigma=3.0):
    sigma2 = F(sigma) ** 2
    diff = np.asarray(pred, dtype=F) - np.asarray(target, dtype=F)
    a = np.abs(diff)
    return np.where(a < F(1.0) / sigma2, F(0.5) * sigma2 * np.square(a),
                    a - F(0.5) / sigma2).sum(axis=1).astype(F)


def softmax_cross_entropy(logits, onehot):
    logits = np.asarray(logits, dtype=F)
    m = logits.max(axis=1, keepdims=True)
    z = logits - m
    lse = np.log(np.exp(z).sum(axis=1, keepdims=True))
    return (-(onehot * (z - lse)).sum(axis=1)).astype(F)


def rpn_loss(rpn_cls_score, rpn_cls_target, rpn_bbox_pred, rpn_bbox_target, l1_sigma=3.0):
    t = np.asarray(rpn_cls_target).reshape(-1).astype(np.int32)
    not_ignored = t != -1
    labels = t[not_ignored]
    ce = softmax_cross_entropy(np.asarray(rpn_cls_score, F)[not_ignored], np.eye(2, dtype=F)[labels])
    pos = t == 1
    reg = smooth_l1_loss(np.asarray(rpn_bbox_pred, F).reshape(-1, 4)[pos],
                         np.asarray(rpn_bbox_target, F).reshape(-1, 4)[pos], sigma=l1_sigma)
    with np.errstate(invalid='ignore'):
        return {'rpn_cls_loss': F(np.mean(ce)) if ce.size else F(np.nan),
                'rpn_reg_loss': F(np.mean(reg)) if reg.size else F(np.nan),
                'cross_entropy_per_anchor': ce, 'reg_loss_per_anchor': reg}


def rcnn_loss(cls_score, bbox_offsets, cls_target, bbox_offsets_target, num_classes, l1_sigma=1.0):
    t = np.asarray(cls_target).reshape(-1).astype(np.int32)
    ni = t >= 0
    ce = softmax_cross_entropy(np.asarray(cls_score, F)[ni], np.eye(num_classes + 1, dtype=F)[t[ni]])
    fg = t > 0
    cls = t[fg] - 1
    bo = np.asarray(bbox_offsets, F)[fg].reshape(-1, num_classes, 4)
    cleaned = bo[np.arange(bo.shape[0]), cls]                        # rcnn.py:353-387
    reg = smooth_l1_loss(cleaned, np.asarray(bbox_offsets_target, F)[fg], sigma=l1_sigma)
    with np.errstate(invalid='ignore'):
        return {'rcnn_cls_loss': F(np.mean(ce)) if ce.size else F(np.nan),
                'rcnn_reg_loss': F(np.mean(reg)) if reg.size else F(np.nan),
                'cross_entropy_per_proposal': ce, 'reg_loss_per_proposal': reg}
