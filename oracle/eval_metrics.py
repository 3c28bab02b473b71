"""Oracle (test infrastructure only): the COCO-style AP / AR of `lumi eval`, restated with plain loops for small
cases (reference: luminoth/eval.py:487-653 `calculate_metrics`; IoU from luminoth/utils/bbox_overlap.py:51-94, the
numpy twin of the +1-pixel-convention overlap whose golden cases are in tests/test_oracle_golden.py).

PARITY UNPINNED: the reference has no test or golden vector for its metrics (the tutorial figures need the COCO
subset and a trained checkpoint); the hand-computable cases in tests/test_eval_metrics.py pin the definition.

Quirk kept on purpose (eval.py:571-596): inside one (image, class) group the TP/FP flags are written at the
detection's ORIGINAL position while the scores stored next to them are SORTED — the two only line up when the
detector already emits that class's boxes by descending score (it does: rcnn_proposal.py / ssd proposal top_k).
"""
import numpy as np

from . import boxes as bx

IOU_THRESHOLDS = np.linspace(0.50, 0.95, 10)
REC_THRESHOLDS = np.linspace(0.00, 1.00, 101)


def match_image_class(boxes, scores, gt_boxes):
    """One (image, class) group -> (flags (D,10) at original positions, scores sorted descending)."""
    order = np.argsort(-scores)
    flags = np.zeros((len(order), len(IOU_THRESHOLDS)))
    if gt_boxes.shape[0] == 0:
        return flags, scores[order]
    taken = np.zeros((gt_boxes.shape[0], len(IOU_THRESHOLDS)), bool)
    ious = bx.bbox_overlap_np(boxes, gt_boxes)
    for d in order:                                   # highest score first
        g = int(np.argmax(ious[d]))                   # its best ground truth, whatever the threshold
        for t, thr in enumerate(IOU_THRESHOLDS):
            if ious[d, g] >= thr and not taken[g, t]:
                flags[d, t] = 1
                taken[g, t] = True
    return flags, scores[order]


def average_precision(flags, scores, num_examples):
    """Ranked flags of one class over the whole split -> (ap (10,), ar (10,))."""
    ap, ar = np.zeros(len(IOU_THRESHOLDS)), np.zeros(len(IOU_THRESHOLDS))
    rank = np.argsort(-scores)
    tp = flags[rank]
    ctp, cfp = np.cumsum(tp, axis=0), np.cumsum(1 - tp, axis=0)
    with np.errstate(divide='ignore', invalid='ignore'):
        recall = ctp.astype(float) / num_examples
        precision = np.divide(ctp.astype(float), ctp + cfp)
    for t in range(len(IOU_THRESHOLDS)):
        p, r = precision[:, t].copy(), recall[:, t]
        for i in range(len(p) - 1, 0, -1):            # make the curve non-increasing from the right
            if p[i] > p[i - 1]:
                p[i - 1] = p[i]
        total = 0.0
        for pos in np.searchsorted(r, REC_THRESHOLDS):
            if pos >= len(r):
                break
            total += p[pos] / len(REC_THRESHOLDS)
        ap[t] = total
        ar[t] = r[-1] if len(r) else 0
    return ap, ar


def calculate_metrics(output_per_batch, num_classes):
    per_class = [[] for _ in range(num_classes)]
    examples = [0] * num_classes
    for i in range(len(output_per_batch['bboxes'])):
        cls_ids, boxes, scores = (np.asarray(output_per_batch[k][i]) for k in ('classes', 'bboxes', 'scores'))
        gt_cls, gt_boxes = np.asarray(output_per_batch['gt_classes'][i]), np.asarray(output_per_batch['gt_bboxes'][i])
        for c in range(num_classes):
            sel, gsel = cls_ids == c, gt_cls == c
            examples[c] += int(gsel.sum())
            per_class[c].append(match_image_class(boxes[sel], scores[sel], gt_boxes[gsel]))
    ap = np.zeros((num_classes, len(IOU_THRESHOLDS)))
    ar = np.zeros((num_classes, len(IOU_THRESHOLDS)))
    for c in range(num_classes):
        flags = np.concatenate([f for f, _ in per_class[c]]) if per_class[c] else np.zeros((0, len(IOU_THRESHOLDS)))
        scores = np.concatenate([s for _, s in per_class[c]]) if per_class[c] else np.zeros((0,))
        ap[c], ar[c] = average_precision(flags, scores, examples[c])
    return ap, ar
