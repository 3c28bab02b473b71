"""`lumi eval` re-hosted on the HIP path (reference: luminoth/eval.py:15-653).

    python -m luminoth_amd.eval -c cfg.yml [--split val] [--no-watch] [--from-global-step N] [--max-detections 100]

The forward pass (detections + losses on the split) runs through the HIP kernels; the metric itself is the
reference's host-side numpy computation — COCO-style AP@[.50:.95] / AR with greedy score-ordered matching and
101-point interpolated precision (eval.py:487-653) — here vectorised over the ten IoU thresholds.  TensorBoard
summaries and image visualisation (eval.py:137-223,343-372) are outside the hot-path scope: metrics are logged with
the reference's lines and returned / written as one JSON line per evaluated checkpoint.
"""
import argparse
import json
import logging
import os
import sys
import time

import numpy as np

log = logging.getLogger('luminoth_amd')

IOU_THRESHOLDS = np.linspace(0.50, 0.95, int(np.round((0.95 - 0.50) / 0.05)) + 1)
REC_THRESHOLDS = np.linspace(0.00, 1.00, int(np.round((1.00 - 0.00) / 0.01)) + 1)   # 101 recall levels (COCO)


def bbox_overlap(bboxes1, bboxes2):
    """utils/bbox_overlap.py:51-94 (+1 pixel convention; dtypes left to numpy promotion like the reference)."""
    b1, b2 = np.asarray(bboxes1), np.asarray(bboxes2)
    w = np.minimum(b1[:, [2]], b2[:, [2]].T) - np.maximum(b1[:, [0]], b2[:, [0]].T) + 1
    h = np.minimum(b1[:, [3]], b2[:, [3]].T) - np.maximum(b1[:, [1]], b2[:, [1]].T) + 1
    intersection = np.maximum(w, 0.) * np.maximum(h, 0.)
    area1 = (b1[:, [2]] - b1[:, [0]] + 1) * (b1[:, [3]] - b1[:, [1]] + 1)
    area2 = (b2[:, [2]] - b2[:, [0]] + 1) * (b2[:, [3]] - b2[:, [1]] + 1)
    union = (area1 + area2.T) - intersection
    iou = np.zeros((b1.shape[0], b2.shape[0]))
    np.divide(intersection, union, out=iou, where=intersection > 0.)
    return iou


def _match(cls_bboxes, cls_scores, cls_gt_bboxes):
    """TP/FP flags of one (image, class) group for all IoU thresholds at once (eval.py:558-596).  Flags are
    stored at the detection's original position, the returned scores are sorted — as the reference does."""
    order = np.argsort(-cls_scores)
    flags = np.zeros((order.shape[0], IOU_THRESHOLDS.shape[0]))
    num_gt = cls_gt_bboxes.shape[0]
    if num_gt == 0 or order.shape[0] == 0:
        return flags, cls_scores[order]
    ious = bbox_overlap(cls_bboxes, cls_gt_bboxes)
    gt_match = np.argmax(ious, axis=1)
    over = ious[np.arange(ious.shape[0]), gt_match][:, None] >= IOU_THRESHOLDS[None, :]      # (D, T)
    detected = np.zeros((num_gt, IOU_THRESHOLDS.shape[0]), bool)
    for d in order:                       # greedy, highest confidence first; each gt is matched once per threshold
        first = over[d] & ~detected[gt_match[d]]
        flags[d] = first
        detected[gt_match[d]] |= first
    return flags, cls_scores[order]


def calculate_metrics(output_per_batch, num_classes):
    """eval.py:487-653 -> (ap_per_class, ar_per_class), both (num_classes, 10)."""
    T = IOU_THRESHOLDS.shape[0]
    groups = [[] for _ in range(num_classes)]
    num_examples = np.zeros(num_classes, np.int64)
    for idx in range(len(output_per_batch['bboxes'])):
        classes = np.asarray(output_per_batch['classes'][idx])
        bboxes = np.asarray(output_per_batch['bboxes'][idx])
        scores = np.asarray(output_per_batch['scores'][idx])
        gt_classes = np.asarray(output_per_batch['gt_classes'][idx])
        gt_bboxes = np.asarray(output_per_batch['gt_bboxes'][idx])
        for cls in range(num_classes):
            sel, gsel = classes == cls, gt_classes == cls
            num_examples[cls] += int(gsel.sum())
            groups[cls].append(_match(bboxes[sel, :], scores[sel], gt_bboxes[gsel, :]))
    ap_per_class = np.zeros((num_classes, T))
    ar_per_class = np.zeros((num_classes, T))
    for cls in range(num_classes):
        labels = np.concatenate([g[0] for g in groups[cls]]) if groups[cls] else np.zeros((0, T))
        scores = np.concatenate([g[1] for g in groups[cls]]) if groups[cls] else np.zeros((0,))
        ranked = labels[np.argsort(-scores), :]
        cum_tp = np.cumsum(ranked, axis=0)
        cum_fp = np.cumsum(1 - ranked, axis=0)
        with np.errstate(divide='ignore', invalid='ignore'):
            recall = cum_tp.astype(float) / num_examples[cls]
            precision = np.divide(cum_tp.astype(float), cum_tp + cum_fp)
        if precision.shape[0] == 0:
            continue
        # interpolated precision: running maximum from the right (eval.py:629-632)
        precision = np.maximum.accumulate(precision[::-1], axis=0)[::-1]
        for t in range(T):
            inds = np.searchsorted(recall[:, t], REC_THRESHOLDS)
            inds = inds[:int(np.argmax(inds >= recall.shape[0]))] if (inds >= recall.shape[0]).any() else inds
            ap_per_class[cls, t] = precision[inds, t].sum() / REC_THRESHOLDS.shape[0] if inds.shape[0] else 0.0
        ar_per_class[cls] = recall[-1]
    return ap_per_class, ar_per_class


def summarize(ap_per_class, ar_per_class):
    """eval.py:401-404."""
    return {'AP@0.50': float(np.mean(ap_per_class[:, 0])), 'AP@0.75': float(np.mean(ap_per_class[:, 5])),
            'AP@[0.50:0.95]': float(np.mean(ap_per_class)), 'AR@[0.50:0.95]': float(np.mean(ar_per_class))}


def prepare_config(config, dataset_split='val', max_detections=100):
    """The config edits of eval.py:52-92."""
    config.dataset.split = dataset_split
    config.dataset.data_augmentation = []
    if config.model.type == 'fasterrcnn':
        if config.model.network.with_rcnn:
            config.model.rcnn.proposals.total_max_detections = max_detections
        else:
            config.model.rpn.proposals.post_nms_top_n = max_detections
        config.model.rcnn.proposals.min_prob_threshold = 0.0
    elif config.model.type == 'ssd':
        config.model.proposals.total_max_detections = max_detections
        config.model.proposals.min_prob_threshold = 0.0
    else:
        raise ValueError("Model type '{}' not supported".format(config.model.type))
    config.train.num_epochs = 1
    config.model.base_network.trainable = False
    return config


def _detections(config, pd, b):
    """Per-image (objects, classes, scores) as numpy (eval.py:101-120)."""
    if config.model.type == 'ssd' or config.model.network.get('with_rcnn', False):
        cp = pd['classification_prediction']
        n = int(cp['num_objects'][b])
        return (cp['objects'][b, :n].cpu().numpy(), cp['labels'][b, :n].cpu().numpy(),
                cp['probs'][b, :n].cpu().numpy())
    rp = pd['rpn_prediction']
    n = int(rp['num_proposals'][b])
    scores = rp['scores'][b, :n].cpu().numpy()
    return rp['proposals'][b, :n].cpu().numpy(), np.zeros(scores.shape, np.int32), scores


def evaluate_once(config, model, dataset, global_step=None, class_labels=None, split='val', outputs=None):
    """eval.py:282-484: one pass over the split -> metrics dict (also logged with the reference's lines)."""
    out = {'bboxes': [], 'classes': [], 'scores': [], 'gt_bboxes': [], 'gt_classes': []}
    loss_sums, batches = {}, 0
    total_evaluated = 0
    start = track_start = time.time()
    track_count = 0
    num_classes = config.model.network.num_classes
    if config.model.type == 'fasterrcnn' and not config.model.network.get('with_rcnn', False):
        num_classes = 1                                                     # eval.py:109-110
    for batch in dataset:
        pd = model(batch['image'], batch['bboxes'], is_training=False)
        losses = model.loss(pd, return_all=True)
        for k, v in losses.items():
            loss_sums[k] = loss_sums.get(k, 0.0) + float(v)
        batches += 1
        for b, gt in enumerate(batch['bboxes']):
            objects, classes, scores = _detections(config, pd, b)
            gt = np.asarray(gt)
            out['bboxes'].append(objects)
            out['classes'].append(classes)
            out['scores'].append(scores)
            out['gt_bboxes'].append(gt[:, :4])
            out['gt_classes'].append(gt[:, 4])
            total_evaluated += 1
            track_count += 1
        now = time.time()
        if now - track_start > 20.:
            log.info('%d processed in %.2fs (global %.2f images/s, period %.2f images/s)', total_evaluated,
                     now - start, total_evaluated / (now - start), track_count / (now - track_start))
            track_count, track_start = 0, now
    if outputs is not None:
        outputs.update(out)
    ap_per_class, ar_per_class = calculate_metrics(out, num_classes)
    res = summarize(ap_per_class, ar_per_class)
    log.info('Finished evaluation at step %s.', global_step)
    log.info('Evaluated %d images.', total_evaluated)
    log.info('Average Precision (AP) @ [0.50] = %.3f', res['AP@0.50'])
    log.info('Average Precision (AP) @ [0.75] = %.3f', res['AP@0.75'])
    log.info('Average Precision (AP) @ [0.50:0.95] = %.3f', res['AP@[0.50:0.95]'])
    log.info('Average Recall (AR) @ [0.50:0.95] = %.3f', res['AR@[0.50:0.95]'])
    for idx, val in enumerate(ap_per_class[:, 0]):
        label = '{} ({})'.format(class_labels[idx], idx) if class_labels else idx
        log.debug('Average Precision (AP) @ [0.50] for %s = %.3f', label, val)
    res.update({'total_evaluated': total_evaluated, 'evaluation_time': time.time() - start,
                'global_step': global_step, 'ap_per_class': ap_per_class.tolist(),
                'ar_per_class': ar_per_class.tolist()})
    for k, v in loss_sums.items():
        res['{}_losses/{}'.format(split, k)] = v / max(batches, 1)
    return res


def evaluate(config_files, override_params=(), dataset_split='val', watch=True, from_global_step=None,
             max_detections=100, output=None, poll_secs=10.0, max_evaluations=None):
    """eval.py:24-279: evaluates every checkpoint of job_dir/run_name newer than the last one evaluated (or only
    the latest with --no-watch).  Returns the list of result dicts."""
    from luminoth_amd.datasets import get_dataset
    from luminoth_amd.models import get_model
    from luminoth_amd.train import list_checkpoints, restore_checkpoint
    from luminoth_amd.utils.config import get_config
    try:
        config = get_config(list(config_files), override_params=list(override_params))
    except KeyError:
        raise KeyError('model.type should be set on the custom config.')
    if not config.train.get('job_dir'):
        raise KeyError('`job_dir` should be set.')
    if not config.train.get('run_name'):
        raise KeyError('`run_name` should be set.')
    run_dir = os.path.join(config.train.job_dir, config.train.run_name)
    prepare_config(config, dataset_split, max_detections)
    class_labels = None
    classes_file = os.path.join(config.dataset.get('dir') or '', 'classes.json')
    if os.path.exists(classes_file):
        with open(classes_file) as f:
            class_labels = json.load(f)
    model = get_model(config.model.type)(config)
    last_global_step = from_global_step
    results = []
    while True:
        ckpts = [(s, p) for s, p in list_checkpoints(run_dir)
                 if last_global_step is None or s > last_global_step]
        if not watch:
            ckpts = ckpts[-1:]
        if not ckpts and not watch:
            log.info('No checkpoints found')
        for step, path in ckpts:
            log.info('Evaluating global_step %d using checkpoint \'%s\'', step, path)
            restore_checkpoint(model, path)
            dataset = get_dataset(config.dataset.type)(config)
            res = evaluate_once(config, model, dataset, global_step=step, class_labels=class_labels,
                                split=dataset_split)
            results.append(res)
            if output is not None:
                output.write(json.dumps({k: v for k, v in res.items() if not k.endswith('_per_class')}) + '\n')
                output.flush()
            last_global_step = step
        if not watch or (max_evaluations is not None and len(results) >= max_evaluations):
            break
        time.sleep(poll_secs)
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(description='Evaluate trained (or training) models (luminoth/eval.py:15-23).')
    ap.add_argument('--split', dest='dataset_split', default='val', help='Dataset split to use.')
    ap.add_argument('--config', '-c', dest='config_files', action='append', required=True, help='Config to use.')
    ap.add_argument('--watch', dest='watch', action='store_true', default=True)
    ap.add_argument('--no-watch', dest='watch', action='store_false')
    ap.add_argument('--from-global-step', type=int, default=None)
    ap.add_argument('--override', '-o', dest='override_params', action='append', default=[])
    ap.add_argument('--max-detections', type=int, default=100)
    ap.add_argument('--debug', action='store_true')
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.DEBUG if args.debug else logging.INFO,
                        format='%(levelname)s:%(name)s:%(message)s')
    evaluate(args.config_files, args.override_params, args.dataset_split, args.watch, args.from_global_step,
             args.max_detections, output=sys.stdout)
    return 0


if __name__ == '__main__':
    sys.exit(main())
