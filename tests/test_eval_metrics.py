"""CPU tests of the `lumi eval` row: the oracle's loop restatement of eval.py:487-653 on
hand-computable cases (the reference ships no test for its metrics: these pin the definition), the vectorised host
implementation against the oracle on random inputs, and the evaluation loop with stand-in model and dataset."""
import numpy as np
import pytest
import torch

from luminoth_amd import eval as E
from oracle import boxes as obx
from oracle import eval_metrics as om

F = np.float32
BOTH = [om.calculate_metrics, E.calculate_metrics]


def batch(dets, gts):
    """dets: [(box, cls, score)], gts: [(box, cls)] for one image."""
    return {'bboxes': [np.array([d[0] for d in dets], F).reshape(-1, 4)], 'classes': [np.array([d[1] for d in dets], np.int32)],
            'scores': [np.array([d[2] for d in dets], F)], 'gt_bboxes': [np.array([g[0] for g in gts], np.int32).reshape(-1, 4)],
            'gt_classes': [np.array([g[1] for g in gts], np.int32)]}


def merge(*bs):
    return {k: sum((b[k] for b in bs), []) for k in bs[0]}


@pytest.mark.parametrize('fn', BOTH)
def test_hand_computed_cases(fn):
    g = [0, 0, 99, 99]
    # perfect detection
    ap, ar = fn(batch([(g, 0, 0.9)], [(g, 0)]), 1)
    np.testing.assert_allclose(ap, np.ones((1, 10)), rtol=1e-12)
    np.testing.assert_allclose(ar, np.ones((1, 10)))
    # IoU = 100*62/10000 = 0.62: true positive for thresholds .50 .55 .60 only
    ap, ar = fn(batch([([0, 0, 99, 61], 0, 0.9)], [(g, 0)]), 1)
    np.testing.assert_allclose(ap[0], [1, 1, 1, 0, 0, 0, 0, 0, 0, 0], rtol=1e-12)
    np.testing.assert_allclose(ar[0], [1, 1, 1, 0, 0, 0, 0, 0, 0, 0])
    # duplicate detection of one gt: second is a false positive, AP stays 1 (recall already 1 at rank 1)
    ap, ar = fn(batch([(g, 0, 0.9), (g, 0, 0.8)], [(g, 0)]), 1)
    np.testing.assert_allclose(ap[0], np.ones(10), rtol=1e-12)
    # a higher-scored false positive first: precision [0, .5] -> interpolated .5 everywhere
    ap, ar = fn(batch([([200, 200, 240, 240], 0, 0.95), (g, 0, 0.8)], [(g, 0)]), 1)
    np.testing.assert_allclose(ap[0], np.full(10, 0.5), rtol=1e-12)
    # two gts, one found: recall .5 -> 51 of the 101 recall levels are reached at precision 1
    ap, ar = fn(batch([(g, 0, 0.9)], [(g, 0), ([300, 300, 340, 340], 0)]), 1)
    np.testing.assert_allclose(ap[0], np.full(10, 51 / 101.), rtol=1e-12)
    np.testing.assert_allclose(ar[0], np.full(10, 0.5))
    # per class: class 1 has a gt and no detection (0/0), class 2 has nothing at all (0/0), class 0 perfect
    ap, ar = fn(batch([(g, 0, 0.9)], [(g, 0), ([300, 300, 340, 340], 1)]), 3)
    np.testing.assert_allclose(ap[:, 0], [1, 0, 0], rtol=1e-12)
    np.testing.assert_allclose(ar[:, 0], [1, 0, 0])
    # a detection of a class without any ground truth: recall is x/0 (NaN AR, like the reference), AP 0
    with np.errstate(all='ignore'):
        ap, ar = fn(batch([(g, 0, 0.9), (g, 1, 0.5)], [(g, 0)]), 2)
    assert ap[1, 0] == 0 and np.isnan(ar[1, 0]) and ap[0, 0] == pytest.approx(1.0)
    # wrong class never matches
    ap, ar = fn(batch([(g, 1, 0.9)], [(g, 0)]), 2)
    assert ap[0].sum() == 0 and ar[0].sum() == 0
    # greedy by score across images: image 2's detection outranks image 1's false positive
    ap, ar = fn(merge(batch([([200, 200, 240, 240], 0, 0.7)], [(g, 0)]), batch([(g, 0, 0.9)], [(g, 0)])), 1)
    np.testing.assert_allclose(ap[0], np.full(10, 51 / 101.), rtol=1e-12)        # P=[1,.5] R=[.5,.5]
    # empty split
    ap, ar = fn({k: [] for k in ('bboxes', 'classes', 'scores', 'gt_bboxes', 'gt_classes')}, 2)
    assert ap.shape == (2, 10) and ap.sum() == 0 and ar.sum() == 0


def test_summary_indices():
    ap = np.arange(20, dtype=float).reshape(2, 10)
    s = E.summarize(ap, ap / 2)
    assert s['AP@0.50'] == 5.0 and s['AP@0.75'] == 10.0 and s['AP@[0.50:0.95]'] == 9.5 and s['AR@[0.50:0.95]'] == 4.75
    np.testing.assert_allclose(E.IOU_THRESHOLDS, [.5, .55, .6, .65, .7, .75, .8, .85, .9, .95])
    assert E.REC_THRESHOLDS.shape == (101,) and E.REC_THRESHOLDS[50] == 0.5


def random_split(rs, images, num_classes, sort_scores=True):
    out = {k: [] for k in ('bboxes', 'classes', 'scores', 'gt_bboxes', 'gt_classes')}
    for _ in range(images):
        G = rs.randint(0, 6)
        xy = rs.randint(0, 300, size=(G, 2))
        wh = rs.randint(20, 120, size=(G, 2))
        gt = np.concatenate([xy, xy + wh], 1).astype(np.int32)
        gc = rs.randint(0, num_classes, size=G).astype(np.int32)
        D = rs.randint(0, 25)
        src = rs.randint(0, max(G, 1), size=D)
        if G:
            det = gt[src].astype(F) + rs.randn(D, 4).astype(F) * rs.choice([1, 6, 25], size=(D, 1))
            dc = np.where(rs.rand(D) < 0.8, gc[src], rs.randint(0, num_classes, size=D)).astype(np.int32)
        else:
            det = (rs.rand(D, 4) * 300).astype(F)
            dc = rs.randint(0, num_classes, size=D).astype(np.int32)
        sc = rs.rand(D).astype(F)
        if D > 3:
            sc[1] = sc[3]                                       # a tie
        if sort_scores:
            o = np.argsort(-sc, kind='stable')
            det, dc, sc = det[o], dc[o], sc[o]
        for k, v in zip(('bboxes', 'classes', 'scores', 'gt_bboxes', 'gt_classes'), (det, dc, sc, gt, gc)):
            out[k].append(v)
    return out


@pytest.mark.parametrize('seed', range(6))
def test_vectorised_host_equals_oracle(seed):
    rs = np.random.RandomState(seed)
    data = random_split(rs, 12, 4, sort_scores=(seed % 2 == 0))    # odd seeds: unsorted -> the misalignment quirk
    with np.errstate(all='ignore'):
        ap, ar = E.calculate_metrics(data, 4)
        ap_o, ar_o = om.calculate_metrics(data, 4)
    np.testing.assert_allclose(ap, ap_o, rtol=1e-12, atol=1e-15, equal_nan=True)
    np.testing.assert_array_equal(ar, ar_o)
    assert 0 < np.nanmean(ap) < 1


def test_overlap_twin_matches_oracle_and_tf_convention():
    rs = np.random.RandomState(3)
    a = (rs.rand(20, 4) * 100).astype(F)
    a[:, 2:] += a[:, :2]
    b = rs.randint(0, 100, size=(7, 4)).astype(np.int32)
    b[:, 2:] += b[:, :2]
    np.testing.assert_array_equal(E.bbox_overlap(a, b), obx.bbox_overlap_np(a, b))
    assert E.bbox_overlap(a, b).dtype == np.float64
    np.testing.assert_allclose(E.bbox_overlap(a, b.astype(F)), obx.bbox_overlap(a, b.astype(F)), rtol=1e-6, atol=1e-7)
    assert E.bbox_overlap(np.array([[0, 0, 10, 10]]), np.array([[11, 11, 20, 20]]))[0, 0] == 0   # bbox_overlap_test.py:44-50


class _GtEchoModel(object):
    """Returns every ground-truth box as a detection (first image: perfect; `miss` drops the last box)."""

    def __init__(self, miss=False):
        self.miss = miss

    def __call__(self, image, gt_boxes, is_training=False):
        assert not is_training
        B = len(gt_boxes)
        cap = max(len(g) for g in gt_boxes)
        obj, lab, prob = torch.zeros(B, cap, 4), torch.full((B, cap), -1, dtype=torch.int32), torch.zeros(B, cap)
        num = torch.zeros(B, dtype=torch.int32)
        for b, g in enumerate(gt_boxes):
            g = np.asarray(g)
            if self.miss:
                g = g[:-1]
            n = len(g)
            obj[b, :n] = torch.tensor(g[:, :4], dtype=torch.float32)
            lab[b, :n] = torch.tensor(g[:, 4].astype(np.int32))
            prob[b, :n] = torch.linspace(0.9, 0.5, n) if n else prob[b, :n]
            num[b] = n
        return {'classification_prediction': {'objects': obj, 'labels': lab, 'probs': prob, 'num_objects': num}}

    def loss(self, pd, return_all=False):
        return {'total_loss': torch.tensor(2.0), 'rpn_cls_loss': torch.tensor(0.5)}


def test_evaluate_once_with_stand_in_model():
    from luminoth_amd.utils.config import get_config
    cfg = E.prepare_config(get_config({'model': {'type': 'fasterrcnn', 'network': {'num_classes': 3}}}), 'val', 50)
    assert cfg.dataset.split == 'val' and cfg.dataset.data_augmentation == [] and cfg.train.num_epochs == 1
    assert cfg.model.rcnn.proposals.total_max_detections == 50 and cfg.model.rcnn.proposals.min_prob_threshold == 0.0
    assert cfg.model.base_network.trainable is False
    rs = np.random.RandomState(0)
    data = []
    for i in range(5):
        xy = rs.randint(0, 200, size=(3, 2))
        g = np.concatenate([xy, xy + rs.randint(30, 90, size=(3, 2)), np.array([[0], [1], [2]])], 1).astype(F)
        data.append({'image': torch.zeros(1, 8, 8, 3), 'bboxes': [g], 'filename': ['f%d' % i]})
    outs = {}
    res = E.evaluate_once(cfg, _GtEchoModel(), data, global_step=12, outputs=outs)
    assert res['AP@0.50'] == pytest.approx(1.0) and res['AP@[0.50:0.95]'] == pytest.approx(1.0)
    assert res['AR@[0.50:0.95]'] == pytest.approx(1.0) and res['total_evaluated'] == 5 and res['global_step'] == 12
    assert res['val_losses/total_loss'] == pytest.approx(2.0) and res['val_losses/rpn_cls_loss'] == pytest.approx(0.5)
    assert len(outs['bboxes']) == 5 and outs['gt_classes'][0].tolist() == [0, 1, 2]
    res = E.evaluate_once(cfg, _GtEchoModel(miss=True), data)          # class 2 is never detected
    assert res['AP@0.50'] == pytest.approx(2 / 3.) and res['AR@[0.50:0.95]'] == pytest.approx(2 / 3.)
    ssd = E.prepare_config(get_config({'model': {'type': 'ssd'}}), 'test', 7)
    assert ssd.model.proposals.total_max_detections == 7 and ssd.dataset.split == 'test'
