"""CPU: the SSD oracle (oracle/ssd.py) on hand-checkable cases, the host-side anchor generator against
it, and the SSD config surface.  The reference ships NO SSD tests (parity unpinned for
S1-S6), so these pin the restatement to the arithmetic spelled out in the cited reference lines."""
import numpy as np
import pytest

from oracle import boxes as obx
from oracle import ssd as oss

F = np.float32


def test_anchor_count_and_geometry_ssd300():
    # ssd/base_config.yml:128-138 + feature maps 37,18,9,5,3,1 (truncated_vgg.py VALID pools) -> 8096 anchors
    shapes = [(37, 37), (18, 18), (9, 9), (5, 5), (3, 3), (1, 1)]
    a = oss.all_anchors(shapes, (300, 300, 3))
    assert a.shape == (8096, 4) and a.dtype == np.float32
    assert a.min() >= 0 and a.max() <= 299                         # clipped to [0, dim-1] (bbox_transform.py:105-122)
    # first anchor of the first cell: side sqrt(.1*.256)*37 feature units, centre 0.5, scaled by 300/37, clipped at 0
    s = np.sqrt(0.1 * (0.1 + 0.78 / 5)) * 37
    np.testing.assert_allclose(a[0], [0, 0, (0.5 + s / 2) * 300 / 37, (0.5 + s / 2) * 300 / 37], rtol=1e-6)
    # ratio 0.5 anchor: h = s/sqrt(.5), w = s*sqrt(.5)
    w, h = 0.1 * np.sqrt(.5) * 37, 0.1 / np.sqrt(.5) * 37
    np.testing.assert_allclose(a[2], [0, 0, (0.5 + w / 2) * 300 / 37, (0.5 + h / 2) * 300 / 37], rtol=1e-6)
    # last map (1x1, single scale): first anchor uses scale*0.99 (utils.py:44-46)
    last = a[-4]
    np.testing.assert_allclose(last, np.clip([(0.5 - .88 * .99 / 2) * 300, (0.5 - .88 * .99 / 2) * 300,
                                              (0.5 + .88 * .99 / 2) * 300, (0.5 + .88 * .99 / 2) * 300], 0, 299),
                               rtol=1e-6)


def test_host_anchor_generator_equals_oracle():
    from luminoth_amd.models.ssd.utils import generate_all_anchors
    for shapes, hw in (([(37, 37), (18, 18), (9, 9), (5, 5), (3, 3), (1, 1)], (300, 300)),
                       ([(20, 27), (10, 13), (5, 7), (3, 4), (1, 2), (1, 1)], (160, 216))):
        a = oss.all_anchors(shapes, (hw[0], hw[1], 3))
        b = generate_all_anchors(shapes, hw, 0.1, 0.88, np.array([1, .5, 2, .333, 3]), [4, 6, 6, 6, 4, 4])
        np.testing.assert_array_equal(a, b)


def _probs(n, c, rs):
    p = rs.rand(n, c + 1).astype(F) + F(0.05)
    return (p / p.sum(1, keepdims=True)).astype(F)


def test_target_labels_best_anchor_and_hard_negatives():
    rs = np.random.RandomState(0)
    anchors = np.array([[0, 0, 9, 9], [0, 0, 19, 19], [30, 30, 49, 49], [100, 100, 119, 119],
                        [200, 200, 209, 209], [250, 250, 259, 259], [60, 60, 64, 64], [150, 150, 160, 160]], F)
    gt = np.array([[0, 0, 19, 19, 3], [32, 32, 47, 47, 7]], F)      # gt0 == anchor 1; gt1 inside anchor 2 (IoU .64)
    probs = _probs(8, 10, rs)
    labels, targets = oss.ssd_target(probs, anchors, gt)
    assert labels[1] == 4 and labels[2] == 8                       # label + 1 (target.py:85-96)
    # #fg = 2 -> num_bg = int(2 * 3.0) = 6 hardest negatives.  Only 5 rows are candidates (IoU <= .2 and
    # label <= 0): anchors 3..7.  Anchor 0 (IoU .25 with gt0: neither fg nor a candidate) has score -1 like
    # the fg rows, and top_k's 6th pick is the LOWEST-index row among those -1 scores: anchor 0 becomes 0 too.
    assert all(labels[i] == 0 for i in (3, 4, 5, 6, 7))
    assert labels[0] == 0 and (labels == 0).sum() == 6
    np.testing.assert_allclose(targets[1], 0, atol=1e-6)           # perfect match encodes to 0
    np.testing.assert_array_equal(targets[[0, 3, 4, 5, 6, 7]], 0)
    np.testing.assert_allclose(targets[2], obx.encode(anchors[2:3], gt[1:2, :4], (0.1, 0.2))[0])


def test_target_best_anchor_override_below_threshold_and_duplicate_gt():
    rs = np.random.RandomState(1)
    anchors = np.array([[0, 0, 99, 99], [200, 200, 219, 219], [300, 300, 309, 309], [400, 400, 419, 419],
                        [500, 500, 519, 519]], F)
    # both gts have anchor 0 as their best anchor with IoU < .5: the LAST gt's label wins (sparse_to_dense)
    gt = np.array([[0, 0, 39, 39, 1], [10, 10, 59, 59, 5]], F)
    labels, _ = oss.ssd_target(_probs(5, 6, rs), anchors, gt)
    assert labels[0] == 6 and (labels[1:] <= 0).all() and (labels == 0).sum() == 3      # num_bg = int(1 * 3.)
    # with too few candidates the same top_k clears the foreground row itself (score -1 rows, lowest index first)
    labels3, _ = oss.ssd_target(_probs(3, 6, rs), anchors[:3], gt)
    assert labels3[0] == 0
    # hard-negative mining may clear a foreground row when num_bg exceeds the candidates (target.py:146-160)
    anchors2 = np.array([[0, 0, 99, 99], [0, 0, 98, 98]], F)
    gt2 = np.array([[0, 0, 99, 99, 2]], F)
    labels2, t2 = oss.ssd_target(_probs(2, 6, rs), anchors2, gt2)    # 2 fg, num_bg = 6 > N? -> clamp to top_k(k<=N)
    assert set(labels2.tolist()) <= {0.0, 3.0}


def test_loss_zero_without_positives_and_known_value():
    C = 3
    cls_pred = np.zeros((4, C + 1), F)
    loc_pred = np.zeros((4, 4), F)
    final, cls, box = oss.ssd_loss(cls_pred, loc_pred, np.array([0, 0, -1, 0], F), np.zeros((4, 4), F), C)
    assert final == 0 and np.isclose(cls, 3 * np.log(4))           # ssd.py:252-270: 0 when there is no positive
    tgt = np.zeros((4, 4), F)
    tgt[1] = [0.05, -0.05, 0.5, -2.0]                               # |d| < 1/9 quadratic, else |d| - 1/18
    final, cls, box = oss.ssd_loss(cls_pred, loc_pred, np.array([0, 2, -1, 0], F), tgt, C)
    exp_box = 2 * 0.5 * 9 * 0.05 ** 2 + (0.5 - 1 / 18.) + (2.0 - 1 / 18.)
    assert np.isclose(box, exp_box, rtol=1e-6) and np.isclose(cls, 3 * np.log(4), rtol=1e-6)
    assert np.isclose(final, (cls + box) / 1.0, rtol=1e-6)


def test_proposal_per_class_nms_and_topk():
    C = 2
    anchors = np.array([[10, 10, 50, 50], [12, 12, 52, 52], [100, 100, 140, 150], [200, 200, 220, 260]], F)
    loc = np.zeros((4, 4), F)
    prob = np.array([[.1, .8, .1], [.2, .7, .1], [.1, .1, .8], [.3, .1, .6]], F)
    out = oss.ssd_proposal(prob, loc, anchors, (300, 300), C)
    # class 0: anchors 0,1 overlap (IoU > .45) -> keep the .8; class 1: two separate boxes
    np.testing.assert_allclose(out['probs'], [.8, .8, .6])
    np.testing.assert_array_equal(out['labels'], [0, 1, 1])
    np.testing.assert_allclose(out['objects'][0], obx.clip_boxes(obx.decode(anchors[:1], loc[:1], (.1, .2)), (300, 300))[0])


def test_ssd_config_surface():
    from luminoth_amd.utils.config import get_config
    from luminoth_amd.models import get_model, get_model_defaults
    cfg = get_config({'model': {'type': 'ssd'}})
    assert cfg.model.anchors.anchors_per_point == [4, 6, 6, 6, 4, 4] and cfg.model.variances == [0.1, 0.2]
    assert cfg.dataset.image_preprocessing.fixed_height == 300 and cfg.train.optimizer.momentum == 0.5
    assert get_model_defaults('ssd')['model']['target']['hard_negative_ratio'] == 3.0
    assert get_model('SSD').__name__ == 'SSD'
