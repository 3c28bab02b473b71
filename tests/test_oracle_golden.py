"""Pins the CPU oracle against the golden vectors / known answers held by the
reference's own unit tests.  Each test cites the reference test
(file:line relative to /root/reference/luminoth/) whose inputs and expected
values it reuses.  No GPU, no HIP library needed.
"""
import os

import numpy as np
import pytest

from oracle import boxes as bx
from oracle import frcnn as of
from oracle import tfops

F = np.float32
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


# ---- A3 anchors -----------------------------------------------------------
def test_anchor_reference_values():
    # utils/anchors_test.py:17-94
    ref = bx.generate_anchors_reference(256, [1.], [1.])
    assert ref.shape == (1, 4)
    np.testing.assert_array_equal(ref[0], [-127.5, -127.5, 127.5, 127.5])
    ref = bx.generate_anchors_reference(256, [1.], np.array([0.5, 1., 2., 4.]))
    np.testing.assert_array_equal(ref, [[-63.5, -63.5, 63.5, 63.5],
                                        [-127.5, -127.5, 127.5, 127.5],
                                        [-255.5, -255.5, 255.5, 255.5],
                                        [-511.5, -511.5, 511.5, 511.5]])
    scales = np.array([0.5, 1., 2.])
    ref = bx.generate_anchors_reference(256, np.array([0.5, 1., 2.]), scales)
    wh = np.column_stack((ref[:, 2] - ref[:, 0] + 1, ref[:, 3] - ref[:, 1] + 1))
    np.testing.assert_allclose(wh[:, 1] / wh[:, 0], [.5, .5, .5, 1, 1, 1, 2, 2, 2])
    np.testing.assert_allclose(np.sqrt(wh[:, 1] * wh[:, 0] / 256 ** 2),
                               [.5, 1, 2, .5, 1, 2, .5, 1, 2])


def test_anchor_reference_too_small():
    # utils/anchors_test.py:96-111
    with pytest.raises(ValueError):
        bx.generate_anchors_reference(1, [0.5], [0.5])


def test_anchor_grid_int_truncation():
    # models/fasterrcnn/fasterrcnn_test.py:256-302 (pins the int32 truncation)
    ref = bx.generate_anchors_reference(16, np.array([0.5, 1, 2]), np.array([0.5, 1, 2]))
    anchors = bx.generate_anchors(ref, 32, 32, 1)
    assert anchors.shape == (9216, 4) and anchors.dtype == np.int32
    w = anchors[:, 2] - anchors[:, 0]
    h = anchors[:, 3] - anchors[:, 1]
    np.testing.assert_array_equal(np.unique(w), np.unique(h))
    assert np.unique(w * h).shape[0] == 6
    assert anchors[:, 0].min() == -22 and anchors[:, 0].max() == 29
    assert anchors[:, 1].min() == -22 and anchors[:, 1].max() == 29
    assert anchors[:, 2].min() == 2 and anchors[:, 2].max() == 53
    assert anchors[:, 3].min() == 2 and anchors[:, 3].max() == 53
    for c in range(4):
        u = np.unique(anchors[:, c])
        np.testing.assert_array_equal(np.diff(u), 1)


def test_default_anchor_refs_truncated():
    # base 256, scales .25,.5,1,2, ratios .5,1,2
    ref = bx.generate_anchors_reference(256, np.array([.5, 1, 2]), np.array([.25, .5, 1, 2]))
    a = bx.generate_anchors(ref, 1, 1, 16)
    assert a.shape == (12, 4)
    np.testing.assert_array_equal(a[0], [-44, -22, 44, 22])
    np.testing.assert_array_equal(a[6], [-127, -127, 127, 127])
    np.testing.assert_array_equal(a[11], [-180, -361, 180, 361])


# ---- A7 IoU ---------------------------------------------------------------
def test_bbox_overlap_known_answers():
    # utils/bbox_overlap_test.py:44-84
    assert bx.bbox_overlap([[0, 0, 10, 10]], [[11, 11, 20, 20]])[0, 0] == 0
    assert bx.bbox_overlap([[0, 0, 10, 10]], [[0, 0, 10, 10]])[0, 0] == 1
    assert bx.bbox_overlap([[10, 10, 0, 0]], [[0, 0, 10, 10]])[0, 0] == 0  # negative area
    # IoU matrix comment at models/fasterrcnn/rpn_proposal_test.py:72-77
    gt = np.array([[10, 10, 26, 36], [10, 10, 20, 22], [10, 11, 20, 21], [19, 30, 33, 38]])
    np.testing.assert_allclose(bx.bbox_overlap(gt, gt), [
        [1., 0.31154684, 0.26361656, 0.10408922],
        [0.31154684, 1., 0.84615385, 0.],
        [0.26361656, 0.84615385, 1., 0.],
        [0.10408922, 0., 0., 1.]], rtol=1e-6)
    # rcnn_target_test.py:109-112
    iou = bx.bbox_overlap([[491, 70, 510, 92], [400, 60, 450, 92], [413, 40, 480, 77],
                           [411, 40, 480, 77]], [[423, 30, 501, 80]])[:, 0]
    np.testing.assert_allclose(iou, [0.0277, 0.1147, 0.4998, 0.4914], atol=1e-4)  # comment values are truncated to 4 digits


# ---- A8 encode / decode / clip --------------------------------------------
def test_bbox_transform_against_reference_numpy_twin():
    # tests/golden/make_golden.py imported luminoth/utils/bbox_transform.py
    g = np.load(os.path.join(GOLD, 'bbox_transform_golden.npz'))
    enc = bx.encode(g['boxes'], g['gt'])
    np.testing.assert_allclose(enc, g['encode'], rtol=2e-5, atol=2e-6)
    dec = bx.decode(g['boxes'], g['deltas'])
    np.testing.assert_allclose(dec, g['decode'], rtol=1e-5, atol=1e-3)
    np.testing.assert_array_equal(bx.clip_boxes(g['clip_in'], g['clip_shape']), g['clip_out'])


def test_bbox_transform_known_answers():
    # utils/bbox_transform_test.py:97-152
    x = np.array([[10, 10, 20, 22], [0, 0, 5, 5]], dtype=F)
    np.testing.assert_array_equal(bx.encode(x, x), np.zeros((2, 4), F))
    rng = np.random.RandomState(0)
    b = np.sort(rng.randint(0, 800, size=(64, 2, 2)), axis=1).transpose(0, 2, 1).reshape(64, 4)
    b = b[:, [0, 2, 1, 3]].astype(F)
    b[:, 2:] += 1
    gt = b[::-1].copy()
    np.testing.assert_allclose(bx.decode(b, bx.encode(b, gt)), gt, atol=1e-2)
    # clip golden :128-146, image (50, 60)
    np.testing.assert_array_equal(
        bx.clip_boxes([[10, 10, 60, 20], [60, 50, 60, 50], [-10, -5, 5, 70]], (50, 60)),
        [[10, 10, 59, 20], [59, 49, 59, 49], [0, 0, 5, 49]])
    # bbox_transform_tf.py:129-152 self-check
    d = bx.encode([[10, 10, 20, 22]], [[11, 13, 34, 31]])
    np.testing.assert_allclose(bx.clip_boxes(bx.decode([[10, 10, 20, 22]], d), (100, 100)),
                               [[11, 13, 34, 31]], atol=1e-4)


# ---- A5 RPNProposal -------------------------------------------------------
CFG5 = dict(pre_nms_top_n=4, post_nms_top_n=3, nms_threshold=1, clip_after_nms=False,
            filter_outside_anchors=False, apply_nms=True, min_prob_threshold=0.0)


def _prop(anchors, prob, gt=None, pred=None, **kw):
    cfg = dict(CFG5)
    cfg.update(kw)
    if pred is None:
        pred = bx.encode(anchors, gt)
    return of.rpn_proposal(np.array(prob, F), pred, np.array(anchors, F), (40, 40), **cfg)


def test_rpn_proposal_nms_threshold():
    # models/fasterrcnn/rpn_proposal_test.py:61-170
    gt = [[10, 10, 26, 36], [10, 10, 20, 22], [10, 11, 20, 21], [19, 30, 33, 38]]
    anchors = [[11, 13, 34, 31], [10, 10, 20, 22], [11, 13, 34, 28], [21, 29, 34, 37]]
    prob = [[.8, .2], [.1, .9], [.4, .6], [.2, .8]]
    r = _prop(anchors, prob, gt, post_nms_top_n=4, nms_threshold=0.0)
    assert r['proposals'].shape == (2, 4)
    np.testing.assert_allclose(r['scores'], [.9, .8])
    for thr in (0.3, 0.6, 0.8):
        r = _prop(anchors, prob, gt, post_nms_top_n=4, nms_threshold=thr)
        assert r['proposals'].shape == (3, 4)
        np.testing.assert_allclose(r['scores'], [.9, .8, .2])
    r = _prop(anchors, prob, gt, post_nms_top_n=4, nms_threshold=1.0)
    assert r['proposals'].shape == (4, 4)


def test_rpn_proposal_outsiders_and_topn():
    # models/fasterrcnn/rpn_proposal_test.py:172-306
    gt = [[10, 10, 20, 22], [10, 10, 20, 22], [10, 10, 20, 50], [10, 10, 20, 22]]
    anchors = [[11, 13, 34, 31], [10, 10, 20, 22], [11, 13, 34, 40], [7, 13, 34, 30]]
    prob = [[.3, .7], [.4, .6], [.9, .1], [.8, .2]]
    r = _prop(anchors, prob, gt)
    assert r['proposals'].shape == (3, 4) and r['unsorted_proposals'].shape == (4, 4)
    np.testing.assert_allclose(r['scores'], [.7, .6, .2])
    r = _prop(anchors, prob, gt, post_nms_top_n=2)
    assert r['proposals'].shape == (2, 4)
    np.testing.assert_allclose(r['scores'], [.7, .6])
    np.testing.assert_allclose(r['sorted_top_scores'], [.7, .6, .2, .1])
    r = _prop(anchors, prob, gt, post_nms_top_n=3, pre_nms_top_n=2)
    assert r['proposals'].shape == (2, 4) and r['sorted_top_proposals'].shape == (2, 4)
    np.testing.assert_allclose(r['sorted_top_scores'], [.7, .6])
    r = _prop(anchors, prob, gt, post_nms_top_n=1, pre_nms_top_n=2)
    assert r['proposals'].shape == (1, 4)
    np.testing.assert_allclose(r['scores'], [.7])


def test_rpn_proposal_negative_area():
    # models/fasterrcnn/rpn_proposal_test.py:308-374
    gt = [[10, 10, 20, 3], [10, 10, 20, 22], [10, 10, 8, 22], [10, 10, 20, 22]]
    anchors = [[11, 13, 12, 16], [10, 10, 20, 22], [11, 13, 12, 19], [7, 13, 34, 30]]
    prob = [[.3, .7], [.4, .6], [.9, .1], [.8, .2]]
    r = _prop(anchors, prob, gt)
    assert r['proposals'].shape == (2, 4) and r['unsorted_proposals'].shape == (2, 4)
    anchors = [[11, 13, 12, 16], [10, 10, 9, 9], [11, 13, 12, 28], [7, 13, 34, 30]]
    r = _prop(anchors, prob, pred=np.zeros((4, 4), F))
    assert r['unsorted_proposals'].shape == (3, 4)


def test_rpn_proposal_clipping_and_outside_filter():
    # models/fasterrcnn/rpn_proposal_test.py:376-504
    anchors = [[-20, -10, 12, 6], [2, -10, 20, 20], [0, 0, 12, 16], [2, -10, 20, 2]]
    prob = [[.3, .7], [.4, .6], [.3, .7], [.1, .9]]
    z = np.zeros((4, 4), F)
    rb = _prop(anchors, prob, pred=z, clip_after_nms=False)
    unclipped = bx.decode(np.array(anchors, F), z)[rb['proposal_filter']]
    np.testing.assert_array_equal(rb['unsorted_proposals'], bx.clip_boxes(unclipped, (40, 40)))
    assert (rb['proposals'] >= 0).all() and (rb['proposals'] < 40).all()
    ra = _prop(anchors, prob, pred=z, clip_after_nms=True)
    np.testing.assert_array_equal(ra['unsorted_proposals'], unclipped)
    assert (ra['proposals'] >= 0).all() and (ra['proposals'] < 40).all()
    gt = [[0, 0, 10, 12], [10, 10, 20, 22], [10, 10, 20, 22], [30, 25, 39, 39], [30, 25, 39, 39]]
    anchors = [[-20, -10, 12, 6], [2, 10, 20, 20], [0, 0, 50, 16], [2, -10, 20, 50], [25, 30, 27, 33]]
    prob = [[.3, .7], [.4, .6], [.3, .7], [.1, .9], [.2, .8]]
    assert _prop(anchors, prob, gt)['all_proposals'].shape == (5, 4)
    assert _prop(anchors, prob, gt, filter_outside_anchors=True)['all_proposals'].shape == (2, 4)


# ---- A6 RPNTarget ---------------------------------------------------------
CFG6 = dict(allowed_border=0, clobber_positives=False, foreground_threshold=0.7,
            background_threshold_high=0.3, foreground_fraction=0.5, minibatch_size=2)
GT6 = np.array([[200, 0, 400, 400]], F)


def test_rpn_target_base_case():
    # models/fasterrcnn/rpn_target_test.py:44-92 ([1,0,-1]; which bg is dropped is TF-RNG specific)
    anchors = np.array([[200, 100, 400, 400], [300, 300, 400, 400], [200, 380, 300, 500]], F)
    labels, tg, mo, pre, _ = of.rpn_target(anchors, GT6, (600, 600), return_pre_subsample=True, **CFG6)
    np.testing.assert_array_equal(pre, [1, 0, 0])
    assert labels[0] == 1 and sorted(labels[1:]) == [-1, 0]
    np.testing.assert_array_equal(tg[1:], np.zeros((2, 4)))
    assert (tg[0] != 0).any() and mo.shape == (3,) and mo[0] >= 0.7 and (mo[1:] <= 0.3).all()


def test_rpn_target_border_outsiders():
    # models/fasterrcnn/rpn_target_test.py:94-152
    anchors = np.array([[200, 100, 400, 400], [300, 300, 400, 400], [200, 380, 300, 500],
                        [500, 500, 600, 650], [200, 100, 400, 400]])
    cfg = dict(CFG6, minibatch_size=5)
    labels, tg, _ = of.rpn_target(anchors, GT6, (600, 600), **cfg)
    np.testing.assert_array_equal(labels, [1, 0, 0, -1, 1])
    assert tg[0][0] == 0 and tg[0][2] == 0 and tg[0][1] != 0 and tg[0][3] != 0
    np.testing.assert_array_equal(tg[0], tg[-1])
    np.testing.assert_array_equal(tg[1:4], np.zeros((3, 4)))
    cfg['foreground_fraction'] = 0.2
    labels, tg, _ = of.rpn_target(anchors, GT6, (600, 600), **cfg)
    assert sorted([labels[0], labels[4]]) == [-1, 1]       # which fg is dropped: RNG-specific
    np.testing.assert_array_equal(labels[1:4], [0, 0, -1])


def test_rpn_target_no_clear_match_and_clobber():
    # models/fasterrcnn/rpn_target_test.py:154-190
    anchors = np.array([[300, 300, 400, 400], [200, 380, 300, 500]])
    labels, tg, _ = of.rpn_target(anchors, GT6, (600, 600), **CFG6)
    np.testing.assert_array_equal(labels, [1, 0])
    assert (tg[0] != 0).all() and (tg[1] == 0).all()
    labels, tg, _ = of.rpn_target(anchors, GT6, (600, 600), **dict(CFG6, clobber_positives=True))
    np.testing.assert_array_equal(labels, [0, 0])
    np.testing.assert_array_equal(tg, np.zeros((2, 4)))


def test_rpn_target_multiple_gt():
    # models/fasterrcnn/rpn_target_test.py:192-277
    anchors = np.array([[300, 300, 400, 390], [300, 300, 400, 400], [100, 310, 120, 380]], F)
    gt = np.array([[200, 0, 400, 400], [100, 300, 120, 375]], F)
    labels, tg, mo, pre, _ = of.rpn_target(anchors, gt, (600, 600), return_pre_subsample=True,
                                           **dict(CFG6, minibatch_size=3))
    np.testing.assert_array_equal(pre, [0, 1, 1])
    assert labels[0] == 0 and sorted(labels[1:]) == [-1, 1]
    anchors = np.array([[0, 0, 10, 10]] * 2 + [[10, 10, 20, 20]] * 2 + [[20, 20, 30, 30]] * 2 +
                       [[30, 30, 40, 40]] * 2 + [[100, 100, 110, 110], [100, 100, 120, 120]] +
                       [[110, 110, 120, 120], [110, 110, 130, 130]] * 3, F)
    gt = np.array([[2, 2, 8, 8], [12, 12, 18, 18], [22, 22, 28, 28], [32, 32, 38, 38]], F)
    labels, _, _ = of.rpn_target(anchors, gt, (600, 600), **dict(CFG6, minibatch_size=8))
    assert (labels == 1).sum() == 4 and (labels == 0).sum() == 4
    assert (labels.argsort(kind='stable')[-4:] < 8).all()


def test_rpn_target_zero_overlap_quirk():
    # a gt whose best IoU is 0 makes every inside anchor with IoU 0 positive.
    anchors = np.array([[0, 0, 10, 10], [20, 20, 30, 30], [40, 40, 50, 50]], F)
    gt = np.array([[100, 100, 120, 120]], F)
    _, _, _, pre, _ = of.rpn_target(anchors, gt, (600, 600), return_pre_subsample=True, **CFG6)
    np.testing.assert_array_equal(pre, [1, 1, 1])


# ---- A10 RCNNTarget -------------------------------------------------------
CFG10 = dict(foreground_threshold=0.5, background_threshold_high=0.5, background_threshold_low=0.1,
             foreground_fraction=0.5, minibatch_size=2)


def test_rcnn_target_basic_and_empty():
    # models/fasterrcnn/rcnn_target_test.py:52-134
    gt = [(20, 20, 80, 100, 3.)]
    props = [(55, 75, 85, 105), (25, 21, 85, 105), (78, 98, 99, 135)]
    label, _ = of.rcnn_target(props, gt, **CFG10)
    np.testing.assert_allclose(label, [0., 4., -1.], atol=1e-3)
    iou = bx.bbox_overlap(np.array(props, F), np.array(gt, F)[:, :4])[:, 0]
    np.testing.assert_allclose(iou, [0.1293, 0.7934, 0.0015], atol=1e-4)   # :68-73
    gt = [(423, 30, 501, 80, 3.)]
    props = [(491, 70, 510, 92), (400, 60, 450, 92), (413, 40, 480, 77), (411, 40, 480, 77)]
    label, _ = of.rcnn_target(props, gt, **CFG10)
    assert abs(label[2] - 4.) < 1e-3 and all(label[i] < 1 for i in (0, 1, 3))
    assert (label >= 0).sum() == 2


def test_rcnn_target_multiple_gt_and_priority():
    # models/fasterrcnn/rcnn_target_test.py:349-398, 475-524
    gt = [(10, 0, 398, 399, 0), (200, 300, 250, 390, 1), (185, 305, 235, 372, 2)]
    props = [(12, 70, 350, 540), (190, 310, 240, 370), (197, 300, 252, 389), (196, 300, 252, 389),
             (197, 303, 252, 394), (180, 310, 235, 370), (0, 0, 400, 400), (197, 302, 252, 389),
             (0, 0, 400, 400)]
    label, tg = of.rcnn_target(props, gt, **dict(CFG10, minibatch_size=18))
    np.testing.assert_allclose(label[1:], np.add([2., 1., 1., 1., 2., 0., 1., 0.], 1), atol=1e-3)
    assert ((label > 0) == (np.abs(tg).sum(1) > 0)).all()                 # :292-347
    gt = [[10, 10, 20, 20, 3.], [10, 10, 30, 30, 4.]]
    props = [[10, 10, 20, 20], [12, 10, 20, 20]]
    label, _ = of.rcnn_target(props, gt, **dict(CFG10, background_threshold_low=0.0, minibatch_size=64))
    assert (label == 4.).sum() == 1 and (label == 5.).sum() == 1


def test_rcnn_target_batch_invariants():
    # models/fasterrcnn/rcnn_target_test.py:136-290,400-473 (invariants over random inputs)
    rs = np.random.RandomState(1)
    for trial in range(20):
        G, P = 5, 300
        xy = rs.randint(0, 500, size=(G, 2))
        wh = rs.randint(20, 200, size=(G, 2))
        gt = np.concatenate([xy, xy + wh, rs.randint(0, 20, size=(G, 1))], 1).astype(F)
        xy = rs.randint(0, 600, size=(P, 2))
        wh = rs.randint(5, 250, size=(P, 2))
        props = np.concatenate([xy, xy + wh], 1).astype(F)
        props[:G] = gt[:, :4]
        label, tg = of.rcnn_target(props, gt, seed=trial, foreground_fraction=0.25, minibatch_size=64,
                                   foreground_threshold=0.5, background_threshold_high=0.5,
                                   background_threshold_low=0.0)
        assert (label > 0).sum() <= 16 and (label > 0).sum() > 0
        assert (label >= 0).sum() <= 64
        assert (np.abs(tg[label <= 0]).sum() == 0)


# ---- A11 ROI pooling ------------------------------------------------------
def _quadrants():
    m = np.block([[np.ones((5, 5)) * 1, np.ones((5, 5)) * 2], [np.ones((5, 5)) * 3, np.ones((5, 5)) * 4]])
    return m[None, :, :, None].astype(F)


def test_roi_pool_quadrants():
    # models/fasterrcnn/roi_pool_test.py:56-175
    fm = _quadrants()
    pooled, crops = of.roi_pool(np.array([[1, 1, 4, 4], [6, 1, 9, 4], [1, 6, 4, 9], [6, 6, 9, 9]]),
                                fm, (10, 10), 2, 2)
    assert crops.shape == (4, 4, 4, 1) and pooled.shape == (4, 2, 2, 1)
    for i in range(4):
        np.testing.assert_array_equal(pooled[i, :, :, 0], np.ones((2, 2)) * (i + 1))
    pooled, _ = of.roi_pool(np.array([[3, 1, 6, 4], [1, 3, 4, 7], [5, 3, 9, 7], [3, 6, 6, 9]]),
                            fm, (10, 10), 2, 2)
    p = pooled[..., 0]
    np.testing.assert_array_equal(p[0], [[1, 2], [1, 2]])
    np.testing.assert_array_equal(p[1], [[1, 1], [3, 3]])
    np.testing.assert_array_equal(p[2], [[2, 2], [4, 4]])
    np.testing.assert_array_equal(p[3], [[3, 4], [3, 4]])


def test_roi_pool_interpolation_bounds():
    # models/fasterrcnn/roi_pool_test.py:177-239
    fm = _quadrants()
    pooled, crops = of.roi_pool(np.array([[4, 1, 7, 4], [1, 4, 4, 8], [5, 4, 9, 8], [4, 6, 7, 9]]),
                                fm, (10, 10), 2, 2)
    lo, hi = [1, 1, 2, 3], [2, 3, 4, 4]
    for i in range(4):
        assert (pooled[i] >= lo[i]).all() and (crops[i] <= hi[i]).all()


# ---- A14 RCNNProposal -----------------------------------------------------
CFG14 = dict(num_classes=3, variances=None, class_max_detections=100, class_nms_threshold=0.6,
             total_max_detections=300, min_prob_threshold=0.0)


def _bbox_pred(props, gt_per_class):
    return np.concatenate([bx.encode(props, np.repeat(np.array(g, F), len(props), 0))
                           for g in gt_per_class], axis=1)


def test_rcnn_proposal_classes_and_nms():
    # models/fasterrcnn/rcnn_proposal_test.py:75-145
    props = np.array([(85, 500, 730, 590), (50, 500, 70, 530), (700, 570, 740, 598)], F)
    gts = [[(101, 101, 201, 249)], [(200, 502, 209, 532)], [(86, 571, 743, 599)]]
    prob = [(0., .3, .3, .4), (.8, 0., 0., 2.), (.35, .3, .2, .15)]
    r = of.rcnn_proposal(props, _bbox_pred(props, gts), prob, (900, 1440), **CFG14)
    assert len(r['objects']) == 3 and set(r['proposal_label']) == {0, 1, 2}
    props = np.array([(85, 500, 730, 590), (50, 500, 740, 570), (700, 570, 740, 598)], F)
    prob = [(0., .1, .3, .6), (.1, .2, .25, .45), (.2, .3, .25, .25)]
    r = of.rcnn_proposal(props, _bbox_pred(props, gts), prob, (900, 1440), **CFG14)
    assert len(r['objects']) == 3


def test_rcnn_proposal_clipping_bboxpred_limits():
    # models/fasterrcnn/rcnn_proposal_test.py:147-292
    props = np.array([(1300, 800, 1435, 870), (10, 1, 30, 7), (2, 870, 80, 898)], F)
    gts = [[(1320, 815, 1455, 912)], [(5, -8, 31, 8)], [(-120, 910, 78, 1040)]]
    prob = [(0., 1., 0., 0.), (.2, .25, .3, .25), (.45, 0., 0., .55)]
    for shape in ((1440, 900), (900, 1440)):
        r = of.rcnn_proposal(props, _bbox_pred(props, gts), prob, shape, **CFG14)
        o = r['objects']
        assert (o >= 0).all() and (o[:, [0, 2]] < shape[1]).all() and (o[:, [1, 3]] < shape[0]).all()
    props = np.array([(200, 315, 400, 370), (56, 0, 106, 4), (15, 15, 20, 20)], F)
    gts = [[(0, 0, 1, 1)], [(5, 5, 10, 10)], [(15, 15, 20, 20)]]
    r = of.rcnn_proposal(props, _bbox_pred(props, gts), prob, (900, 1440), **CFG14)
    objs = np.array([g[0] for g in gts], F)
    order = np.array(prob)[:, 1:].max(axis=1).argsort()[::-1]
    np.testing.assert_allclose(r['objects'], objs[order], atol=1e-3)
    props = np.array([(0, 0, 1, 1), (5, 5, 10, 10), (15, 15, 20, 20), (25, 25, 30, 30), (35, 35, 40, 40),
                      (38, 40, 65, 65), (70, 50, 90, 90), (95, 95, 100, 100), (105, 105, 110, 110)], F)
    prob = [(0., 1., 0.), (0., .2, .8), (0., .45, .55), (0., .55, .45), (1., 0., 0.), (1., 0., 0.),
            (0., .95, .05), (1., 0., 0.), (0., .495, .505)]
    r = of.rcnn_proposal(props, np.zeros((9, 8), F), prob, (900, 1440), num_classes=2,
                         variances=None, class_max_detections=2, class_nms_threshold=0.6, total_max_detections=3,
                         min_prob_threshold=0.0)
    lab = r['proposal_label']
    assert (lab == 0).sum() <= 2 and (lab == 1).sum() <= 2 and lab.shape[0] <= 3


# ---- losses ---------------------------------------------------------------
def test_losses_perfect_prediction_is_zero():
    # models/fasterrcnn/rpn_test.py:265-304, rcnn_test.py:303-402
    t = np.array([1, 0, -1, 1, 0], F)
    score = np.where(np.eye(2)[np.maximum(t, 0).astype(int)] > 0, 100., -100.).astype(F)
    tg = np.random.RandomState(0).randn(5, 4).astype(F)
    r = of.rpn_loss(score, t, tg, tg)
    assert r['rpn_cls_loss'] == 0 and r['rpn_reg_loss'] == 0
    C = 4
    ct = np.array([0, 2, -1, 4], F)
    cs = np.where(np.eye(C + 1)[np.maximum(ct, 0).astype(int)] > 0, 100., -100.).astype(F)
    off = np.zeros((4, 4 * C), F)
    tgt = np.zeros((4, 4), F)
    tgt[1] = [.1, .2, .3, .4]
    off[1, 4:8] = tgt[1]
    tgt[3] = [.5, .6, .7, .8]
    off[3, 12:16] = tgt[3]
    off[3, 0:4] = 9.  # wrong-class slots must be ignored
    r = of.rcnn_loss(cs, off, ct, tgt, C)
    assert abs(r['rcnn_cls_loss']) < 1e-3 and r['rcnn_reg_loss'] == 0


def test_smooth_l1_known_answer():
    # utils/losses.py:36-49 vectors; hand-computed with sigma=3: 1/9 threshold
    p = np.array([[0.47450006, -0.80413032, -0.26595005, 0.17124325]], F)
    t = np.array([[0.10058594, 0.07910156, 0.10555581, -0.1224325]], F)
    d = np.abs(p - t)[0].astype(np.float64)
    expect = sum(0.5 * 9 * x * x if x < 1 / 9. else x - 0.5 / 9 for x in d)
    np.testing.assert_allclose(of.smooth_l1_loss(p, t, 3.0)[0], expect, rtol=1e-6)


# ---- TF op restatements: documented behaviour -----------------------------
def test_topk_ties_lower_index_first():
    v, i = tfops.top_k(np.array([.5, .9, .5, .9, .1], F), 4)
    np.testing.assert_array_equal(i, [1, 3, 0, 2])


def test_nms_strict_greater_and_scalar_twin():
    b = np.array([[0, 0, 10, 10], [0, 0, 10, 5], [20, 20, 30, 30], [0, 0, 0, 10]], F)
    s = np.array([.9, .8, .7, .6], F)
    # IoU(0,1) = 0.5 exactly: suppressed only when thr < 0.5 (strict >)
    np.testing.assert_array_equal(tfops.non_max_suppression(b, s, 10, 0.5), [0, 1, 2, 3])
    np.testing.assert_array_equal(tfops.non_max_suppression(b, s, 10, 0.49), [0, 2, 3])
    assert tfops.nms_iou_greater(b[0], b[1], 0.49) and not tfops.nms_iou_greater(b[0], b[1], 0.5)
    assert not tfops.nms_iou_greater(b[0], b[3], 0.0)       # zero-area box never suppresses
    np.testing.assert_array_equal(tfops.non_max_suppression(b, s, 2, 0.49), [0, 2])
