"""The oracle against fixtures produced by RUNNING the reference's TensorFlow-graph code (tests/golden/
make_golden_ref_tf.py: luminoth/models/{fasterrcnn,ssd}/*.py and luminoth/utils/*_tf.py loaded by path on top of the
eager numpy `tf` stand-in tests/golden/tf_numpy_shim.py).  Index / label / keep decisions: bit-exact; fp32 values:
bit-exact where the oracle performs the same fp32 operations in the same order, 1e-6 relative otherwise (stated at
each assert).  The HIP kernels replay the same fixtures in tests/test_gpu_ref_tf_golden.py."""
import os

import numpy as np
import pytest

from oracle import boxes as obx
from oracle import frcnn as of
from oracle import ssd as ossd
from oracle import tfops

F = np.float32
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'ref_tf_golden.npz')


@pytest.fixture(scope='module')
def G():
    return np.load(GOLD)


def names(prefix):
    z = np.load(GOLD)
    return sorted({k.split('/')[1] for k in z.files if k.startswith(prefix + '/')})


def test_fixture_is_fresh_when_reference_present():
    """In the build container the committed fixture must be what the generator produces from /root/reference today."""
    if not os.path.isdir('/root/reference/luminoth'):
        pytest.skip('reference tree not present (GPU box)')
    import subprocess
    import sys
    import tempfile
    gen = os.path.join(os.path.dirname(GOLD), 'make_golden_ref_tf.py')
    with tempfile.TemporaryDirectory() as d:
        code = ("import sys, runpy, numpy as np, os\n"
                "sys.argv=['x']\n"
                "ns = runpy.run_path(%r)\n"
                "m = ns['load_reference'](); out = {}\n"
                "for f in ('gen_box_utils','gen_rpn_target','gen_rcnn_target','gen_rpn_proposal','gen_rcnn_proposal',"
                "'gen_roi_pool','gen_losses','gen_ssd','gen_heads'): ns[f](m, out)\n"
                "np.savez(%r, **{k: np.asarray(v) for k, v in out.items()})\n") % (gen, os.path.join(d, 'x.npz'))
        subprocess.check_call([sys.executable, '-c', code], stdout=subprocess.DEVNULL)
        new, old = np.load(os.path.join(d, 'x.npz')), np.load(GOLD)
        assert sorted(new.files) == sorted(old.files)
        for k in old.files:
            np.testing.assert_array_equal(new[k], old[k], err_msg=k)


# ------------------------------------------------------------------------------------------ A7 / A8 / losses ----
def test_box_transforms_match_reference_tf_code(G):
    a, g, d = G['box/a'], G['box/g'], G['box/d']
    np.testing.assert_array_equal(obx.encode(a, g), G['box/encode'])
    np.testing.assert_array_equal(obx.encode(a, g, variances=(0.1, 0.2)), G['box/encode_var'])
    np.testing.assert_array_equal(obx.decode(a, d), G['box/decode'])
    np.testing.assert_array_equal(obx.decode(a, d, variances=(0.1, 0.2)), G['box/decode_var'])
    np.testing.assert_array_equal(obx.clip_boxes(G['box/wild'], (600, 800)), G['box/clip'])
    np.testing.assert_array_equal(a[:, [1, 0, 3, 2]], G['box/change_order'])


def test_iou_matches_reference_tf_code(G):
    np.testing.assert_array_equal(obx.bbox_overlap(G['box/a'], G['box/g'][:9]), G['box/iou'])
    np.testing.assert_array_equal(obx.bbox_overlap(G['box/neg'], G['box/g'][:9]), G['box/iou_neg'])
    assert (G['box/iou_neg'][::7] == 0).all()          # negative-area rows: 0 through the outer max(., 0)


def test_smooth_l1_matches_reference_tf_code(G):
    np.testing.assert_array_equal(of.smooth_l1_loss(G['box/sl1_p'], G['box/sl1_t'], sigma=3.0), G['box/sl1_s3'])
    np.testing.assert_array_equal(of.smooth_l1_loss(G['box/sl1_p'], G['box/sl1_t'], sigma=1.0), G['box/sl1_s1'])


# ---------------------------------------------------------------------------------------------------- A6 ----
@pytest.mark.parametrize('name', names('rpn_target'))
def test_rpn_target_matches_reference_graph(G, name):
    k = 'rpn_target/%s/' % name
    fh, fw, stride, H, W = [int(v) for v in G[k + 'geom']]
    border, clobber, fg_thr, bg_thr, fg_frac, mb = G[k + 'cfg']
    ref = G[k + 'ref_i32']
    anchors = (ref[None, None] + (np.stack(np.meshgrid(np.arange(fw), np.arange(fh)), -1) * stride)[
        ..., [0, 1, 0, 1]][:, :, None]).reshape(-1, 4).astype(np.int32)
    np.testing.assert_array_equal(anchors, obx.generate_anchors(ref.astype(np.float64), fh, fw, stride))
    labels, targets, max_ov = of.rpn_target(
        anchors, G[k + 'gt'], (H, W), seed=int(G[k + 'seed'][0]), allowed_border=int(border),
        clobber_positives=bool(clobber), foreground_threshold=fg_thr, background_threshold_high=bg_thr,
        foreground_fraction=fg_frac, minibatch_size=int(mb))
    np.testing.assert_array_equal(labels, G[k + 'labels'])          # incl. WHICH anchors the subsample dropped
    np.testing.assert_array_equal(max_ov, G[k + 'max_ov'])
    np.testing.assert_array_equal(targets, G[k + 'targets'])


def test_rpn_target_zero_overlap_gt_quirk_is_in_the_fixture(G):
    """Appendix B.4 (rpn_target.py:155-178): a gt whose best IoU is 0 makes EVERY inside anchor tie its column
    maximum; the reference then subsamples 128 of them and leaves no room for... 128 backgrounds."""
    lab = G['rpn_target/zero_overlap_gt/labels']
    assert (lab == 1).sum() == 128 and (lab == 0).sum() == 0


# --------------------------------------------------------------------------------------------------- A10 ----
@pytest.mark.parametrize('name', names('rcnn_target'))
def test_rcnn_target_matches_reference_graph(G, name):
    k = 'rcnn_target/%s/' % name
    fg_frac, mb, fg_thr, bg_hi, bg_lo = G[k + 'cfg']
    labels, targets = of.rcnn_target(G[k + 'proposals'], G[k + 'gt'], seed=int(G[k + 'seed'][0]),
                                     foreground_fraction=fg_frac, minibatch_size=int(mb), foreground_threshold=fg_thr,
                                     background_threshold_high=bg_hi, background_threshold_low=bg_lo)
    np.testing.assert_array_equal(labels, G[k + 'labels'])
    np.testing.assert_array_equal(targets, G[k + 'targets'])


# ---------------------------------------------------------------------------------------------------- A5 ----
@pytest.mark.parametrize('name', names('rpn_proposal'))
def test_rpn_proposal_matches_reference_graph(G, name):
    k = 'rpn_proposal/%s/' % name
    fh, fw, stride, H, W = [int(v) for v in G[k + 'geom']]
    pre, post, apply_nms, thr, filt, clip_after, min_prob = G[k + 'cfg']
    anchors = obx.generate_anchors(G[k + 'ref_i32'].astype(np.float64), fh, fw, stride)
    np.testing.assert_array_equal(tfops.softmax(G[k + 'score']), G[k + 'prob'])
    r = of.rpn_proposal(G[k + 'prob'], G[k + 'pred'], anchors, (H, W), pre_nms_top_n=int(pre),
                        post_nms_top_n=int(post), nms_threshold=thr, apply_nms=bool(apply_nms),
                        clip_after_nms=bool(clip_after), filter_outside_anchors=bool(filt),
                        min_prob_threshold=min_prob)
    np.testing.assert_array_equal(r['sorted_top_scores'], G[k + 'sorted_top_scores'])
    np.testing.assert_array_equal(r['scores'], G[k + 'scores'])
    np.testing.assert_array_equal(r['proposals'], G[k + 'proposals'])


# --------------------------------------------------------------------------------------------------- A14 ----
@pytest.mark.parametrize('name', names('rcnn_proposal'))
def test_rcnn_proposal_matches_reference_graph(G, name):
    k = 'rcnn_proposal/%s/' % name
    C, H, W, cmax, cthr, tmax, minp = G[k + 'cfg']
    r = of.rcnn_proposal(G[k + 'proposals'], G[k + 'pred'], G[k + 'prob'], (int(H), int(W)), int(C),
                         class_max_detections=int(cmax), class_nms_threshold=cthr, total_max_detections=int(tmax),
                         min_prob_threshold=minp)
    np.testing.assert_array_equal(r['proposal_label'], G[k + 'labels'])
    np.testing.assert_array_equal(r['proposal_label_prob'], G[k + 'probs'])
    np.testing.assert_array_equal(r['objects'], G[k + 'objects'])


# --------------------------------------------------------------------------------------------------- A11 ----
def test_roi_pool_matches_reference_graph(G):
    pooled, crops = of.roi_pool(G['roi_pool/rois'], G['roi_pool/feat'], tuple(G['roi_pool/im_shape']))
    np.testing.assert_array_equal(of.roi_normalised_boxes(G['roi_pool/rois'], tuple(G['roi_pool/im_shape'])),
                                  G['roi_pool/bboxes'])
    np.testing.assert_array_equal(crops, G['roi_pool/crops'])
    np.testing.assert_array_equal(pooled, G['roi_pool/pooled'])


# ---------------------------------------------------------------------------------------------- A9 / A15 ----
def test_rpn_loss_matches_reference_graph(G):
    r = of.rpn_loss(G['rpn_loss/rpn_cls_score'], G['rpn_loss/rpn_cls_target'], G['rpn_loss/rpn_bbox_pred'],
                    G['rpn_loss/rpn_bbox_target'], l1_sigma=3.0)
    # means of <= 600 fp32 terms: summation order differs (numpy pairwise vs ours) -> 1e-6 relative
    np.testing.assert_allclose(r['rpn_cls_loss'], G['rpn_loss/rpn_cls_loss'], rtol=1e-6)
    np.testing.assert_allclose(r['rpn_reg_loss'], G['rpn_loss/rpn_reg_loss'], rtol=1e-6)


def test_rcnn_loss_matches_reference_graph(G):
    r = of.rcnn_loss(G['rcnn_loss/cls_score'], G['rcnn_loss/bbox_offsets'], G['rcnn_loss/cls_target'],
                     G['rcnn_loss/bbox_target'], 20, l1_sigma=1.0)
    np.testing.assert_allclose(r['rcnn_cls_loss'], G['rcnn_loss/rcnn_cls_loss'], rtol=1e-6)
    np.testing.assert_allclose(r['rcnn_reg_loss'], G['rcnn_loss/rcnn_reg_loss'], rtol=1e-6)


# ---------------------------------------------------------------------------------------------------- SSD ----
@pytest.mark.parametrize('name', names('ssd_target'))
def test_ssd_target_matches_reference_graph(G, name):
    k = 'ssd_target/%s/' % name
    ratio, fg_thr, bg_hi = G[k + 'cfg']
    labels, targets = ossd.ssd_target(G[k + 'probs'], G['ssd/anchors'], G[k + 'gt'], hard_negative_ratio=ratio,
                                      foreground_threshold=fg_thr, background_threshold_high=bg_hi)
    np.testing.assert_array_equal(labels, G[k + 'labels'])
    np.testing.assert_array_equal(targets, G[k + 'targets'])


@pytest.mark.parametrize('name', names('ssd_proposal'))
def test_ssd_proposal_matches_reference_graph(G, name):
    k = 'ssd_proposal/%s/' % name
    C, thr, cmax, tmax, minp = G[k + 'cfg']
    r = ossd.ssd_proposal(G[k + 'prob'], G[k + 'loc'], G['ssd/anchors'], (150, 150), int(C), class_nms_threshold=thr,
                          class_max_detections=int(cmax), total_max_detections=int(tmax), min_prob_threshold=minp)
    np.testing.assert_array_equal(r['labels'], G[k + 'labels'])
    np.testing.assert_array_equal(r['probs'], G[k + 'probs'])
    np.testing.assert_array_equal(r['objects'], G[k + 'objects'])
    np.testing.assert_array_equal(r['anchors'], G[k + 'anchors_out'])
    np.testing.assert_array_equal(r['raw_proposals'], G[k + 'raw_proposals'])


@pytest.mark.parametrize('name', ['mixed', 'no_positives'])
def test_ssd_loss_matches_reference_graph(G, name):
    k = 'ssd_loss/%s/' % name
    final, cls_loss, bbox_loss = ossd.ssd_loss(G[k + 'cls_pred'], G[k + 'loc_pred'], G[k + 'cls_target'],
                                               G[k + 'bbox_target'], 20, loc_loss_weight=float(G[k + 'loc_weight']))
    np.testing.assert_allclose(cls_loss, G[k + 'cls_loss'], rtol=1e-6)
    np.testing.assert_allclose(bbox_loss, G[k + 'bbox_loss'], rtol=1e-6)
    np.testing.assert_allclose(final + G[k + 'reg'], G[k + 'total_loss'], rtol=1e-6)      # + regularisation collection
    if name == 'no_positives':
        assert float(final) == 0.0                                                         # ssd.py:262-270


# ------------------------------------------------------------------------- A4 / A13 / S2: head layouts (round 5) ----
def _seeded(name, var, shape):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    from tf_numpy_shim import seeded_variable
    return seeded_variable(name, var, shape)


def test_rpn_head_layout_matches_reference_build(G):
    """rpn.py:148-172 executed by the generator (numpy Sonnet Conv2D): the oracle's `rpn_head` puts the same anchor /
    class / coordinate in the same place of `(N,2)` / `(N,4)`; values 1e-5 (two float32 convolution orders)."""
    import torch
    from oracle.model import OracleFasterRCNN
    k = 'heads/rpn/'
    feat = G[k + 'feat']
    fh, fw, stride, H, W, ch = (int(v) for v in G[k + 'geom'])
    cin, A = feat.shape[3], G[k + 'ref_i32'].shape[0]
    p = 'fasterrcnn/rpn'
    sd = {p + '/conv/w': _seeded('conv', 'w', (3, 3, cin, ch)), p + '/conv/b': _seeded('conv', 'b', (ch,)),
          p + '/cls_conv/w': _seeded('cls_conv', 'w', (1, 1, ch, 2 * A)), p + '/cls_conv/b': _seeded('cls_conv', 'b', (2 * A,)),
          p + '/bbox_conv/w': _seeded('bbox_conv', 'w', (1, 1, ch, 4 * A)), p + '/bbox_conv/b': _seeded('bbox_conv', 'b', (4 * A,))}
    o = OracleFasterRCNN(sd, num_classes=5)
    cls, box = o.rpn_head(torch.from_numpy(feat))
    np.testing.assert_allclose(cls.numpy(), G[k + 'rpn_cls_score'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(box.numpy(), G[k + 'rpn_bbox_pred'], rtol=1e-5, atol=1e-5)
    # a layout error is O(1): swapping the two columns or two anchors must fail this comparison by a wide margin
    assert np.abs(G[k + 'rpn_cls_score'][:, 0] - G[k + 'rpn_cls_score'][:, 1]).mean() > 0.1
    # ... and the rest of _build on those scores: anchor targets (bit-exact) and the proposals of its own softmax
    anchors = obx.generate_anchors(obx.generate_anchors_reference(64, np.array([0.5, 1, 2]), np.array([0.25, 0.5, 1, 2])),
                                   fh, fw, stride)
    labels, targets, _ = of.rpn_target(anchors, G[k + 'gt'], (H, W), seed=int(G[k + 'seed'][0]))
    np.testing.assert_array_equal(labels, G[k + 'rpn_cls_target'])
    np.testing.assert_allclose(targets, G[k + 'rpn_bbox_target'], rtol=1e-6, atol=1e-6)
    r = of.rpn_proposal(G[k + 'rpn_cls_prob'], G[k + 'rpn_bbox_pred'], anchors, (H, W), pre_nms_top_n=12000,
                        post_nms_top_n=2000, nms_threshold=0.7)
    np.testing.assert_allclose(r['proposals'], G[k + 'proposals'], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('case', ['mean', 'flatten_fc'])
def test_rcnn_head_layout_matches_reference_build(G, case):
    """rcnn.py:149-239 executed by the generator: targets + training-batch compaction (order of the kept proposals),
    crop pooling, spatial mean OR row-major flatten of (7,7,C), the FC stack, `(R,C+1)` / `(R,4C)`."""
    import torch
    from oracle import torch_ops as ot
    k = 'heads/rcnn_%s/' % case
    feat, props, gt = G[k + 'feat'], G[k + 'proposals'], G[k + 'gt']
    geom = [int(v) for v in G[k + 'geom']]
    H, W, C, use_mean, sizes = geom[0], geom[1], geom[2], bool(geom[3]), geom[4:]
    lab, tg = of.rcnn_target(props, gt, seed=int(G[k + 'seed'][0]), minibatch_size=32)
    keep = lab >= 0
    np.testing.assert_array_equal(lab[keep], G[k + 'target_cls'])
    np.testing.assert_allclose(tg[keep], G[k + 'target_bbox'], rtol=1e-6, atol=1e-6)
    rois = torch.from_numpy(props[keep])
    pooled = ot.roi_pool(torch.from_numpy(feat), rois, torch.zeros(rois.shape[0], dtype=torch.long), (H, W))
    net = pooled.mean(dim=(1, 2)) if use_mean else pooled.reshape(pooled.shape[0], -1)
    for i, size in enumerate(sizes):
        net = torch.relu(net @ torch.from_numpy(_seeded('fc_%d' % i, 'w', (net.shape[1], size))) +
                         torch.from_numpy(_seeded('fc_%d' % i, 'b', (size,))))
    cls = net @ torch.from_numpy(_seeded('fc_classifier', 'w', (net.shape[1], C + 1))) + \
        torch.from_numpy(_seeded('fc_classifier', 'b', (C + 1,)))
    box = net @ torch.from_numpy(_seeded('fc_bbox', 'w', (net.shape[1], 4 * C))) + torch.from_numpy(_seeded('fc_bbox', 'b', (4 * C,)))
    np.testing.assert_allclose(cls.numpy(), G[k + 'cls_score'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(box.numpy(), G[k + 'bbox_offsets'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(torch.softmax(cls, 1).numpy(), G[k + 'cls_prob'], rtol=1e-5, atol=1e-6)


def _ssd_head_variables(G, scope='ssd'):
    geom = [int(v) for v in G['heads/ssd/geom']]
    C, app = geom[2], geom[3:]
    names = ['vgg_16/conv4/conv4_3', 'vgg_16/fc7', 'conv6_2', 'conv7_2', 'conv8_2', 'conv9_2']
    maps = [G['heads/ssd/fmap/' + n.replace('/', '.')] for n in names]
    sd = {}
    for i, (fm, a) in enumerate(zip(maps, app)):
        for kind, cout in (('offsets', a * 4), ('classes', a * (C + 1))):
            n = 'MultiBox_%d_%s_conv' % (i, kind)
            sd['%s/%s/w' % (scope, n)] = _seeded(n, 'w', (3, 3, fm.shape[3], cout))
            sd['%s/%s/b' % (scope, n)] = _seeded(n, 'b', (cout,))
    return sd, maps, C, app, geom[:2]


def test_ssd_head_layout_matches_reference_build(G):
    """ssd/ssd.py:73-195 executed by the generator over given feature maps: six multibox head pairs, `[-1,4]` /
    `[-1,C+1]` reshapes, concat order, anchors in the same order, targets + hard-negative filter, proposals."""
    import torch
    from oracle.ssd_model import OracleSSD
    sd, maps, C, app, (H, W) = _ssd_head_variables(G)
    o = OracleSSD(sd, num_classes=C, anchors_per_point=app)
    loc, cls = o.heads([torch.from_numpy(m) for m in maps])
    np.testing.assert_allclose(cls.numpy(), G['heads/ssd_predict/cls_pred'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(loc.numpy(), G['heads/ssd_predict/loc_pred'], rtol=1e-5, atol=1e-5)
    anchors = ossd.all_anchors([(m.shape[1], m.shape[2]) for m in maps], (H, W, 3), anchors_per_point=app)
    probs = torch.softmax(cls, 1).numpy()
    labels, targets = ossd.ssd_target(probs, anchors, G['heads/ssd/gt'], variances=(0.1, 0.2), hard_negative_ratio=3.0,
                                      foreground_threshold=0.5, background_threshold_high=0.2)
    keep = labels >= 0
    np.testing.assert_array_equal(labels[keep], G['heads/ssd_train/target_cls'])
    np.testing.assert_array_equal(anchors[keep], G['heads/ssd_train/target_anchors'])
    np.testing.assert_allclose(targets[keep], G['heads/ssd_train/target_bbox'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(cls.numpy()[keep], G['heads/ssd_train/cls_pred'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(loc.numpy()[keep], G['heads/ssd_train/loc_pred'], rtol=1e-5, atol=1e-5)
