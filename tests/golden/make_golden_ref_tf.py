"""Generates tests/golden/ref_tf_golden.npz by RUNNING the reference's own TensorFlow-graph code for the box stages
of the hot path, in the build container, on top of the eager numpy stand-in `tests/golden/tf_numpy_shim.py`.

Reference code executed (loaded BY PATH from /root/reference/luminoth, never copied):

    utils/bbox_transform_tf.py:4-126      get_width_upright / encode / decode / clip_boxes / change_order   (A8)
    utils/bbox_overlap.py:7-48            bbox_overlap_tf                                                  (A7)
    utils/losses.py:4-32                  smooth_l1_loss
    models/fasterrcnn/rpn_target.py:73-335    RPNTarget._build  — INCLUDING the fg / bg subsample           (A6)
    models/fasterrcnn/rcnn_target.py:48-299   RCNNTarget._build — INCLUDING the fg / bg subsample           (A10)
    models/fasterrcnn/rpn_proposal.py:41-197  RPNProposal._build                                           (A5)
    models/fasterrcnn/rcnn_proposal.py:46-164 RCNNProposal._build                                          (A14)
    models/fasterrcnn/roi_pool.py:37-95       ROIPoolingLayer._build (crop mode)                           (A11)
    models/fasterrcnn/rpn.py:219-309          RPN.loss                                                     (A9)
    models/fasterrcnn/rcnn.py:255-411         RCNN.loss                                                    (A15)
    models/ssd/target.py:35-200               SSDTarget._build                                             (S4)
    models/ssd/proposal.py:41-171             SSDProposal._build                                           (S5)
    models/ssd/ssd.py:197-300                 SSD.loss                                                     (S6)
    models/fasterrcnn/rpn.py:23-217           RPN.__init__ / _instantiate_layers / _build                  (A4)
    models/fasterrcnn/rcnn.py:41-250          RCNN.__init__ / _instantiate_layers / _build                 (A13)
    models/ssd/ssd.py:21-195                  SSD.__init__ / _build (multibox heads, anchors, targets,      (S2)
                                              hard-negative filter, proposals) over GIVEN feature maps
    utils/vars.py:1-130                       get_initializer / get_activation_function / summaries
    models/fasterrcnn/fasterrcnn.py:22-358    FasterRCNN.__init__ / _build / loss / _generate_anchors /             (A16)
                                              get_trainable_vars / get_base_network_checkpoint_vars
    models/base/base_network.py:39-259        BaseNetwork (arg_scope, network, _build, preprocess, variable lists)   (A1, A2 names)
    models/base/truncated_base_network.py:28-169  TruncatedBaseNetwork (_build, _build_tail, get_trainable_vars)
    utils/anchors.py:4-52                     generate_anchors_reference                                            (A3)

The three head `_build`s run on numpy `snt.Conv2D` / `snt.Linear` (tf_numpy_shim.py) whose variables are a fixed
function of the module NAME (`seeded_variable`), so the replaying tests rebuild the same weights; what they pin is the
reference's composition: which output channel is which anchor / class / coordinate after its reshapes and concats
(`(N,2)` / `(N,4)`, `(R,C+1)` / `(R,4C)`, the flatten order in front of the FC stack, the order of the six multibox
heads and of their anchors).  `SSDFeatureExtractor` (slim VGG: third party, absent) is replaced by a stand-in that returns
the fixture's feature maps; `base_network._build_tail` by the identity (the ResNet-50 / VGG case: no tail).

`tf.random_shuffle` (rpn_target.py:206,243; rcnn_target.py:172,223) is handed the permutation "descending
(hash(seed, stream, index), index)" of the shared counter RNG (oracle/rng.py == csrc/lmh_common.h lmh_hash_u32): the
reference drops the FIRST n-k shuffled entries, so it keeps exactly the k smallest keys — the subset the oracle and
the HIP kernels keep.  Any permutation is a legal outcome of tf.random_shuffle; this one makes the reference's
post-subsample outputs comparable bit for bit.

/root/reference does not exist on the GPU box, so the outputs are committed; tests/test_ref_tf_golden.py replays
them through oracle/{boxes,frcnn,ssd}.py (CPU) and tests/test_gpu_ref_tf_golden.py through the HIP kernels.

    python tests/golden/make_golden_ref_tf.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import tf_numpy_shim as tf  # noqa: E402
from oracle import boxes as obx  # noqa: E402
from oracle import rng as orng  # noqa: E402
from oracle import ssd as ossd  # noqa: E402

REF = '/root/reference/luminoth'
F = np.float32
ED = tf.EasyDict


def load_reference():
    """Loads the reference modules by path under their real dotted names (so their own cross imports resolve to the
    reference code, not to stand-ins)."""
    tf.install()
    for name in ('luminoth', 'luminoth.utils', 'luminoth.models', 'luminoth.models.fasterrcnn', 'luminoth.models.ssd',
                 'luminoth.models.base', 'luminoth.utils.vars', 'luminoth.models.ssd.feature_extractor',
                 'luminoth.utils.anchors', 'luminoth.models.ssd.utils'):
        if name not in sys.modules or not isinstance(sys.modules[name], types.ModuleType):
            sys.modules[name] = tf._Inert(name)
        sys.modules[name].__path__ = []
    mods = {}
    for rel in ('utils/vars.py', 'models/ssd/utils.py',
                'utils/bbox_transform_tf.py', 'utils/bbox_transform.py', 'utils/bbox_overlap.py', 'utils/losses.py',
                'models/fasterrcnn/rpn_target.py', 'models/fasterrcnn/rpn_proposal.py',
                'models/fasterrcnn/rcnn_target.py', 'models/fasterrcnn/rcnn_proposal.py',
                'models/fasterrcnn/roi_pool.py', 'models/fasterrcnn/rpn.py', 'models/fasterrcnn/rcnn.py',
                'models/ssd/target.py', 'models/ssd/proposal.py', 'models/ssd/ssd.py'):
        name = 'luminoth.' + rel[:-3].replace('/', '.')
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        mods[rel[:-3].split('/')[-1] if 'ssd' not in rel else 'ssd_' + rel[:-3].split('/')[-1]] = mod
    return mods


# --------------------------------------------------------------------------------------------- input makers ----
def rand_boxes(rs, n, W, H, smin, smax):
    w = rs.randint(smin, smax, size=n)
    h = rs.randint(smin, smax, size=n)
    x = np.array([rs.randint(0, max(1, W - wi)) for wi in w])
    y = np.array([rs.randint(0, max(1, H - hi)) for hi in h])
    return np.stack([x, y, x + w - 1, y + h - 1], 1).astype(F)


def anchor_grid(base, ratios, scales, fh, fw, stride):
    ref = obx.generate_anchors_reference(base, np.array(ratios), np.array(scales))
    ref_i32 = np.trunc(ref).astype(np.int32)                       # fasterrcnn.py:299-302 int32 quirk
    return ref_i32, obx.generate_anchors(ref, fh, fw, stride).astype(np.int32)


def shuffle_by_counter_rng(seed_u32, streams, index_map=None):
    """The tf.random_shuffle stand-in: descending (hash, index) so that the reference's `[:n-k]` prefix is the set
    with the LARGEST keys and the k smallest survive (oracle/rng.py keep_k_smallest)."""
    def hook(value, seed, caller):
        stream = streams[caller]
        idx = np.asarray(value).reshape(value.shape[0], -1)[:, 0].astype(np.int64)
        full = idx if index_map is None else index_map[idx]
        keys = (orng.hash_u32(seed_u32, stream, full).astype(np.uint64) << np.uint64(32)) | full.astype(np.uint64)
        order = np.argsort(keys, kind='stable')[::-1]
        return np.asarray(value)[order]
    return hook


def lattice_probs(rs, n):
    """Foreground probabilities (perm+0.5)/n: pairwise gaps of 1/n survive the few-ulp differences between numpy's
    and the device's expf in the kernel's own softmax, so order decisions are comparable on fixed inputs."""
    p = (rs.permutation(n).astype(np.float64) + 0.5) / n
    logit = np.log(p / (1 - p))
    return np.stack([np.zeros(n), logit], 1).astype(F)


# --------------------------------------------------------------------------------------------------- cases ----
def gen_box_utils(m, out):
    rs = np.random.RandomState(100)
    bt = m['bbox_transform_tf']
    a = rand_boxes(rs, 64, 800, 600, 8, 300)
    g = rand_boxes(rs, 64, 800, 600, 8, 300)
    d = (rs.randn(64, 4) * 0.4).astype(F)
    out['box/a'], out['box/g'], out['box/d'] = a, g, d
    out['box/encode'] = bt.encode(a, g)
    out['box/encode_var'] = bt.encode(a, g, variances=[0.1, 0.2])
    out['box/decode'] = bt.decode(a, d)
    out['box/decode_var'] = bt.decode(a, d, variances=[0.1, 0.2])
    wild = (a + rs.randint(-200, 200, size=a.shape)).astype(F)
    out['box/wild'] = wild
    out['box/clip'] = bt.clip_boxes(wild, np.array([600, 800], np.int32))
    out['box/change_order'] = bt.change_order(a)
    out['box/iou'] = m['bbox_overlap'].bbox_overlap_tf(a, g[:9])
    neg = a.copy()
    neg[::7, 2] = neg[::7, 0] - 5                                   # negative-width boxes (bbox_overlap_test.py:69-84)
    out['box/neg'] = neg
    out['box/iou_neg'] = m['bbox_overlap'].bbox_overlap_tf(neg, g[:9])
    p, t = (rs.randn(40, 4)).astype(F), (rs.randn(40, 4) * 0.3).astype(F)
    out['box/sl1_p'], out['box/sl1_t'] = p, t
    out['box/sl1_s3'] = m['losses'].smooth_l1_loss(p, t)
    out['box/sl1_s1'] = m['losses'].smooth_l1_loss(p, t, sigma=1.0)


RPN_TARGET_CASES = [
    # (name, feat_h, feat_w, base, G, cfg overrides, special)
    ('default', 20, 25, 64, 5, {}, None),
    ('clobber', 20, 25, 64, 5, {'clobber_positives': True}, None),
    ('border', 20, 25, 64, 3, {'allowed_border': 24}, None),
    ('small_batch', 20, 25, 64, 8, {'minibatch_size': 64, 'foreground_fraction': 0.25}, None),
    ('thresholds', 16, 16, 48, 6, {'foreground_threshold': 0.5, 'background_threshold_high': 0.4}, None),
    ('one_gt', 12, 16, 64, 1, {}, None),
    ('zero_overlap_gt', 12, 16, 64, 2, {}, 'outside_gt'),
    ('many_gt', 20, 25, 64, 40, {'minibatch_size': 128}, None),
    ('big_anchors', 20, 25, 256, 4, {}, None),                      # default refs: most anchors cross the border
]


def gen_rpn_target(m, out):
    for ci, (name, fh, fw, base, G, over, special) in enumerate(RPN_TARGET_CASES):
        rs = np.random.RandomState(200 + ci)
        stride = 16
        H, W = fh * stride, fw * stride
        ref_i32, anchors = anchor_grid(base, [0.5, 1, 2], [0.25, 0.5, 1, 2], fh, fw, stride)
        gt = np.concatenate([rand_boxes(rs, G, W, H, 16, min(H, W) // 2), rs.randint(0, 20, size=(G, 1))], 1).astype(F)
        if special == 'outside_gt':
            gt[1, :4] = [W + 40, H + 30, W + 90, H + 100]           # IoU 0 with every inside anchor
        cfg = dict(allowed_border=0, clobber_positives=False, foreground_threshold=0.7,
                   background_threshold_high=0.3, foreground_fraction=0.5, minibatch_size=256)
        cfg.update(over)
        seed = orng.image_seed(ci, 3, 1)
        b = cfg['allowed_border']
        inside = np.where((anchors[:, 0] >= -b) & (anchors[:, 1] >= -b) & (anchors[:, 2] < W + b) &
                          (anchors[:, 3] < H + b))[0]
        tf.set_random_shuffle(shuffle_by_counter_rng(
            seed, {'subsample_positive': orng.STREAM_RPN_FG, 'subsample_negative': orng.STREAM_RPN_BG}, inside))
        mod = m['rpn_target'].RPNTarget(ref_i32.shape[0], ED(cfg), seed=None)
        labels, targets, max_ov = mod(anchors, gt, np.array([H, W], np.int32))
        k = 'rpn_target/%s/' % name
        out[k + 'ref_i32'], out[k + 'gt'] = ref_i32, gt
        out[k + 'geom'] = np.array([fh, fw, stride, H, W], np.int32)
        out[k + 'seed'] = np.array([seed], np.uint32)
        out[k + 'cfg'] = np.array([cfg['allowed_border'], int(cfg['clobber_positives']), cfg['foreground_threshold'],
                                   cfg['background_threshold_high'], cfg['foreground_fraction'],
                                   cfg['minibatch_size']], np.float64)
        out[k + 'labels'], out[k + 'targets'], out[k + 'max_ov'] = labels, targets, max_ov
        assert labels.dtype == F and targets.dtype == F and max_ov.dtype == F
        print('rpn_target %-16s N=%5d inside=%5d fg=%3d bg=%3d' % (name, anchors.shape[0], inside.shape[0],
                                                                    (labels == 1).sum(), (labels == 0).sum()))


RCNN_TARGET_CASES = [
    ('default', 300, 5, {}),
    ('full', 2000, 8, {}),
    ('few', 40, 3, {}),
    ('small_batch', 600, 6, {'minibatch_size': 64, 'foreground_fraction': 0.5}),
    ('bg_low', 500, 4, {'background_threshold_low': 0.1}),
    ('thresholds', 500, 10, {'foreground_threshold': 0.6, 'background_threshold_high': 0.4}),
    ('dup_best', 200, 6, {}),                                       # one proposal is the best of several gts
    ('all_bg_disabled', 120, 2, {'minibatch_size': 16, 'foreground_fraction': 1.0}),   # max_bg reaches 0 (B.6)
]


def gen_rcnn_target(m, out):
    for ci, (name, P, G, over) in enumerate(RCNN_TARGET_CASES):
        rs = np.random.RandomState(300 + ci)
        W, H = 640, 480
        gt = np.concatenate([rand_boxes(rs, G, W, H, 30, 240), rs.randint(0, 20, size=(G, 1))], 1).astype(F)
        # proposals: jittered copies of the gts (foregrounds) + random boxes (backgrounds / ignored)
        nfg = P // 3
        jit = gt[rs.randint(0, G, size=nfg), :4] + rs.randint(-25, 26, size=(nfg, 4))
        props = np.concatenate([jit, rand_boxes(rs, P - nfg, W, H, 10, 300)], 0).astype(F)
        props = props[rs.permutation(P)]
        if name == 'dup_best':
            gt[1, :4] = gt[0, :4] + [1, 0, 1, 0]                    # near-identical gts share their best proposal
            gt[2, :4] = gt[0, :4]
        cfg = dict(foreground_fraction=0.25, minibatch_size=256, foreground_threshold=0.5,
                   background_threshold_high=0.5, background_threshold_low=0.0)
        cfg.update(over)
        seed = orng.image_seed(ci, 9, 0)
        tf.set_random_shuffle(shuffle_by_counter_rng(
            seed, {'disable_some_fgs': orng.STREAM_RCNN_FG, 'disable_some_bgs': orng.STREAM_RCNN_BG}))
        mod = m['rcnn_target'].RCNNTarget(20, ED(cfg), variances=[0.1, 0.2], seed=None)
        labels, targets = mod(props, gt)
        k = 'rcnn_target/%s/' % name
        out[k + 'proposals'], out[k + 'gt'] = props, gt
        out[k + 'seed'] = np.array([seed], np.uint32)
        out[k + 'cfg'] = np.array([cfg['foreground_fraction'], cfg['minibatch_size'], cfg['foreground_threshold'],
                                   cfg['background_threshold_high'], cfg['background_threshold_low']], np.float64)
        out[k + 'labels'], out[k + 'targets'] = labels, targets
        assert labels.dtype == F and targets.dtype == F
        print('rcnn_target %-16s P=%4d fg=%3d bg=%3d disabled_fg=%3d' % (name, P, (labels > 0).sum(), (labels == 0).sum(),
                                                                         (labels < -1).sum()))


RPN_PROPOSAL_CASES = [
    ('default', 20, 25, {}),
    ('small_topn', 20, 25, {'pre_nms_top_n': 600, 'post_nms_top_n': 50}),
    ('clip_after', 16, 16, {'clip_after_nms': True}),
    ('filter_outside', 20, 25, {'filter_outside_anchors': True}),
    ('no_nms', 12, 16, {'apply_nms': False, 'pre_nms_top_n': 300}),
    ('min_prob', 20, 25, {'min_prob_threshold': 0.4, 'nms_threshold': 0.5}),
]


def gen_rpn_proposal(m, out):
    for ci, (name, fh, fw, over) in enumerate(RPN_PROPOSAL_CASES):
        rs = np.random.RandomState(400 + ci)
        stride = 16
        H, W = fh * stride, fw * stride
        ref_i32, anchors = anchor_grid(64, [0.5, 1, 2], [0.25, 0.5, 1, 2], fh, fw, stride)
        N = anchors.shape[0]
        score = lattice_probs(rs, N)
        pred = (rs.randn(N, 4) * 0.3).astype(F)
        pred[rs.rand(N) < 0.05, 2] = -30.0                          # exp(-30)*w: collapses to (almost) zero width
        cfg = dict(pre_nms_top_n=12000, post_nms_top_n=2000, apply_nms=True, nms_threshold=0.7, min_size=0,
                   filter_outside_anchors=False, clip_after_nms=False, min_prob_threshold=0.0)
        cfg.update(over)
        prob = tf.nn.softmax(score)
        mod = m['rpn_proposal'].RPNProposal(ref_i32.shape[0], ED(cfg), debug=True)
        r = mod(prob, pred, anchors, np.array([H, W], np.int32))
        k = 'rpn_proposal/%s/' % name
        out[k + 'ref_i32'], out[k + 'score'], out[k + 'prob'], out[k + 'pred'] = ref_i32, score, prob, pred
        out[k + 'geom'] = np.array([fh, fw, stride, H, W], np.int32)
        out[k + 'cfg'] = np.array([cfg['pre_nms_top_n'], cfg['post_nms_top_n'], int(cfg['apply_nms']),
                                   cfg['nms_threshold'], int(cfg['filter_outside_anchors']),
                                   int(cfg['clip_after_nms']), cfg['min_prob_threshold']], np.float64)
        out[k + 'proposals'], out[k + 'scores'] = r['proposals'], r['scores']
        out[k + 'sorted_top_scores'] = r['sorted_top_scores']
        assert r['proposals'].dtype == F and r['scores'].dtype == F
        print('rpn_proposal %-16s N=%5d valid=%5d out=%4d' % (name, N, r['unsorted_scores'].shape[0],
                                                              r['proposals'].shape[0]))


RCNN_PROPOSAL_CASES = [
    ('default', 300, 20, {}),
    ('low_thr', 200, 6, {'min_prob_threshold': 0.05, 'class_nms_threshold': 0.3}),
    ('caps', 400, 4, {'min_prob_threshold': 0.0, 'class_max_detections': 15, 'total_max_detections': 40}),
    ('no_thr', 64, 3, {'min_prob_threshold': None, 'total_max_detections': 300}),
]


def gen_rcnn_proposal(m, out):
    for ci, (name, R, C, over) in enumerate(RCNN_PROPOSAL_CASES):
        rs = np.random.RandomState(500 + ci)
        W, H = 900, 600
        props = rand_boxes(rs, R, W, H, 20, 400)
        props = (props + rs.randint(-30, 31, size=props.shape)).astype(F)      # some stick out of the image
        pred = (rs.randn(R, 4 * C) * 0.5).astype(F)
        logits = (rs.randn(R, C + 1) * 3.0).astype(F)
        prob = tf.nn.softmax(logits)
        cfg = dict(class_max_detections=100, class_nms_threshold=0.5, total_max_detections=300, min_prob_threshold=0.5)
        cfg.update(over)
        mod = m['rcnn_proposal'].RCNNProposal(C, ED(cfg), variances=[0.1, 0.2])
        r = mod(props, pred, prob, np.array([H, W], np.int32))
        k = 'rcnn_proposal/%s/' % name
        out[k + 'proposals'], out[k + 'pred'], out[k + 'prob'] = props, pred, prob
        out[k + 'cfg'] = np.array([C, H, W, cfg['class_max_detections'], cfg['class_nms_threshold'],
                                   cfg['total_max_detections'], cfg['min_prob_threshold'] or 0.0], np.float64)
        out[k + 'objects'], out[k + 'labels'], out[k + 'probs'] = r['objects'], r['proposal_label'], \
            r['proposal_label_prob']
        print('rcnn_proposal %-12s R=%4d C=%2d -> %3d detections' % (name, R, C, r['objects'].shape[0]))


def gen_roi_pool(m, out):
    rs = np.random.RandomState(600)
    feat = rs.randn(1, 12, 16, 8).astype(F)
    H, W = 192, 256
    rois = rand_boxes(rs, 24, W, H, 8, 180)
    rois[3] = [-20, -10, 60, 70]                                    # partly outside: extrapolation value 0
    rois[5] = [200, 150, 300, 260]
    rois[7] = [40, 40, 40, 40]                                      # degenerate
    mod = m['roi_pool'].ROIPoolingLayer(ED(pooling_mode='crop', pooled_width=7, pooled_height=7, padding='VALID'),
                                        debug=True)
    r = mod(rois, feat, np.array([H, W], np.int32))
    out['roi_pool/feat'], out['roi_pool/rois'] = feat, rois
    out['roi_pool/im_shape'] = np.array([H, W], np.int32)
    out['roi_pool/bboxes'], out['roi_pool/crops'], out['roi_pool/pooled'] = r['bboxes'], r['crops'], r['roi_pool']
    assert r['roi_pool'].shape == (24, 7, 7, 8) and r['roi_pool'].dtype == F


def gen_losses(m, out):
    rs = np.random.RandomState(700)
    # RPN.loss (rpn.py:219-309)
    N = 600
    rpn = object.__new__(m['rpn'].RPN)
    rpn._l1_sigma = 3.0
    labels = rs.choice([-1, 0, 1], size=N, p=[.6, .3, .1]).astype(F)
    pd = {'rpn_cls_score': (rs.randn(N, 2) * 2).astype(F), 'rpn_cls_target': labels,
          'rpn_bbox_target': (rs.randn(N, 4) * 0.5).astype(F) * (labels == 1)[:, None],
          'rpn_bbox_pred': (rs.randn(N, 4) * 0.5).astype(F)}
    for k_, v in pd.items():
        out['rpn_loss/' + k_] = v
    r = rpn.loss(dict(pd))
    out['rpn_loss/rpn_cls_loss'], out['rpn_loss/rpn_reg_loss'] = r['rpn_cls_loss'], r['rpn_reg_loss']
    # RCNN.loss (rcnn.py:255-411)
    R, C = 256, 20
    rcnn = object.__new__(m['rcnn'].RCNN)
    rcnn._l1_sigma, rcnn._num_classes, rcnn._debug = 1.0, C, False
    cls_t = rs.choice(np.arange(-1, C + 1), size=R).astype(F)
    cls_t[:40] = 0
    pd = {'rcnn': {'cls_score': (rs.randn(R, C + 1) * 2).astype(F), 'bbox_offsets': rs.randn(R, 4 * C).astype(F)},
          'target': {'cls': cls_t, 'bbox_offsets': (rs.randn(R, 4)).astype(F) * (cls_t > 0)[:, None]}}
    out['rcnn_loss/cls_score'], out['rcnn_loss/bbox_offsets'] = pd['rcnn']['cls_score'], pd['rcnn']['bbox_offsets']
    out['rcnn_loss/cls_target'], out['rcnn_loss/bbox_target'] = pd['target']['cls'], pd['target']['bbox_offsets']
    r = rcnn.loss(pd)
    out['rcnn_loss/rcnn_cls_loss'], out['rcnn_loss/rcnn_reg_loss'] = r['rcnn_cls_loss'], r['rcnn_reg_loss']
    # SSD.loss (ssd.py:197-300) on the rows ssd.py:146-161 keeps (target >= 0) — and the all-background case
    for name, npos in (('mixed', None), ('no_positives', 0)):
        tf.reset_losses()
        tf.add_regularization_loss(0.75)                            # get_total_loss adds the regularisation collection
        ssd = object.__new__(m['ssd_ssd'].SSD)
        ssd._num_classes, ssd._loc_loss_weight, ssd._losses_collections = C, 1.5, ['ssd_losses']
        n = 400
        t = rs.randint(0, C + 1, size=n).astype(F)
        t[rs.rand(n) < 0.6] = 0
        if npos == 0:
            t[:] = 0
        pd = {'cls_pred': (rs.randn(n, C + 1) * 2).astype(F), 'loc_pred': rs.randn(n, 4).astype(F),
              'target': {'cls': t, 'bbox_offsets': rs.randn(n, 4).astype(F) * (t > 0)[:, None]}}
        r = ssd.loss(pd, return_all=True)
        k = 'ssd_loss/%s/' % name
        out[k + 'cls_pred'], out[k + 'loc_pred'] = pd['cls_pred'], pd['loc_pred']
        out[k + 'cls_target'], out[k + 'bbox_target'] = t, pd['target']['bbox_offsets']
        out[k + 'total_loss'], out[k + 'cls_loss'], out[k + 'bbox_loss'] = r['total_loss'], r['cls_loss'], r['bbox_loss']
        out[k + 'reg'] = np.float32(0.75)
        out[k + 'loc_weight'] = np.float32(1.5)
    tf.reset_losses()


SSD_FEATS = [(18, 18), (9, 9), (5, 5), (3, 3), (2, 2), (1, 1)]
SSD_TARGET_CASES = [
    ('default', 4, {}, None),
    ('one_gt', 1, {}, None),
    ('thresholds', 6, {'foreground_threshold': 0.4, 'background_threshold_high': 0.3, 'hard_negative_ratio': 2.0}, None),
    ('crowded', 12, {'hard_negative_ratio': 3.0, 'background_threshold_high': 0.05}, 'big'),   # top_k reaches the -1 rows
    ('dup_best', 5, {}, 'dup'),
    ('frac_ratio', 3, {'hard_negative_ratio': 2.5}, None),
]


def gen_ssd(m, out):
    C = 5
    anchors = ossd.all_anchors(SSD_FEATS, (150, 150))
    N = anchors.shape[0]
    out['ssd/anchors'] = anchors
    for ci, (name, G, over, special) in enumerate(SSD_TARGET_CASES):
        rs = np.random.RandomState(800 + ci)
        lo, hi = (60, 140) if special == 'big' else (12, 90)
        gt = np.concatenate([rand_boxes(rs, G, 150, 150, lo, hi), rs.randint(0, C, size=(G, 1))], 1).astype(F)
        if special == 'dup':
            gt[1, :4] = gt[0, :4]
            gt[2, :4] = gt[0, :4] + [0, 1, 0, 1]
        probs = tf.nn.softmax((rs.randn(N, C + 1) * 2).astype(F))
        cfg = dict(hard_negative_ratio=3.0, foreground_threshold=0.5, background_threshold_high=0.2)
        cfg.update(over)
        mod = m['ssd_target'].SSDTarget(C, ED(cfg), [0.1, 0.2])
        labels, targets = mod(probs, anchors, gt)
        k = 'ssd_target/%s/' % name
        out[k + 'gt'], out[k + 'probs'] = gt, probs
        out[k + 'cfg'] = np.array([cfg['hard_negative_ratio'], cfg['foreground_threshold'],
                                   cfg['background_threshold_high']], np.float64)
        out[k + 'labels'], out[k + 'targets'] = labels, targets
        assert labels.dtype == F and targets.dtype == F
        print('ssd_target %-12s N=%d fg=%3d bg=%4d' % (name, N, (labels > 0).sum(), (labels == 0).sum()))
    for ci, (name, over) in enumerate([('default', {}), ('low_thr', {'min_prob_threshold': 0.1, 'class_nms_threshold': 0.3}),
                                       ('caps', {'min_prob_threshold': 0.02, 'class_max_detections': 10,
                                                 'total_max_detections': 25})]):
        rs = np.random.RandomState(900 + ci)
        prob = tf.nn.softmax((rs.randn(N, C + 1) * 2.5).astype(F))
        loc = (rs.randn(N, 4) * 0.8).astype(F)
        cfg = dict(class_nms_threshold=0.45, class_max_detections=100, total_max_detections=100,
                   min_prob_threshold=0.5, filter_outside_anchors=False)
        cfg.update(over)
        mod = m['ssd_proposal'].SSDProposal(C, ED(cfg), [0.1, 0.2])
        r = mod(prob, loc, anchors, np.array([150., 150.], F))      # ssd.py:186 passes a float32 shape
        k = 'ssd_proposal/%s/' % name
        out[k + 'prob'], out[k + 'loc'] = prob, loc
        out[k + 'cfg'] = np.array([C, cfg['class_nms_threshold'], cfg['class_max_detections'],
                                   cfg['total_max_detections'], cfg['min_prob_threshold']], np.float64)
        out[k + 'objects'], out[k + 'labels'], out[k + 'probs'] = r['objects'], r['labels'], r['probs']
        out[k + 'anchors_out'], out[k + 'raw_proposals'] = r['anchors'], r['raw_proposals']
        print('ssd_proposal %-10s -> %3d detections' % (name, r['objects'].shape[0]))


# ------------------------------------------------------------------------------------------ head layouts ----
INIT = {'type': 'random_normal_initializer', 'mean': 0.0, 'stddev': 0.01}


def gen_heads(m, out):
    """RPN._build, RCNN._build and SSD._build executed as they are."""
    # ---- RPN (rpn.py:96-217): 3x3 conv + relu6, 1x1 cls / bbox convs, reshape to (N,2) / (N,4), softmax, proposals, targets
    rs = np.random.RandomState(1000)
    fh, fw, cin, stride = 6, 8, 16, 16
    H, W = fh * stride, fw * stride
    ref_i32, anchors = anchor_grid(64, [0.5, 1, 2], [0.25, 0.5, 1, 2], fh, fw, stride)
    feat = rs.randn(1, fh, fw, cin).astype(F)
    gt = np.concatenate([rand_boxes(rs, 3, W, H, 16, 64), rs.randint(0, 5, size=(3, 1))], 1).astype(F)
    prop_cfg = dict(pre_nms_top_n=12000, post_nms_top_n=2000, apply_nms=True, nms_threshold=0.7, min_size=0,
                    filter_outside_anchors=False, clip_after_nms=False, min_prob_threshold=0.0)
    tgt_cfg = dict(allowed_border=0, clobber_positives=False, foreground_threshold=0.7,
                   background_threshold_high=0.3, foreground_fraction=0.5, minibatch_size=256)
    cfg = ED(num_channels=24, kernel_shape=[3, 3], rpn_initializer=INIT, cls_initializer=INIT, bbox_initializer=INIT,
             l2_regularization_scale=0.0005, l1_sigma=3.0, activation_function='relu6', proposals=prop_cfg, target=tgt_cfg)
    seed = orng.image_seed(7, 0, 0)
    inside = np.where((anchors[:, 0] >= 0) & (anchors[:, 1] >= 0) & (anchors[:, 2] < W) & (anchors[:, 3] < H))[0]
    tf.set_random_shuffle(shuffle_by_counter_rng(
        seed, {'subsample_positive': orng.STREAM_RPN_FG, 'subsample_negative': orng.STREAM_RPN_BG}, inside))
    rpn = m['rpn'].RPN(ref_i32.shape[0], cfg, debug=False, seed=None)
    r = rpn(feat, np.array([H, W], np.int32), anchors, gt_boxes=gt, is_training=True)
    k = 'heads/rpn/'
    out[k + 'feat'], out[k + 'gt'], out[k + 'ref_i32'] = feat, gt, ref_i32
    out[k + 'geom'] = np.array([fh, fw, stride, H, W, cfg.num_channels], np.int32)
    out[k + 'seed'] = np.array([seed], np.uint32)
    for key in ('rpn_cls_score', 'rpn_cls_prob', 'rpn_bbox_pred', 'proposals', 'scores', 'rpn_cls_target', 'rpn_bbox_target'):
        out[k + key] = r[key]
    assert r['rpn_cls_score'].shape == (fh * fw * 12, 2) and r['rpn_bbox_pred'].shape == (fh * fw * 12, 4)
    print('heads/rpn: N=%d proposals=%d fg=%d' % (anchors.shape[0], r['proposals'].shape[0], (r['rpn_cls_target'] == 1).sum()))

    # ---- RCNN (rcnn.py:112-250), the two configurations of the sample configs: spatial mean and no FC stack (ResNet),
    # flattened 7x7xC features through two FC layers (VGG)
    class NoTail(object):          # truncated_base_network.py:56-63 returns its input unless the architecture is resnet_v1_101
        @staticmethod
        def _build_tail(inputs, is_training=False):
            return inputs
    C = 5
    for name, use_mean, sizes, cf in (('mean', True, [], 12), ('flatten_fc', False, [32, 24], 8)):
        rs = np.random.RandomState(1100 + len(name))
        fh, fw = 10, 12
        H, W = fh * 16, fw * 16
        feat = rs.randn(1, fh, fw, cf).astype(F)
        G, P = 4, 90
        gt = np.concatenate([rand_boxes(rs, G, W, H, 24, 96), rs.randint(0, C, size=(G, 1))], 1).astype(F)
        jit = gt[rs.randint(0, G, size=P // 3), :4] + rs.randint(-8, 9, size=(P // 3, 4))
        props = np.concatenate([jit, rand_boxes(rs, P - P // 3, W, H, 12, 120)], 0).astype(F)[rs.permutation(P)]
        t_cfg = dict(foreground_fraction=0.25, minibatch_size=32, foreground_threshold=0.5,
                     background_threshold_high=0.5, background_threshold_low=0.0)
        p_cfg = dict(class_max_detections=100, class_nms_threshold=0.5, total_max_detections=300, min_prob_threshold=0.5)
        cfg = ED(layer_sizes=sizes, activation_function='relu', dropout_keep_prob=1.0, use_mean=use_mean,
                 target_normalization_variances=[0.1, 0.2], rcnn_initializer=INIT, cls_initializer=INIT,
                 bbox_initializer=INIT, l2_regularization_scale=0.0005, l1_sigma=1.0,
                 roi=dict(pooling_mode='crop', pooled_width=7, pooled_height=7, padding='VALID'), target=t_cfg,
                 proposals=p_cfg)
        seed = orng.image_seed(8, 1, 0)
        tf.set_random_shuffle(shuffle_by_counter_rng(
            seed, {'disable_some_fgs': orng.STREAM_RCNN_FG, 'disable_some_bgs': orng.STREAM_RCNN_BG}))
        rcnn = m['rcnn'].RCNN(C, cfg, debug=False, seed=None)
        r = rcnn(feat, props, np.array([H, W], np.int32), NoTail(), gt_boxes=gt, is_training=True)
        k = 'heads/rcnn_%s/' % name
        out[k + 'feat'], out[k + 'gt'], out[k + 'proposals'] = feat, gt, props
        out[k + 'geom'] = np.array([H, W, C, int(use_mean)] + sizes, np.int32)
        out[k + 'seed'] = np.array([seed], np.uint32)
        out[k + 'target_cls'], out[k + 'target_bbox'] = r['target']['cls'], r['target']['bbox_offsets']
        out[k + 'cls_score'], out[k + 'cls_prob'] = r['rcnn']['cls_score'], r['rcnn']['cls_prob']
        out[k + 'bbox_offsets'] = r['rcnn']['bbox_offsets']
        n = r['target']['cls'].shape[0]
        assert r['rcnn']['cls_score'].shape == (n, C + 1) and r['rcnn']['bbox_offsets'].shape == (n, 4 * C)
        print('heads/rcnn_%-10s rois=%d fg=%d' % (name, n, (r['target']['cls'] > 0).sum()))

    # ---- SSD (ssd/ssd.py:37-195) over given feature maps: six multibox head pairs, reshape + concat, anchors, targets with
    # the hard-negative filter (training) / proposals (inference)
    import collections
    C = 4
    shapes = [(5, 5, 8), (3, 3, 12), (2, 2, 8), (2, 2, 6), (1, 1, 6), (1, 1, 4)]
    app = [4, 6, 6, 6, 4, 4]
    names = ['vgg_16/conv4/conv4_3', 'vgg_16/fc7', 'conv6_2', 'conv7_2', 'conv8_2', 'conv9_2']
    rs = np.random.RandomState(1200)
    fmaps = collections.OrderedDict((n, tf.Tensor((rs.randn(1, h, w, c) * 1.5).astype(F))) for n, (h, w, c) in zip(names, shapes))
    ssd_mod = m['ssd_ssd']
    ssd_mod.SSDFeatureExtractor = lambda cfg, parent_name=None: (lambda image, is_training=False: fmaps)
    gt = np.concatenate([rand_boxes(rs, 3, 96, 96, 20, 70), rs.randint(0, C, size=(3, 1))], 1).astype(F)
    for mode in ('train', 'predict'):
        tf.reset_losses()
        cfg = ED(model=dict(network=dict(num_classes=C),
                            anchors=dict(max_scale=0.88, min_scale=0.1, ratios=[1, 0.5, 2, 0.333, 3], anchors_per_point=app),
                            loss=dict(localization_loss_weight=1.0), base_network={}, variances=[0.1, 0.2],
                            target=dict(hard_negative_ratio=3.0, foreground_threshold=0.5, background_threshold_high=0.2),
                            proposals=dict(class_nms_threshold=0.45, class_max_detections=100, total_max_detections=100,
                                           min_prob_threshold=0.3, filter_outside_anchors=False)),
                 train=dict(debug=False, seed=None),
                 dataset=dict(image_preprocessing=dict(fixed_height=96, fixed_width=96)))
        ssd = ssd_mod.SSD(cfg)
        image = tf.Tensor(np.zeros((96, 96, 3), F))
        if mode == 'train':
            r = ssd(image, gt_boxes=gt, is_training=True)
        else:
            r = ssd(image, is_training=False)
        k = 'heads/ssd_%s/' % mode
        out[k + 'cls_pred'], out[k + 'loc_pred'] = r['cls_pred'], r['loc_pred']
        if mode == 'train':
            out[k + 'target_cls'], out[k + 'target_bbox'] = r['target']['cls'], r['target']['bbox_offsets']
            out[k + 'target_anchors'] = r['target']['anchors']
        else:
            cp = r['classification_prediction']
            out[k + 'objects'], out[k + 'labels'], out[k + 'probs'] = cp['objects'], cp['labels'], cp['probs']
        print('heads/ssd_%-8s rows=%d' % (mode, r['cls_pred'].shape[0]))
    for n, fm in fmaps.items():
        out['heads/ssd/fmap/' + n.replace('/', '.')] = np.asarray(fm)
    out['heads/ssd/gt'] = gt
    out['heads/ssd/geom'] = np.array([96, 96, C] + app, np.int32)
    tf.reset_losses()


TOPLEVEL_CFG = {            # overrides of the reference's own models/fasterrcnn/base_config.yml (small shapes; replayed by the tests)
    'train': {'seed': 7, 'debug': False},
    'model': {'network': {'num_classes': 5},
              'anchors': {'base_size': 64},
              'base_network': {'architecture': 'resnet_v1_50'},
              'rpn': {'num_channels': 24},
              'rcnn': {'target': {'minibatch_size': 32}}}}


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def load_toplevel():
    """The reference's top-level modules (fasterrcnn.py, base_network.py, truncated_base_network.py, utils/anchors.py) by path,
    on the shim + the slim stand-in."""
    import slim_standin
    slim_standin.install()
    for name in ('luminoth.utils.checkpoint_downloader', 'luminoth.models.base.truncated_vgg'):
        sys.modules[name] = tf._Inert(name)
    mods = {}
    for rel in ('utils/anchors.py', 'models/base/base_network.py', 'models/base/truncated_base_network.py',
                'models/fasterrcnn/fasterrcnn.py'):
        name = 'luminoth.' + rel[:-3].replace('/', '.')
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        if rel.endswith('truncated_base_network.py'):
            sys.modules['luminoth.models.base'].BaseNetwork = mods['base_network'].BaseNetwork
        if rel.endswith('fasterrcnn.py'):
            sys.modules['luminoth.models.base'].TruncatedBaseNetwork = mods['truncated_base_network'].TruncatedBaseNetwork
        spec.loader.exec_module(mod)
        mods[rel[:-3].split('/')[-1]] = mod
    return mods, slim_standin


def toplevel_config(over=None):
    import copy
    import yaml
    cfg = yaml.safe_load(open(os.path.join(REF, 'models/fasterrcnn/base_config.yml')))
    _merge(cfg, copy.deepcopy(TOPLEVEL_CFG))
    if over:
        _merge(cfg, over)
    return ED(cfg)


def gen_toplevel(m, out):
    """the reference's TOP-LEVEL composition executed as it is — FasterRCNN.__init__ / _build / loss /
    get_trainable_vars (fasterrcnn.py:22-259,337-358) and BaseNetwork._get_base_network_vars / get_trainable_vars /
    get_base_network_checkpoint_vars + TruncatedBaseNetwork.get_trainable_vars (base_network.py:196-259,
    truncated_base_network.py:96-144) — over a slim stand-in that creates the real variables (names, order, shapes,
    collections, weight decay: tests/golden/slim_standin.py) and returns the fixture's feature map.  Pinned: the int32 anchor
    grid the composition itself forms, stop_gradient'ed proposals into the RCNN, loss weights and what enters total_loss
    (the loss dict), WHICH variables train per `fine_tune_from`, which are regularised, and the checkpoint name map."""
    import json
    tl, slim_standin = load_toplevel()
    k = 'toplevel/'
    rs = np.random.RandomState(2000)
    fh, fw, stride = 6, 8, 16
    H, W = fh * stride, fw * stride
    feat = (rs.randn(1, fh, fw, 1024) * 0.5).astype(F)
    gt = np.concatenate([rand_boxes(rs, 3, W, H, 16, 64), rs.randint(0, 5, size=(3, 1))], 1).astype(F)
    image = tf.Tensor((rs.rand(H, W, 3) * 255).astype(F))
    cfg = toplevel_config()
    seed = orng.image_seed(cfg.train.seed, 0, 0)

    def build(cfg_):
        tf.reset_variables()
        tf.reset_losses()
        tf.track_regularizers(True)
        slim_standin.set_feature_map(feat)
        model = tl['fasterrcnn'].FasterRCNN(cfg_)
        ref = np.trunc(model._anchor_reference).astype(np.int32)
        anchors = obx.generate_anchors(model._anchor_reference, fh, fw, stride).astype(np.int32)
        inside = np.where((anchors[:, 0] >= 0) & (anchors[:, 1] >= 0) & (anchors[:, 2] < W) & (anchors[:, 3] < H))[0]
        rpn_hook = shuffle_by_counter_rng(seed, {'subsample_positive': orng.STREAM_RPN_FG,
                                                 'subsample_negative': orng.STREAM_RPN_BG}, inside)
        rcnn_hook = shuffle_by_counter_rng(seed, {'disable_some_fgs': orng.STREAM_RCNN_FG,
                                                  'disable_some_bgs': orng.STREAM_RCNN_BG})
        tf.set_random_shuffle(lambda v, s, caller: (rpn_hook if caller.startswith('subsample') else rcnn_hook)(v, s, caller))
        pred = model(image, gt_boxes=gt, is_training=True)
        return model, pred, ref, anchors

    model, pred, ref_i32, anchors = build(cfg)
    losses = model.loss(pred, return_all=True)
    rp, cp = pred['rpn_prediction'], pred['classification_prediction']
    out[k + 'cfg'] = np.array(json.dumps(TOPLEVEL_CFG, sort_keys=True))
    out[k + 'feat'], out[k + 'gt'], out[k + 'ref_i32'] = feat, gt, ref_i32
    out[k + 'geom'] = np.array([fh, fw, stride, H, W], np.int32)
    out[k + 'seed'] = np.array([seed], np.uint32)
    for key in ('rpn_cls_score', 'rpn_bbox_pred', 'proposals', 'rpn_cls_target', 'rpn_bbox_target'):
        out[k + key] = rp[key]
    out[k + 'rcnn_target_cls'], out[k + 'rcnn_target_bbox'] = cp['target']['cls'], cp['target']['bbox_offsets']
    out[k + 'rcnn_cls_score'], out[k + 'rcnn_bbox_offsets'] = cp['rcnn']['cls_score'], cp['rcnn']['bbox_offsets']
    for name, v in losses.items():
        out[k + 'loss/' + name] = np.float32(v)
    assert set(losses) == {'total_loss', 'no_reg_loss', 'regularization_loss', 'rpn_cls_loss', 'rpn_reg_loss',
                           'rcnn_cls_loss', 'rcnn_reg_loss'}
    strip = lambda vs: np.array([v.name[:-2] for v in vs])          # noqa: E731  ('<name>:0' -> '<name>')
    out[k + 'names/all_variables'] = np.array([v.op.name for v in tf._variables])
    out[k + 'names/all_shapes'] = np.array([list(v.shape) + [0] * (4 - len(v.shape)) for v in tf._variables], np.int32)
    out[k + 'names/all_trainable'] = np.array([tf.GraphKeys.TRAINABLE_VARIABLES in v.collections for v in tf._variables])
    out[k + 'names/regularized'] = np.array(tf.regularized_names())
    ck = model.get_base_network_checkpoint_vars()
    out[k + 'names/checkpoint_keys'] = np.array(list(ck))
    out[k + 'names/checkpoint_vars'] = np.array([v.op.name for v in ck.values()])
    out[k + 'names/trainable/block2'] = strip(model.get_trainable_vars())
    n_all, n_reg, n_tr = len(tf._variables), len(tf.regularized_names()), len(model.get_trainable_vars())
    # the other fine_tune_from settings of the config surface, and ResNet-101 (whose block4 tail trains: use_tail, not
    # freeze_tail — truncated_base_network.py:127-142): names only
    for tag, over in (('none', {'model': {'base_network': {'fine_tune_from': None}}}),
                      ('block3_unit_2', {'model': {'base_network': {'fine_tune_from': 'block3/unit_2'}}}),
                      ('not_trainable', {'model': {'base_network': {'trainable': False}}}),
                      ('resnet_v1_101', {'model': {'base_network': {'architecture': 'resnet_v1_101'}}})):
        m2, _, _, _ = build(toplevel_config(over))
        out[k + 'names/trainable/' + tag] = strip(m2.get_trainable_vars())
    tf.track_regularizers(False)
    tf.reset_variables()
    tf.reset_losses()
    print('toplevel: %d variables, %d regularised, %d trainable (fine_tune_from block2); total_loss %.6f = no_reg %.6f + reg %.6f'
          % (n_all, n_reg, n_tr, float(losses['total_loss']), float(losses['no_reg_loss']), float(losses['regularization_loss'])))


def main():
    m = load_reference()
    out = {}
    gen_box_utils(m, out)
    gen_rpn_target(m, out)
    gen_rcnn_target(m, out)
    gen_rpn_proposal(m, out)
    gen_rcnn_proposal(m, out)
    gen_roi_pool(m, out)
    gen_losses(m, out)
    gen_ssd(m, out)
    gen_heads(m, out)
    gen_toplevel(m, out)
    for k, v in out.items():
        v = np.asarray(v)
        assert v.dtype != np.float64 or k.endswith('/cfg'), (k, v.dtype)     # nothing silently promoted
        out[k] = v
    path = os.path.join(HERE, 'ref_tf_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote %s: %d arrays, %.1f KB' % (path, len(out), os.path.getsize(path) / 1024.))


if __name__ == '__main__':
    main()
