"""GPU parity tests: every HIP kernel (through the C ABI) vs the CPU oracle on
identical seeded inputs.  Bit-exact for index / label / keep-mask work,
1e-4..1e-5 for fp32 (tolerances written at each assert).  Run with `-m gpu`.
"""
import numpy as np
import pytest
import torch

from oracle import boxes as obx
from oracle import frcnn as of
from oracle import rng as orng
from oracle import tfops
from oracle import torch_ops as ot

pytestmark = pytest.mark.gpu
F = np.float32


def dev():
    return torch.device('cuda:0')


def T(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev()).contiguous()


def rand_boxes(rs, n, lim=1024, smin=16, smax=400):
    wh = rs.randint(smin, smax, size=(n, 2))
    xy = np.stack([rs.randint(0, lim - wh[:, 0]), rs.randint(0, lim - wh[:, 1])], 1)
    return np.concatenate([xy, xy + wh - 1], 1).astype(F)


@pytest.fixture(scope='module')
def K():
    from luminoth_amd import kernels
    return kernels


# ------------------------------------------------------------------ sort ----
@pytest.mark.parametrize('n', [2, 64, 4096, 8192, 16384, 32768, 65536, 262144])   # 1, 2, 3, 4 and 4 + 2 fused global passes
def test_sort_u64(K, n):
    rs = np.random.RandomState(n)
    keys = rs.randint(0, 2 ** 62, size=(3, n), dtype=np.int64)
    keys[1, : n // 2] = keys[1, n // 2:][: n // 2]       # duplicates
    out = K.sort_u64(T(keys)).cpu().numpy()
    np.testing.assert_array_equal(out.view(np.uint64), np.sort(keys.view(np.uint64), axis=1))


# ------------------------------------------------------------------- NMS ----
def test_nms_matches_oracle_bit_exact(K):
    rs = np.random.RandomState(0)
    B, Kn = 3, 3000
    boxes = np.zeros((B, Kn, 4), F)
    counts = np.array([3000, 1777, 64], np.int32)
    for b in range(B):
        base = rand_boxes(rs, 60, 600, 40, 200)
        jit = base[rs.randint(0, 60, size=Kn)] + rs.randint(-12, 13, size=(Kn, 4))
        boxes[b] = jit.astype(F)
    boxes[0, 5] = [10, 10, 10, 50]        # zero-area box
    boxes[0, 7] = [50, 60, 20, 30]        # inverted corners (TF normalises them)
    for thr, max_out in ((0.7, 300), (0.3, 2000), (0.0, 50)):
        keep, kc = K.nms(T(boxes), T(counts), thr, max_out)
        keep, kc = keep.cpu().numpy(), kc.cpu().numpy()
        for b in range(B):
            scores = np.arange(counts[b], 0, -1).astype(F)      # already sorted, descending
            ref = tfops.non_max_suppression(boxes[b, :counts[b]][:, [1, 0, 3, 2]], scores, max_out, thr)
            assert kc[b] == ref.shape[0]
            np.testing.assert_array_equal(keep[b, :kc[b]], ref)
            assert (keep[b, kc[b]:] == -1).all()


def test_nms_full_size_super_chunks(K):
    """12 000 candidates (12 super-chunks of 1024 in k_nms_reduce), clustered boxes so that suppression reaches far
    down the list; max_out both inside (2000) and beyond (5000) the LDS mirror of the kept list; ragged counts."""
    rs = np.random.RandomState(4)
    B, Kn = 2, 12000
    boxes = np.zeros((B, Kn, 4), F)
    counts = np.array([12000, 9001], np.int32)
    for b in range(B):
        base = rand_boxes(rs, 400 + 2000 * b, 1000, 30, 300)
        jit = base[rs.randint(0, base.shape[0], size=Kn)] + rs.randint(-10, 11, size=(Kn, 4))
        boxes[b] = jit.astype(F)
    for thr, max_out in ((0.7, 2000), (0.5, 5000), (0.9, 300)):
        keep, kc = K.nms(T(boxes), T(counts), thr, max_out)
        keep, kc = keep.cpu().numpy(), kc.cpu().numpy()
        for b in range(B):
            scores = np.arange(counts[b], 0, -1).astype(F)
            ref = tfops.non_max_suppression(boxes[b, :counts[b]][:, [1, 0, 3, 2]], scores, max_out, thr)
            assert kc[b] == ref.shape[0], (thr, max_out, b, kc[b], ref.shape[0])
            np.testing.assert_array_equal(keep[b, :kc[b]], ref)
            assert (keep[b, kc[b]:] == -1).all()


@pytest.mark.parametrize('stage_mult', [0, 2, 1])
def test_nms_scan_variants_agree(K, stage_mult):
    """The pipelined scan (k_nms_reduce_p: row prefetch, look-ahead gather, wave 0 folding the next word itself) and its
    two-stage split (stage B continues from the saved state: the first gather of the pipelined
    kernel then covers the whole keep list) give the oracle's keep list bit for bit: even and odd row lengths (the
    16-byte and the 8-byte loads), ragged counts, a count that ends inside the first chunk of a super-chunk, max_out
    reached inside a chunk, keep lists inside and beyond the LDS mirror."""
    rs = np.random.RandomState(11)
    old = K.get_option('nms_stage_mult')
    K.set_option('nms_stage_mult', stage_mult)
    try:
        for Kn, counts, cases in ((12000, [12000, 1025, 7000], ((0.7, 2000), (0.5, 2500))),
                                  (4160, [4160, 4097, 1], ((0.7, 700), (0.3, 4160)))):       # W = 188 / 65
            B = len(counts)
            boxes = np.zeros((B, Kn, 4), F)
            for b in range(B):
                base = rand_boxes(rs, 300 + 900 * b, 1000, 30, 300)
                boxes[b] = (base[rs.randint(0, base.shape[0], size=Kn)] + rs.randint(-10, 11, size=(Kn, 4))).astype(F)
            cnt = np.array(counts, np.int32)
            for thr, max_out in cases:
                keep, kc = K.nms(T(boxes), T(cnt), thr, max_out)
                keep, kc = keep.cpu().numpy(), kc.cpu().numpy()
                for b in range(B):
                    scores = np.arange(cnt[b], 0, -1).astype(F)
                    ref = tfops.non_max_suppression(boxes[b, :cnt[b]][:, [1, 0, 3, 2]], scores, max_out, thr)
                    assert kc[b] == ref.shape[0], (Kn, thr, max_out, b, kc[b], ref.shape[0])
                    np.testing.assert_array_equal(keep[b, :kc[b]], ref)
                    assert (keep[b, kc[b]:] == -1).all()
    finally:
        K.set_option('nms_stage_mult', old)


# ---------------------------------------------------------- RPN proposal ----
def _proposal_case(K, feat, A, stride, im, pre, post, thr, zero_wh, seed, **kw):
    rs = np.random.RandomState(seed)
    B = 2
    ref = obx.generate_anchors_reference(256, np.array([.5, 1, 2]), np.array([.25, .5, 1, 2]))[:A]
    ref_i32 = np.trunc(ref).astype(np.int32)
    N = feat * feat * A
    score = (rs.randn(B, N, 2) * 2).astype(F)
    pred = (rs.randn(B, N, 4) * 0.2).astype(F)
    if zero_wh:
        pred[..., 2:] = 0           # exp(0) == 1 exactly: the whole chain is IEEE-exact
    prob, props, scores, cnt = K.rpn_proposal(T(score), T(pred), T(ref_i32), feat, feat, stride, im,
                                              pre_nms_top_n=pre, post_nms_top_n=post, nms_threshold=thr, **kw)
    prob, props, scores, cnt = [t.cpu().numpy() for t in (prob, props, scores, cnt)]
    np.testing.assert_allclose(prob, tfops.softmax(score), rtol=2e-6, atol=1e-7)   # fp32 softmax
    anchors = obx.generate_anchors(ref, feat, feat, stride)
    for b in range(B):
        # the oracle consumes the kernel's own probabilities so ordering decisions are comparable
        r = of.rpn_proposal(prob[b], pred[b], anchors, im, pre_nms_top_n=pre, post_nms_top_n=post,
                            nms_threshold=thr, **kw)
        assert cnt[b] == r['proposals'].shape[0]
        if zero_wh:
            np.testing.assert_array_equal(props[b, :cnt[b]], r['proposals'])       # bit-exact
        else:
            np.testing.assert_allclose(props[b, :cnt[b]], r['proposals'], rtol=1e-6, atol=1e-4)
        np.testing.assert_array_equal(scores[b, :cnt[b]], r['scores'])
        assert (props[b, cnt[b]:] == 0).all()


def test_rpn_proposal_small_exact(K):
    _proposal_case(K, 16, 12, 16, (256, 256), 600, 100, 0.7, True, 1)
    _proposal_case(K, 16, 12, 16, (256, 256), 6000, 3072, 0.7, True, 2)          # k > n_valid
    _proposal_case(K, 16, 12, 16, (256, 256), 600, 100, 0.7, True, 3, clip_after_nms=True)
    _proposal_case(K, 16, 12, 16, (256, 256), 600, 100, 0.7, True, 4, filter_outside_anchors=True)
    _proposal_case(K, 16, 12, 16, (256, 256), 600, 100, 0.7, True, 5, apply_nms=False)
    _proposal_case(K, 16, 12, 16, (256, 256), 600, 100, 0.7, True, 6, min_prob_threshold=0.4)


def test_rpn_proposal_full_size(K):
    # BASELINE config 2 geometry: 1024x1024 image, 64x64x12 = 49152 anchors, 12000 -> 2000
    _proposal_case(K, 64, 12, 16, (1024, 1024), 12000, 2000, 0.7, True, 7)
    _proposal_case(K, 64, 12, 16, (1024, 1024), 12000, 2000, 0.7, False, 8)


def test_rpn_proposal_reference_vectors(K):
    # luminoth/models/fasterrcnn/rpn_proposal_test.py:61-170 through the kernel: arbitrary anchors
    # are expressed as a 1x1 grid whose "reference" IS the anchor list (stride 0).
    gt = np.array([[10, 10, 26, 36], [10, 10, 20, 22], [10, 11, 20, 21], [19, 30, 33, 38]], F)
    anchors = np.array([[11, 13, 34, 31], [10, 10, 20, 22], [11, 13, 34, 28], [21, 29, 34, 37]], np.int32)
    prob = np.array([[.8, .2], [.1, .9], [.4, .6], [.2, .8]], F)
    score = np.log(prob)[None]
    pred = obx.encode(anchors, gt)[None]
    for thr, n in ((0.0, 2), (0.3, 3), (0.6, 3), (0.8, 3), (1.0, 4)):
        _, props, scores, cnt = K.rpn_proposal(T(score), T(pred), T(anchors), 1, 1, 0, (40, 40),
                                               pre_nms_top_n=4, post_nms_top_n=4, nms_threshold=thr)
        assert int(cnt[0]) == n
        np.testing.assert_allclose(scores[0, :n].cpu().numpy(), [.9, .8, .2, .1][:n] if n < 4 else
                                   [.9, .8, .6, .2], rtol=1e-5)


# ------------------------------------------------------------ RPN target ----
def test_rpn_target_full_size_bit_exact(K):
    rs = np.random.RandomState(11)
    B, Gmax, feat, A, stride = 2, 16, 64, 12, 16
    ref = obx.generate_anchors_reference(256, np.array([.5, 1, 2]), np.array([.25, .5, 1, 2]))
    ref_i32 = np.trunc(ref).astype(np.int32)
    gt = np.zeros((B, Gmax, 5), F)
    counts = np.array([8, 3], np.int32)
    for b in range(B):
        gt[b, :counts[b], :4] = rand_boxes(rs, counts[b], 1024, 32, 512)
        gt[b, :counts[b], 4] = rs.randint(0, 80, size=counts[b])
    seeds = np.array([orng.image_seed(0, 5, b) for b in range(B)], np.uint32)
    labels, targets, mo, pre = K.rpn_target(T(ref_i32), feat, feat, stride, T(gt), T(counts),
                                            T(seeds.view(np.int32)), (1024, 1024), want_pre=True)
    labels, targets, mo, pre = [t.cpu().numpy() for t in (labels, targets, mo, pre)]
    anchors = obx.generate_anchors(ref, feat, feat, stride)
    for b in range(B):
        ol, ot_, om, opre, _ = of.rpn_target(anchors, gt[b, :counts[b]], (1024, 1024), seed=int(seeds[b]),
                                             return_pre_subsample=True)
        np.testing.assert_array_equal(pre[b], opre)          # labels before subsampling: bit-exact
        np.testing.assert_array_equal(mo[b], om)             # IoU: bit-exact fp32
        np.testing.assert_array_equal(labels[b], ol)         # shared counter RNG: bit-exact
        np.testing.assert_allclose(targets[b], ot_, rtol=1e-5, atol=1e-6)   # logf: 1e-5
        assert (labels[b] == 1).sum() <= 128 and (labels[b] >= 0).sum() == 256


def test_rpn_target_reference_vectors(K):
    # luminoth/models/fasterrcnn/rpn_target_test.py:94-152,154-190 via the 1x1-grid trick
    gt = np.array([[[200, 0, 400, 400, 0]]], F)
    cnt = np.array([1], np.int32)
    seeds = np.array([7], np.int32)
    anchors = np.array([[200, 100, 400, 400], [300, 300, 400, 400], [200, 380, 300, 500],
                        [500, 500, 600, 650], [200, 100, 400, 400]], np.int32)
    labels, targets, _, _ = K.rpn_target(T(anchors), 1, 1, 0, T(gt), T(cnt), T(seeds), (600, 600),
                                         minibatch_size=5)
    np.testing.assert_array_equal(labels[0].cpu().numpy(), [1, 0, 0, -1, 1])
    t = targets[0].cpu().numpy()
    assert t[0][0] == 0 and t[0][2] == 0 and t[0][1] != 0 and t[0][3] != 0
    np.testing.assert_array_equal(t[1:4], np.zeros((3, 4)))
    anchors = np.array([[300, 300, 400, 400], [200, 380, 300, 500]], np.int32)
    labels, _, _, _ = K.rpn_target(T(anchors), 1, 1, 0, T(gt), T(cnt), T(seeds), (600, 600), minibatch_size=2)
    np.testing.assert_array_equal(labels[0].cpu().numpy(), [1, 0])
    labels, _, _, _ = K.rpn_target(T(anchors), 1, 1, 0, T(gt), T(cnt), T(seeds), (600, 600), minibatch_size=2,
                                   clobber_positives=True)
    np.testing.assert_array_equal(labels[0].cpu().numpy(), [0, 0])


# ----------------------------------------------------------- RCNN target ----
def test_rcnn_target_bit_exact(K):
    rs = np.random.RandomState(21)
    B, Pn, Gmax = 3, 2000, 16
    props = np.zeros((B, Pn, 4), F)
    pc = np.array([2000, 700, 40], np.int32)
    gt = np.zeros((B, Gmax, 5), F)
    gc = np.array([8, 2, 1], np.int32)
    for b in range(B):
        gt[b, :gc[b], :4] = rand_boxes(rs, gc[b], 1024, 32, 512)
        gt[b, :gc[b], 4] = rs.randint(0, 80, size=gc[b])
        p = rand_boxes(rs, pc[b], 1024, 8, 600)
        near = gt[b, rs.randint(0, gc[b], size=pc[b] // 3), :4] + rs.randint(-20, 21, size=(pc[b] // 3, 4))
        p[: pc[b] // 3] = near
        props[b, :pc[b]] = p
    seeds = np.array([orng.image_seed(1, 9, b) for b in range(B)], np.uint32)
    r = K.rcnn_target(T(props), T(pc), T(gt), T(gc), T(seeds.view(np.int32)), want_pre=True)
    r = {k: v.cpu().numpy() for k, v in r.items()}
    for b in range(B):
        ol, ot_, opre, _, _ = of.rcnn_target(props[b, :pc[b]], gt[b, :gc[b]], seed=int(seeds[b]),
                                             return_pre_subsample=True)
        np.testing.assert_array_equal(r['labels_pre'][b, :pc[b]], opre)
        np.testing.assert_array_equal(r['labels'][b, :pc[b]], ol)
        assert (r['labels'][b, pc[b]:] == -1).all()
        np.testing.assert_allclose(r['bbox_targets'][b, :pc[b]], ot_, rtol=1e-5, atol=1e-6)
        keep = ol >= 0                                      # rcnn.py:156-167 compaction
        n = int(keep.sum())
        assert r['roi_count'][b] == n and n <= 256
        np.testing.assert_array_equal(r['rois'][b, :n], props[b, :pc[b]][keep])
        np.testing.assert_array_equal(r['roi_labels'][b, :n], ol[keep])
        np.testing.assert_allclose(r['roi_targets'][b, :n], ot_[keep], rtol=1e-5, atol=1e-6)
        assert (r['roi_labels'][b, n:] == -1).all() and (r['rois'][b, n:] == 0).all()


def test_rcnn_target_reference_vectors(K):
    # luminoth/models/fasterrcnn/rcnn_target_test.py:349-398 (labels) and :475-524 (priority)
    gt = np.array([[(10, 0, 398, 399, 0), (200, 300, 250, 390, 1), (185, 305, 235, 372, 2)]], F)
    props = np.array([[(12, 70, 350, 540), (190, 310, 240, 370), (197, 300, 252, 389), (196, 300, 252, 389),
                       (197, 303, 252, 394), (180, 310, 235, 370), (0, 0, 400, 400), (197, 302, 252, 389),
                       (0, 0, 400, 400)]], F)
    r = K.rcnn_target(T(props), T(np.array([9], np.int32)), T(gt), T(np.array([3], np.int32)),
                      T(np.array([0], np.int32)), minibatch_size=18, foreground_fraction=0.5,
                      background_threshold_low=0.1)
    np.testing.assert_array_equal(r['labels'][0, 1:].cpu().numpy(), np.add([2., 1., 1., 1., 2., 0., 1., 0.], 1))
    gt = np.array([[[10, 10, 20, 20, 3.], [10, 10, 30, 30, 4.]]], F)
    props = np.array([[[10, 10, 20, 20], [12, 10, 20, 20]]], F)
    r = K.rcnn_target(T(props), T(np.array([2], np.int32)), T(gt), T(np.array([2], np.int32)),
                      T(np.array([0], np.int32)), minibatch_size=64, foreground_fraction=0.5)
    lab = r['labels'][0].cpu().numpy()
    assert (lab == 4.).sum() == 1 and (lab == 5.).sum() == 1


# ------------------------------------------- edge cases / unbounded sizes ----
def test_rpn_target_edge_cases_match_oracle(K):
    """GPU twins of the oracle-only edge cases: B.4 a gt whose best IoU is 0 turns every inside
    anchor positive (rpn_target.py:155-178); an image without gt boxes; every anchor outside the image."""
    ref = obx.generate_anchors_reference(64, np.array([.5, 1, 2]), np.array([.25, .5, 1, 2]))
    ref_i32 = np.trunc(ref).astype(np.int32)
    fh, fw, stride, im = 12, 16, 16, (192, 256)
    anchors = obx.generate_anchors(ref, fh, fw, stride)
    gt = np.zeros((3, 4, 5), F)
    gt[0, 0] = [300, 220, 340, 260, 1]                 # outside the image: IoU 0 with every inside anchor   (B.4)
    gt[0, 1] = [20, 30, 90, 120, 2]
    gt[2, :2] = [[10, 10, 60, 60, 0], [100, 50, 200, 150, 3]]
    counts = np.array([2, 0, 2], np.int32)              # image 1: no gt at all
    seeds = np.array([orng.image_seed(2, 1, b) for b in range(3)], np.uint32)
    labels, targets, mo, pre = K.rpn_target(T(ref_i32), fh, fw, stride, T(gt), T(counts), T(seeds.view(np.int32)), im,
                                            want_pre=True)
    labels, targets, mo, pre = [t.cpu().numpy() for t in (labels, targets, mo, pre)]
    for b in range(3):
        ol, ot_, om, opre, _ = of.rpn_target(anchors, gt[b, :counts[b]], im, seed=int(seeds[b]),
                                             return_pre_subsample=True)
        np.testing.assert_array_equal(pre[b], opre)
        np.testing.assert_array_equal(labels[b], ol)
        np.testing.assert_array_equal(mo[b], om)
        np.testing.assert_allclose(targets[b], ot_, rtol=1e-5, atol=1e-6)
    inside = pre[0] >= 0
    assert inside.sum() > 256 and (pre[0][inside] == 1).all()       # B.4: every inside anchor is a positive
    assert (labels[0] == 1).sum() == 128 and (labels[0] == 0).sum() == 0
    assert (pre[1][pre[1] >= 0] == 0).all() and (labels[1] == 0).sum() == 256     # no gt: backgrounds only
    # every anchor outside: a 40x40 image is smaller than every anchor of this set
    labels, targets, mo, _ = K.rpn_target(T(ref_i32 * 4), 3, 3, 16, T(gt[2:3]), T(counts[2:3]), T(seeds[:1].view(np.int32)),
                                          (40, 40))
    assert (labels.cpu().numpy() == -1).all() and (targets.cpu().numpy() == 0).all() and (mo.cpu().numpy() == 0).all()


def test_rcnn_target_max_bg_zero_and_no_gt(K):
    """B.6 (rcnn_target.py:211-250): `#bg >= max_bg` with max_bg == 0 disables EVERY background; fg disabled -> -label."""
    rs = np.random.RandomState(5)
    gt = np.zeros((1, 4, 5), F)
    gt[0, :2] = [[50, 50, 200, 220, 4], [300, 100, 460, 300, 9]]
    props = np.concatenate([gt[0, rs.randint(0, 2, size=60), :4] + rs.randint(-15, 16, size=(60, 4)),
                            rand_boxes(rs, 60, 600, 10, 250)]).astype(F)[None]
    seeds = np.array([77], np.int32)
    r = K.rcnn_target(T(props), T(np.array([120], np.int32)), T(gt), T(np.array([2], np.int32)), T(seeds),
                      minibatch_size=16, foreground_fraction=1.0)
    ol, ot_ = of.rcnn_target(props[0], gt[0, :2], seed=77, minibatch_size=16, foreground_fraction=1.0)
    lab = r['labels'][0].cpu().numpy()
    np.testing.assert_array_equal(lab, ol)
    assert (lab > 0).sum() == 16 and (lab == 0).sum() == 0 and (lab < -1).sum() > 0
    assert int(r['roi_count'][0]) == 16


def test_targets_are_unbounded_in_gt_and_proposals(K):
    """The reference bounds neither the gt boxes nor the proposals (rcnn_target.py:48-66): 300 gt boxes stream through
    LDS in three chunks, 6000 proposals keep their state in the workspace (round 2 returned LMH_ERR_INVALID)."""
    rs = np.random.RandomState(17)
    G = 300
    ref = obx.generate_anchors_reference(64, np.array([.5, 1, 2]), np.array([.25, .5, 1, 2]))
    ref_i32 = np.trunc(ref).astype(np.int32)
    fh, fw, stride, im = 32, 32, 16, (512, 512)
    anchors = obx.generate_anchors(ref, fh, fw, stride)
    gt = np.zeros((1, G, 5), F)
    gt[0, :, :4] = rand_boxes(rs, G, 512, 12, 160)
    gt[0, :, 4] = rs.randint(0, 80, size=G)
    gt[0, 200, :4] = gt[0, 3, :4]                        # duplicates across chunks: argmax keeps the FIRST, best-of-gt the LAST
    gt[0, 290, :4] = gt[0, 140, :4]
    cnt = np.array([G], np.int32)
    seeds = np.array([orng.image_seed(4, 2, 0)], np.uint32)
    labels, targets, mo, pre = K.rpn_target(T(ref_i32), fh, fw, stride, T(gt), T(cnt), T(seeds.view(np.int32)), im,
                                            want_pre=True)
    ol, ot_, om, opre, _ = of.rpn_target(anchors, gt[0], im, seed=int(seeds[0]), return_pre_subsample=True)
    np.testing.assert_array_equal(pre[0].cpu().numpy(), opre)
    np.testing.assert_array_equal(labels[0].cpu().numpy(), ol)
    np.testing.assert_array_equal(mo[0].cpu().numpy(), om)
    np.testing.assert_allclose(targets[0].cpu().numpy(), ot_, rtol=1e-5, atol=1e-6)
    for P in (2000, 6000):
        props = np.concatenate([gt[0, rs.randint(0, G, size=P // 2), :4] + rs.randint(-10, 11, size=(P // 2, 4)),
                                rand_boxes(rs, P - P // 2, 512, 8, 300)]).astype(F)[rs.permutation(P)][None]
        r = K.rcnn_target(T(props), T(np.array([P], np.int32)), T(gt), T(cnt), T(seeds.view(np.int32)), want_pre=True)
        ol, ot_, opre, _, _ = of.rcnn_target(props[0], gt[0], seed=int(seeds[0]), return_pre_subsample=True)
        np.testing.assert_array_equal(r['labels_pre'][0].cpu().numpy(), opre)
        np.testing.assert_array_equal(r['labels'][0].cpu().numpy(), ol)
        np.testing.assert_allclose(r['bbox_targets'][0].cpu().numpy(), ot_, rtol=1e-5, atol=1e-6)
        keep = ol >= 0
        assert int(r['roi_count'][0]) == keep.sum() == 256
        np.testing.assert_array_equal(r['rois'][0].cpu().numpy(), props[0][keep])


def test_nms_beyond_32768_candidates(K):
    """Round 2 refused K > 32768; the only structural limit left is the grid (64 * 65535)."""
    rs = np.random.RandomState(23)
    Kn = 40000
    base = rand_boxes(rs, 3000, 1000, 20, 200)
    boxes = (base[rs.randint(0, 3000, size=Kn)] + rs.randint(-8, 9, size=(Kn, 4))).astype(F)[None]
    counts = np.array([Kn], np.int32)
    keep, kc = K.nms(T(boxes), T(counts), 0.6, 1500)
    ref = tfops.non_max_suppression(boxes[0][:, [1, 0, 3, 2]], np.arange(Kn, 0, -1).astype(F), 1500, 0.6)
    assert int(kc[0]) == ref.shape[0]
    np.testing.assert_array_equal(keep[0, :ref.shape[0]].cpu().numpy(), ref)


# -------------------------------------------------------------- ROI pool ----
def test_roi_pool_fwd_bwd(K):
    rs = np.random.RandomState(31)
    B, FH, FW, C, R = 2, 20, 24, 64, 24
    feat = rs.randn(B, FH, FW, C).astype(F)
    rois = np.zeros((B, R, 4), F)
    cnt = np.array([24, 10], np.int32)
    for b in range(B):
        rois[b] = rand_boxes(rs, R, 320, 8, 200)
    rois[0, 0] = [-30, -20, 100, 90]        # partly outside: extrapolation 0
    rois[0, 1] = [300, 200, 420, 380]       # beyond the image
    im = (320, 384)
    out, am = K.roi_pool_fwd(T(feat), T(rois), T(cnt), im)
    out = out.cpu().numpy().reshape(B, R, 7, 7, C)
    for b in range(B):
        ref, _ = of.roi_pool(rois[b, :cnt[b]], feat[b:b + 1], im, 7, 7)
        np.testing.assert_array_equal(out[b, :cnt[b]], ref)          # bit-exact bilinear + max
        assert (out[b, cnt[b]:] == 0).all()
    # backward vs torch autograd of the torch restatement
    ft = torch.tensor(feat, requires_grad=True)
    sel = [(b, r) for b in range(B) for r in range(cnt[b])]
    rr = torch.tensor(np.stack([rois[b, r] for b, r in sel]))
    bi = torch.tensor([b for b, _ in sel])
    pooled = ot.roi_pool(ft, rr, bi, im)
    g = rs.randn(B, R, 7, 7, C).astype(F)
    gsel = torch.tensor(np.stack([g[b, r] for b, r in sel]))
    pooled.backward(gsel)
    dfeat = K.roi_pool_bwd(T(g.reshape(B * R, 7, 7, C)), am, T(rois), T(cnt), (B, FH, FW, C), im)
    np.testing.assert_allclose(dfeat.cpu().numpy(), ft.grad.numpy(), rtol=1e-4, atol=1e-5)  # atomics order
    # reference quadrant vectors, roi_pool_test.py:56-175
    m = np.block([[np.ones((5, 5)) * 1, np.ones((5, 5)) * 2], [np.ones((5, 5)) * 3, np.ones((5, 5)) * 4]])
    fm = np.repeat(m[None, :, :, None], 4, axis=3).astype(F)
    rq = np.array([[[3, 1, 6, 4], [1, 3, 4, 7], [5, 3, 9, 7], [3, 6, 6, 9]]], F)
    o, _ = K.roi_pool_fwd(T(fm), T(rq), T(np.array([4], np.int32)), (10, 10), ph=2, pw=2)
    o = o.cpu().numpy()[..., 0]
    np.testing.assert_array_equal(o[0], [[1, 2], [1, 2]])
    np.testing.assert_array_equal(o[1], [[1, 1], [3, 3]])
    np.testing.assert_array_equal(o[2], [[2, 2], [4, 4]])
    np.testing.assert_array_equal(o[3], [[3, 4], [3, 4]])


@pytest.mark.parametrize('shape', [(2, 20, 24, 64, 24), (2, 64, 64, 128, 256), (1, 50, 84, 32, 64), (1, 64, 75, 16, 32)])
def test_roi_pool_mean_fused_equals_pool_then_mean(K, shape):
    """k_roi_pool_mean_fwd / the MEAN backward == roi_pool + spatial_mean, bit for bit (forward: same arithmetic in the
    same order; backward: the fixed-point slab sum does not depend on order).  Last shape: C % 8 == 0 but the map only
    fits the 4-channel slab of the backward."""
    B, FH, FW, C, R = shape
    rs = np.random.RandomState(7)
    feat = rs.randn(B, FH, FW, C).astype(F)
    im = (FH * 16, FW * 16)
    rois = np.zeros((B, R, 4), F)
    for b in range(B):
        rois[b] = rand_boxes(rs, R, min(im), 8, min(im) // 2)
    rois[0, 0] = [-30, -20, 100, 90]
    rois[0, 1] = [im[1] - 20, im[0] - 30, im[1] + 80, im[0] + 60]
    cnt = np.array([R, max(1, R // 3)][:B], np.int32)
    assert K.roi_pool_mean_supported((B, FH, FW, C))
    out, am = K.roi_pool_fwd(T(feat), T(rois), T(cnt), im)
    ref_mean = K.spatial_mean_fwd(out)
    mean, am2 = K.roi_pool_mean_fwd(T(feat), T(rois), T(cnt), im)
    assert torch.equal(mean, ref_mean)
    assert torch.equal(am, am2)
    # every slab width of the forward kernel (8 / 4 channels: one block per (half a) compute unit; 2 / 1: 256 * CS threads
    # and <= 32 KB, round 5) is the same arithmetic per (ROI, channel)
    for cs in (4, 2, 1):
        K.set_option('roi_mean_cs', cs)
        try:
            m_cs, am_cs = K.roi_pool_mean_fwd(T(feat), T(rois), T(cnt), im)
        finally:
            K.set_option('roi_mean_cs', -1)
        assert torch.equal(m_cs, ref_mean) and torch.equal(am_cs, am), cs
    dy = T(rs.randn(B * R, C).astype(F))
    ref_d = K.roi_pool_bwd(K.spatial_mean_bwd(dy, tuple(out.shape)), am, T(rois), T(cnt), (B, FH, FW, C), im)
    d = K.roi_pool_mean_bwd(dy, am2, T(rois), T(cnt), (B, FH, FW, C), im)
    assert torch.equal(d, ref_d)
    assert not K.roi_pool_mean_supported((1, 200, 200, 64))          # 40000 pixels: the unfused pair is the path


def test_spatial_mean(K):
    x = torch.randn(37, 7, 7, 128, device=dev())
    np.testing.assert_allclose(K.spatial_mean_fwd(x).cpu().numpy(), x.cpu().mean(dim=(1, 2)).numpy(),
                               rtol=1e-5, atol=1e-6)
    dy = torch.randn(37, 128, device=dev())
    dx = K.spatial_mean_bwd(dy, x.shape).cpu().numpy()
    np.testing.assert_allclose(dx, np.broadcast_to(dy.cpu().numpy()[:, None, None, :] / 49, x.shape), rtol=1e-6)


# ---------------------------------------------------------------- losses ----
def test_rpn_loss(K):
    rs = np.random.RandomState(41)
    B, N = 2, 49152
    score = rs.randn(B, N, 2).astype(F)
    pred = (rs.randn(B, N, 4) * 0.5).astype(F)
    labels = np.full((B, N), -1, F)
    tg = np.zeros((B, N, 4), F)
    for b in range(B):
        idx = rs.choice(N, 256, replace=False)
        labels[b, idx[:100 + 20 * b]] = 1
        labels[b, idx[100 + 20 * b:]] = 0
        tg[b, idx[:100 + 20 * b]] = (rs.randn(100 + 20 * b, 4) * 0.5).astype(F)
    losses, per, dc, db = K.rpn_loss(T(score), T(pred), T(labels), T(tg), sigma=3.0)
    st, pt = torch.tensor(score, requires_grad=True), torch.tensor(pred, requires_grad=True)
    tot = 0
    for b in range(B):
        o = of.rpn_loss(score[b], labels[b], pred[b], tg[b], 3.0)
        np.testing.assert_allclose(per[b, :2].cpu().numpy(), [o['rpn_cls_loss'], o['rpn_reg_loss']], rtol=1e-5)
        c, r = ot.rpn_loss(st[b], pt[b], torch.tensor(labels[b]), torch.tensor(tg[b]), 3.0)
        tot = tot + (c + r) / B
    tot.backward()
    np.testing.assert_allclose(float(losses.sum()), float(tot), rtol=1e-5)       # loss: 1e-5 rel
    np.testing.assert_allclose(dc.cpu().numpy(), st.grad.numpy(), rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(db.cpu().numpy(), pt.grad.numpy(), rtol=1e-4, atol=1e-8)


def test_rcnn_loss(K):
    rs = np.random.RandomState(43)
    B, R, C = 2, 256, 80
    score = rs.randn(B, R, C + 1).astype(F)
    off = (rs.randn(B, R, 4 * C) * 0.5).astype(F)
    labels = np.full((B, R), -1, F)
    tg = np.zeros((B, R, 4), F)
    for b in range(B):
        n = 200 + 30 * b
        labels[b, :n] = 0
        fg = rs.choice(n, 50, replace=False)
        labels[b, fg] = rs.randint(1, C + 1, size=50)
        tg[b, fg] = rs.randn(50, 4).astype(F)
    losses, per, dc, do = K.rcnn_loss(T(score), T(off), T(labels), T(tg), C, sigma=1.0)
    st, ot_ = torch.tensor(score, requires_grad=True), torch.tensor(off, requires_grad=True)
    tot = 0
    for b in range(B):
        o = of.rcnn_loss(score[b], off[b], labels[b], tg[b], C, 1.0)
        np.testing.assert_allclose(per[b, :2].cpu().numpy(), [o['rcnn_cls_loss'], o['rcnn_reg_loss']], rtol=1e-5)
        c, r = ot.rcnn_loss(st[b], ot_[b], torch.tensor(labels[b]), torch.tensor(tg[b]), C, 1.0)
        tot = tot + (c + r) / B
    tot.backward()
    np.testing.assert_allclose(float(losses.sum()), float(tot), rtol=1e-5)
    np.testing.assert_allclose(dc.cpu().numpy(), st.grad.numpy(), rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(do.cpu().numpy(), ot_.grad.numpy(), rtol=1e-4, atol=1e-8)
    # the gradient half alone (it counts its own normalisers) and the value half alone give the same bits
    dc2, do2 = K.rcnn_loss_grad(T(score), T(off), T(labels), T(tg), C, sigma=1.0)
    assert torch.equal(dc2, dc) and torch.equal(do2, do)
    losses2, per2, _, _ = K.rcnn_loss(T(score), T(off), T(labels), T(tg), C, sigma=1.0, want_grad=False)
    assert torch.equal(losses2, losses) and torch.equal(per2, per)
    y = K.softmax(T(score)).cpu().numpy()
    np.testing.assert_allclose(y, tfops.softmax(score), rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------ conv ----
CONV_CASES = [
    # N, H, W, C, K, R, stride, dil, padding, act
    (2, 32, 32, 64, 256, 1, 1, 1, 'SAME', 'relu'),
    (2, 32, 32, 256, 64, 3, 1, 1, 'SAME', 'relu'),
    (1, 33, 47, 128, 128, 3, 2, 1, 'SAME_EXPLICIT', 'relu'),
    (1, 64, 64, 3, 64, 7, 2, 1, 'SAME_EXPLICIT', 'relu'),
    (2, 19, 19, 512, 1024, 3, 1, 6, 'SAME', 'relu'),
    (1, 16, 16, 512, 72, 1, 1, 1, 'VALID', None),
    (1, 16, 32, 1024, 404, 1, 1, 1, 'VALID', None),
    (1, 64, 64, 1024, 512, 3, 1, 1, 'SAME', 'relu6'),
    (1, 10, 10, 256, 256, 3, 2, 1, 'SAME', 'relu'),
    (1, 5, 5, 128, 256, 3, 1, 1, 'VALID', 'relu'),
    (1, 1, 512, 1024, 81, 1, 1, 1, 'VALID', None),      # fc_classifier: K % 4 != 0 scalar paths
    (1, 1, 500, 1024, 320, 1, 1, 1, 'VALID', 'relu'),
    (2, 1, 301, 1024, 81, 1, 1, 1, 'VALID', None),      # Linear layer, 81 columns: rows % 8 != 0, odd row count
    (1, 1, 37, 256, 21, 1, 1, 1, 'VALID', 'relu'),      # ... with an activation, 21 columns (VOC classes + 1)
    (2, 24, 40, 512, 24, 1, 1, 1, 'VALID', None),       # RPN cls head (2 x 12 anchors): k_skinny_bwd_data, K = 24
    (2, 23, 41, 512, 48, 1, 1, 1, 'VALID', None),       # RPN bbox head (4 x 12): K = 48, pixel count % 64 != 0
    (1, 9, 9, 288, 48, 1, 1, 1, 'SAME', 'relu'),        # skinny bwd-data with a partial 256-channel tile (C = 288)
    (2, 18, 18, 1024, 24, 3, 1, 1, 'SAME', None),       # SSD multibox offsets head: 3x3, K % 32 != 0
    (2, 9, 9, 512, 126, 3, 1, 1, 'SAME', None),         # SSD multibox classes head: 3x3, K % 4 != 0
    (1, 19, 19, 512, 1024, 3, 1, 6, 'SAME', 'relu'),    # SSD conv6: rate 6
    (2, 5, 5, 128, 256, 3, 1, 1, 'VALID', 'relu'),      # SSD conv10_2
    # stride-2 3x3 with even sizes: bwd_data walks the pixels parity class by parity class and skips dead taps
    (2, 64, 64, 128, 128, 3, 2, 1, 'SAME_EXPLICIT', 'relu'),   # ResNet block2 unit4 (explicit pad 1/1)
    (1, 32, 32, 64, 128, 3, 2, 1, 'SAME', 'relu'),             # TF SAME on an even size: pad 0 before, 1 after
    (2, 32, 64, 96, 64, 3, 2, 1, 'SAME_EXPLICIT', None),       # C % 64 != 0 (partial column tile), no activation
]


@pytest.mark.parametrize('M,C,N', [(512, 1024, 81), (512, 1024, 320), (37, 2048, 81), (1000, 512, 1), (4096, 4096, 33),
                                   (33, 1024, 512)])
def test_linear_head_kernel(K, M, C, N):
    """k_head_fwd (split-reduction MFMA kernel of the RCNN heads, rcnn.py:221-228): ragged row / column counts, one column,
    the largest routed shape; against float64, and bit-identical to itself with the option that routes it off — the tiled
    kernels — within fp32 round-off; integer data: exact, whatever the summation order."""
    rs = np.random.RandomState(M + C + N)
    x = rs.randn(1, 1, M, C).astype(F)
    w = (rs.randn(1, 1, C, N) * np.sqrt(1.0 / C)).astype(F)
    b = rs.randn(N).astype(F)
    d = K.conv_desc(x.shape, w.shape, 1, 1, 'VALID', None)
    y = K.conv2d_fwd(d, T(x), T(w), None, T(b)).cpu().numpy().reshape(M, N)
    ref = x.reshape(M, C).astype(np.float64) @ w.reshape(C, N).astype(np.float64) + b
    np.testing.assert_allclose(y, ref, rtol=1e-5, atol=2e-5 * max(1.0, float(np.abs(ref).max())))
    K.set_option('head_gemm', 0)
    try:
        y0 = K.conv2d_fwd(d, T(x), T(w), None, T(b)).cpu().numpy().reshape(M, N)
    finally:
        K.set_option('head_gemm', 1)
    np.testing.assert_allclose(y, y0, rtol=1e-5, atol=2e-5 * max(1.0, float(np.abs(ref).max())))
    xi = rs.randint(-8, 9, size=x.shape).astype(F)
    wi = rs.randint(-8, 9, size=w.shape).astype(F)
    yi = K.conv2d_fwd(d, T(xi), T(wi)).cpu().numpy().reshape(M, N)
    np.testing.assert_array_equal(yi, xi.reshape(M, C) @ wi.reshape(C, N))


PP_CASES = [
    # N, H, W, C, K, residual, scale/shift, act, act_bits, which kernel takes it
    (2, 64, 64, 256, 1024, True, True, 'relu', True, 'pp'),      # block3 expand of the benchmark step: 512 tiles, 2 per block
    (2, 128, 128, 128, 512, True, True, 'relu', True, 'pp'),     # block2 expand: 1024 tiles, 4 per block, 4 stages per tile
    (2, 63, 65, 128, 1024, True, True, 'relu6', True, 'pp'),     # ragged rows: the last row tile has 126 rows
    (2, 64, 64, 256, 1024, False, False, None, False, 'pp'),     # no residual, no BatchNorm, no activation, no mask
    (1, 128, 96, 160, 640, True, False, 'relu', True, 'pp'),     # 5 stages per tile; 480 tiles over 240 blocks
    (2, 63, 65, 128, 1000, True, True, 'relu6', False, 'tiled'),  # K % 128 != 0: the tiled kernel keeps it
    (2, 64, 64, 512, 1024, True, True, 'relu', True, 'tiled'),   # 16 stages per tile: the tiled kernel keeps it
]


@pytest.mark.parametrize('case', PP_CASES)
def test_conv1x1_pp_equals_tiled_kernel(K, case):
    """k_conv1x1_pp (csrc/conv_pp.h: one block per compute unit, tiles software-pipelined, epilogue straight from the
    accumulator registers) == k_conv_fwd (one block per tile, epilogue through LDS), bit for bit: output AND activation
    bit mask; and both within the fp32 contract of the torch-CPU reference."""
    import ctypes
    N, H, W, C, Kc, use_res, use_bn, act, want_bits, taker = case
    rs = np.random.RandomState(PP_CASES.index(case) + 900)
    x = T(rs.randn(N, H, W, C).astype(F))
    w = T((rs.randn(1, 1, C, Kc) * np.sqrt(2.0 / C)).astype(F))
    scale = T((1 + 0.1 * rs.randn(Kc)).astype(F)) if use_bn else None
    shift = T((0.1 * rs.randn(Kc)).astype(F)) if use_bn else None
    res = T(rs.randn(N, H, W, Kc).astype(F)) if use_res else None
    d = K.conv_desc(x.shape, w.shape, 1, 1, 'SAME', act)
    lib = K._lib.load()
    out = {}
    for pp in (1, 0):
        K.set_option('conv_pp', pp)
        try:
            bits = K.new_act_bits(N * H * W, Kc, x.device).fill_(-1) if want_bits else None
            fl = ctypes.c_double(0.0)
            e0, e1 = lib.lmh_event_create(), lib.lmh_event_create()
            lib.lmh_conv2d_profile_next(e0, e1)
            y = K.conv2d_fwd(d, x, w, scale, shift, res, act_bits=bits)
            name = lib.lmh_conv2d_profile_last(ctypes.byref(fl)).decode()
            torch.cuda.synchronize()
            lib.lmh_event_destroy(e0)
            lib.lmh_event_destroy(e1)
            out[pp] = (y, bits, name)
        finally:
            K.set_option('conv_pp', 1)
    assert (out[1][2].startswith('k_conv1x1_pp<') if taker == 'pp' else out[1][2] == out[0][2]) and out[0][2].startswith('k_conv_fwd<'), (out[1][2], out[0][2])
    assert torch.equal(out[1][0], out[0][0])
    if want_bits:
        assert torch.equal(out[1][1], out[0][1])
    yt = ot.conv2d_nhwc(x.cpu(), w.cpu(), 1, 1, 'SAME')
    if use_bn:
        yt = yt * scale.cpu() + shift.cpu()
    if use_res:
        yt = yt + res.cpu()
    if act:
        yt = torch.clamp(yt, 0, 6) if act == 'relu6' else torch.relu(yt)
    np.testing.assert_allclose(out[1][0].cpu().numpy(), yt.numpy(), rtol=1e-4, atol=2e-5 * max(1.0, float(yt.abs().max())))


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_bwd(K, case, monkeypatch):
    monkeypatch.setattr(K, 'WINOGRAD', False)      # the direct kernels are under test here (bit-identity checks)
    N, H, W, C, Kc, R, stride, dil, padding, act = case
    rs = np.random.RandomState(CONV_CASES.index(case) + 100)     # (hash() of a str-bearing tuple is per-process)
    x = rs.randn(N, H, W, C).astype(F)
    w = (rs.randn(R, R, C, Kc) * np.sqrt(2.0 / (R * R * C))).astype(F)
    scale = (1 + 0.1 * rs.randn(Kc)).astype(F)
    shift = (0.1 * rs.randn(Kc)).astype(F)
    in_sub = np.array([.3, -.2, .1], F) if C == 3 else None
    d = K.conv_desc(x.shape, w.shape, stride, dil, padding, act)
    res = rs.randn(N, d.OH, d.OW, Kc).astype(F)
    y = K.conv2d_fwd(d, T(x), T(w), T(scale), T(shift), T(res), None if in_sub is None else T(in_sub))
    xt = torch.tensor(x, requires_grad=True)
    wt = torch.tensor(w, requires_grad=True)
    xin = xt - torch.tensor(in_sub) if in_sub is not None else xt
    conv = ot.conv2d_nhwc(xin, wt, stride, dil, padding)
    yt = conv * torch.tensor(scale) + torch.tensor(shift) + torch.tensor(res)
    if act == 'relu':
        yt = torch.relu(yt)
    elif act == 'relu6':
        yt = torch.clamp(yt, 0, 6)
    assert tuple(y.shape) == tuple(yt.shape)
    # fp32 MFMA == fmaf chain; reference sums in another order: 2e-5 of the output scale
    tol = 2e-5 * max(1.0, float(yt.abs().max()))
    np.testing.assert_allclose(y.cpu().numpy(), yt.detach().numpy(), rtol=1e-4, atol=tol)
    # backward through the fused layer
    gy = rs.randn(*yt.shape).astype(F)
    g = K.act_bwd(T(gy), y, act) if act else T(gy)
    colsum = torch.zeros(Kc, device=dev())
    K.act_bwd(T(gy), y, act, want_g=False, colsum=colsum)
    yg = y.cpu().numpy()      # mask from the kernel's own output (values within 1 ulp of 0 / 6 may differ)
    gref = gy * ((yg > 0) & ((yg < 6) if act == 'relu6' else True)) if act else gy
    np.testing.assert_array_equal(g.cpu().numpy(), gref)
    (conv * torch.tensor(scale)).backward(torch.tensor(gref))     # reference backward under the same mask
    np.testing.assert_allclose(colsum.cpu().numpy(), gref.reshape(-1, Kc).sum(0), rtol=1e-3, atol=1e-3)
    dx = K.conv2d_bwd_data(d, g, T(w), T(scale))
    tolx = 2e-5 * max(1.0, float(xt.grad.abs().max()))
    np.testing.assert_allclose(dx.cpu().numpy(), xt.grad.numpy(), rtol=1e-4, atol=tolx)
    add = rs.randn(*x.shape).astype(F)
    dx2 = K.conv2d_bwd_data(d, g, T(w), T(scale), addend=T(add))
    np.testing.assert_allclose(dx2.cpu().numpy(), xt.grad.numpy() + add, rtol=1e-4, atol=tolx)
    xw = x - in_sub if in_sub is not None else x     # (the mean subtraction is part of the forward gather only)
    dw = K.conv2d_bwd_weight(d, T(xw), g)           # raw: w.r.t. the un-scaled conv output
    dw_ref = wt.grad.numpy()
    dws = dw.cpu().numpy() * scale[None, None, None, :]
    tolw = 5e-5 * max(1.0, float(np.abs(dw_ref).max()))
    np.testing.assert_allclose(dws, dw_ref, rtol=1e-3, atol=tolw)
    # activation bit mask (round 3): the forward epilogue (or the stand-alone pass for kernels without it) writes one bit
    # per output element; a consumer's backward-data epilogue applies a mask of ITS input: dx * act'(x) — bit-identical
    # to masking afterwards
    if act and Kc % 32 == 0:
        bits = K.new_act_bits(N * d.OH * d.OW, Kc, dev())
        y2 = K.conv2d_fwd(d, T(x), T(w), T(scale), T(shift), T(res), None if in_sub is None else T(in_sub),
                          act_bits=bits)
        assert torch.equal(y2, y)
        ref_bits = np.packbits(((yg > 0) & ((yg < 6) if act == 'relu6' else True)).reshape(-1, Kc // 32, 32),
                               axis=-1, bitorder='little').view(np.uint32).reshape(-1, Kc // 32)
        np.testing.assert_array_equal(bits.cpu().numpy().view(np.uint32), ref_bits)
        assert torch.equal(K.act_bits(y, act), bits)
    if C % 32 == 0:
        xm = rs.rand(N * H * W, C) > 0.4                                    # any mask of the layer input
        xbits = T(np.packbits(xm.reshape(-1, C // 32, 32), axis=-1, bitorder='little').view(np.int32).reshape(-1, C // 32))
        dx_m = K.conv2d_bwd_data(d, g, T(w), T(scale), addend=T(add), xbits=xbits)
        np.testing.assert_array_equal(dx_m.cpu().numpy(), np.where(xm.reshape(x.shape), dx2.cpu().numpy(), 0))
    # fused activation backward: the kernels apply act'(y) while loading dy -> bit-identical results,
    # and bwd_weight emits the per-channel sums of g (dbeta / dbias)
    if K.conv_fused_colsum_ok(d):
        cs = torch.full((Kc,), 7.0, device=dev())
        dw_c = K.conv2d_bwd_weight(d, T(xw), g, colsum=cs)         # same kernel as `dw`, column sums riding along
        assert torch.equal(dw_c, dw)
        np.testing.assert_allclose(cs.cpu().numpy(), gref.reshape(-1, Kc).sum(0), rtol=1e-3, atol=1e-3)
        if act and K.conv_fused_act_ok(d):
            cs2 = torch.full((Kc,), 7.0, device=dev())
            dx_f = K.conv2d_bwd_data(d, T(gy), T(w), T(scale), yact=y)
            assert torch.equal(dx_f, dx)
            dw_f = K.conv2d_bwd_weight(d, T(xw), T(gy), yact=y, colsum=cs2)
            if R == 1 and stride == 1:      # 1x1: `dw` came from the direct-to-LDS kernel, the yact form from the
                np.testing.assert_allclose(dw_f.cpu().numpy(), dw.cpu().numpy(), rtol=1e-4, atol=tolw)   # register-staged one
            else:
                assert torch.equal(dw_f, dw)
            np.testing.assert_allclose(cs2.cpu().numpy(), gref.reshape(-1, Kc).sum(0), rtol=1e-3, atol=1e-3)


WINO_CASES = [
    # N, H, W, C, K, act
    (2, 16, 16, 64, 96, 'relu'),        # several GEMM tiles per transformed plane, K not a multiple of 64
    (1, 64, 64, 1024, 512, 'relu6'),    # the RPN convolution (rpn.py:69-75) at the benchmark size
    (2, 6, 10, 32, 32, None),           # tiny, H != W, no activation
    (2, 38, 38, 512, 512, 'relu'),      # VGG conv4 block at SSD-like size
    (2, 37, 37, 256, 256, 'relu'),      # odd sizes (SSD conv4_3 map): half-empty last tile row / column
    (1, 5, 7, 32, 64, None),
]


WG1_CASES = [
    # N, H, W, C, K: 1x1 / stride-1 weight gradients = TN GEMMs over P = N*H*W pixels (conv_wgrad1x1.h)
    (2, 16, 16, 64, 64),        # one tile, P = 512
    (1, 13, 17, 96, 160),       # ragged P = 221, partial tiles on both axes
    (2, 32, 32, 256, 128),      # ResNet block2 conv1 geometry, scaled down
    (1, 1, 500, 1024, 320),     # Sonnet Linear (fc_bbox): P = ROIs
    (1, 7, 9, 32, 36),          # smaller than one tile everywhere
]


@pytest.mark.parametrize('case', WG1_CASES)
def test_wgrad_1x1_direct_to_lds_variants(K, case):
    """Every tile shape x LDS ring depth x split count of the direct-to-LDS 1x1 weight-gradient kernel against a
    plain fp32 reference, and bit-identical results across ring depths (same summation order)."""
    N, H, W, C, Kc = case
    rs = np.random.RandomState(7 + WG1_CASES.index(case))
    x = rs.randn(N, H, W, C).astype(F)
    g = rs.randn(N, H, W, Kc).astype(F)
    d = K.conv_desc(x.shape, (1, 1, C, Kc), 1, 1, 'VALID', None)
    ref = torch.tensor(x).reshape(-1, C).double().t() @ torch.tensor(g).reshape(-1, Kc).double()
    scale = float(ref.abs().max())
    lib = K._lib.load()
    assert lib.lmh_conv2d_bwd_weight_fuses_colsum(d) == 1       # per-channel sums ride along in both fp32 fast paths
    try:
        for bm, bn in ((64, 64), (128, 64), (64, 128), (128, 128)):
            for splits in (0, 1, 3):
                outs = []
                for nbuf in (2, 3, 4):
                    lib.lmh_conv2d_force_config(bm, bn, splits)
                    lib.lmh_conv2d_force_wgrad_variant(nbuf)
                    dw = K.conv2d_bwd_weight(d, T(x), T(g)).cpu()
                    np.testing.assert_allclose(dw.reshape(C, Kc).numpy(), ref.numpy(), rtol=1e-4, atol=2e-5 * scale,
                                               err_msg='tile %dx%d splits %d nbuf %d' % (bm, bn, splits, nbuf))
                    outs.append(dw)
                assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        # the register-staged kernel (variant -1) agrees too
        lib.lmh_conv2d_force_config(0, 0, 0)
        lib.lmh_conv2d_force_wgrad_variant(-1)
        dw_old = K.conv2d_bwd_weight(d, T(x), T(g)).cpu()
        np.testing.assert_allclose(dw_old.reshape(C, Kc).numpy(), ref.numpy(), rtol=1e-4, atol=2e-5 * scale)
    finally:
        lib.lmh_conv2d_force_config(0, 0, 0)
        lib.lmh_conv2d_force_wgrad_variant(0)


@pytest.fixture(params=[2, 4], ids=['F2x2', 'F4x4'])
def wino_m(request, K):
    """Both Winograd output tiles: F(4x4,3x3) is the default of round 3, F(2x2,3x3) stays selectable."""
    K.set_option('wino_m', request.param)
    yield request.param
    K.set_option('wino_m', 4)


@pytest.mark.parametrize('case', WINO_CASES)
def test_conv_winograd_equals_direct(K, case, wino_m):
    """Winograd F(2x2,3x3) / F(4x4,3x3) forward / backward-data / weight gradient against the torch fp32 reference AND
    the direct HIP kernels (same operands, same epilogues).  F(2x2): the transforms only reorder fp32 roundings (4e-5 of
    the output scale).  F(4x4): coefficients up to 8 and 1/24 amplify round-off ~20x (measured against float64 on
    post-ReLU data with a 1024-channel reduction: max 1.8e-5, rms 1.3e-6 of the output scale) — bound 1e-4, north_star's."""
    N, H, W, C, Kc, act = case
    wtol = 1.0 if wino_m == 2 else 2.5
    rs = np.random.RandomState(WINO_CASES.index(case) + 900)
    x = rs.randn(N, H, W, C).astype(F)
    w = (rs.randn(3, 3, C, Kc) * np.sqrt(2.0 / (9 * C))).astype(F)
    scale = (1 + 0.1 * rs.randn(Kc)).astype(F)
    shift = (0.1 * rs.randn(Kc)).astype(F)
    res = rs.randn(N, H, W, Kc).astype(F)
    d = K.conv_desc(x.shape, w.shape, 1, 1, 'SAME', act)
    assert K.winograd_ok(d)
    assert not K.winograd_ok(K.conv_desc(x.shape, w.shape, 2, 1, 'SAME', act))
    assert not K.winograd_ok(K.conv_desc(x.shape, w.shape, 1, 2, 'SAME', act))
    y_dir = K.conv2d_fwd(d, T(x), T(w), T(scale), T(shift), T(res))
    y_win = K.conv2d_fwd_winograd(d, T(x), T(w), T(scale), T(shift), T(res))
    xt = torch.tensor(x, requires_grad=True)
    conv = ot.conv2d_nhwc(xt, torch.tensor(w), 1, 1, 'SAME')
    yt = conv * torch.tensor(scale) + torch.tensor(shift) + torch.tensor(res)
    yt = torch.relu(yt) if act == 'relu' else (torch.clamp(yt, 0, 6) if act == 'relu6' else yt)
    tol = wtol * 4e-5 * max(1.0, float(yt.abs().max()))
    np.testing.assert_allclose(y_win.cpu().numpy(), yt.detach().numpy(), rtol=2e-4, atol=tol)
    np.testing.assert_allclose(y_win.cpu().numpy(), y_dir.cpu().numpy(), rtol=2e-4, atol=tol)
    plain = K.conv2d_fwd_winograd(d, T(x), T(w))                       # no scale / shift / residual
    ref_plain = torch.relu(conv) if act == 'relu' else (torch.clamp(conv, 0, 6) if act == 'relu6' else conv)
    np.testing.assert_allclose(plain.cpu().numpy(), ref_plain.detach().numpy(), rtol=2e-4, atol=tol)
    g = rs.randn(N, H, W, Kc).astype(F)
    (conv * torch.tensor(scale)).backward(torch.tensor(g))
    add = rs.randn(*x.shape).astype(F)
    dx_dir = K.conv2d_bwd_data(d, T(g), T(w), T(scale), addend=T(add))
    dx_win = K.conv2d_bwd_data_winograd(d, T(g), T(w), T(scale), addend=T(add))
    tolx = wtol * 4e-5 * max(1.0, float(xt.grad.abs().max()))
    np.testing.assert_allclose(dx_win.cpu().numpy(), xt.grad.numpy() + add, rtol=2e-4, atol=tolx)
    np.testing.assert_allclose(dx_win.cpu().numpy(), dx_dir.cpu().numpy(), rtol=2e-4, atol=tolx)
    dw_dir = K.conv2d_bwd_weight(d, T(x), T(g)) if not K.WINOGRAD else None
    dw_win = K.conv2d_bwd_weight_winograd(d, T(x), T(g))               # raw: w.r.t. the un-scaled conv output
    wt = torch.tensor(w, requires_grad=True)
    ot.conv2d_nhwc(torch.tensor(x), wt, 1, 1, 'SAME').backward(torch.tensor(g))
    tolw = wtol * 1e-4 * max(1.0, float(wt.grad.abs().max()))
    np.testing.assert_allclose(dw_win.cpu().numpy(), wt.grad.numpy(), rtol=1e-3, atol=tolw)
    if dw_dir is not None:
        np.testing.assert_allclose(dw_win.cpu().numpy(), dw_dir.cpu().numpy(), rtol=1e-3, atol=tolw)
    # round 3: the forward pass may keep B^T x B for the weight gradient (bit-identical result, one transform less), and
    # the per-channel sums of g come from the (1,1) plane of the transformed gradient (every tile's pixel sum)
    xk = T(x)
    K.conv2d_fwd_winograd(d, xk, T(w), T(scale), T(shift), T(res), keep_v=True)
    assert getattr(xk, '_lmh_wino_v', None) is not None
    cs = torch.full((Kc,), 7.0, device=dev())
    dw_kept = K.conv2d_bwd_weight_winograd(d, xk, T(g), colsum=cs)
    assert torch.equal(dw_kept, dw_win)
    np.testing.assert_allclose(cs.cpu().numpy(), g.reshape(-1, Kc).sum(0), rtol=1e-4, atol=1e-4 * np.abs(g).sum(axis=(0, 1, 2)).max())
    # activation bit masks through the output transform (round 3): emitted for y, applied to dx
    if act and Kc % 32 == 0:
        bits = K.new_act_bits(N * H * W, Kc, dev())
        y_b = K.conv2d_fwd_winograd(d, T(x), T(w), T(scale), T(shift), T(res), act_bits=bits)
        assert torch.equal(y_b, y_win) and torch.equal(bits, K.act_bits(y_win, act))
    xm = rs.rand(N * H * W, C) > 0.4
    xbits = T(np.packbits(xm.reshape(-1, C // 32, 32), axis=-1, bitorder='little').view(np.int32).reshape(-1, C // 32))
    dx_m = K.conv2d_bwd_data_winograd(d, T(g), T(w), T(scale), addend=T(add), xbits=xbits)
    np.testing.assert_array_equal(dx_m.cpu().numpy(), np.where(xm.reshape(x.shape), dx_win.cpu().numpy(), 0))
    dx0 = K.conv2d_bwd_data_winograd(d, T(g), T(w))                    # no kscale, no addend
    xt.grad = None
    ot.conv2d_nhwc(xt, torch.tensor(w), 1, 1, 'SAME').backward(torch.tensor(g))
    np.testing.assert_allclose(dx0.cpu().numpy(), xt.grad.numpy(), rtol=2e-4, atol=tolx)


def test_conv_stem_kernel(K):
    """ResNet conv1 (7x7/2, 3 -> 64, conv2d_same + mean subtraction + BN + ReLU): the dedicated persistent
    kernel (k_conv_stem7x7s2) vs the oracle, incl. ragged tile edges."""
    rs = np.random.RandomState(21)
    for (N, H, W) in ((2, 64, 96), (1, 70, 45), (1, 256, 256)):
        x = (rs.rand(N, H, W, 3) * 255).astype(F)
        w = (rs.randn(7, 7, 3, 64) * np.sqrt(2.0 / 147)).astype(F)
        scale = (0.01 * (1 + 0.1 * rs.randn(64))).astype(F)
        shift = (0.1 * rs.randn(64)).astype(F)
        in_sub = np.array([123.68, 116.78, 103.94], F)
        d = K.conv_desc(x.shape, w.shape, 2, 1, 'SAME_EXPLICIT', 'relu')
        assert K._lib.load().lmh_conv2d_kernel_id(d, 0) == 7007
        y = K.conv2d_fwd(d, T(x), T(w), T(scale), T(shift), None, T(in_sub))
        yt = torch.relu(ot.conv2d_nhwc(torch.tensor(x) - torch.tensor(in_sub), torch.tensor(w), 2, 1, 'SAME_EXPLICIT') *
                        torch.tensor(scale) + torch.tensor(shift))
        assert tuple(y.shape) == tuple(yt.shape)
        np.testing.assert_allclose(y.cpu().numpy(), yt.numpy(), rtol=1e-4, atol=2e-5 * max(1.0, float(yt.abs().max())))


def test_bn_param_grads(K):
    rs = np.random.RandomState(5)
    rsc, Kc = 9 * 64, 96
    w = rs.randn(rsc, Kc).astype(F)
    dwr = rs.randn(rsc, Kc).astype(F)
    dbeta, mean = rs.randn(Kc).astype(F), rs.randn(Kc).astype(F)
    rstd, scale = rs.rand(Kc).astype(F) + .5, rs.rand(Kc).astype(F) + .5
    dwt = T(dwr.copy())
    dg = K.bn_param_grads(T(w), dwt, T(dbeta), T(mean), T(rstd), T(scale))
    np.testing.assert_allclose(dg.cpu().numpy(), rstd * ((w * dwr).sum(0) - mean * dbeta), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dwt.cpu().numpy(), dwr * scale, rtol=1e-6)


def test_maxpool(K):
    rs = np.random.RandomState(6)
    for (H, W, k, s, pad) in ((33, 31, 3, 2, 'SAME'), (20, 20, 2, 2, 'VALID'), (18, 18, 3, 1, 'SAME')):
        x = rs.randn(2, H, W, 64).astype(F)
        y, geom = K.maxpool_fwd(T(x), k, s, pad)
        xt = torch.tensor(x, requires_grad=True)
        yt = ot.max_pool_nhwc(xt, k, s, pad)
        np.testing.assert_array_equal(y.cpu().numpy(), yt.detach().numpy())
        gy = rs.randn(*yt.shape).astype(F)
        yt.backward(torch.tensor(gy))
        dx = K.maxpool_bwd(T(x), y, T(gy), k, s, geom)
        np.testing.assert_allclose(dx.cpu().numpy(), xt.grad.numpy(), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------- optimizer ----
def test_sgd_momentum_and_l2(K):
    rs = np.random.RandomState(8)
    sizes = [1000, 37, 4096, 5]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    wd = np.array([5e-4, 0.0, 1e-3, 0.0], F)
    n = int(off[-1])
    w, g, v = rs.randn(n).astype(F), rs.randn(n).astype(F), rs.randn(n).astype(F)
    wt, gt_, vt = T(w.copy()), T(g), T(v.copy())
    reg = K.l2_reg_loss(wt, T(off), T(wd))
    wde = np.repeat(wd, sizes)
    np.testing.assert_allclose(float(reg), float((wde * w.astype(np.float64) ** 2 / 2).sum()), rtol=1e-5)
    # the reported regulariser is the same bits every call (per-block partials added in block order; it used to be an atomic
    # sum in arrival order): a tensor large enough for the 1024-block grid, twenty calls
    big = T(rs.randn(3_000_001).astype(F))
    boff, bwd = T(np.array([0, 1_000_000, 3_000_001], np.int64)), T(np.array([5e-4, 1e-4], F))
    vals = set(float(K.l2_reg_loss(big, boff, bwd)) for _ in range(20))
    assert len(vals) == 1, vals
    bw = big.double().cpu().numpy()
    np.testing.assert_allclose(vals.pop(), 5e-4 * (bw[:1_000_000] ** 2).sum() / 2 + 1e-4 * (bw[1_000_000:] ** 2).sum() / 2, rtol=1e-5)
    K.sgd_momentum(wt, gt_, vt, T(off), T(wd), lr=3e-4, momentum=0.9, gscale=0.5)
    gg = g * F(0.5) + wde * w
    vv = F(0.9) * v + gg
    np.testing.assert_allclose(vt.cpu().numpy(), vv, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(wt.cpu().numpy(), w - F(3e-4) * vv, rtol=1e-6, atol=1e-7)


# ------------------------------------------------------- optimizer surface ----
def _np_clip_factors(w, g, off, wd, gscale, clip):
    f = np.ones(len(wd), F)
    for s in range(len(wd)):
        gp = g[off[s]:off[s + 1]] * F(gscale) + F(wd[s]) * w[off[s]:off[s + 1]]
        nrm = np.sqrt(np.sum(gp.astype(np.float64) ** 2))
        f[s] = clip / max(nrm, clip)
    return f


def test_optimizer_variants_match_tf_formulas(K):
    """lmh_grad_clip_factors + lmh_optimizer_step against numpy restatements of tf.clip_by_norm (training.py:84-120),
    tf.train.MomentumOptimizer / AdamOptimizer / RMSPropOptimizer (OPTIMIZERS table, training.py:6-11)."""
    rs = np.random.RandomState(5)
    off = np.array([0, 1000, 1004, 5003], np.int64)
    n = int(off[-1])
    wd = np.array([5e-4, 0.0, 1e-3], F)
    w0 = rs.randn(n).astype(F)
    grads = [(rs.randn(n) * sc).astype(F) for sc in (0.5, 0.01)]     # step 1 is clipped, step 2 is not
    seg_of = np.repeat(np.arange(3), np.diff(off))
    dev_off, dev_wd = T(off), T(wd)
    gscale, clip, lr = 0.5, 10.0, 0.01
    # --- clip factors
    fac = torch.empty(3, device=dev())
    K.grad_clip_factors(T(w0), T(grads[0]), dev_off, dev_wd, gscale, clip, fac)
    want = _np_clip_factors(w0, grads[0], off, wd, gscale, clip)
    assert want.min() < 0.9 and want.max() == 1.0       # segment 1 (4 elements) is below the clip norm
    np.testing.assert_allclose(fac.cpu().numpy(), want, rtol=1e-6)
    # kinds 3 / 4: use_nesterov=True (TF ApplyMomentum) and centered=True (TF ApplyCenteredRMSProp), the keyword
    # arguments training.py:64-81 forwards to the TF constructors
    for kind, p1, p2, eps in ((0, 0.9, 0.0, 0.0), (1, 0.9, 0.999, 1e-8), (2, 0.9, 0.1, 1e-10), (3, 0.9, 0.0, 0.0),
                              (4, 0.9, 0.1, 1e-10)):
        w = w0.copy()
        s1 = np.ones(n, F) if kind in (2, 4) else np.zeros(n, F)
        s2 = np.zeros(n, F)
        s3 = np.zeros(n, F)
        dw, ds1, ds2, ds3 = T(w), T(s1), T(s2), T(s3)
        for t, g in enumerate(grads, 1):
            f = _np_clip_factors(w, g, off, wd, gscale, clip)
            gp = (g * F(gscale) + wd[seg_of] * w) * f[seg_of]
            lr_t = lr
            if kind == 0:
                s1 = F(p1) * s1 + gp
                w = w - F(lr) * s1
            elif kind == 1:
                lr_t = lr * np.sqrt(1 - p2 ** t) / (1 - p1 ** t)
                s1 = F(p1) * s1 + F(1 - p1) * gp
                s2 = F(p2) * s2 + F(1 - p2) * gp * gp
                w = w - F(lr_t) * s1 / (np.sqrt(s2) + F(eps))
            elif kind == 2:
                s1 = F(p1) * s1 + F(1 - p1) * gp * gp
                s2 = F(p2) * s2 + F(lr) * gp / np.sqrt(s1 + F(eps))
                w = w - s2
            elif kind == 3:
                s1 = F(p1) * s1 + gp
                w = w - (gp * F(lr) + s1 * F(p1) * F(lr))
            else:
                s3 = F(p1) * s3 + F(1 - p1) * gp
                s1 = F(p1) * s1 + F(1 - p1) * gp * gp
                s2 = F(p2) * s2 + F(lr) * gp / np.sqrt(s1 - s3 * s3 + F(eps))
                w = w - s2
            K.grad_clip_factors(dw, T(g), dev_off, dev_wd, gscale, clip, fac)
            K.optimizer_step(kind, dw, T(g), ds1, ds2 if kind not in (0, 3) else None, dev_off, dev_wd, fac, lr_t, p1, p2, eps,
                             gscale, slot3=ds3 if kind == 4 else None)
        np.testing.assert_allclose(dw.cpu().numpy(), w, rtol=2e-5, atol=1e-6, err_msg='kind %d' % kind)
        np.testing.assert_allclose(ds1.cpu().numpy(), s1, rtol=2e-5, atol=1e-7, err_msg='kind %d slot1' % kind)
    # the fused default kernel (lmh_sgd_momentum) == kind 0 without clipping
    a, va = T(w0), torch.zeros(n, device=dev())
    b, vb = T(w0), torch.zeros(n, device=dev())
    K.sgd_momentum(a, T(grads[0]), va, dev_off, dev_wd, lr, 0.9, gscale)
    K.optimizer_step(0, b, T(grads[0]), vb, None, dev_off, dev_wd, None, lr, 0.9, 0.0, 0.0, gscale)
    assert torch.equal(a, b) and torch.equal(va, vb)


def test_get_optimizer_builds_every_reference_optimizer(K):
    from luminoth_amd.utils import training as TR
    from luminoth_amd.utils.config import Config

    class Store(object):
        pass

    class Model(object):
        pass
    n = 1024
    for kind, cls in (('momentum', TR.MomentumOptimizer), ('gradient_descent', TR.MomentumOptimizer),
                      ('adam', TR.AdamOptimizer), ('rmsprop', TR.RMSPropOptimizer)):
        m = Model()
        m.store = Store()
        m.store.flat = torch.ones(n, device=dev())
        m.store.grad = torch.full((n,), 0.5, device=dev())
        m.store.mom = torch.zeros(n, device=dev())
        m.store.seg_offset = T(np.array([0, n], np.int64))
        m.store.seg_wd = T(np.array([0.0], F))
        cfg = Config({'learning_rate': {'decay_method': None, 'learning_rate': 0.1}, 'optimizer': {'type': kind},
                      'clip_by_norm': True})
        opt = TR.get_optimizer(cfg, m)
        assert type(opt) is cls and opt.clip_norm == 10.0
        opt.step()
        w = m.store.flat.cpu().numpy()
        clipped = 0.5 * 10.0 / np.sqrt(n * 0.25)                       # ||g|| = 16 > 10
        expect = {'momentum': 1 - 0.1 * clipped, 'gradient_descent': 1 - 0.1 * clipped,
                  'adam': 1 - 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9) * (0.1 * clipped) / (np.sqrt(0.001 * clipped ** 2) + 1e-8),
                  'rmsprop': 1 - 0.1 * clipped / np.sqrt(0.9 + 0.1 * clipped ** 2 + 1e-10)}[kind]
        np.testing.assert_allclose(w, expect, rtol=1e-5, err_msg=kind)


def test_dropout_matches_oracle_mask(K):
    from oracle import rng as orng
    rs = np.random.RandomState(9)
    x = rs.randn(513, 37).astype(F)
    for keep, seed in ((0.5, 1234), (0.9, 7), (1.0, 3)):
        y = K.dropout(T(x), keep, seed).cpu().numpy()
        mask = orng.dropout_mask(x.size, keep, seed).reshape(x.shape)
        np.testing.assert_array_equal(y, np.where(mask, x * F(1.0 / keep), F(0)))
        assert abs(mask.mean() - keep) < 0.02
        dy = rs.randn(*x.shape).astype(F)
        np.testing.assert_array_equal(K.dropout(T(dy), keep, seed).cpu().numpy(), np.where(mask, dy * F(1.0 / keep), F(0)))


# ------------------------------------------------------- deferred tails ----
@pytest.mark.parametrize('shape', [(2, 24, 24, 64, 128, 1), (1, 20, 20, 64, 64, 3), (1, 9, 11, 256, 36, 1)])
def test_deferred_weight_gradient_tails_equal_immediate_path(K, shape, monkeypatch):
    """csrc/tail.hip: queueing the split-K reduction, BatchNorm scaling + dgamma and dbeta of several layers and
    finishing them with lmh_wgrad_tail_batch gives the results of the per-layer launches (same slab order: the weight
    gradient is bit-identical; dgamma / dbeta sum their partial rows in a different tree: 1e-5)."""
    monkeypatch.setattr(K, 'WINOGRAD', False)
    N, H, W, C, Kc, R = shape
    rs = np.random.RandomState(3)
    x = T(rs.randn(N, H, W, C).astype(F))
    y = T(rs.randn(N, H, W, Kc).astype(F))
    dy = T(rs.randn(N, H, W, Kc).astype(F))
    w = T((rs.randn(R, R, C, Kc) * 0.1).astype(F))
    scale = T((1 + 0.1 * rs.randn(Kc)).astype(F))
    mean, rstd = T(rs.randn(Kc).astype(F)), T((1 + 0.1 * rs.rand(Kc)).astype(F))
    d = K.conv_desc(x.shape, w.shape, 1, 1, 'SAME', 'relu')

    def run(deferred):
        dw = torch.zeros_like(w)
        dbeta, dgamma = torch.zeros(Kc, device=dev()), torch.zeros(Kc, device=dev())
        dw2 = torch.zeros_like(w)                           # a second, bias-only layer in the same batch
        dbias2 = torch.zeros(Kc, device=dev())
        if deferred:
            K.TAILS.begin()
        try:
            g = K.act_bwd(dy, y, 'relu', want_g=True, colsum=dbeta, defer='layer_a' if deferred else None)
            K.conv2d_bwd_weight(d, x, g, out=dw, defer='layer_a' if deferred else None)
            g2 = K.act_bwd(dy, None, None, want_g=False, colsum=dbias2, defer='layer_b' if deferred else None)
            assert g2 is None
            K.conv2d_bwd_weight(d, x, dy, out=dw2, defer='layer_b' if deferred else None)
            if deferred:
                K.TAILS.entry('layer_a')['bn'] = dict(w=w, scale=scale, mean=mean, rstd=rstd, dgamma=dgamma)
                assert len(K.TAILS.order) == 2
                K.TAILS.flush()
                assert not K.TAILS.order
            else:
                K.bn_param_grads(w, dw, dbeta, mean, rstd, scale, out=dgamma)
        finally:
            K.TAILS.active = False
        torch.cuda.synchronize()
        return [t.cpu().numpy() for t in (dw, dbeta, dgamma, dw2, dbias2)]

    a, b = run(False), run(True)
    np.testing.assert_array_equal(a[3], b[3])              # plain layer: pure split-K reduction, same order
    for i, name in enumerate(('dw (BN-scaled)', 'dbeta', 'dgamma', 'dw2', 'dbias2')):
        sc = max(1.0, float(np.abs(a[i]).max()))
        np.testing.assert_allclose(b[i], a[i], rtol=1e-5, atol=1e-5 * sc, err_msg=name)


@pytest.mark.parametrize('act', ['elu', 'selu', 'softplus', 'softsign', 'sigmoid', 'tanh', 'leaky_relu', 'relu', 'relu6'])
def test_activation_pass_matches_torch(K, act):
    """lmh_act_fwd / lmh_act_bwd for every activation id: values and the derivative (from the output; from the input for softplus / softsign) against
    torch in float64 (TF 1.x definitions: leaky_relu alpha 0.2, selu's two constants), 2e-6 of the scale; ragged sizes,
    the per-channel sums of g included."""
    import torch.nn.functional as TF
    f = {'elu': TF.elu, 'selu': TF.selu, 'softplus': TF.softplus, 'softsign': TF.softsign, 'sigmoid': torch.sigmoid,
         'tanh': torch.tanh, 'leaky_relu': lambda t: TF.leaky_relu(t, 0.2), 'relu': torch.relu,
         'relu6': lambda t: torch.clamp(t, 0, 6)}[act]
    rs = np.random.RandomState(3)
    for rows, C in ((257, 36), (64, 128), (5, 3)):
        z = (rs.randn(rows, C) * 4).astype(F)
        z[0, :3] = (-30.0, 30.0, 0.0)
        zt = torch.tensor(z, dtype=torch.float64, requires_grad=True)
        ref = f(zt)
        dy = rs.randn(rows, C).astype(F)
        ref.backward(torch.tensor(dy, dtype=torch.float64))
        zd = T(z.copy())
        from_input = act in K.ACT_GRAD_FROM_INPUT          # softplus / softsign: a separate output, z kept for the backward
        y = K.act_fwd(zd, act, out=None if from_input else zd)
        assert (y.data_ptr() == zd.data_ptr()) == (not from_input)
        np.testing.assert_allclose(y.cpu().numpy(), ref.detach().numpy(), rtol=2e-6, atol=2e-6)
        cs = torch.zeros(C, device=y.device)
        g = K.act_bwd(T(dy), zd if from_input else y, act, want_g=True, colsum=cs)
        gref = zt.grad.numpy()
        if act in ('relu', 'relu6', 'leaky_relu', 'elu', 'selu'):       # kinks: compare away from them
            ok = np.abs(z) > 1e-3
            if act == 'relu6':
                ok &= np.abs(z - 6) > 1e-3
        else:
            ok = np.ones_like(z, bool)
        np.testing.assert_allclose(g.cpu().numpy()[ok], gref[ok], rtol=2e-5, atol=2e-6)
        if ok.all():
            np.testing.assert_allclose(cs.cpu().numpy(), gref.sum(0), rtol=1e-4, atol=1e-4)

