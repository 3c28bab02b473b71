"""End-to-end GPU parity: the HIP Faster R-CNN train step (through the model
module API) vs the CPU oracle on identical seeded inputs and identical weights.
Continuous quantities within the north_star tolerance (1e-4 fp32 for box
coords and losses); discrete ones (labels, keep sets) bit-exact on identical
inputs.  Run with `-m gpu`."""
import os

import numpy as np
import pytest
import torch

import sys

from oracle import frcnn as of

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
F = np.float32


from e2e_util import (compare_step_with_oracle, condition_like_pretrained, free_running_agreement, make_config,   # noqa: E402
                      synth)


@pytest.fixture(scope='module')
def setup():
    from luminoth_amd.models import get_model
    cfg = make_config()
    model = condition_like_pretrained(get_model('fasterrcnn')(cfg), 'resnet_v1_50')
    images, gts = synth(2, 320, 384, 4, 80, 3)
    return cfg, model, images, gts


def _winograd_everywhere(monkeypatch, on):
    from luminoth_amd import kernels as KK
    monkeypatch.setattr(KK, 'WINOGRAD', on)
    monkeypatch.setattr(KK, 'WINOGRAD_MIN_CK', 64 * 64)
    monkeypatch.setattr(KK, 'WINOGRAD_WGRAD_MIN_CK', 64 * 64)


@pytest.mark.parametrize('fused', [False, True], ids=['module_api', 'train_step'])
@pytest.mark.parametrize('winograd', [False, True], ids=['direct', 'winograd'])
def test_train_step_matches_oracle(setup, winograd, fused, monkeypatch):
    """Whole step against the oracle, once with the direct 3x3 kernels and once with the Winograd F(2x2,3x3) path
    on every layer it covers (the default routes RPN + block3 through it).  Gradients with pinned ReLU branches:
    every element within 1e-3 of the tensor's scale on BOTH paths (tests/e2e_util.py)."""
    _winograd_everywhere(monkeypatch, winograd)
    cfg, model, images, gts = setup
    # module_api: model(...) -> loss() -> backward() (torch.autograd routes the activation gradients); train_step: the
    # production step (plain kernel calls on three streams, what bench.py and luminoth_amd.train.run issue)
    compare_step_with_oracle(model, images, gts, 80, fused=fused)


# The two arithmetics of the convolutions bench.py can put on its line: the native fp32 MFMA and bf16x3
# (every fp32 operand split exactly into three bf16 pieces, six bf16 MFMA products, fp32 accumulation: csrc/conv_x3.h).
# bf16x3 runs under the SAME bounds as fp32 everywhere below — it is fp32 arithmetic, not a reduced precision.
ARITHMETICS = pytest.mark.parametrize('compute', [None, 'bf16x3'], ids=['f32', 'bf16x3'])


def _arith(compute):
    return {} if compute is None else {'model.base_network.compute_dtype': compute}


def _check_arith(model, compute):
    layers = model.base_network.trunk.all_layers()
    if compute is not None:
        assert all(l.compute == compute for l in layers if hasattr(l, 'k') and l.k > 0) and model._rpn._rpn.compute == compute
        from luminoth_amd import kernels as KK
        assert KK.get_option('wino_m') == 4 and KK.X3_WINOGRAD_MODE == '3'       # F(4x4,3x3) with bf16x3 GEMMs, like fp32


@ARITHMETICS
def test_train_step_matches_oracle_at_benchmark_shape(compute):
    """BASELINE configs[1] itself: ResNet-50, 2 x 1024 x 1024, 80 classes, 8 gt boxes / image, default kernel
    routing (Winograd F(4x4,3x3) on RPN + block2 + block3) — the shape bench.py times, in both arithmetics it reports."""
    from luminoth_amd.models import get_model
    import bench
    cfg = make_config(**_arith(compute))
    model = get_model('fasterrcnn')(cfg)
    _check_arith(model, compute)
    bench.condition_weights(model, 'resnet_v1_50')
    images, (gt, cnt) = bench.synth_batch(2, 1024, 1024, 8, 80, 100, 'cpu')
    gts = [gt[b, :int(cnt[b])].numpy() for b in range(2)]
    compare_step_with_oracle(model, images, gts, 80, fused=True)      # the production step: what bench.py times


@pytest.mark.parametrize('fused', [False, True], ids=['module_api', 'train_step'])
def test_train_batch_norm_matches_oracle(fused):
    """`model.base_network.train_batch_norm: True` (base_network.py:82-93; VERDICT r3 missing #1): every BatchNorm
    normalises with the statistics of the batch and the step advances the moving averages (train.py:87-88).  One image
    (the reference's batch): head outputs, losses and every gradient — including dgamma / dbeta THROUGH the batch
    statistics — against the oracle under the fp32 bounds; the moving statistics after the step against
    v -= (v - batch) * (1 - 0.997) with the unbiased batch variance."""
    from luminoth_amd.models import get_model
    from oracle.model import OracleFasterRCNN
    from parity_log import check_close
    cfg = make_config(**{'model.base_network.train_batch_norm': True})
    model = condition_like_pretrained(get_model('fasterrcnn')(cfg), 'resnet_v1_50')
    images, gts = synth(1, 320, 384, 4, 80, 11)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    stats = {}
    try:
        # gradients THROUGH the batch statistics are projections (dz is orthogonal to 1 and to x-hat over the batch): the
        # weight gradients are small differences of large sums, so fp32 summation order shows at ~1e-3 of a tensor's scale
        # instead of ~1e-4 (the observed values are recorded: profiles/r05_parity_observed.json)
        compare_step_with_oracle(model, images, gts, 80, oracle_kwargs={'train_bn': True}, fused=fused, stats=stats,
                                 grad_tight=2e-3, grad_max=5e-3)
    finally:
        print('train_batch_norm step vs oracle: observed %s' % {k: '%.2e' % v for k, v in stats.items()})
    sd1 = model.state_dict()
    oracle = OracleFasterRCNN(sd0, arch='resnet_v1_50', num_classes=80, seed=0, train_bn=True)
    with torch.no_grad():
        oracle.backbone(images[0:1])
    want = oracle.moving_statistics_after_step()
    assert len(want) == 2 * (1 + 3 * 3 + 1 + 4 * 3 + 1 + 6 * 3 + 1)          # conv1 + block1..3 (with their shortcuts)
    for name, ref in want.items():
        assert not torch.equal(sd1[name], sd0[name]), name                   # the step moved it
        check_close('train_batch_norm/moving_statistics', sd1[name].numpy(), ref.numpy(), rtol=1e-5,
                    atol=1e-6 * max(1.0, float(ref.abs().max())))
    # back to inference: the frozen-statistics path folds the NEW moving averages (BNTable reload)
    pred = model(images, is_training=False)
    assert torch.isfinite(pred['rpn_prediction']['rpn_cls_score']).all()
    assert not any(l.bn_train for l in model.base_network.trunk.all_layers())


@pytest.mark.parametrize('fused', [False, True], ids=['module_api', 'train_step'])
def test_resnet_v2_matches_oracle(fused):
    """`architecture: resnet_v2_50` (base_network.py:18-27,94-101; VERDICT r3 missing #2): pre-activation bottlenecks
    (`preact` BatchNorm + ReLU ahead of the convolutions, biased shortcut / conv3 without BatchNorm, un-activated unit
    outputs), BatchNorm in training mode while training (the reference hands `is_training` to slim for v2).  One image:
    the training step against the oracle (outputs, losses, every gradient incl. preact gamma / beta and the biases), then
    the inference forward (moving statistics, just advanced by the step) against the oracle's inference forward."""
    from luminoth_amd.models import get_model
    from oracle.model import OracleFasterRCNN
    from parity_log import check_close
    cfg = make_config('resnet_v2_50', 20)
    model = condition_like_pretrained(get_model('fasterrcnn')(cfg), 'resnet_v2_50')
    names = model.get_trainable_vars()
    assert 'truncated_base_network/resnet_v2_50/block2/unit_1/bottleneck_v2/preact/gamma' in names
    assert 'truncated_base_network/resnet_v2_50/block3/unit_6/bottleneck_v2/conv3/biases' in names
    assert 'truncated_base_network/resnet_v2_50/block1/unit_3/bottleneck_v2/conv3/biases' not in names      # fine_tune_from block2
    assert 'truncated_base_network/resnet_v2_50/postnorm/gamma' in model.store.specs                          # exists, unused
    images, gts = synth(1, 256, 320, 3, 20, 17)
    stats = {}
    try:
        compare_step_with_oracle(model, images, gts, 20, arch='resnet_v2_50', oracle_kwargs={'train_bn': True}, fused=fused,
                                 stats=stats, grad_tight=2e-3, grad_max=5e-3, min_checked=80, grad_floor_rel=1e-3)
    finally:
        print('resnet_v2_50 step vs oracle: observed %s' % {k: '%.2e' % v for k, v in stats.items()})
    # inference: frozen (moving) statistics, folded scale / shift for the convolutions, lmh_bn_apply for the preacts
    pred = model(images, is_training=False)
    oracle = OracleFasterRCNN(model.state_dict(), arch='resnet_v2_50', num_classes=20, seed=0)
    with torch.no_grad():
        cls, box = oracle.rpn_head(oracle.backbone(images[0:1]))
    got = pred['rpn_prediction']['rpn_cls_score'].cpu().numpy().reshape(-1, 2)
    check_close('resnet_v2_50/inference/rpn_cls_score', got, cls.numpy().reshape(-1, 2), rtol=1e-3,
                atol=1e-4 * max(1.0, float(cls.abs().max())))


@ARITHMETICS
def test_free_running_agreement_at_benchmark_shape(compute):
    """the step comparison is teacher-forced stage by stage.  Here the oracle runs FREE on its own
    upstream outputs at the benchmark shape; reported (printed) and bounded: the proposal lists and the sampled ROI sets
    must coincide almost everywhere (near-ties of scores at the last bit may reorder a few), every loss within 1e-4."""
    from luminoth_amd.models import get_model
    import bench
    cfg = make_config(**_arith(compute))
    model = get_model('fasterrcnn')(cfg)
    _check_arith(model, compute)
    bench.condition_weights(model, 'resnet_v1_50')
    images, (gt, cnt) = bench.synth_batch(2, 1024, 1024, 8, 80, 100, 'cpu')
    gts = [gt[b, :int(cnt[b])].numpy() for b in range(2)]
    rep = free_running_agreement(model, images, gts, 80)
    print('free-running agreement @ 2x1024^2: proposals at the same rank %s, as sets %s, sampled ROI sets %s, losses %s'
          % (rep['same_rank'], rep['same_set'], rep['roi_set'], rep['losses']))
    from parity_log import note as note_
    note = lambda k, *a: note_(k if compute is None else k.replace('free_running@', 'free_running[%s]@' % compute), *a)
    note('free_running@2x1024x1024/proposals_not_at_same_rank', 1.0 - min(rep['same_rank']))
    note('free_running@2x1024x1024/proposal_set_mismatch', 1.0 - min(rep['same_set']), 0.02)
    note('free_running@2x1024x1024/sampled_roi_set_mismatch', 1.0 - min(rep['roi_set']), 0.05)
    assert min(rep['same_set']) >= 0.98 and min(rep['roi_set']) >= 0.95
    for k, (got, ref) in rep['losses'].items():
        # north_star's 1e-4 on every loss of the free run (round 3 allowed 1e-3).  RPN losses see no discrete decision of
        # the free run (anchor targets depend on gt only: observed 6e-8); RCNN losses are means over the SAMPLED ROI set —
        # a proposal pair within an ulp of the NMS threshold, or two scores equal to the last bit, swaps a few of the 256
        # ROIs of an image (observed: 3 % of the sampled set, 1.7 % of the proposal set) and each swapped ROI moves the
        # mean by |loss_i - loss_j| / 256: observed 2.1e-5 (profiles/r05_parity_observed.json)
        tol = 1e-4
        note('free_running@2x1024x1024/loss:' + k, abs(got - ref) / max(1.0, abs(ref)), tol)
        assert abs(got - ref) <= tol * max(1.0, abs(ref)), (k, got, ref)


def test_resnet101_tail_at_config4_size():
    """BASELINE configs[3] per-GPU share at its own size: ResNet-101, 2 x 1024 x 1024, 80 classes, RCNN minibatch 256
    (12 544 GEMM rows through the block4 tail per image).  Losses, head outputs and integer stages only — the
    small-shape test below covers every gradient of the same code path; here the oracle's backward alone would run
    minutes on the host."""
    from luminoth_amd.models import get_model
    import bench
    cfg = make_config('resnet_v1_101', 80)
    model = get_model('fasterrcnn')(cfg)
    bench.condition_weights(model, 'resnet_v1_101')
    assert model._rcnn._rcnn_target._minibatch_size == 256 and model.base_network.tail is not None
    images, (gt, cnt) = bench.synth_batch(2, 1024, 1024, 8, 80, 100, 'cpu')
    gts = [gt[b, :int(cnt[b])].numpy() for b in range(2)]
    compare_step_with_oracle(model, images, gts, 80, arch='resnet_v1_101', check_grads=False)


@pytest.mark.parametrize('winograd', [False, True], ids=['direct', 'winograd'])
def test_resnet101_block4_tail_matches_oracle(winograd, monkeypatch):
    """A12 (truncated_base_network.py:56-95): ResNet-101, block4 applied to the pooled ROIs with the trunk's block4
    variables; 23-unit block3.  RCNN minibatch 64 keeps the CPU oracle's tail (12.5k GEMM rows at 256) short."""
    from luminoth_amd.models import get_model
    _winograd_everywhere(monkeypatch, winograd)
    cfg = make_config('resnet_v1_101', 20, **{'model.rcnn.target.minibatch_size': 64})
    model = condition_like_pretrained(get_model('fasterrcnn')(cfg), 'resnet_v1_101')
    assert model.base_network.tail is not None and model.base_network.tail_channels == 2048
    images, gts = synth(2, 256, 320, 3, 20, 5)
    compare_step_with_oracle(model, images, gts, 20, arch='resnet_v1_101', fused=winograd)   # direct: module API; winograd: train_step
    # the tail's variables are trained (use_tail, not freeze_tail): their gradients were part of the check
    g = model.store.grads['truncated_base_network/resnet_v1_101/block4/unit_3/bottleneck_v1/conv2/weights']
    assert float(g.abs().max()) > 0


@pytest.mark.parametrize('winograd', [False, True], ids=['direct', 'winograd'])
@pytest.mark.parametrize('hw', [(600, 800), (600, 900)], ids=['600x800', '600x900'])
def test_vgg16_fasterrcnn_matches_oracle(hw, winograd, monkeypatch):
    """BASELINE configs[0]: Faster R-CNN VGG-16 on Pascal-VOC-shape images (375x500 / 333x500 resized by the
    600/1024 rule -> 600x800 / 600x900), 20 classes, batch 1 like the reference; conv3..conv5 trainable.
    Direct and Winograd (default routing sends conv3-conv5 + RPN through Winograd) both within 1e-4."""
    from luminoth_amd.models import get_model
    from luminoth_amd import kernels as KK
    monkeypatch.setattr(KK, 'WINOGRAD', winograd)
    cfg = make_config('vgg_16', 20, **{'model.base_network.fine_tune_from': 'conv3'})
    model = condition_like_pretrained(get_model('fasterrcnn')(cfg), 'vgg_16')
    images, gts = synth(1, hw[0], hw[1], 3, 20, 7)
    compare_step_with_oracle(model, images, gts, 20, arch='vgg_16', oracle_kwargs={'fine_tune_from': 'conv3'},
                             min_checked=20, fused=winograd)        # direct: module API; winograd: train_step
    frozen = 'truncated_base_network/vgg_16/conv2/conv2_2/weights'
    assert frozen not in model.get_trainable_vars()


@ARITHMETICS
def test_resnet101_free_running_agreement(compute):
    """VERDICT r4 weak #1b, third architecture: ResNet-101 WITH the block4 tail on the pooled ROIs (BASELINE configs[3]'s
    model; one 384 x 512 image, RCNN minibatch 64 so that the CPU oracle's tail stays short), the oracle running FREE on its
    own proposals / sampled ROIs.  Same bounds as the ResNet-50 and VGG-16 runs."""
    from luminoth_amd.models import get_model
    from parity_log import note as note_
    note = lambda k, *a: note_(k if compute is None else k.replace('_free_running@', '_free_running[%s]@' % compute), *a)
    cfg = make_config('resnet_v1_101', 20, **dict({'model.rcnn.target.minibatch_size': 64}, **_arith(compute)))
    model = condition_like_pretrained(get_model('fasterrcnn')(cfg), 'resnet_v1_101')
    _check_arith(model, compute)
    images, gts = synth(1, 384, 512, 3, 20, 11)
    rep = free_running_agreement(model, images, gts, 20, arch='resnet_v1_101', oracle_kwargs={'rcnn': {'minibatch_size': 64}})
    print('ResNet-101 free-running agreement @ 1x384x512: proposals at the same rank %s, as sets %s, sampled ROI sets %s, '
          'losses %s' % (rep['same_rank'], rep['same_set'], rep['roi_set'], rep['losses']))
    note('resnet101_free_running@1x384x512/proposal_set_mismatch', 1.0 - min(rep['same_set']), 0.02)
    note('resnet101_free_running@1x384x512/sampled_roi_set_mismatch', 1.0 - min(rep['roi_set']), 0.05)
    assert min(rep['same_set']) >= 0.98 and min(rep['roi_set']) >= 0.95
    for k, (got, ref) in rep['losses'].items():
        note('resnet101_free_running@1x384x512/loss:' + k, abs(got - ref) / max(1.0, abs(ref)), 1e-4)
        assert abs(got - ref) <= 1e-4 * max(1.0, abs(ref)), (k, got, ref)


@ARITHMETICS
def test_vgg16_free_running_agreement_at_config1_shape(compute):
    """VERDICT r4 weak #1b: BASELINE configs[0] (Faster R-CNN VGG-16, one Pascal-VOC-shape image, 20 classes) with the
    oracle running FREE on its own upstream outputs — its probabilities, its NMS, its sampled ROIs — instead of the
    kernels'.  Bounded like the ResNet-50 run at the benchmark shape: proposal sets / sampled ROI sets coincide almost
    everywhere, every loss within north_star's 1e-4."""
    from luminoth_amd.models import get_model
    from parity_log import note as note_
    note = lambda k, *a: note_(k if compute is None else k.replace('_free_running@', '_free_running[%s]@' % compute), *a)
    cfg = make_config('vgg_16', 20, **dict({'model.base_network.fine_tune_from': 'conv3'}, **_arith(compute)))
    model = condition_like_pretrained(get_model('fasterrcnn')(cfg), 'vgg_16')
    images, gts = synth(1, 600, 800, 3, 20, 7)
    rep = free_running_agreement(model, images, gts, 20, arch='vgg_16', oracle_kwargs={'fine_tune_from': 'conv3'})
    print('VGG-16 free-running agreement @ 1x600x800: proposals at the same rank %s, as sets %s, sampled ROI sets %s, '
          'losses %s' % (rep['same_rank'], rep['same_set'], rep['roi_set'], rep['losses']))
    note('vgg16_free_running@1x600x800/proposal_set_mismatch', 1.0 - min(rep['same_set']), 0.02)
    note('vgg16_free_running@1x600x800/sampled_roi_set_mismatch', 1.0 - min(rep['roi_set']), 0.05)
    assert min(rep['same_set']) >= 0.98 and min(rep['roi_set']) >= 0.95
    for k, (got, ref) in rep['losses'].items():
        note('vgg16_free_running@1x600x800/loss:' + k, abs(got - ref) / max(1.0, abs(ref)), 1e-4)
        assert abs(got - ref) <= 1e-4 * max(1.0, abs(ref)), (k, got, ref)


def test_inference_prediction_dict_keys(setup):
    cfg, model, images, gts = setup
    pred = model(images[0], is_training=False)
    rp, cp = pred['rpn_prediction'], pred['classification_prediction']
    assert set(['rpn_cls_prob', 'rpn_cls_score', 'rpn_bbox_pred', 'proposals', 'scores']) <= set(rp)
    assert set(['objects', 'labels', 'probs', 'rcnn']) <= set(cp)
    n = rp['proposals'].shape[0]
    assert rp['proposals'].shape == (n, 4) and rp['scores'].shape == (n,)
    assert cp['rcnn']['cls_prob'].shape == (n, 81) and cp['rcnn']['bbox_offsets'].shape == (n, 320)
    d = cp['objects'].shape[0]
    assert cp['objects'].shape == (d, 4) and cp['labels'].shape == (d,) and d <= 300
    # detections vs the oracle's RCNNProposal on the kernel's own head outputs
    probs = torch.softmax(cp['rcnn']['cls_score'].cpu(), dim=1).numpy()
    np.testing.assert_allclose(cp['rcnn']['cls_prob'].cpu().numpy(), probs, rtol=1e-5, atol=1e-7)


def test_rcnn_proposal_kernel_vs_oracle():
    from luminoth_amd import kernels as K
    rs = np.random.RandomState(3)
    B, R, C = 2, 300, 20
    props = np.zeros((B, R, 4), F)
    cnt = np.array([300, 123], np.int32)
    for b in range(B):
        wh = rs.randint(16, 200, size=(R, 2))
        xy = rs.randint(0, 400, size=(R, 2))
        props[b] = np.concatenate([xy, xy + wh], 1)
    pred = (rs.randn(B, R, 4 * C) * 0.5).astype(F)
    pred[..., 2::4] = 0
    pred[..., 3::4] = 0           # exp(0): exact chain
    logits = rs.randn(B, R, C + 1).astype(F) * 3
    prob = torch.softmax(torch.tensor(logits), dim=2).numpy()
    dev = torch.device('cuda:0')
    o, l, p, n = K.rcnn_proposal(torch.tensor(props).to(dev), torch.tensor(cnt).to(dev), torch.tensor(pred).to(dev),
                                 torch.tensor(prob).to(dev), (480, 640), C, class_max_detections=10,
                                 class_nms_threshold=0.5, total_max_detections=50, min_prob_threshold=0.2)
    o, l, p, n = o.cpu().numpy(), l.cpu().numpy(), p.cpu().numpy(), n.cpu().numpy()
    for b in range(B):
        r = of.rcnn_proposal(props[b, :cnt[b]], pred[b, :cnt[b]], prob[b, :cnt[b]], (480, 640), C,
                             class_max_detections=10, class_nms_threshold=0.5, total_max_detections=50,
                             min_prob_threshold=0.2)
        assert n[b] == r['objects'].shape[0]
        np.testing.assert_array_equal(o[b, :n[b]], r['objects'])
        np.testing.assert_array_equal(l[b, :n[b]], r['proposal_label'])
        np.testing.assert_array_equal(p[b, :n[b]], r['proposal_label_prob'])
        assert (l[b, n[b]:] == -1).all()


def test_optimizer_step_changes_weights_like_oracle(setup):
    from luminoth_amd.utils.training import get_optimizer
    cfg, model, images, gts = setup
    before = model.state_dict()
    model._step = 0
    opt = get_optimizer(cfg.train, model)
    pred = model(images, gts, is_training=True)
    total = model.loss(pred)
    model.backward(total)
    g = {n: t.clone() for n, t in model.store.grads.items()}
    opt.step()
    after = model.state_dict()
    wd = {n: sp.wd for n, sp in model.store.specs.items()}
    for n in ('fasterrcnn/rpn/conv/w', 'fasterrcnn/rcnn/fc_bbox/b',
              'truncated_base_network/resnet_v1_50/block3/unit_2/bottleneck_v1/conv2/weights',
              'truncated_base_network/resnet_v1_50/block2/unit_1/bottleneck_v1/conv1/BatchNorm/gamma'):
        exp = before[n] - 3e-4 * (g[n].cpu() + wd[n] * before[n])      # first step: v = g'
        np.testing.assert_allclose(after[n].numpy(), exp.numpy(), rtol=1e-6, atol=1e-8, err_msg=n)
    frozen = 'truncated_base_network/resnet_v1_50/block1/unit_1/bottleneck_v1/conv1/weights'
    np.testing.assert_array_equal(after[frozen].numpy(), before[frozen].numpy())
    model.load_state_dict(before)


def test_fused_two_stream_step_equals_plain_step(setup):
    """FasterRCNN.train_step (two-stream schedule) == __call__ + loss + backward: same losses, same
    targets, same gradients (ROI-pool backward adds in LDS-atomic order: 1e-5 of the gradient scale)."""
    cfg, model, images, gts = setup
    model._step = 0
    pred = model(images, gts, is_training=True)
    losses = model.loss(pred, return_all=True)
    model.backward(losses['total_loss'])
    torch.cuda.synchronize()
    g_plain = model.store.grad.clone()
    model._step = 0
    total, pred2 = model.train_step(images, gts)
    torch.cuda.synchronize()
    g_fused = model.store.grad.clone()
    np.testing.assert_allclose(float(total), float(losses['total_loss']), rtol=1e-6)
    for k in ('rpn_cls_loss', 'rpn_reg_loss', 'rcnn_cls_loss', 'rcnn_reg_loss'):
        np.testing.assert_allclose(float(model._last_losses[k]), float(losses[k]), rtol=1e-6)
    a, b = pred['rpn_prediction'], pred2['rpn_prediction']
    for k in ('rpn_cls_target', 'rpn_bbox_target', 'proposals', 'scores', 'num_proposals'):
        assert torch.equal(a[k], b[k]), k
    ca, cb = pred['classification_prediction'], pred2['classification_prediction']
    assert torch.equal(ca['target']['cls'], cb['target']['cls'])
    assert torch.equal(ca['proposals'], cb['proposals'])
    scale = float(g_plain.abs().max())
    np.testing.assert_allclose(g_fused.cpu().numpy(), g_plain.cpu().numpy(), rtol=1e-4, atol=1e-5 * scale)


def test_gradient_buckets_reduce_every_element_exactly_once(setup):
    """Data-parallel overlap (utils/training.py GradientBuckets) on one GPU: a stand-in reduce that doubles its
    range, issued with the production stream protocol, must leave grad == 2 x the plain step's gradient — every
    element handed over exactly once and never before the kernels that write it have finished."""
    from luminoth_amd.utils import training as T
    cfg, model, images, gts = setup
    model._step = 0
    model.train_step(images, gts)
    torch.cuda.synchronize()
    ref = model.store.grad.clone()
    calls = []

    def doubling(t):
        calls.append(int(t.numel()))
        t.mul_(2.0)

    buckets = T.GradientBuckets(model.store, reduce_fn=doubling, bucket_bytes=4 << 20)
    T.install_buckets(buckets)
    try:
        for rep in range(3):                       # repeated: plans are cached, per-step state is reset
            model._step = 0
            del calls[:]
            model.train_step(images, gts)
            early = len(calls)
            buckets.finish()
            torch.cuda.synchronize()
            assert early >= 3 and len(calls) > early, (early, len(calls))     # heads + trunk runs, then the rest
            assert sum(calls) == model.store.grad.numel()
            got = model.store.grad
            scale = float(ref.abs().max())
            np.testing.assert_allclose(got.cpu().numpy(), 2.0 * ref.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale)
        # a path that never arms the buckets (plain call sequence): finish() is the single whole-buffer reduce
        model._step = 0
        del calls[:]
        pred = model(images, gts, is_training=True)
        model.backward(model.loss(pred))
        buckets.finish()
        torch.cuda.synchronize()
        assert calls == [model.store.grad.numel()]
        np.testing.assert_allclose(model.store.grad.cpu().numpy(), 2.0 * ref.cpu().numpy(), rtol=1e-4,
                                   atol=2e-5 * float(ref.abs().max()))
    finally:
        T.install_buckets(None)


def test_rccl_single_rank_bucketed_step(setup, tmp_path):
    """The real RCCL path on one rank (world_size 1, forced): process group init, async all_reduce of the buckets
    from the autograd thread on the communication stream, finish() + fused update — identical weights to the
    un-bucketed optimizer step."""
    import torch.distributed as dist
    from luminoth_amd.utils import training as T
    cfg, model, images, gts = setup
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    opt = T.get_optimizer(cfg.train, model)
    assert opt.buckets is None
    model._step = 0
    model.store.mom.zero_()
    T.train_step(model, opt, images, gts)
    torch.cuda.synchronize()
    want = model.store.flat.clone()
    model.load_state_dict(sd0)
    model.store.mom.zero_()
    os.environ['LUMINOTH_AMD_FORCE_BUCKETS'] = '1'
    dist.init_process_group('nccl', init_method='file://%s' % (tmp_path / 'rdzv'), world_size=1, rank=0)
    try:
        opt2 = T.get_optimizer(cfg.train, model)
        assert opt2.buckets is not None and T.ACTIVE_BUCKETS is opt2.buckets
        model._step = 0
        T.train_step(model, opt2, images, gts)
        torch.cuda.synchronize()
        scale = float(want.abs().max())
        np.testing.assert_allclose(model.store.flat.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-7 * scale)
    finally:
        T.install_buckets(None)
        os.environ.pop('LUMINOTH_AMD_FORCE_BUCKETS', None)
        dist.destroy_process_group()
        model.load_state_dict(sd0)


def test_next_image_prefetch_equals_plain_steps(setup):
    """train_step(next_image=...) computes the frozen trunk prefix of the following batch ahead of time: three steps over
    alternating batches end in bit-identical weights and losses with and without the look-ahead, the cached prefix is
    dropped when the next call gets another tensor, and an in-place change of the announced batch invalidates it."""
    from luminoth_amd.utils import training as T
    cfg, model, images, gts = setup
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    opt = T.get_optimizer(cfg.train, model)
    a = images.to(model.device)
    b = torch.flip(a, dims=[2]).contiguous()
    seq = [a, b, a]

    def reset():
        model.load_state_dict(sd0)
        model.store.mom.zero_()
        model._step = 0
        opt.global_step = 0

    try:
        reset()
        plain = [float(T.train_step(model, opt, x, gts)[0]) for x in seq]
        want = model.store.flat.clone()
        reset()
        ahead = []
        for i, x in enumerate(seq):
            nxt = seq[i + 1] if i + 1 < len(seq) else a
            ahead.append(float(T.train_step(model, opt, x, gts, next_image=nxt, next_gt=gts)[0]))
            st = [S for S in model._step_state.values() if S['pf'] is not None]
            assert len(st) == 1 and st[0]['pf']['image'] is nxt and st[0]['pf']['gt'] is gts     # next step's prefix + anchor targets
        assert ahead == plain
        assert torch.equal(model.store.flat, want)
        # announced `a` with one gt list, but `b` arrives with another list object (same values): nothing stale is used
        reset()
        T.train_step(model, opt, a, gts, next_image=a, next_gt=gts)
        l_b = float(T.train_step(model, opt, b, [g.copy() for g in gts])[0])
        assert all(S['pf'] is None for S in model._step_state.values())
        assert l_b == plain[1]
        # announced tensor modified in place afterwards
        reset()
        c = a.clone()
        T.train_step(model, opt, a, gts, next_image=c)
        c.copy_(b)
        assert float(T.train_step(model, opt, c, gts)[0]) == plain[1]
    finally:
        model.load_state_dict(sd0)
        model.store.mom.zero_()
        model._step_state = {}


def test_non_default_config_surface_trains(monkeypatch):
    """Keys the reference accepts and round 1 rejected with NotImplementedError: RCNN FC stack + dropout
    (rcnn.py:196-218), train.clip_by_norm, the Adam optimizer.  Two steps must run, stay finite and move the weights;
    with keep_prob 1.0 the dropout layers are exact identities (same loss as the default model)."""
    from luminoth_amd.models import get_model
    from luminoth_amd.utils import training as T
    from luminoth_amd.utils.config import get_config
    cfg = get_config({'model': {'type': 'fasterrcnn', 'network': {'num_classes': 20},
                                'base_network': {'architecture': 'resnet_v1_50'},
                                'rcnn': {'dropout_keep_prob': 0.5, 'layer_sizes': [256]}},
                      'train': {'seed': 0, 'clip_by_norm': True, 'optimizer': {'type': 'adam'}}})
    model = condition_like_pretrained(get_model('fasterrcnn')(cfg), 'resnet_v1_50')
    images, gts = synth(2, 256, 320, 3, 20, 13)
    opt = T.get_optimizer(cfg.train, model)
    assert isinstance(opt, T.AdamOptimizer) and opt.clip_norm == 10.0
    w0 = model.store.flat.clone()
    l1, _ = T.train_step(model, opt, images, gts)
    l2, _ = T.train_step(model, opt, images, gts)
    assert np.isfinite(float(l1)) and np.isfinite(float(l2))
    moved = (model.store.flat - w0).abs()
    assert float(moved.max()) > 1e-5 and float(moved.max()) < 1e-2      # Adam: |step| ~ lr per element
    assert model._rcnn._dropout_calls == 4                               # after pooling + after fc_0, two steps


@pytest.mark.parametrize('fused', [False, True], ids=['module_api', 'train_step'])
@pytest.mark.parametrize('act', ['elu', 'selu', 'softplus', 'softsign', 'sigmoid', 'tanh', 'leaky_relu', 'relu'])
def test_rpn_activation_function_of_tf_nn(act, fused):
    """`model.rpn.activation_function` is any tf.nn.<name> in the reference (luminoth/utils/vars.py:80-88 -> rpn.py:57-59;
    the default is relu6).  The activations no convolution epilogue fuses run as an in-place pass behind the RPN
    convolution and are differentiated from their output (csrc/elementwise.hip): whole step against the oracle under the
    fp32 bounds, through both paths."""
    from luminoth_amd.models import get_model
    cfg = make_config(**{'model.rpn.activation_function': act})
    model = condition_like_pretrained(get_model('fasterrcnn')(cfg), 'resnet_v1_50')
    assert model._rpn._rpn.act == act
    # pre-activations of O(1): on this unnormalised feature map the default initializer gives pre-activations far above 1, where every one of
    # these functions is saturated or linear (nothing of the activation would be tested), softsign's peaked derivative
    # turns the 1e-5-of-scale error of the Winograd forward into 1.5e-3 of the weight gradient, and an unbounded activation
    # (the reference's default relu6 is bounded) feeds box deltas whose exp() cancels in the decoded corners at 1e-4 px
    sd = model.state_dict()
    sd['fasterrcnn/rpn/conv/w'].mul_(0.02)
    model.load_state_dict(sd)
    images, gts = synth(2, 256, 320, 3, 80, 5)
    compare_step_with_oracle(model, images, gts, 80, fused=fused, oracle_kwargs={'rpn': {'activation_function': act}})


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """`bench.py --gpus 2` end to end on THIS box every round — the self-spawn through
    torch.distributed.run, rank-sharded synthetic inputs and seeds, the bucketed gradient exchange under the trunk
    backward (on device tensors, through the production stream protocol), the collectives of the profiling steps and
    the max-over-ranks timing — with the gloo backend so that both ranks may share the one GPU (RCCL needs a GPU per
    rank; the driver's 8-GPU run is the first RCCL world > 1)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LUMINOTH_AMD_DIST_BACKEND='gloo')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    # 6 + 6 steps: eager, recorded and REPLAYED steps of the launch plan, with the bucket exchange as host work between
    # the parts of the plan (luminoth_amd/plan.py host_call)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '6',
                          '--no-cpu-baseline', '--no-other-configs'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 4 and d['config']['parallelism'] == 'dp2'
    assert d['dist']['world_size'] == 2 and d['dist']['backend'] == 'gloo'
    assert d['dist']['replicas_identical_after_timed_steps'] is True
    # the collective was measured before the first step (all-reduce of the flat gradient and of one 12 MB bucket) and the
    # bucket size of the exchange follows from it
    pr = d['dist']['allreduce_probe']
    assert pr['ranks_seen'] == 2 and {m['what'] for m in pr['messages']} == {'flat_gradient', 'bucket'}
    assert all(m['correct'] and m['ms'] > 0 for m in pr['messages'])
    assert d['dist']['buckets']['bucket_mb'] == pr['bucket_bytes_from_probe'] / float(1 << 20)
    assert len(d['dist']['buckets']['early_ranges_mb'][0]) >= 2
    assert d['config']['launch_plan']['enabled'] and d['config']['launch_plan']['kernel_launches_per_step'] > 150
    assert np.isfinite(d['config']['final_total_loss']) and d['value'] > 0
    assert d['roofline'] is not None and d['roofline']['whole_step']['executed_flops'] > 0


def test_bench_two_ranks_fallback_ladder():
    """first-contact safety of `bench.py --gpus N`.  The bucketed exchange is made to raise on its first
    early bucket (both ranks, inside the step being recorded into a launch plan); the timed block is re-run with ONE
    all-reduce after the backward, the line says which mode produced the number and what failed before it, and the
    replicas end bit-identical."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LUMINOTH_AMD_DIST_BACKEND='gloo')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'LUMINOTH_AMD_BUCKETED_ALLREDUCE', 'LUMINOTH_AMD_PLAN'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '4',
                          '--no-cpu-baseline', '--no-other-configs', '--no-roofline', '--inject-bucket-failure'],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['dist']['world_size'] == 2
    assert len(d['dist']['fallbacks']) == 1 and 'injected failure' in d['dist']['fallbacks'][0]['error']
    assert d['dist']['fallbacks'][0]['bucketed_allreduce_under_backward'] is True
    assert d['dist']['mode'] == {'bucketed_allreduce_under_backward': False, 'launch_plan': True}
    assert d['dist']['buckets'] is None
    assert d['dist']['replicas_identical_after_timed_steps'] is True
    assert d['dist']['GPU_MAX_HW_QUEUES'] == '8' and d['dist']['collective_timeout_s'] == 180
    assert np.isfinite(d['config']['final_total_loss']) and d['value'] > 0


def test_bench_two_ranks_over_rccl():
    """the same two-rank run over RCCL (backend `nccl`) whenever the box has two GPUs — the driver's
    8-GPU node exercises the real collective layer (communication stream, RCCL's own stream, the launch-plan cuts)
    inside `-m gpu` before its scaling bench; one-GPU boxes skip."""
    import json
    import subprocess
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (RCCL wants one device per rank)')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'LUMINOTH_AMD_DIST_BACKEND'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '6',
                          '--no-cpu-baseline', '--no-other-configs'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['dist']['world_size'] == 2 and d['dist']['backend'] == 'nccl'
    assert d['dist']['replicas_identical_after_timed_steps'] is True
    assert np.isfinite(d['config']['final_total_loss']) and d['value'] > 0
