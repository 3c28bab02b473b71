"""Shared body of the end-to-end GPU parity tests: one HIP Faster R-CNN train step (through the model-module API)
against the CPU oracle on identical seeded inputs and identical weights, for any backbone.

Tolerances (north_star): losses and box coordinates within 1e-4 (fp32), labels / keep sets bit-exact on identical
inputs, head outputs within 1e-4 of their scale.  Gradients are compared with the oracle's ReLU / ReLU6 decisions
PINNED to the kernels' own activations (oracle/model.py `masks`): both sides then differentiate the same
piecewise-linear branch and every element must agree to 1e-3 of the tensor's gradient scale (99.5 % to 2e-4).
"""
import numpy as np
import torch

from oracle import boxes as obx
from oracle import frcnn as of
from oracle import rng as orng
from oracle.model import OracleFasterRCNN

F = np.float32


def make_config(arch='resnet_v1_50', num_classes=80, **over):
    from luminoth_amd.utils.config import get_config
    cfg = {'model': {'type': 'fasterrcnn', 'network': {'num_classes': num_classes},
                     'base_network': {'architecture': arch}},
           'train': {'seed': 0}}
    return get_config(cfg, ['%s=%s' % kv for kv in over.items()])


def synth(B, H, W, G, num_classes, seed):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand((B, H, W, 3), generator=g) * 255.0
    rs = np.random.RandomState(seed)
    gts = []
    for b in range(B):
        wh = rs.randint(min(32, H // 4), max(H // 2, 40), size=(G, 2))
        xy = np.stack([rs.randint(0, W - wh[:, 0]), rs.randint(0, H - wh[:, 1])], 1)
        gts.append(np.concatenate([xy, xy + wh, rs.randint(0, num_classes, size=(G, 1))], 1).astype(F))
    return images, gts


def condition_like_pretrained(model, arch):
    """Random-init stand-in for pretrained statistics.  Without them raw 0..255 pixels drive the activations to
    O(1e2..1e3) and fp32 round-off alone is ~1e-3 absolute on the RPN logits (identically for the CPU oracle), so a
    1e-4 comparison would measure the conditioning of the synthetic weights, not the kernels.  ResNet: conv1's frozen
    BatchNorm variance = pixel variance x fan-in gain; VGG (no BN): conv1_1 weights / pixel std."""
    sd = model.state_dict()
    if arch.startswith('resnet_v2'):     # the root convolution has no BatchNorm: scale its weights by the pixel std instead
        sd['truncated_base_network/%s/conv1/weights' % arch].mul_(1.0 / 73.6)
    elif arch.startswith('resnet'):
        sd['truncated_base_network/%s/conv1/BatchNorm/moving_variance' % arch].fill_(73.6 ** 2 * 2)
        if arch != 'resnet_v1_50':      # 33 residual adds: keep the deep trunk O(1) (bench.py does the same)
            for k in sd:
                if k.endswith('conv3/BatchNorm/moving_variance'):
                    sd[k].fill_(16.0)
    else:
        sd['truncated_base_network/%s/conv1/conv1_1/weights' % arch].mul_(1.0 / 73.6)
    model.load_state_dict(sd)
    return model


def _stat(stats, name, got, ref):
    if stats is not None:
        stats[name] = max(stats.get(name, 0.0), float(np.abs(got - ref).max() / max(1.0, np.abs(got).max())))


def _log_stats(model, images, arch, fused, oracle_kwargs, stats):
    """The errors this comparison observed -> tests/parity_log.py (profiles/r06_parity_observed.json)."""
    from parity_log import note
    bn = model.base_network
    tag = 'e2e/%s/%dx%dx%d/%s%s%s' % (arch, images.shape[0], images.shape[1], images.shape[2],
                                       'train_step' if fused else 'module_api',
                                       '/storage_' + bn.storage_dtype if getattr(bn, 'storage_dtype', None) else '',
                                       '/compute_' + str(bn.compute_dtype) if getattr(bn, 'compute_dtype', None) else '')
    from luminoth_amd import kernels as KK
    tag += '/winograd' if KK.WINOGRAD else '/direct'
    for k, v in stats.items():
        note('%s/%s' % (tag, k), v if k != 'grad_tight_min' else 1.0 - v)


def run_step_with_tap(model, images, gts, fused=False):
    """model(...) -> loss -> backward with every conv layer's output recorded (CPU copies).  fused: through
    `model.train_step` instead — the production schedule (three streams, no torch.autograd: FasterRCNN._step_body)."""
    from luminoth_amd.models.base import layers as L
    model._step = 0
    L.ACT_TAP = {}
    try:
        if fused:
            model._step_state = {}          # a fresh (eager) step: the activation tap is host code
            _, pred = model.train_step(images, gts)
            losses = dict(model._last_losses)
        else:
            pred = model(images, gts, is_training=True)
        tap = {k: v.detach().cpu() for k, v in L.ACT_TAP.items()}
    finally:
        L.ACT_TAP = None
    if not fused:
        losses = model.loss(pred, return_all=True)
        model.backward(losses['total_loss'])
    torch.cuda.synchronize()
    return pred, losses, tap


def compare_step_with_oracle(model, images, gts, num_classes, arch='resnet_v1_50', oracle_kwargs=None,
                             check_grads=True, min_checked=100, out_tol=1e-4, loss_tol=1e-4, grad_tight=2e-4,
                             grad_max=1e-3, stats=None, fused=False, grad_floor_rel=0.0):
    """out_tol / loss_tol / grad_tight / grad_max: the fp32 bounds by default; the mixed-precision tests pass theirs
    (stated in tests/test_gpu_half.py).  stats (dict, optional): filled with the errors actually observed."""
    B, H, W = images.shape[0], images.shape[1], images.shape[2]
    stats_local = stats if stats is not None else {}
    stats = stats_local
    pred, losses, tap = run_step_with_tap(model, images, gts, fused=fused)
    oracle = OracleFasterRCNN(model.state_dict(), arch=arch, num_classes=num_classes, seed=0,
                              **(oracle_kwargs or {}))
    rp, cp = pred['rpn_prediction'], pred['classification_prediction']
    names = oracle.trainable_names()
    if check_grads:                       # (no autograd graph otherwise: ResNet-101 at 1024^2 is forward-only)
        for n in names:
            oracle.v[n].requires_grad_(True)
    per = {k: 0.0 for k in ('rpn_cls_loss', 'rpn_reg_loss', 'rcnn_cls_loss', 'rcnn_reg_loss')}
    R = cp['proposals'].shape[1]
    stride = model._anchor_stride
    for b in range(B):
        seed = orng.image_seed(0, 0, b)
        n_roi = int(cp['num_proposals'][b])
        rois = cp['proposals'][b, :n_roi].cpu().numpy()
        ov = dict(rois=rois, roi_labels=cp['target']['cls'][b, :n_roi].cpu().numpy(),
                  roi_targets=cp['target']['bbox_offsets'][b, :n_roi].cpu().numpy())
        # the kernels' own activations of image b: trunk / RPN layers are (B,h,w,c), tail layers (B*R,7,7,c)
        oracle.masks = {k: (v[b:b + 1] if v.shape[0] == B else v[b * R:b * R + n_roi]) for k, v in tap.items()}
        o = oracle.forward_image(images[b], gts[b], seed, overrides=ov)
        sc = rp['rpn_cls_score'][b].detach().cpu().numpy()
        _stat(stats, 'rpn_cls_score', sc, o['rpn_cls_score'].detach().numpy())
        np.testing.assert_allclose(sc, o['rpn_cls_score'].detach().numpy(), rtol=10 * out_tol,
                                   atol=out_tol * max(1.0, np.abs(sc).max()))
        bp = rp['rpn_bbox_pred'][b].detach().cpu().numpy()
        _stat(stats, 'rpn_bbox_pred', bp, o['rpn_bbox_pred'].detach().numpy())
        np.testing.assert_allclose(bp, o['rpn_bbox_pred'].detach().numpy(), rtol=10 * out_tol,
                                   atol=out_tol * max(1.0, np.abs(bp).max()))
        # anchor labels: bit-exact (functions of anchors + gt only)
        np.testing.assert_array_equal(rp['rpn_cls_target'][b].cpu().numpy(), o['rpn_labels'])
        np.testing.assert_allclose(rp['rpn_bbox_target'][b].cpu().numpy(), o['rpn_targets'], rtol=1e-5, atol=1e-6)
        # proposals on identical inputs (the kernel's own probabilities / deltas)
        fh, fw = model.base_network.feature_hw(H, W)
        anchors = obx.generate_anchors(oracle.anchor_ref, fh, fw, stride)
        pr = of.rpn_proposal(rp['rpn_cls_prob'][b].cpu().numpy(), bp, anchors, (H, W))
        n_p = int(rp['num_proposals'][b])
        assert n_p == pr['proposals'].shape[0]
        np.testing.assert_allclose(rp['proposals'][b, :n_p].cpu().numpy(), pr['proposals'], rtol=1e-6, atol=1e-4)
        # proposal targets on identical proposals: labels bit-exact
        lab, tg = of.rcnn_target(rp['proposals'][b, :n_p].cpu().numpy(), gts[b], seed=seed,
                                 minibatch_size=model._rcnn._rcnn_target._minibatch_size)
        keep = lab >= 0
        assert n_roi == int(keep.sum())
        np.testing.assert_array_equal(ov['roi_labels'], lab[keep])
        np.testing.assert_array_equal(rois, rp['proposals'][b, :n_p].cpu().numpy()[keep])
        np.testing.assert_allclose(ov['roi_targets'], tg[keep], rtol=1e-5, atol=1e-6)
        # RCNN head (ROI pooling, block4 tail for ResNet-101, FCs) on identical rois
        cs = cp['rcnn']['cls_score'][b, :n_roi].detach().cpu().numpy()
        _stat(stats, 'rcnn_cls_score', cs, o['rcnn_cls_score'].detach().numpy())
        np.testing.assert_allclose(cs, o['rcnn_cls_score'].detach().numpy(), rtol=10 * out_tol,
                                   atol=out_tol * max(1.0, np.abs(cs).max()))
        bo = cp['rcnn']['bbox_offsets'][b, :n_roi].detach().cpu().numpy()
        _stat(stats, 'rcnn_bbox_offsets', bo, o['rcnn_bbox_offsets'].detach().numpy())
        np.testing.assert_allclose(bo, o['rcnn_bbox_offsets'].detach().numpy(), rtol=10 * out_tol,
                                   atol=out_tol * max(1.0, np.abs(bo).max()))
        for k in per:
            per[k] = per[k] + o[k] / B
    # losses within 1e-4 (north_star)
    for k in per:
        got, ref = float(losses[k].detach()), float(per[k])
        if stats is not None:
            stats['loss:' + k] = abs(got - ref) / max(1.0, abs(ref))
        assert abs(got - ref) <= loss_tol * max(1.0, abs(ref)), (k, got, ref)
    reg = float(oracle.regularization_loss())
    assert abs(float(losses['regularization_loss']) - reg) <= 1e-4 * reg
    total = sum(per.values())
    assert abs(float(losses['no_reg_loss']) - float(total)) <= loss_tol * max(1.0, float(total))
    if not check_grads:
        _log_stats(model, images, arch, fused, oracle_kwargs, stats_local)
        return losses, per
    # gradients (data loss only; the L2 term is folded into the optimizer kernel), ReLU branches pinned
    total.backward()
    grads = model.store.grads
    checked, worst = 0, (0.0, None)
    # grad_floor_rel: tensors whose exact gradient is ZERO (a bias in front of a training-mode BatchNorm: adding a
    # per-channel constant changes nothing downstream) hold nothing but rounding residue on both sides; their scale is
    # floored at this fraction of the largest gradient in the network instead of at the residue itself
    floor = grad_floor_rel * max([float(oracle.v[n].grad.abs().max()) for n in names if oracle.v[n].grad is not None] or [0.0])
    for n in names:
        g_ref = oracle.v[n].grad
        if g_ref is None:
            continue
        g = grads[n].cpu().numpy().reshape(g_ref.shape)
        scale = max(1e-6, floor, float(g_ref.abs().max()))
        err = np.abs(g - g_ref.numpy())
        tight = err <= grad_tight * scale + 10 * grad_tight * np.abs(g_ref.numpy())
        if stats is not None:
            stats['grad_tight_min'] = min(stats.get('grad_tight_min', 1.0), float(tight.mean()))
            stats['grad_max'] = max(stats.get('grad_max', 0.0), float(err.max() / scale))
        assert tight.mean() >= 0.995, (n, float(tight.mean()))
        assert err.max() <= grad_max * scale, (n, float(err.max()), scale)
        if err.max() / scale > worst[0]:
            worst = (float(err.max() / scale), n)
        checked += 1
    assert checked >= min_checked, checked
    _log_stats(model, images, arch, fused, oracle_kwargs, stats_local)
    return losses, per


def free_running_agreement(model, images, gts, num_classes, arch='resnet_v1_50', oracle_kwargs=None):
    """What the teacher-forced comparison above cannot say: how far a REAL step drifts from the reference when the
    oracle runs on its OWN upstream outputs (its probabilities -> its proposals -> its sampled ROIs) instead of being
    handed the kernels'.  Decisions on near-ties (scores equal to the last bits) may then fall differently.  Returns per
    image the fraction of proposals that coincide (same box to 1e-3 px at the same rank / anywhere in the list), the
    fraction of the sampled ROI set shared, and both values of every loss."""
    pred = model(images, gts, is_training=True)
    losses = model.loss(pred, return_all=True)
    torch.cuda.synchronize()
    oracle = OracleFasterRCNN(model.state_dict(), arch=arch, num_classes=num_classes, seed=0, **(oracle_kwargs or {}))
    rp, cp = pred['rpn_prediction'], pred['classification_prediction']
    B = images.shape[0]
    rep = {'same_rank': [], 'same_set': [], 'roi_set': [], 'losses': {}}
    acc = {k: 0.0 for k in ('rpn_cls_loss', 'rpn_reg_loss', 'rcnn_cls_loss', 'rcnn_reg_loss')}
    with torch.no_grad():
        for b in range(B):
            o = oracle.forward_image(images[b], gts[b], orng.image_seed(0, 0, b))          # no overrides
            n_p = int(rp['num_proposals'][b])
            mine = rp['proposals'][b, :n_p].cpu().numpy()
            theirs = o['proposals']
            n = min(n_p, theirs.shape[0])
            # "the same box": within 1e-3 px (the device's expf and numpy's differ in the last bits of a coordinate);
            # set membership through a 0.01 px grid
            rep['same_rank'].append(float((np.abs(mine[:n] - theirs[:n]).max(axis=1) < 1e-3).mean()))
            q = lambda a_: set(map(tuple, np.round(np.asarray(a_, np.float64) * 100).astype(np.int64).tolist()))   # noqa: E731
            a, bset = q(mine), q(theirs)
            rep['same_set'].append(len(a & bset) / float(max(1, len(a | bset))))
            n_roi = int(cp['num_proposals'][b])
            ra, rb = q(cp['proposals'][b, :n_roi].cpu().numpy()), q(o['rois'])
            rep['roi_set'].append(len(ra & rb) / float(max(1, len(ra | rb))))
            for k in acc:
                acc[k] += float(o[k]) / B
    for k in acc:
        rep['losses'][k] = (float(losses[k].detach()), acc[k])
    return rep
