"""Launch plans (csrc/plan.hip, luminoth_amd/plan.py): a recorded train step re-issued with one host call must be the
step the host code issues eagerly — same kernels, same arguments, same cross-stream order.  Pinned here by running the
SAME sequence of steps with plans on and off and demanding bit-identical weights, and by the data-parallel stand-in of
tests/test_gpu_model.py under replay.  Run with `-m gpu`."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

from e2e_util import condition_like_pretrained, make_config, synth      # noqa: E402


@pytest.fixture(scope='module')
def setup():
    from luminoth_amd.models import get_model
    cfg = make_config(**{'train.learning_rate.learning_rate': 1e-5})     # 12 updates of this random-init model stay finite
    model = condition_like_pretrained(get_model('fasterrcnn')(cfg), 'resnet_v1_50')
    a, gts_a = synth(2, 320, 384, 4, 80, 3)
    b, gts_b = synth(2, 320, 384, 4, 80, 5)
    dev = model.device
    return cfg, model, [(a.to(dev), gts_a), (b.to(dev), gts_b)]


def _run(model, cfg, batches, steps, plan_on, lookahead=True):
    from luminoth_amd import plan as P
    from luminoth_amd.utils import training as T
    P.ENABLED = plan_on
    model._step_state = {}
    model._step = 0
    model.store.mom.zero_()
    opt = T.get_optimizer(cfg.train, model)
    losses = []
    for i in range(steps):
        cur, nxt = batches[i % 2], batches[(i + 1) % 2]
        if lookahead:
            total, _ = T.train_step(model, opt, cur[0], cur[1], next_image=nxt[0], next_gt=nxt[1])
        else:
            total, _ = T.train_step(model, opt, cur[0], cur[1])
        losses.append(dict((k, float(v)) for k, v in model._last_losses.items()))
    torch.cuda.synchronize()
    plans = [pl for S in model._step_state.values() for pl in S['plans'].values()]
    return model.store.flat.clone(), losses, plans


@pytest.mark.parametrize('lookahead', [True, False], ids=['lookahead', 'plain'])
def test_replayed_steps_equal_eager_steps(setup, lookahead):
    """12 steps over two alternating batches: with launch plans the first steps run eagerly, two are recorded (one per
    step parity) and the rest are replays — weights after the last update bit-identical to 12 eager steps, every loss
    equal (the reported L2 term is an fp32 atomic sum over 13 M terms: 1e-5)."""
    from luminoth_amd import plan as P
    cfg, model, batches = setup
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    try:
        w_eager, l_eager, plans0 = _run(model, cfg, batches, 12, False, lookahead)
        assert plans0 == []
        model.load_state_dict(sd0)
        w_plan, l_plan, plans = _run(model, cfg, batches, 12, True, lookahead)
        assert len(plans) == 2 and all(pl.replays >= 3 for pl in plans), [(pl.replays, pl.n_kernels) for pl in plans]
        assert all(150 < pl.n_kernels < 400 for pl in plans), [pl.n_kernels for pl in plans]
        assert bool(torch.isfinite(w_eager).all()) and torch.equal(w_plan, w_eager)
        for a, b in zip(l_plan, l_eager):
            for k in a:
                assert abs(a[k] - b[k]) <= 1e-5 * max(1.0, abs(b[k])), (k, a[k], b[k])
    finally:
        P.ENABLED = True
        model.load_state_dict(sd0)
        model._step_state = {}


def test_unannounced_batch_falls_back_to_its_own_prefix(setup):
    """A replayed step that was promised batch B but receives another tensor copies it in and computes its own prefix /
    anchor targets (another plan variant): same weights as eager steps over the same sequence."""
    from luminoth_amd import plan as P
    from luminoth_amd.utils import training as T
    cfg, model, batches = setup
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    (a, ga), (b, gb) = batches
    c = torch.flip(a, dims=[2]).contiguous()
    seq = [(a, ga, b, gb), (b, gb, a, ga), (a, ga, b, gb), (b, gb, a, ga), (a, ga, b, gb), (b, gb, a, ga),
           (c, ga, b, gb),          # announced `a`, `c` arrives
           (b, gb, a, ga), (a, ga, b, gb)]

    def run(plan_on):
        P.ENABLED = plan_on
        model.load_state_dict(sd0)
        model._step_state, model._step = {}, 0
        model.store.mom.zero_()
        opt = T.get_optimizer(cfg.train, model)
        for x, g, nx, ng in seq:
            T.train_step(model, opt, x, g, next_image=nx, next_gt=ng)
        torch.cuda.synchronize()
        return model.store.flat.clone()

    try:
        want = run(False)
        got = run(True)
        assert bool(torch.isfinite(want).all()) and torch.equal(got, want)
    finally:
        P.ENABLED = True
        model.load_state_dict(sd0)
        model._step_state = {}


def test_variable_shapes_keep_lookahead_and_replay(setup):
    """real data hands the step a new image size and a new number of gt boxes almost every batch.
    The gt count is not part of a shape's identity (fixed buffers have a bucketed capacity, `gt_count` carries the real
    counts), a next batch of ANOTHER image size still gets its frozen prefix and anchor targets computed one step ahead
    (into that shape's own state), a shape that comes back replays its plan, and all of it leaves the same bits as plain
    eager steps without any look-ahead."""
    from luminoth_amd import plan as P
    from luminoth_amd.utils import training as T
    cfg, model, _ = setup
    dev = model.device
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    shapes = [(256, 320, 3), (256, 320, 5), (320, 256, 2), (256, 320, 4), (256, 320, 6), (256, 320, 3), (256, 320, 7),
              (320, 256, 5), (256, 320, 2), (256, 320, 4), (256, 320, 3), (256, 320, 5)]
    data = []
    for i, (h, w, g) in enumerate(shapes):
        im, gts = synth(2, h, w, g, 80, 40 + i)
        gts[1] = gts[1][:max(1, g - 1)]                     # ragged counts inside a batch as well
        data.append((im.to(dev), gts))

    def run(plan_on, lookahead):
        P.ENABLED = plan_on
        model.load_state_dict(sd0)
        model._step_state, model._step = {}, 0
        model.store.mom.zero_()
        opt = T.get_optimizer(cfg.train, model)
        for i, (im, gts) in enumerate(data):
            if lookahead and i + 1 < len(data):
                T.train_step(model, opt, im, gts, next_image=data[i + 1][0], next_gt=data[i + 1][1])
            else:
                T.train_step(model, opt, im, gts)
        torch.cuda.synchronize()
        return model.store.flat.clone()

    try:
        want = run(False, False)
        got = run(True, True)
        states = model._step_state
        assert len(states) == 2, list(states)                                   # two image sizes, ONE gt bucket each
        assert all(k[3] == 8 for k in states)
        served = sum(n for S in states.values() for v, n in S['seen'].items() if v[1])
        replays = sum(pl.replays for S in states.values() for pl in S['plans'].values())
        # every batch but the first was announced: its prefix / targets were waiting in its slot, also across the size change
        assert served + replays >= len(data) - 1 and served >= 3, (served, replays)
        assert replays >= 1, [(v, pl.replays) for S in states.values() for v, pl in S['plans'].items()]
        assert bool(torch.isfinite(want).all()) and torch.equal(got, want)
        # the gt capacity of a model only GROWS.  A batch with 12 boxes moves the 256 x 320 batches to a capacity-16
        # state; a following batch with 3 boxes stays there (keyed by its own bucket it would bounce back to the capacity-8
        # state and every switch would run eagerly), and the counters show what was replayed / run eagerly / evicted
        model.plan_stats.clear()
        opt = T.get_optimizer(cfg.train, model)
        for g in (12, 3, 5, 12, 2, 4):
            im, gts = synth(2, 256, 320, g, 80, 70 + g)
            T.train_step(model, opt, im.to(dev), gts)
        torch.cuda.synchronize()
        caps = sorted(k[3] for k in model._step_state if k[1:3] == (256, 320))
        assert caps == [8, 16] and model._gt_cap_seen == 16, caps
        S16 = [S for k, S in model._step_state.items() if k[1:3] == (256, 320) and k[3] == 16][0]
        assert S16['n'] == 6                                    # all six steps, whatever their own box count
        assert model.plan_stats['replayed_steps'] + model.plan_stats['eager_steps'] == 6
        assert model.plan_stats['replayed_steps'] >= 2 and model.plan_stats['evicted_states'] == 0, dict(model.plan_stats)
    finally:
        P.ENABLED = True
        model.load_state_dict(sd0)
        model._step_state = {}
        model._gt_cap_seen = 8


@pytest.mark.parametrize('decay', [False, True], ids=['constant_lr', 'exponential_decay'])
def test_early_range_updates_equal_one_update(setup, decay, monkeypatch):
    """VERDICT r4 next #7 (opt-in, LUMINOTH_AMD_EARLY_UPDATE=1): on one GPU the momentum update of a finished gradient range is
    issued under the rest of the backward pass (`EarlyUpdates`: recordable launches with the learning rate in device memory)
    and only the last ranges are left for `optimizer.step()`.  Same arithmetic per element: weights AND momentum bit-identical to the single launch over the whole
    buffer, through eager, recorded and replayed steps, also when the learning rate changes every step."""
    from luminoth_amd import plan as P
    from luminoth_amd.utils import training as T
    from luminoth_amd.utils.config import get_config
    cfg, model, batches = setup
    lr = {'_replace': True, 'decay_method': 'exponential_decay', 'learning_rate': 2e-5, 'decay_steps': 3, 'decay_rate': 0.7} \
        if decay else {'_replace': True, 'decay_method': None, 'learning_rate': 1e-5}
    tcfg = get_config({'model': {'type': 'fasterrcnn'}, 'train': {'seed': 0, 'learning_rate': lr}}).train
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}

    def run(early):
        monkeypatch.setenv('LUMINOTH_AMD_EARLY_UPDATE', '1' if early else '0')
        P.ENABLED = True
        model.load_state_dict(sd0)
        model._step_state, model._step = {}, 0
        model.store.mom.zero_()
        opt = T.get_optimizer(tcfg, model)
        assert (opt.early is not None) == early
        for i in range(9):
            cur, nxt = batches[i % 2], batches[(i + 1) % 2]
            T.train_step(model, opt, cur[0], cur[1], next_image=nxt[0], next_gt=nxt[1])
        torch.cuda.synchronize()
        plans = [pl for S in model._step_state.values() for pl in S['plans'].values()]
        return model.store.flat.clone(), model.store.mom.clone(), opt, plans

    try:
        w0, m0, _, _ = run(False)
        w1, m1, opt, plans = run(True)
        assert sum(pl.replays for pl in plans) >= 4
        ranges = [r for plan in opt.early._plans.values() for r in plan.values()]
        covered = sum(hi - lo for lo, hi in ranges)
        assert len(ranges) >= 2 and covered * 2 > int(model.store.flat.numel()), (ranges, covered)     # most of the buffer goes early
        assert bool(torch.isfinite(w0).all()) and torch.equal(w1, w0) and torch.equal(m1, m0)
    finally:
        T.install_buckets(None)
        P.ENABLED = True
        model.load_state_dict(sd0)
        model._step_state = {}


def test_gradient_buckets_under_replay(setup):
    """The data-parallel exchange is host work between two parts of a plan (plan.host_call): with a stand-in reduce that
    doubles its range, replayed steps must leave exactly 2 x the plain gradient — every element handed over once per
    step, never before its producers finished — like the eager protocol (tests/test_gpu_model.py)."""
    from luminoth_amd import plan as P
    from luminoth_amd.utils import training as T
    cfg, model, batches = setup
    x, g = batches[0]
    P.ENABLED = False
    model._step_state, model._step = {}, 0
    model.train_step(x, g)
    torch.cuda.synchronize()
    ref = model.store.grad.clone()
    calls = []

    def doubling(t):
        calls.append(int(t.numel()))
        t.mul_(2.0)

    buckets = T.GradientBuckets(model.store, reduce_fn=doubling, bucket_bytes=4 << 20)
    T.install_buckets(buckets)
    P.ENABLED = True
    try:
        for rep in range(8):
            model._step = 0
            del calls[:]
            model.train_step(x, g)
            early = len(calls)
            buckets.finish()
            torch.cuda.synchronize()
            assert early >= 3 and len(calls) > early, (rep, early, len(calls))
            assert sum(calls) == model.store.grad.numel()
            scale = float(ref.abs().max())
            np.testing.assert_allclose(model.store.grad.cpu().numpy(), 2.0 * ref.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale)
        plans = [pl for S in model._step_state.values() for pl in S['plans'].values()]
        assert plans and sum(pl.replays for pl in plans) >= 4 and all(len(pl.cuts) >= 3 for pl in plans)
    finally:
        T.install_buckets(None)
        P.ENABLED = True
        model._step_state = {}


def test_plan_c_abi_records_and_replays_a_memset_and_a_kernel():
    """The C ABI alone: record {memset, bn_refresh kernel, stream wait}, change the inputs, replay -> the outputs follow."""
    from luminoth_amd import _lib, kernels as K
    lib = _lib.load()
    dev = torch.device('cuda')
    n = 1000
    gamma, beta, mean, rstd = (torch.rand(n, device=dev) + 0.5 for _ in range(4))
    scale, shift = torch.empty(n, device=dev), torch.empty(n, device=dev)
    junk = torch.ones(64, device=dev)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    assert lib.lmh_plan_recording() == 0 and lib.lmh_plan_position() == -1
    assert lib.lmh_plan_begin() == 0
    K.zero_(junk)
    K.bn_refresh(gamma, beta, mean, rstd, scale, shift)
    K.stream_wait(side, torch.cuda.current_stream())
    assert lib.lmh_plan_position() == 4          # memset, kernel, event record, stream wait
    plan = lib.lmh_plan_end()
    assert plan and lib.lmh_plan_size(ctypes.c_void_p(plan)) == 4
    assert lib.lmh_plan_kernel_count(ctypes.c_void_p(plan), 0, -1) == 1
    gamma.mul_(3.0)
    junk.fill_(7.0)
    assert lib.lmh_plan_run(ctypes.c_void_p(plan), 0, -1) == 0
    torch.cuda.synchronize()
    s = gamma * rstd
    assert torch.equal(scale, s) and torch.equal(shift, beta - mean * s) and float(junk.abs().max()) == 0.0
    lib.lmh_plan_destroy(ctypes.c_void_p(plan))
