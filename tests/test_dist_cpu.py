"""CPU, world_size 2, gloo: the N>1 host logic of the train step — ONE all-reduce of the
flat gradient buffer, mean folded into the update, replicas stay bit-identical.  The HIP
update kernel itself needs a GPU; here a numpy twin of `k_sgd_momentum` stands in for it so the
collective plumbing is what is under test."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Store(object):
    def __init__(self, n, rank):
        g = torch.Generator().manual_seed(0)
        self.flat = torch.randn(n, generator=g)            # identical on every rank (seeded init)
        self.mom = torch.zeros(n)
        gr = torch.Generator().manual_seed(100 + rank)
        self.grad = torch.randn(n, generator=gr)           # per-rank gradients
        self.seg_offset = torch.tensor([0, n // 2, n], dtype=torch.int64)
        self.seg_wd = torch.tensor([5e-4, 0.0])


class _Model(object):
    pass


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from luminoth_amd.utils import training
    from luminoth_amd.utils.config import Config

    def sgd_twin(w, g, v, seg_offset, seg_wd, lr, momentum, gscale):     # numpy twin of k_sgd_momentum
        wd = torch.zeros_like(w)
        for s in range(seg_wd.numel()):
            wd[int(seg_offset[s]):int(seg_offset[s + 1])] = seg_wd[s]
        gi = g * gscale + wd * w
        v.mul_(momentum).add_(gi)
        w.sub_(lr * v)
    training.K.sgd_momentum = sgd_twin
    m = _Model()
    m.store = _Store(1000, rank)
    local_grad = m.store.grad.clone()
    cfg = Config({'learning_rate': {'decay_method': None, 'learning_rate': 0.01},
                  'optimizer': {'type': 'momentum', 'momentum': 0.9}, 'clip_by_norm': False})
    opt = training.get_optimizer(cfg, m)
    w0 = m.store.flat.clone()
    opt.step()
    q.put((rank, local_grad.numpy(), m.store.grad.numpy().copy(), m.store.flat.numpy().copy(), w0.numpy()))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, g0, s0, w0_new, w0), (_, g1, s1, w1_new, _) = res
    np.testing.assert_allclose(s0, g0 + g1, rtol=1e-6)       # gradients were summed once
    np.testing.assert_array_equal(s0, s1)
    np.testing.assert_array_equal(w0_new, w1_new)             # replicas stay bit-identical
    wd = np.concatenate([np.full(500, 5e-4, np.float32), np.zeros(500, np.float32)])
    exp = w0 - 0.01 * ((g0 + g1) * 0.5 + wd * w0)
    np.testing.assert_allclose(w0_new, exp, rtol=1e-5, atol=1e-7)


class _FakeLayer(object):
    def __init__(self, scope, with_bn):
        self.w_name = scope + '/weights'
        self._names = [self.w_name] + ([scope + '/BatchNorm/beta', scope + '/BatchNorm/gamma'] if with_bn else [])

    def var_names(self):
        return self._names


class _FakeNode(object):
    def __init__(self, layers):
        self.layers = layers


class _FakeTrunk(object):
    def __init__(self, nodes):
        self.nodes = nodes


def _bucket_worker(rank, world, port, q):
    """The path N > 1 actually takes: MomentumOptimizer installs GradientBuckets, the (simulated) trunk backward
    finishes the flat gradient buffer from the back and calls the node hook, early buckets are all-reduced
    asynchronously (gloo) while 'earlier nodes' are still being written, finish() reduces the rest."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LUMINOTH_AMD_FORCE_BUCKETS='1', LUMINOTH_AMD_BUCKET_MB='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from luminoth_amd.models.base import layers as L
    from luminoth_amd.utils import training
    from luminoth_amd.utils.config import Config

    def sgd_twin(w, g, v, seg_offset, seg_wd, lr, momentum, gscale):
        v.mul_(momentum).add_(g * gscale)
        w.sub_(lr * v)
    training.K.sgd_momentum = sgd_twin
    # flat layout like the real store: conv weights node after node, then the BatchNorm block, then the heads
    sizes = [40, 24, 64, 16, 100]
    nodes = [_FakeNode([_FakeLayer('n%d' % i, True)]) for i in range(len(sizes))]
    offsets, o = {}, 0
    for i, sz in enumerate(sizes):
        offsets['n%d/weights' % i] = (True, o, sz)
        o += sz
    for kind in ('gamma', 'beta'):
        for i in range(len(sizes)):
            offsets['n%d/BatchNorm/%s' % (i, kind)] = (True, o, 4)
            o += 4
    heads_lo = o
    n = o + 60                                            # RPN / RCNN variables
    m = _Model()
    m.store = _Store(n, rank)
    m.store.offsets = offsets
    m.store.grad = torch.zeros(n)
    g_local = torch.randn(n, generator=torch.Generator().manual_seed(100 + rank))
    cfg = Config({'learning_rate': {'decay_method': None, 'learning_rate': 0.01},
                  'optimizer': {'type': 'momentum', 'momentum': 0.9}, 'clip_by_norm': False})
    opt = training.get_optimizer(cfg, m)
    assert opt.buckets is not None and training.ACTIVE_BUCKETS is opt.buckets and L.BACKWARD_HOOK is not None
    calls = []
    real = opt.buckets.reduce_fn

    def counting(t):
        calls.append(int(t.numel()))
        return real(t)
    opt.buckets.reduce_fn = counting
    trunk = _FakeTrunk(nodes)
    w0 = m.store.flat.clone()
    for step in range(2):
        del calls[:]
        m.store.grad.zero_()
        m.store.grad[heads_lo:] = g_local[heads_lo:]      # head gradients complete before the trunk backward
        opt.buckets.arm(trunk)
        L.BACKWARD_HOOK(nodes, len(nodes))
        for j in range(len(nodes) - 1, -1, -1):           # Trunk.backward: node j's gradients, then the hook
            _, lo, cnt = offsets['n%d/weights' % j]
            m.store.grad[lo:lo + cnt] = g_local[lo:lo + cnt]
            for kind in ('gamma', 'beta'):
                _, lo, cnt = offsets['n%d/BatchNorm/%s' % (j, kind)]
                m.store.grad[lo:lo + cnt] = g_local[lo:lo + cnt]
            L.BACKWARD_HOOK(nodes, j)
        early = len(calls)
        opt.buckets.disarm()
        opt.step()                                        # finish() + update
        if step == 0:
            first = (early, list(calls), m.store.grad.clone())
    training.install_buckets(None)
    q.put((rank, g_local.numpy(), first[0], first[1], first[2].numpy(), m.store.flat.numpy().copy(), w0.numpy(), n))
    dist.destroy_process_group()


def test_gradient_buckets_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, g0, early0, calls0, s0, w0_new, w0, n), (_, g1, early1, calls1, s1, w1_new, _, _) = res
    assert early0 == early1 >= 3 and len(calls0) > early0          # heads + trunk runs early, the BN block in finish()
    assert calls0 == calls1 and sum(calls0) == n                    # every element exactly once, same order on all ranks
    np.testing.assert_allclose(s0, g0 + g1, rtol=1e-6)
    np.testing.assert_array_equal(s0, s1)
    np.testing.assert_array_equal(w0_new, w1_new)                   # replicas stay bit-identical after two steps
    v1 = (g0 + g1) * 0.5
    exp = w0 - 0.01 * v1 - 0.01 * (0.9 * v1 + v1)
    np.testing.assert_allclose(w0_new, exp, rtol=1e-5, atol=1e-6)


def test_learning_rate_schedules():
    sys.path.insert(0, ROOT)
    from luminoth_amd.utils.config import Config
    from luminoth_amd.utils.training import get_learning_rate, get_optimizer
    import pytest
    c = Config({'learning_rate': {'decay_method': None, 'learning_rate': 3e-4}})
    assert get_learning_rate(c, 10) == 3e-4
    c = Config({'learning_rate': {'decay_method': 'piecewise_constant', 'boundaries': [10, 20],
                                  'values': [1.0, 0.1, 0.01]}})
    assert [get_learning_rate(c, s) for s in (0, 10, 11, 20, 21)] == [1.0, 1.0, 0.1, 0.1, 0.01]
    c = Config({'learning_rate': {'decay_method': 'exponential_decay', 'learning_rate': 1.0, 'decay_steps': 10,
                                  'decay_rate': 0.5, 'staircase': True}})
    assert get_learning_rate(c, 25) == 0.25
    with pytest.raises(ValueError):
        get_learning_rate(Config({'learning_rate': {'decay_method': 'bogus', 'learning_rate': 1.0}}), 0)
    with pytest.raises(ValueError):
        get_optimizer(Config({'optimizer': {'type': 'bogus'}, 'learning_rate': {}}), None)


# ------------------------------------------------------------------------------------------------------------
def _driver_worker(rank, world, port, tmpdir, q):
    """luminoth_amd.train.run as one of `world` ranks (gloo, CPU): mock model / step, synthetic dataset."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LUMINOTH_AMD_DIST_BACKEND='gloo')
    from luminoth_amd import train as T
    from luminoth_amd.utils import training
    from luminoth_amd.utils.config import get_config
    seen = []

    class Model(object):
        def __init__(self, config):
            self.w = torch.zeros(2)

        def state_dict(self):
            return {'mock/w': self.w.clone()}

        def load_state_dict(self, sd, strict=True):
            self.w = torch.as_tensor(np.asarray(sd['mock/w'])).clone()

    class Opt(object):
        global_step = 0

    def step(model, optimizer, image, gt_boxes):
        seen.append(float(image.sum()))
        t = torch.tensor([float(image.mean())])
        dist.all_reduce(t)                       # every rank must reach every step: a skipped step would hang here
        model.w = model.w + t / world
        optimizer.global_step += 1
        return t[0], {}

    training.get_optimizer = lambda cfg, model: Opt()
    training.broadcast_parameters = lambda model: None
    cfg = get_config({'model': {'type': 'fasterrcnn'}},
                     ['train.num_epochs=1', 'dataset.type=synthetic', 'dataset.num_images=7', 'dataset.height=32',
                      'dataset.width=48', 'train.save_checkpoint_secs=0', 'train.job_dir=%s' % tmpdir,
                      'train.run_name=dp', 'train.seed=5'])
    steps = T.run(cfg, get_model_fn=lambda t: Model, train_step_fn=step)
    files = sorted(os.listdir(os.path.join(tmpdir, 'dp'))) if os.path.isdir(os.path.join(tmpdir, 'dp')) else []
    q.put((rank, steps, seen, files))
    dist.destroy_process_group()


def test_train_driver_world2_gloo(tmp_path):
    """`lumi train` re-host under data parallelism (train.py:282-326 replaced by one process per GPU): the 7 synthetic
    batches are dealt rank::world and trimmed to a common length (3 steps each — no rank runs a step the other skips),
    the shards are disjoint, and only rank 0 writes the checkpoint."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_driver_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, seen0, files0), (r1, s1, seen1, files1) = res
    assert (s0, s1) == (3, 3) and len(seen0) == len(seen1) == 3
    assert not set(seen0) & set(seen1)                       # disjoint shards
    assert any(f.startswith('model.ckpt-3') for f in files0) and 'checkpoint' in files0


def _probe_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from luminoth_amd.utils import training
    p = training.allreduce_probe(300000, torch.device('cpu'), reps=3, bucket_mb=1)
    # the bucket size the exchange of this process would use now
    training.PROBED_BUCKET_BYTES = p['bucket_bytes_from_probe']
    m = _Model()
    m.store = _Store(1000, rank)
    b = training.GradientBuckets(m.store, reduce_fn=lambda t: None)
    q.put((rank, p, b.bucket_bytes))
    dist.destroy_process_group()


def test_allreduce_probe_world2():
    """before the first multi-rank step bench.py times the all-reduce of the flat gradient and of one
    bucket and derives the bucket size of the exchange from what it measured; `ranks_seen` is the collective's own result.
    World size 2 over gloo (the code path of the RCCL run, host tensors)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_probe_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, pr, bucket in res:
        assert pr['ranks_seen'] == 2 and pr['world_size'] == 2 and pr['backend'] == 'gloo'
        whats = {m['what']: m for m in pr['messages']}
        assert whats['flat_gradient']['bytes'] == 1200000 and whats['bucket']['bytes'] == 1 << 20
        assert all(m['correct'] and m['ms'] > 0 and m['GB/s_per_rank'] > 0 for m in pr['messages'])
        assert (4 << 20) <= pr['bucket_bytes_from_probe'] <= (64 << 20) and pr['bucket_bytes_from_probe'] % (1 << 20) == 0
        assert bucket == pr['bucket_bytes_from_probe']          # GradientBuckets takes the measured size


def test_bucket_size_from_measured_allreduce_cost():
    """t(bytes) = alpha + bytes / beta from two timings; the bucket is the smallest size whose fixed cost is <= 25 % of its
    transfer time (1 MB steps, [4, 64] MB)."""
    from luminoth_amd.utils.training import bucket_bytes_from_probe
    mb = 1 << 20
    # 100 GB/s and 40 us per collective: alpha * beta / 0.25 = 16 MB
    t = lambda n: 40e-6 + n / 100e9      # noqa: E731
    b, alpha, beta = bucket_bytes_from_probe(t(12 * mb), 12 * mb, t(54 * mb), 54 * mb)
    assert abs(alpha - 40e-6) < 1e-9 and abs(beta - 100e9) / 100e9 < 1e-9 and b == 16 * mb
    # a fast fabric with a cheap collective: clamped to 4 MB; a slow start-up: clamped to 64 MB
    t2 = lambda n: 2e-6 + n / 300e9      # noqa: E731
    assert bucket_bytes_from_probe(t2(12 * mb), 12 * mb, t2(54 * mb), 54 * mb)[0] == 4 * mb
    t3 = lambda n: 1e-3 + n / 100e9      # noqa: E731
    assert bucket_bytes_from_probe(t3(12 * mb), 12 * mb, t3(54 * mb), 54 * mb)[0] == 64 * mb
    # degenerate timings (the larger message measured faster): still a valid size
    b4, a4, be4 = bucket_bytes_from_probe(1e-3, 12 * mb, 0.9e-3, 54 * mb)
    assert 4 * mb <= b4 <= 64 * mb and be4 > 0 and a4 >= 0
