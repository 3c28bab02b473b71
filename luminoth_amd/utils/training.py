"""Optimizer / learning-rate factory + the data-parallel train step.

Reference: luminoth/utils/training.py:20-120 (`get_learning_rate`,
`get_optimizer`, `clip_gradients_by_norm`) and luminoth/train.py:66-91.

The optimizer is ONE fused kernel over the flat parameter buffer
(lmh_sgd_momentum); under torch.distributed (RCCL) the flat gradient buffer is
all-reduced with one collective before the update (replaces the reference's
asynchronous parameter-server exchange, train.py:46,282-326).
"""
import itertools
import os

import torch
import torch.distributed as dist

from luminoth_amd import kernels as K

FUSED_STEP = os.environ.get('LUMINOTH_AMD_FUSED_STEP', '1') != '0'
OPTIMIZERS = {'momentum', 'gradient_descent', 'adam', 'rmsprop'}
LEARNING_RATE_DECAY_METHODS = {'piecewise_constant', 'exponential_decay'}


def get_learning_rate(train_config, global_step=0):
    """training.py:20-61: constant, piecewise_constant(boundaries, values) or
    exponential_decay(decay_steps, decay_rate, staircase) schedule; host side."""
    lr_config = dict(train_config.learning_rate)
    lr_config.pop('_replace', None)
    decay = lr_config.pop('decay_method', None)
    if not decay or decay == 'none':
        return float(lr_config.get('learning_rate', lr_config.get('value')))
    if decay not in LEARNING_RATE_DECAY_METHODS:
        raise ValueError('Invalid learning_rate method "{}"'.format(decay))
    if decay == 'piecewise_constant':
        bounds, values = lr_config['boundaries'], lr_config['values']
        for b, v in zip(bounds, values):
            if global_step <= b:
                return float(v)
        return float(values[-1])
    base = float(lr_config['learning_rate'])
    p = global_step / float(lr_config['decay_steps'])
    if lr_config.get('staircase'):
        p = int(p)
    return base * float(lr_config['decay_rate']) ** p


class _Done(object):
    def wait(self):
        return True


class GradientBuckets(object):
    """Overlap of the data-parallel gradient exchange with the trunk backward.

    The flat gradient buffer is laid out in registration order (trunk conv weights block by block, the BatchNorm
    gamma/beta block, then RPN and RCNN) while the backward pass finishes it from the back: RPN / RCNN first,
    then the trunk nodes last to first.  Instead of ONE all-reduce after the whole backward, finished contiguous
    ranges are all-reduced (RCCL, async) on a communication stream while the earlier nodes are still running:

      * `[end of trunk parameters, end)` — the heads — when the trunk backward starts,
      * runs of trunk nodes of >= `bucket_bytes` of conv weights as soon as their last node is enqueued,
      * whatever is left (first nodes, BatchNorm block) in `finish()`, right before the update.

    The model arms it around the one backward call whose preconditions hold (`FasterRCNN.train_step`: every head
    gradient is complete and joined before the trunk backward); everything else falls through to `finish()`,
    which then is the single all-reduce of the whole buffer.  Every element is reduced exactly once per step
    (tested on one GPU with a stand-in reduce that doubles the range: tests/test_gpu_model.py)."""

    _generations = itertools.count(1)

    def __init__(self, store, reduce_fn=None, bucket_bytes=None):
        self.store = store
        # identifies THIS object in the keys of recorded launch plans (whose cut callbacks are bound to it): never reused
        self.generation = next(GradientBuckets._generations)
        self.reduce_fn = reduce_fn or self._all_reduce
        if bucket_bytes is None:
            mb = os.environ.get('LUMINOTH_AMD_BUCKET_MB')
            if mb is not None:
                bucket_bytes = int(mb) << 20
            elif PROBED_BUCKET_BYTES:
                # measured on this node before the first step (allreduce_probe): what the collective actually costs
                bucket_bytes = int(PROBED_BUCKET_BYTES)
            else:
                # no measurement: a ring all-reduce over N ranks pays 2 (N - 1) hop latencies per collective whatever its
                # size, so the bucket that amortises them grows with the ring: 6 MB at 2 ranks, 12 MB at 4, 24 MB at 8
                # (a GUESS — xGMI ~100 GB/s per link and direction, ~10 us per hop)
                world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
                bucket_bytes = int(min(24, max(6, 3 * world))) << 20
        self.bucket_bytes = int(bucket_bytes)
        self._armed = None
        self._plans = {}
        self._works = []
        self._done = []          # [lo, hi) ranges already handed to reduce_fn this step
        self._comm = None

    def abort(self):
        """A step raised after some ranges were handed out: forget them.  The next step starts with no range
        marked done, no pending work handles and no armed plan; what the failed step half-did to the gradient buffer (or,
        with EarlyUpdates, to the weights of the ranges it had already updated) is the caller's to judge — the step's
        exception propagates."""
        self._done, self._works, self._armed = [], [], None

    @staticmethod
    def _all_reduce(t):
        return dist.all_reduce(t, async_op=True)

    # ---- planning (host, once per trunk) -------------------------------------------------
    def _plan(self, nodes):
        key = tuple(id(n) for n in nodes)
        plan = self._plans.get(key)
        if plan is not None:
            return plan
        off = self.store.offsets
        numel = int(self.store.grad.numel())
        ranges, trunk_hi = [], 0
        for n in nodes:
            lo, hi = None, None
            for l in n.layers:
                for name in l.var_names():
                    t, o, cnt = off.get(name, (False, 0, 0))
                    if not t:
                        continue
                    trunk_hi = max(trunk_hi, o + cnt)
                    if name == l.w_name:
                        lo = o if lo is None else min(lo, o)
                        hi = o + cnt if hi is None else max(hi, o + cnt)
            ranges.append((lo, hi))
        plan = {}
        ok = True
        prev_lo = None
        for lo, hi in reversed([r for r in ranges if r[0] is not None]):
            if prev_lo is not None and hi > prev_lo:
                ok = False               # not laid out node after node: no early buckets, finish() does it all
            prev_lo = lo
        if ok:
            heads_lo = (trunk_hi + 3) // 4 * 4
            if heads_lo < numel:
                plan[len(nodes)] = (heads_lo, numel)
            cur_hi = None
            for j in range(len(nodes) - 1, -1, -1):
                lo, hi = ranges[j]
                if lo is None:
                    continue
                if cur_hi is None:
                    cur_hi = hi
                if (cur_hi - lo) * 4 >= self.bucket_bytes:
                    plan[j] = (lo, cur_hi)
                    cur_hi = None
        self._plans[key] = plan
        return plan

    def describe(self):
        """What the exchange looks like (bench.py prints it into the JSON line's `dist`)."""
        return {'bucket_mb': self.bucket_bytes / float(1 << 20),
                'early_ranges_mb': [[round((hi - lo) * 4 / float(1 << 20), 2) for lo, hi in
                                     sorted(plan.values(), key=lambda r: -r[0])] for plan in self._plans.values()],
                'grad_mb': round(int(self.store.grad.numel()) * 4 / float(1 << 20), 2)}

    # ---- per step ----------------------------------------------------------------------------
    def arm(self, trunk):
        """Early buckets are allowed for the next backward of `trunk` (or of a suffix of its nodes)."""
        self._armed = trunk.nodes[-1]

    def disarm(self):
        self._armed = None

    def on_node(self, nodes, j):
        if self._armed is None or nodes[-1] is not self._armed:
            return
        rng = self._plan(nodes).get(j)
        if rng is not None:
            self._launch(*rng)

    def _launch(self, lo, hi):
        from luminoth_amd.models.base.layers import SideStream
        grad = self.store.grad
        if not grad.is_cuda:          # host tensors (gloo tests): program order is the only order
            self._works.append(self.reduce_fn(grad[lo:hi]) or _Done())
            self._done.append((lo, hi))
            return
        from luminoth_amd import plan as P
        cur = torch.cuda.current_stream(grad.device)
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=grad.device)
        comm = self._comm
        K.stream_wait(comm, cur)                      # data-gradient stream up to this node
        for st in SideStream._streams.values():       # every weight-gradient chain enqueued so far
            K.stream_wait(comm, st)
        with K.launch_on(comm):
            K.TAILS.flush()       # the queued split-K / BatchNorm tails of everything enqueued so far, on the comm stream

        def exchange():
            # the collective is not a launch of the kernel library: under a recorded launch plan the plan is cut here
            # and this runs between its two parts at every replay (luminoth_amd/plan.py: host_call)
            with torch.cuda.stream(comm):
                self._works.append(self.reduce_fn(grad[lo:hi]) or _Done())
            self._done.append((lo, hi))
        P.host_call(exchange)

    def finish(self):
        """Called on the update stream after the backward (side streams joined): waits for the early buckets and
        reduces every range they did not cover."""
        grad = self.store.grad
        numel = int(grad.numel())
        todo, pos = [], 0
        for lo, hi in sorted(self._done):
            # every element exactly once per step: an early bucket that overlaps another (or runs past the buffer)
            # would be summed twice over the replicas — a silent 2x on those gradients
            if lo < pos or hi > numel or lo >= hi:
                raise RuntimeError('GradientBuckets: early buckets %r overlap or leave [0, %d)' % (sorted(self._done), numel))
            if lo > pos:
                todo.append((pos, lo))
            pos = hi
        if pos < numel:
            todo.append((pos, numel))
        covered = sum(hi - lo for lo, hi in self._done) + sum(hi - lo for lo, hi in todo)
        assert covered == numel, (covered, numel)
        for w in self._works:
            w.wait()
        if self._comm is not None and self._works and grad.is_cuda:
            torch.cuda.current_stream(grad.device).wait_stream(self._comm)
        for lo, hi in todo:
            w = self.reduce_fn(grad[lo:hi]) or _Done()
            w.wait()
        self._works, self._done = [], []
        self._armed = None


class EarlyUpdates(GradientBuckets):
    """ONE GPU, opt-in: the same walk over finished gradient ranges, but instead of all-reducing a range it is UPDATED right
    away (the 48 us optimizer launch at the end of the step, with the main stream idle, shrinks to the
    ranges the backward finishes last — and the backward pays for it: see MomentumOptimizer.__init__).  A range [lo, hi) is complete once the trunk backward has enqueued its nodes: the
    update is a recordable launch (`lmh_sgd_momentum_range`, learning rate read from device memory) on the bucket stream,
    behind every stream that produced those gradients and behind the weight-gradient tails of those layers (which read the
    weights: dgamma of a frozen BatchNorm) — and behind the proposal stream, whose reported L2 term reads ALL weights.
    Nothing reads the weights of a finished node again in the step (the Winograd-transformed copies, the BatchNorm scale
    table and the 16-bit working copies were made at the start of the step).  Same arithmetic per element as the one
    launch over the whole buffer: bit-identical weights (tests/test_gpu_plan.py)."""

    def __init__(self, store, optimizer, bucket_bytes=None):
        if bucket_bytes is None:
            bucket_bytes = int(os.environ.get('LUMINOTH_AMD_EARLY_UPDATE_MB', '8')) << 20
        super(EarlyUpdates, self).__init__(store, reduce_fn=lambda t: None, bucket_bytes=bucket_bytes)
        self.opt = optimizer

    def _launch(self, lo, hi):
        from luminoth_amd.models.base.layers import SideStream
        from luminoth_amd.models.fasterrcnn.fasterrcnn import FasterRCNN
        st = self.store
        grad = st.grad
        cur = torch.cuda.current_stream(grad.device)
        # the proposal / RCNN stream is idle during the trunk backward, and a FOURTH stream would alias one of the three onto
        # the same hardware queue (HIP creates 4: measured 10.9 instead of 6.6 ms per step with a stream of its own here).
        # Its earlier work of the step (the reported L2 term reads ALL weights) is in front of the update by stream order.
        comm = FasterRCNN._AUX_STREAMS.get(str(grad.device))
        if comm is None:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=grad.device)
            comm = self._comm
        self._comm = comm
        K.stream_wait(comm, cur)
        for s_ in SideStream._streams.values():
            if s_ is not comm:
                K.stream_wait(comm, s_)
        hi4 = hi if hi == int(grad.numel()) else hi // 4 * 4
        lo4 = (lo + 3) // 4 * 4
        if hi4 <= lo4:
            return
        from luminoth_amd import plan as P
        with K.launch_on(comm):
            K.TAILS.flush()
            K.sgd_momentum_range(st.flat, grad, st.mom, st.seg_offset, st.seg_wd, lo4, hi4, self.opt.lr_dev,
                                 self.opt.momentum, 1.0, early=True)
        # host bookkeeping of what the final update may skip: under a recorded plan it has to happen at every replay too
        P.host_call(lambda: self._done.append((lo4, hi4)))

    def finish(self):
        """-> the ranges no early update covered, after ordering the calling stream behind the bucket stream."""
        grad = self.store.grad
        numel = int(grad.numel())
        todo, pos = [], 0
        for lo, hi in sorted(self._done):
            if lo < pos or hi > numel or lo >= hi:
                raise RuntimeError('EarlyUpdates: ranges %r overlap or leave [0, %d)' % (sorted(self._done), numel))
            if lo > pos:
                todo.append((pos, lo))
            pos = hi
        if pos < numel:
            todo.append((pos, numel))
        if self._comm is not None and self._done:
            K.stream_wait(torch.cuda.current_stream(grad.device), self._comm)
        self._done = []
        self._armed = None
        return todo


# ---- the collective, measured: before the first multi-rank step the driver times the all-reduce of
# the real flat gradient and of one 12 MB bucket; the line reports what RCCL delivered and how many ranks it saw, and the
# bucket size follows from the measured latency / bandwidth instead of the constants above.
PROBED_BUCKET_BYTES = None


def bucket_bytes_from_probe(t_small_s, small_bytes, t_large_s, large_bytes, lo=4 << 20, hi=64 << 20, share=0.25):
    """Cost model of ONE all-reduce from two measurements, t(bytes) = alpha + bytes / beta (alpha: what a collective costs
    whatever its size — launch, ring hops —, beta: sustained bytes per second), and the bucket it implies: the smallest
    size whose fixed cost is at most `share` of its transfer time, alpha <= share * bytes / beta, rounded up to 1 MB and
    clamped to [lo, hi].  -> (bucket_bytes, alpha_s, beta_Bps).  Degenerate measurements (the larger message not slower)
    fall back to beta = large_bytes / t_large and alpha = the small message's whole time."""
    if large_bytes > small_bytes and t_large_s > t_small_s > 0:
        beta = (large_bytes - small_bytes) / (t_large_s - t_small_s)
        alpha = max(0.0, t_small_s - small_bytes / beta)
    else:
        beta = large_bytes / max(t_large_s, 1e-9)
        alpha = max(t_small_s, 0.0)
    want = alpha * beta / share
    mb = 1 << 20
    bucket = int(min(hi, max(lo, (int(want) + mb - 1) // mb * mb)))
    return bucket, alpha, beta


def allreduce_probe(flat_numel, device, reps=5, bucket_mb=12):
    """Times dist.all_reduce (sum, fp32) of a tensor as large as the model's flat gradient and of one `bucket_mb` MB bucket:
    one untimed round, then the median of `reps`; plus an all-reduce of ones — `ranks_seen` comes from the collective's own
    result.  Every rank calls this at the same point.  -> dict for the bench line (`dist.allreduce_probe`); None when
    torch.distributed is not initialised or has one rank."""
    import time
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return None
    world = dist.get_world_size()
    on_gpu = dist.get_backend() == 'nccl'
    dev = device if on_gpu else torch.device('cpu')
    ones = torch.ones(1, dtype=torch.float32, device=dev)
    dist.all_reduce(ones)
    out = {'ranks_seen': int(round(float(ones[0]))), 'world_size': world, 'backend': dist.get_backend(), 'reps': reps,
           'messages': []}
    times = {}
    for label, nbytes in (('flat_gradient', int(flat_numel) * 4), ('bucket', int(bucket_mb) << 20)):
        t = torch.ones(max(1, nbytes // 4), dtype=torch.float32, device=dev)
        dist.all_reduce(t)                      # connection set-up, buffer registration
        samples = []
        for _ in range(reps):
            t.fill_(1.0)
            if on_gpu:
                torch.cuda.synchronize(dev)
            dist.barrier()
            t0 = time.perf_counter()
            dist.all_reduce(t)
            if on_gpu:
                torch.cuda.synchronize(dev)
            samples.append(time.perf_counter() - t0)
        med = sorted(samples)[len(samples) // 2]
        times[label] = (med, t.numel() * 4)
        # a ring moves 2 (N - 1) / N x bytes through every rank: the usual "bus bandwidth" of an all-reduce
        out['messages'].append({'what': label, 'bytes': t.numel() * 4, 'ms': med * 1e3, 'ms_min': min(samples) * 1e3,
                                'GB/s_per_rank': t.numel() * 4 / med / 1e9,
                                'bus_GB/s': 2.0 * (world - 1) / world * t.numel() * 4 / med / 1e9,
                                'correct': bool(abs(float(t[0]) - world) < 1e-3)})
        del t
    (ts, bs), (tl, bl) = sorted(times.values(), key=lambda v: v[1])
    bucket, alpha, beta = bucket_bytes_from_probe(ts, bs, tl, bl)
    out.update({'alpha_us': alpha * 1e6, 'beta_GB/s': beta / 1e9, 'bucket_bytes_from_probe': bucket,
                'model': 't(bytes) = alpha + bytes / beta from the two messages; bucket = smallest size with alpha <= 25 % of '
                         'its transfer time, 1 MB steps, clamped to [4, 64] MB'})
    return out


ACTIVE_BUCKETS = None          # the GradientBuckets of the optimizer in use (None: single GPU)


def install_buckets(buckets):
    """Routes Trunk.backward's node hook to `buckets` (None uninstalls)."""
    global ACTIVE_BUCKETS
    from luminoth_amd.models.base import layers as L
    ACTIVE_BUCKETS = buckets
    L.BACKWARD_HOOK = buckets.on_node if buckets is not None else None


class MomentumOptimizer(object):
    """tf.train.MomentumOptimizer (non-Nesterov): v = m*v + g ; w -= lr*v, with
    g = grad/world + wd*w (the L2 regulariser of total_loss).  One fused kernel over the flat buffer; with
    `train.clip_by_norm` the per-variable clip factors of `clip_gradients_by_norm` (training.py:84-120) are computed
    by one reduction launch in front of it."""
    KIND = 0

    def __init__(self, model, train_config, momentum=0.9, use_nesterov=False):
        self.model, self.store, self.cfg = model, model.store, train_config
        self.momentum = float(momentum)
        self.use_nesterov = bool(use_nesterov)     # TF ApplyMomentum: w -= g*lr + v*momentum*lr (v updated first)
        self.clip_norm = 10.0 if train_config.get('clip_by_norm') else None      # training.py:108: clip_by_norm(g, 10.)
        self.global_step = 0
        self.buckets = None
        self._slot2 = None
        self._factors = None
        self.early = None
        self.lr_dev, self._lr_dev_value = None, None
        if dist.is_available() and dist.is_initialized() and \
                (dist.get_world_size() > 1 or os.environ.get('LUMINOTH_AMD_FORCE_BUCKETS') == '1'):
            if os.environ.get('LUMINOTH_AMD_BUCKETED_ALLREDUCE', '1') != '0' and \
                    (self.store.grad.is_cuda or os.environ.get('LUMINOTH_AMD_FORCE_BUCKETS') == '1'):
                self.buckets = GradientBuckets(self.store)
                install_buckets(self.buckets)
        elif (type(self) is MomentumOptimizer and not self.use_nesterov and self.clip_norm is None and
              self.store.grad.is_cuda and hasattr(model, '_step_body') and
              os.environ.get('LUMINOTH_AMD_EARLY_UPDATE', '0') != '0'):
            # one GPU, plain momentum SGD: finished gradient ranges are updated under the rest of the backward pass.  OPT-IN
            # (LUMINOTH_AMD_EARLY_UPDATE=1): measured on the benchmark step the end of the step gets 0.043 ms shorter and the
            # backward 0.066 ms longer (the HBM-bound update competes with it): 6.547 -> 6.570 ms, profiles/r05_schedule_ab.md
            self.lr_dev = torch.zeros(1, dtype=torch.float32, device=self.store.grad.device)
            self.early = EarlyUpdates(self.store, self)
            install_buckets(self.early)

    def reduce_gradients(self):
        """Sum the flat gradient buffer over the data-parallel replicas; returns the factor that turns the sum
        into the mean (applied inside the update kernel).  With buckets installed most of the buffer was already
        all-reduced under the trunk backward and only the remainder is exchanged here; otherwise this is ONE
        collective over the whole flat buffer."""
        st = self.store
        if self.buckets is not None:
            self.buckets.finish()
            return 1.0 / dist.get_world_size()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(st.grad)                 # ONE bucket: the flat gradient buffer
            return 1.0 / dist.get_world_size()
        return 1.0

    def _clip_factors(self, gscale):
        if self.clip_norm is None:
            return None
        st = self.store
        if self._factors is None:
            self._factors = torch.empty_like(st.seg_wd)
        K.grad_clip_factors(st.flat, st.grad, st.seg_offset, st.seg_wd, gscale, self.clip_norm, self._factors)
        return self._factors

    def _update(self, lr, gscale, factors):
        st = self.store
        if factors is None and not self.use_nesterov:
            K.sgd_momentum(st.flat, st.grad, st.mom, st.seg_offset, st.seg_wd, lr, self.momentum, gscale)
        else:
            K.optimizer_step(3 if self.use_nesterov else 0, st.flat, st.grad, st.mom, None, st.seg_offset, st.seg_wd,
                             factors, lr, self.momentum, 0.0, 0.0, gscale)

    def prepare_step(self):
        """Before the step's launches: the learning rate of THIS step in device memory (early updates read it from there;
        written only when it changes)."""
        if self.early is not None:
            lr = get_learning_rate(self.cfg, self.global_step)
            if lr != self._lr_dev_value:
                self.lr_dev.fill_(lr)
                self._lr_dev_value = lr

    def step(self):
        if self.early is not None:
            self.prepare_step()          # (a caller that did not go through train_step: nothing was updated early then)
            st = self.store
            for lo, hi in self.early.finish():
                K.sgd_momentum_range(st.flat, st.grad, st.mom, st.seg_offset, st.seg_wd, lo, hi, self.lr_dev, self.momentum, 1.0)
            self.global_step += 1
            return
        gscale = self.reduce_gradients()
        lr = get_learning_rate(self.cfg, self.global_step)
        self._update(lr, gscale, self._clip_factors(gscale))
        self.global_step += 1


class AdamOptimizer(MomentumOptimizer):
    """tf.train.AdamOptimizer(learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8): slots m, v start at zero, the
    bias-corrected rate lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t) is formed on the host (TF: _prepare/_finish)."""
    KIND = 1

    def __init__(self, model, train_config, beta1=0.9, beta2=0.999, epsilon=1e-8):
        super(AdamOptimizer, self).__init__(model, train_config, momentum=0.0)
        self.beta1, self.beta2, self.epsilon = float(beta1), float(beta2), float(epsilon)
        self._slot2 = torch.zeros_like(self.store.mom)

    @property
    def _t(self):
        """TF keeps beta1_power / beta2_power as global (non-slot) variables, which the reference's saver DOES
        checkpoint (train.py:108-112 saves everything but the slots): after a resume the bias correction continues at
        the restored step while m and v restart at zero.  The counter is therefore a function of `global_step` (which
        train.run restores), not a private count of the calls made in this process."""
        return self.global_step + 1

    def _update(self, lr, gscale, factors):
        st = self.store
        lr_t = lr * (1.0 - self.beta2 ** self._t) ** 0.5 / (1.0 - self.beta1 ** self._t)
        K.optimizer_step(1, st.flat, st.grad, st.mom, self._slot2, st.seg_offset, st.seg_wd, factors, lr_t,
                         self.beta1, self.beta2, self.epsilon, gscale)


class RMSPropOptimizer(MomentumOptimizer):
    """tf.train.RMSPropOptimizer(learning_rate, decay=0.9, momentum=0.0, epsilon=1e-10): the mean-square slot starts
    at ONE (TF's initialiser), the momentum slot at zero."""
    KIND = 2

    def __init__(self, model, train_config, decay=0.9, momentum=0.0, epsilon=1e-10, centered=False):
        super(RMSPropOptimizer, self).__init__(model, train_config, momentum=momentum)
        self.decay, self.epsilon = float(decay), float(epsilon)
        self.store.mom.fill_(1.0)                    # slot 1 = ms
        self._slot2 = torch.zeros_like(self.store.mom)
        self.centered = bool(centered)               # TF ApplyCenteredRMSProp: a third slot mg (mean gradient), starts at 0
        self._slot3 = torch.zeros_like(self.store.mom) if self.centered else None

    def _update(self, lr, gscale, factors):
        st = self.store
        K.optimizer_step(4 if self.centered else 2, st.flat, st.grad, st.mom, self._slot2, st.seg_offset, st.seg_wd,
                         factors, lr, self.decay, self.momentum, self.epsilon, gscale, slot3=self._slot3)


def get_optimizer(train_config, model):
    """training.py:64-81: OPTIMIZERS[type](learning_rate, **remaining optimizer config)."""
    opt = dict(train_config.optimizer)
    opt.pop('_replace', None)
    kind = opt.pop('type')
    if kind not in OPTIMIZERS:
        raise ValueError('Invalid optimizer type "{}"'.format(kind))
    opt.pop('use_locking', None)
    opt.pop('name', None)

    def take(allowed):
        """The reference forwards the remaining keys to the TF optimizer constructor, which raises TypeError on an
        unknown keyword; a key TF knows but no kernel here implements must not be swallowed either."""
        kw = {k: opt.pop(k) for k in list(opt) if k in allowed}
        if opt:
            raise TypeError('optimizer "%s": unsupported argument(s) %s' % (kind, sorted(opt)))
        return kw

    if kind == 'momentum':
        return MomentumOptimizer(model, train_config, **take(('momentum', 'use_nesterov')))
    if kind == 'gradient_descent':
        take(())
        return MomentumOptimizer(model, train_config, momentum=0.0)
    if kind == 'adam':
        return AdamOptimizer(model, train_config, **take(('beta1', 'beta2', 'epsilon')))
    return RMSPropOptimizer(model, train_config, **take(('decay', 'momentum', 'epsilon', 'centered')))


def issue_from_high_priority_stream(device):
    """Make a HIGH-priority HIP stream the calling thread's current stream on `device` (drivers call this once, before
    building the model).  The train step issues its critical path — forward, data gradients, update — from the current
    stream and creates the weight-gradient and proposal / RCNN streams itself at the default (lowest) priority, so when
    several streams have workgroups ready the dispatcher serves the critical path first and the other two fill what is
    left (measured on one MI355X box: 8.35 -> 8.28 ms/step).  LUMINOTH_AMD_MAIN_PRIORITY=0 keeps the default stream."""
    if os.environ.get('LUMINOTH_AMD_MAIN_PRIORITY', '1') == '0' or not torch.cuda.is_available():
        return None
    spec = os.environ.get('LUMINOTH_AMD_MAIN_CU_MASK', '')       # experiment: 'period:lo:hi' (kernels.cu_range_stream)
    st = K.cu_range_stream(spec, device) if spec else torch.cuda.Stream(device=device, priority=-1)
    st.wait_stream(torch.cuda.current_stream(device))
    torch.cuda.set_stream(st)
    return st


def broadcast_parameters(model, src=0):
    """Identical replicas at start (seeded init already makes them identical; this is the belt)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(model.store.flat, src)
        dist.broadcast(model.store.frozen, src)


def train_step(model, optimizer, image, gt_boxes, next_image=None, next_gt=None):
    """One step of train.py:66-91: forward, loss, backward, (all-reduce), update.  `next_image` / `next_gt` (optional): the
    batch of the FOLLOWING step, if the caller already has it — the model computes that batch's frozen trunk prefix and
    anchor targets ahead of time in otherwise idle slots of this step (same arithmetic; see FasterRCNN.train_step)."""
    if hasattr(optimizer, 'prepare_step'):
        optimizer.prepare_step()
    if FUSED_STEP and hasattr(model, 'train_step'):
        if next_image is not None and getattr(model, 'accepts_next_image', False):
            total, pred = model.train_step(image, gt_boxes, next_image=next_image, next_gt=next_gt)
        else:
            total, pred = model.train_step(image, gt_boxes)     # same arithmetic, two-stream schedule
    else:
        pred = model(image, gt_boxes, is_training=True)
        total = model.loss(pred)
        model.backward(total)
    optimizer.step()
    return total, pred
