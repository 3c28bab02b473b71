"""Launch plans — ONE host call per train step (csrc/plan.hip).

The reference's host pays one call per step, `sess.run(train_op)` over a graph built once
(luminoth/train.py:235-247).  The host code of this package walks its layer lists every step instead: ~250 kernel
launches, each a Python -> ctypes round trip.  A `StepPlan` is the counterpart of the built graph for one input shape:
the first steps run eagerly; one step is then run with the C library RECORDING every launch, memset, copy, event record
and stream wait it issues (the step executes normally while it is recorded); later steps of the same shape re-issue the
recorded sequence from C — the same kernels with the same arguments on the same streams, nothing skipped.

What makes a recorded step replayable is a property of the HOST code around it, kept here and in
`FasterRCNN._planned_step`:
  * every tensor whose address entered a recorded launch is kept alive by the plan (`kernels._TLS.keep`), so an address
    never goes back to the allocator and never means another tensor;
  * per-step inputs (images, gt boxes, seeds) and what one step hands to the next (the frozen trunk prefix and the
    anchor targets computed one step ahead) live in buffers at fixed addresses, double-buffered by step parity;
  * scalars that change from step to step (the learning rate) stay outside the recorded region (the optimizer update
    is issued after the plan);
  * host work that must happen between two launches of the step — the asynchronous all-reduce of a finished gradient
    bucket under data parallelism — is registered with `host_call`: the plan is cut there and the callback runs between
    the two parts at every replay.
"""
import ctypes
import os

from luminoth_amd import _lib
from luminoth_amd import kernels as K

# LUMINOTH_AMD_PLAN=0: every step is issued eagerly by the host code (the schedule of rounds 1-3)
ENABLED = os.environ.get('LUMINOTH_AMD_PLAN', '1') != '0'
# eager steps of one shape / variant before it is recorded: allocations, workspaces and lazily built constants settle
WARM_STEPS = int(os.environ.get('LUMINOTH_AMD_PLAN_WARM', '1'))

import threading


class _TLS(threading.local):
    active = None      # the StepPlan being recorded on THIS thread (the C recorder, csrc/plan.hip g_rec, is thread-local too)


_tls = _TLS()


def active():
    return _tls.active


class StepPlan(object):
    def __init__(self):
        self.handle = None
        self.keep = []           # tensors (and anything else) that must outlive the plan
        self.events = []         # library events owned by the plan
        self.cuts = []           # (node index, callable): host work between two parts of the plan
        self.result = None       # what the recorded step returned
        self.n_nodes = 0
        self.n_kernels = 0
        self.replays = 0

    # ---- recording ----------------------------------------------------------------------------------
    def __enter__(self):
        if _tls.active is not None:
            raise RuntimeError('a launch plan is already being recorded')
        _lib.check(_lib.load().lmh_plan_begin(), 'lmh_plan_begin')
        _tls.active = self
        K._TLS.keep = self.keep
        return self

    def __exit__(self, exc_type, exc, tb):
        lib = _lib.load()
        _tls.active = None
        K._TLS.keep = None
        if exc_type is not None:
            lib.lmh_plan_abort()
            self.destroy()
            return False
        self.handle = lib.lmh_plan_end()
        if not self.handle:
            raise _lib.LuminothHipError('lmh_plan_end failed: %s' % lib.lmh_last_error().decode())
        self.n_nodes = lib.lmh_plan_size(ctypes.c_void_p(self.handle))
        self.n_kernels = lib.lmh_plan_kernel_count(ctypes.c_void_p(self.handle), 0, -1)
        return False

    def new_event(self):
        ev = _lib.load().lmh_event_create()
        if not ev:
            raise _lib.LuminothHipError('lmh_event_create failed')
        self.events.append(ev)
        return ev

    # ---- replay --------------------------------------------------------------------------------------
    def run(self):
        lib = _lib.load()
        h = ctypes.c_void_p(self.handle)
        pos = 0
        for at, fn in self.cuts:
            if at > pos:
                _lib.check(lib.lmh_plan_run(h, pos, at), 'lmh_plan_run')
            fn()
            pos = at
        _lib.check(lib.lmh_plan_run(h, pos, -1), 'lmh_plan_run')
        self.replays += 1
        return self.result

    def destroy(self):
        lib = _lib.load()
        if self.handle:
            lib.lmh_plan_destroy(ctypes.c_void_p(self.handle))
            self.handle = None
        for ev in self.events:
            lib.lmh_event_destroy(ctypes.c_void_p(ev))
        self.events, self.keep, self.cuts, self.result = [], [], [], None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def pinned_bytes(plan):
    """Bytes of device memory `plan` keeps alive (distinct storages among the tensors it holds; cached on the plan)."""
    n = getattr(plan, '_pinned', None)
    if n is None:
        import torch
        seen, n = set(), 0
        stack = list(plan.keep)
        while stack:
            o = stack.pop()
            if torch.is_tensor(o):
                st = o.untyped_storage()
                if st.data_ptr() not in seen:
                    seen.add(st.data_ptr())
                    n += st.nbytes()
            elif isinstance(o, dict):
                stack.extend(o.values())
            elif isinstance(o, (list, tuple)):
                stack.extend(o)
        plan._pinned = n
    return n


def recording():
    return _tls.active is not None


def host_call(fn):
    """Run `fn()` now; while a plan is being recorded also register it to run at this very position of every replay
    (host work that launches through something else than this library: a collective)."""
    if _tls.active is not None:
        _tls.active.cuts.append((_lib.load().lmh_plan_position(), fn))
    return fn()


def keep(*objs):
    """Objects the plan being recorded must keep alive."""
    if _tls.active is not None:
        _tls.active.keep.extend(objs)
