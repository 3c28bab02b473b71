"""Tensor-level launchers: torch device tensors in, C-ABI calls out.

Every function here enqueues hand-written gfx950 kernels from
libluminoth_hip.so on the caller's current HIP stream and returns torch tensors
that merely own the memory.  Nothing in this file computes with torch ops, and
nothing falls back to the CPU: tensors must live on a ROCm device.
"""
import ctypes
import threading
import os

import torch

from . import _lib
from ._lib import (ConvDesc, RpnProposalDesc, RpnTargetDesc, RcnnTargetDesc, RcnnProposalDesc, SsdTargetDesc, check)

ACT = {None: 0, 'none': 0, 'relu': 1, 'relu6': 2,
       # not fused into a convolution epilogue: applied in place behind it (act_fwd_) and differentiated from the output
       # (act_bwd); include/luminoth_hip.h lmh_act_fwd, luminoth/utils/vars.py:80-88
       'elu': 3, 'selu': 4, 'softplus': 5, 'softsign': 6, 'sigmoid': 7, 'tanh': 8, 'leaky_relu': 9}
FUSED_ACTS = (None, 'none', 'relu', 'relu6')      # what a convolution descriptor (and an activation bit mask) can carry


# While a launch plan is being recorded (luminoth_amd/plan.py) every tensor whose address enters a launch is appended
# here and kept alive by the plan: a recorded pointer must stay valid — and must never be handed to another tensor —
# for as long as the plan is replayed.
# per THREAD, like the C recorder (csrc/plan.hip g_rec): tensors whose addresses enter launches of the plan this thread records
class _PlanTLS(threading.local):
    keep = None


_TLS = _PlanTLS()


def plan_keep(*tensors):
    """Tensors whose data_ptr() goes into a launch without passing through _p (descriptor arrays)."""
    keep = _TLS.keep
    if keep is not None:
        keep.extend(t for t in tensors if t is not None)


def _p(t):
    if t is None:
        return None
    keep = _TLS.keep
    if keep is not None:
        keep.append(t)
    if not t.is_cuda:
        raise _lib.LuminothHipError('luminoth_amd kernels need ROCm device tensors (got %s); '
                                    'there is no CPU fallback' % t.device)
    if not t.is_contiguous():
        raise _lib.LuminothHipError('non-contiguous tensor passed to a HIP kernel')
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)   # ~10x cheaper than current_stream()


_stream_override = None      # raw hipStream_t every launch of this module goes to while set (launch_on)
_stream_override_obj = None  # ... and its torch.cuda.Stream (allocations that must belong to that stream)


class launch_on(object):
    """`with launch_on(stream):` — the kernels of this module launched inside go to `stream` (a torch.cuda.Stream) WITHOUT
    making it torch's current stream: ~5 us cheaper per use than `with torch.cuda.stream(...)`, which the train step would
    pay once per trainable layer.  Only for code that launches through this module and allocates nothing stream-bound
    (persistent workspaces are keyed by the effective stream)."""
    __slots__ = ('h', 'stream', 'prev')

    def __init__(self, stream):
        self.h, self.stream = stream.cuda_stream, stream

    def __enter__(self):
        global _stream_override, _stream_override_obj
        self.prev = (_stream_override, _stream_override_obj)
        _stream_override, _stream_override_obj = self.h, self.stream

    def __exit__(self, *a):
        global _stream_override, _stream_override_obj
        _stream_override, _stream_override_obj = self.prev


def stream_wait(waiter, signaler):
    """waiter.wait_stream(signaler) through the C library (one ctypes call, no Event object)."""
    check(_lib.load().lmh_stream_wait_stream(ctypes.c_void_p(waiter.cuda_stream), ctypes.c_void_p(signaler.cuda_stream)),
          'lmh_stream_wait_stream')


def cu_range_stream(spec, device):
    """'period:lo:hi' (or 'period:keep' = 'period:0:keep') -> a torch stream whose kernels only occupy, of every XCD's
    compute units c = 0..31, those with lo <= c % period < hi (hipExtStreamCreateWithCUMask; experiments only)."""
    v = [int(x) for x in spec.split(':')]
    period, lo, hi = (v[0], 0, v[1]) if len(v) == 2 else v
    h = _lib.load().lmh_stream_create_cu_range(period, lo, hi)
    if not h:
        raise _lib.LuminothHipError('lmh_stream_create_cu_range(%d, %d, %d) failed' % (period, lo, hi))
    return torch.cuda.ExternalStream(h, device=device)


def event_record(event, stream):
    """hipEventRecord of a library event (lmh_event_create handle) on a torch stream — recordable in a launch plan."""
    check(_lib.load().lmh_event_record(ctypes.c_void_p(event), ctypes.c_void_p(stream.cuda_stream)), 'lmh_event_record')


def stream_wait_event(stream, event):
    check(_lib.load().lmh_stream_wait_event(ctypes.c_void_p(stream.cuda_stream), ctypes.c_void_p(event)),
          'lmh_stream_wait_event')


def zero_(t):
    """t <- 0 on the launch stream (async memset through the library: recordable, no framework fill kernel)."""
    assert t.is_contiguous()
    check(_lib.load().lmh_memset(_p(t), 0, ctypes.c_size_t(t.numel() * t.element_size()), _stream()), 'lmh_memset')
    return t


def copy_(dst, src):
    """dst <- src (same dtype / number of elements, both contiguous device tensors) on the launch stream."""
    assert dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and dst.numel() == src.numel(), \
        (dst.shape, src.shape, dst.dtype, src.dtype)
    check(_lib.load().lmh_memcpy_d2d(_p(dst), _p(src), ctypes.c_size_t(dst.numel() * dst.element_size()), _stream()),
          'lmh_memcpy_d2d')
    return dst


def bn_refresh(gamma, beta, mean, rstd, scale, shift):
    check(_lib.load().lmh_bn_refresh(_p(gamma), _p(beta), _p(mean), _p(rstd), gamma.numel(), _p(scale), _p(shift),
                                     _stream()), 'lmh_bn_refresh')


def bn_train_fwd(z, gamma, beta, moving_mean, moving_var, residual=None, act=None, eps=1e-5, decay=0.997,
                 update_moving=True):
    """BatchNorm with the statistics of the batch over the RAW convolution output z (N,H,W,K) -> (y, mean, rstd); the
    moving statistics are advanced in place (lmh_bn_train_fwd)."""
    lib = _lib.load()
    Kc = z.shape[-1]
    rows = z.numel() // Kc
    y = torch.empty_like(z)
    mean = torch.empty((Kc,), dtype=torch.float32, device=z.device)
    rstd = torch.empty((Kc,), dtype=torch.float32, device=z.device)
    ws = _workspace(lib.lmh_bn_train_workspace_bytes(rows, Kc), z.device, 'bn_train')
    check(lib.lmh_bn_train_fwd(_p(_f32(z)), rows, Kc, _p(gamma), _p(beta), float(eps), float(decay), _p(moving_mean),
                               _p(moving_var), int(bool(update_moving)), _p(residual), ACT[act], _p(y), _p(mean), _p(rstd),
                               _p(ws), ctypes.c_size_t(ws.numel()), _stream()), 'lmh_bn_train_fwd')
    return y, mean, rstd


def bn_train_bwd(g, z, mean, rstd, gamma, dgamma, dbeta, addend=None, frozen=False, need_dz=True):
    """-> dz (gradient of the BatchNorm input; + addend); dgamma / dbeta (K,) are written.  frozen: mean / rstd are the
    moving statistics (constants): dz = gamma * rstd * g."""
    lib = _lib.load()
    Kc = z.shape[-1]
    rows = z.numel() // Kc
    dz = torch.empty_like(z) if need_dz else None
    ws = _workspace(lib.lmh_bn_train_workspace_bytes(rows, Kc), z.device, 'bn_train')
    check(lib.lmh_bn_train_bwd(_p(_f32(g)), _p(z), _p(mean), _p(rstd), _p(gamma), rows, Kc, _p(addend), int(bool(frozen)),
                               _p(dgamma), _p(dbeta), _p(dz), _p(ws), ctypes.c_size_t(ws.numel()), _stream()),
          'lmh_bn_train_bwd')
    return dz


def bn_apply(z, scale, shift, residual=None, act=None):
    """y = act(z * scale + shift (+ residual)): a frozen-statistics BatchNorm that does not follow a convolution."""
    Kc = z.shape[-1]
    y = torch.empty_like(z)
    check(_lib.load().lmh_bn_apply(_p(_f32(z)), z.numel() // Kc, Kc, _p(scale), _p(shift), _p(residual), ACT[act], _p(y),
                                   _stream()), 'lmh_bn_apply')
    return y


def loss_sums(terms, reg_a=None, reg_b=None, out=None):
    """-> out (3,) = [total, no_reg, regularization]: no_reg = ((t0 + t1) + t2) + ..., regularization = reg_a + reg_b
    (fasterrcnn.py:203-259) in ONE single-thread launch.  `terms`: 0-d / 1-element device tensors."""
    dev = terms[0].device
    out = out if out is not None else torch.empty((3,), dtype=torch.float32, device=dev)
    arr = (ctypes.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
    plan_keep(*terms)
    check(_lib.load().lmh_loss_sums(arr, len(terms), _p(reg_a), _p(reg_b), _p(out), _stream()), 'lmh_loss_sums')
    return out


def _stream_id(device=None):
    """Raw hipStream_t (int) of torch's CURRENT stream on `device` (default: current device)."""
    if _stream_override is not None:
        return _stream_override
    if _raw_stream is not None:
        idx = torch.cuda.current_device() if device is None or device.index is None else device.index
        return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream


def _stream():
    return ctypes.c_void_p(_stream_id())


def effective_stream(device=None):
    """The torch.cuda.Stream launches of this module go to right now: the launch_on override when one is set, else torch's
    current stream (what record_stream-style protection must be issued against)."""
    return _stream_override_obj if _stream_override_obj is not None else torch.cuda.current_stream(device)


def _f32(t):
    assert t.dtype == torch.float32, t.dtype
    return t


def keep_alive(t, stream):
    """tensor.record_stream(stream) — the caching allocator must not recycle `t` while `stream` still reads it.
    Skipped while the step is being captured into a HIP graph: the graph's private pool keeps every block it
    handed out alive (and at a fixed address) for as long as the graph exists."""
    if t is not None and not torch.cuda.is_current_stream_capturing():
        t.record_stream(stream)


_ws_cache = {}


def _workspace(nbytes, device, tag):
    """Persistent scratch per (tag, device, stream): no allocation in the steady state, and two
    streams never share a scratch buffer."""
    key = (tag, device, _stream_id(device))
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        # (re)allocated from the pool of the stream that uses it: the block a grown workspace replaces is then only ever
        # recycled in that stream's order — under launch_on torch's current stream is not the launch stream
        if _stream_override_obj is not None:
            with torch.cuda.stream(_stream_override_obj):
                ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        else:
            ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


# ------------------------------------------------------- deferred tails ----
class TailQueue(object):
    """Backlog of weight-gradient tails (csrc/tail.hip): while `TAILS.active`, ConvLayer backward passes launch only the
    MFMA weight-gradient kernel (and the g = dy*act'(y) pass) and queue what is left — split-K reduction, BatchNorm
    scaling + dgamma, dbeta / dbias — as one descriptor per layer; `flush()` finishes the whole backlog with two
    launches on the current stream.  The caller flushes after joining every stream that produced gradients and before
    anything reads them (optimizer, gradient all-reduce).  Entries keep their tensors alive until the flush."""

    EARLY_MIN = int(os.environ.get('LUMINOTH_AMD_EARLY_TAILS', '8'))     # 0: only the final flush

    def __init__(self):
        self.active = False
        self.entries = {}      # layer key -> dict of fields
        self.order = []
        self.early = None      # (idle stream, callable -> streams that produce gradients) while a step allows it
        self.early_used = False

    def begin(self, early=None):
        """Start queueing (drops anything a failed step may have left behind).  `early`: see maybe_flush_early."""
        self.entries, self.order = {}, []
        self.active = True
        self.early = early if self.EARLY_MIN > 0 else None
        self.early_used = False

    def maybe_flush_early(self):
        """The tails are HBM-bound and tiny next to the MFMA kernels still to come, so a backlog of >= EARLY_MIN layers
        is finished right away on an otherwise IDLE stream (the proposal / RCNN stream after its branch is done), ordered
        behind every stream that produced those gradients — instead of serially at the very end of the step, where the
        main stream has nothing to hide them under.  The caller joins that stream before anything reads the gradients."""
        if self.early is None or len(self.order) < self.EARLY_MIN:
            return
        stream, producers = self.early
        stream_wait(stream, torch.cuda.current_stream(stream.device))
        for st in producers():
            stream_wait(stream, st)
        with launch_on(stream):
            self.flush()
        self.early_used = True

    def abort(self):
        """A step failed part-way: stop queueing and drop the backlog, so that a later backward which does not go
        through begin() / flush() (plain autograd on `model(...)` outputs) runs its tails immediately again."""
        self.active = False
        self.early = None
        self.early_used = False
        self.entries, self.order = {}, []

    def entry(self, key):
        e = self.entries.get(key)
        if e is None:
            e = self.entries[key] = {}
            self.order.append(key)
        return e

    def flush(self):
        if not self.order:
            return
        lib = _lib.load()
        n = len(self.order)
        arr = (_lib.WgradTail * n)()
        dev = None
        for i, key in enumerate(self.order):
            e = self.entries[key]
            t = arr[i]
            dw = e['dw']
            dev = dw.device
            t.dw = dw.data_ptr()
            t.n = dw.numel()
            t.K = dw.shape[-1]
            t.slabs, t.splits = e.get('slabs', 0) or 0, e.get('splits', 0)
            bn = e.get('bn')
            if bn is not None:
                t.w, t.scale, t.mean, t.rstd, t.dgamma = (bn[k].data_ptr() for k in ('w', 'scale', 'mean', 'rstd', 'dgamma'))
            t.colpart, t.colrows = e.get('colpart', 0) or 0, e.get('colrows', 0)
            cs = e.get('colsum')
            t.colsum = cs.data_ptr() if cs is not None else 0
            if _TLS.keep is not None:
                plan_keep(dw, cs, e.get('_ws'), e.get('_cws'), *(bn.values() if bn is not None else ()))
        ws = _workspace(lib.lmh_wgrad_tail_batch_workspace_bytes(arr, n), dev, 'tails')
        check(lib.lmh_wgrad_tail_batch(arr, n, _p(ws), ctypes.c_size_t(ws.numel()), _stream()), 'lmh_wgrad_tail_batch')
        self.entries, self.order = {}, []


TAILS = TailQueue()


def _last_plan():
    lib = _lib.load()
    slabs, colpart = ctypes.c_void_p(), ctypes.c_void_p()
    splits, colrows = ctypes.c_int(), ctypes.c_int()
    lib.lmh_tail_last_plan(ctypes.byref(slabs), ctypes.byref(splits), ctypes.byref(colpart), ctypes.byref(colrows))
    return slabs.value or 0, splits.value, colpart.value or 0, colrows.value


# ------------------------------------------------------------- profiling ----
class _Profile(object):
    """Optional per-launch timing of the convolution MFMA kernels (bench.py roofline leg).  The HIP events are
    recorded by the C library on the launch stream directly around the implicit-GEMM kernel of each conv call
    (lmh_conv2d_profile_next); the library also names the kernel as rocprofv3 prints it and reports the FLOPs
    the launch executed (for a Winograd stacked GEMM: the 16 transformed-domain products, not the direct count)."""
    enabled = False
    records = []   # (kernel, executed flops, compulsory HBM bytes, direct-convolution flops of the layer, e0, e1)

    @classmethod
    def start(cls):
        cls.enabled, cls.records = True, []

    overhead_ms = 0.0   # of the last stop(): what an event pair adds to a kernel's own duration (lmh_event_pair_overhead_ms)

    @classmethod
    def stop(cls):
        """-> {kernel: {launches, flops, bytes, direct_flops, ms, ms_raw}}: `ms_raw` sums the event intervals as measured,
        `ms` the same minus the calibrated interval of an event pair around an empty kernel (the dispatch latency
        between the start event and the kernel's first wave is not kernel time; rocprofv3's durations exclude it too)."""
        cls.enabled = False
        torch.cuda.synchronize()
        lib = _lib.load()
        cls.overhead_ms = max(0.0, float(lib.lmh_event_pair_overhead_ms(32, _stream())))
        out = {}
        for name, flops, nbytes, direct, e0, e1 in cls.records:
            r = out.setdefault(name, {'launches': 0, 'flops': 0.0, 'bytes': 0.0, 'direct_flops': 0.0, 'ms': 0.0,
                                      'ms_raw': 0.0})
            r['launches'] += 1
            r['flops'] += flops
            r['bytes'] += nbytes
            r['direct_flops'] += direct
            raw = lib.lmh_event_elapsed_ms(e0, e1)
            r['ms_raw'] += raw
            r['ms'] += max(raw - cls.overhead_ms, 0.25 * raw)
            lib.lmh_event_destroy(e0)
            lib.lmh_event_destroy(e1)
        cls.records = []
        return out


def _conv_flops(d):
    return 2.0 * d.N * d.OH * d.OW * d.K * d.R * d.S * d.C


class _timed(object):
    def __init__(self, d, op=None, wino=False):
        self.on = _Profile.enabled
        self.d = d

    def __enter__(self):
        if self.on:
            lib = _lib.load()
            self.e0, self.e1 = lib.lmh_event_create(), lib.lmh_event_create()
            lib.lmh_conv2d_profile_next(self.e0, self.e1)

    def __exit__(self, *a):
        if self.on:
            fl = ctypes.c_double(0.0)
            lib = _lib.load()
            name = lib.lmh_conv2d_profile_last(ctypes.byref(fl))
            name = name.decode() if name else ''
            if name:
                _Profile.records.append((name, fl.value, lib.lmh_conv2d_profile_last_bytes(), _conv_flops(self.d),
                                         self.e0, self.e1))


# ------------------------------------------------------------------ conv ----
def same_pads(in_size, k, stride, dilation=1):
    """TF 'SAME' padding: out = ceil(in/stride); leading pad = total // 2."""
    out = -(-in_size // stride)
    eff = (k - 1) * dilation + 1
    total = max((out - 1) * stride + eff - in_size, 0)
    return out, total // 2


COMPUTE = {None: 0, 'f32': 0, 'fp32': 0, 'float32': 0, 'f16': 1, 'fp16': 1, 'float16': 1, 'bf16': 2, 'bfloat16': 2,
           'bf16x3': 3, 'f32x3': 3}    # bf16x3: fp32 arithmetic as an exact 3-way bf16 split, six MFMA products (conv_half.h)


def conv_desc(x_shape, w_shape, stride=1, dilation=1, padding='SAME', act=None, compute=None):
    """padding: 'SAME' | 'VALID' | 'SAME_EXPLICIT' (slim conv2d_same: pad
    (k_eff-1)//2 before, VALID after) | (pad_top, pad_left, OH, OW).
    compute: MFMA operand arithmetic — None / 'f32' (parity dtype), 'f16', 'bf16' (fp32 tensors, half-precision
    operands, fp32 accumulate: BASELINE configs[4])."""
    N, H, W, C = x_shape
    R, S, C2, K = w_shape
    assert C == C2, (x_shape, w_shape)
    if padding == 'SAME':
        OH, pt = same_pads(H, R, stride, dilation)
        OW, pl = same_pads(W, S, stride, dilation)
    elif padding == 'VALID':
        OH = (H - ((R - 1) * dilation + 1)) // stride + 1
        OW = (W - ((S - 1) * dilation + 1)) // stride + 1
        pt = pl = 0
    elif padding == 'SAME_EXPLICIT':
        keh, kew = (R - 1) * dilation + 1, (S - 1) * dilation + 1
        pt, pl = (keh - 1) // 2, (kew - 1) // 2
        OH = (H + keh - 1 - keh) // stride + 1
        OW = (W + kew - 1 - kew) // stride + 1
    else:
        pt, pl, OH, OW = padding
    return ConvDesc(N, H, W, C, K, R, S, OH, OW, stride, dilation, pt, pl, ACT[act], COMPUTE[compute])


# Winograd F(2x2,3x3) for the wide stride-1 3x3 layers, above a C*K threshold: RPN 1024->512
# 652 -> 334 us, block3 256->256 89 -> 65 us, 128->128 break-even (scripts/bench_winograd.py); whole step
# 9.99 -> 9.31 ms with the threshold at 256*256 (9.45 at 512*512, 9.35 at 128*128).
WINOGRAD = os.environ.get('LUMINOTH_AMD_WINOGRAD', '1') == '1'
# routing threshold on C*K: F(4x4,3x3) (round 3) pays off from 128 x 128 channels on (ResNet block2's 3x3 at 128^2:
# 85 us direct -> 59 us; whole step 7.81 -> 7.54 ms on one box); F(2x2,3x3) only broke even there (256 x 256 in round 2)
WINOGRAD_MIN_CK = int(os.environ.get('LUMINOTH_AMD_WINOGRAD_MIN_CK', str(128 * 128)))


def winograd_ok(d):
    return bool(_lib.load().lmh_conv2d_winograd_ok(ctypes.byref(d)))


# bf16x3 layers that qualify for Winograd F(2x2,3x3): '1' = run them as native fp32 Winograd (fp32 GEMMs), '3' = Winograd
# with the 16 transformed-domain GEMMs in bf16x3, '0' = direct bf16x3 convolution
X3_WINOGRAD_MODE = os.environ.get('LUMINOTH_AMD_X3_WINOGRAD', '3')     # measured 7.83 / 8.16 / 8.23 ms per step for '3' / '1' / '0'


def _use_winograd(d):
    return WINOGRAD and (d.compute == 0 or (d.compute == 3 and X3_WINOGRAD_MODE == '3')) and d.R == 3 and \
        d.C * d.K >= WINOGRAD_MIN_CK and winograd_ok(d)


def winograd_transform_weights(d, w, kscale, backward, out):
    check(_lib.load().lmh_conv2d_winograd_transform_weights(ctypes.byref(d), _p(_f32(w)), _p(kscale),
                                                            int(bool(backward)), _p(out), _stream()),
          'lmh_conv2d_winograd_transform_weights')
    return out


def set_option(name, value):
    """Tuning option of the C library for the CALLING THREAD (include/luminoth_hip.h: lmh_set_option; the process default
    other threads see is lmh_set_default_option, which _lib.load() uses for LMH_OPT_* variables)."""
    global OPTION_VERSION
    check(_lib.load().lmh_set_option(name.encode(), int(value)), 'lmh_set_option')
    OPTION_VERSION += 1


OPTION_VERSION = 0      # bumped by every set_option: launch plans recorded under other options are not replayed


def get_option(name):
    v = ctypes.c_int(0)
    check(_lib.load().lmh_get_option(name.encode(), ctypes.byref(v)), 'lmh_get_option')
    return v.value


def act_bits_ok(channels, act):
    """A layer output can carry an activation bit mask (one bit per element, 32 channels per word)."""
    return bool(act) and act in FUSED_ACTS and channels % 32 == 0


def new_act_bits(rows, channels, device):
    return torch.empty((rows, channels // 32), dtype=torch.int32, device=device)


def _desc_key(d):
    return tuple(getattr(d, f) for f, _ in d._fields_)


def winograd_weights_batch(jobs, backward):
    """jobs: [(w (3,3,C,K), kscale or None, u)] -> every transformed weight tensor `u` in ONE launch (forward: G g G^T;
    backward: of the flipped, BatchNorm-scaled weights).  `u` tensors: new_winograd_u(C, K)."""
    if not jobs:
        return
    arr = (_lib.WinoWeightJob * len(jobs))()
    for i, (w, ks, u) in enumerate(jobs):
        arr[i].w, arr[i].u = w.data_ptr(), u.data_ptr()
        arr[i].kscale = ks.data_ptr() if ks is not None else None
        arr[i].C, arr[i].K = w.shape[2], w.shape[3]
        plan_keep(w, ks, u)
    check(_lib.load().lmh_winograd_transform_weights_batch(arr, len(jobs), int(bool(backward)), _stream()),
          'lmh_winograd_transform_weights_batch')


def new_x3_weights(rs, C, K, device, backward=False):
    """Buffer for the pre-split bf16x3 planes of a (rs, C, K) weight tensor (include/luminoth_hip.h lmh_x3_weights_bytes)."""
    n = _lib.load().lmh_x3_weights_bytes(int(rs), int(C), int(K), int(bool(backward)))
    if n == 0:
        raise _lib.LuminothHipError('pre-split bf16x3 weights need C %% 32 == 0 and K %% 32 == 0 (got %d, %d)' % (C, K))
    return torch.empty((n // 4,), dtype=torch.int32, device=device)


def x3_split_weights_batch(jobs, backward=False):
    """jobs: [(w (..., C, K) fp32 contiguous, rs, out)] -> every layer's three exact bf16 pieces in MFMA fragment order, ONE
    launch (csrc/conv_x3.h k_x3_split_w).  `out`: new_x3_weights(rs, C, K)."""
    if not jobs:
        return
    arr = (_lib.X3WeightJob * len(jobs))()
    for i, (w, rs, out) in enumerate(jobs):
        arr[i].w, arr[i].out = w.data_ptr(), out.data_ptr()
        arr[i].rs, arr[i].C, arr[i].K = int(rs), w.shape[-2], w.shape[-1]
        plan_keep(w, out)
    check(_lib.load().lmh_x3_split_weights_batch(arr, len(jobs), int(bool(backward)), _stream()), 'lmh_x3_split_weights_batch')


def conv2d_fwd_x3w_ok(d):
    return bool(_lib.load().lmh_conv2d_fwd_x3w_supported(ctypes.byref(d)))


def conv2d_fwd_x3w(d, x, w3, scale=None, shift=None, residual=None, out=None, act_bits=None):
    """conv2d_fwd for compute bf16x3 with the layer's PRE-SPLIT weights `w3` (x3_split_weights_batch): bit-identical."""
    y = out if out is not None else torch.empty((d.N, d.OH, d.OW, d.K), dtype=torch.float32, device=x.device)
    with _timed(d, 0):
        check(_lib.load().lmh_conv2d_fwd_x3w(ctypes.byref(d), _p(_f32(x)), _p(w3), _p(scale), _p(shift), _p(residual), _p(y),
                                             _p(act_bits), _stream()), 'lmh_conv2d_fwd_x3w')
    return y


def conv2d_bwd_data_x3w_ok(d):
    return bool(_lib.load().lmh_conv2d_bwd_data_x3w_supported(ctypes.byref(d)))


def conv2d_bwd_data_x3w(d, dy, w3, kscale=None, addend=None, xbits=None, out=None):
    """conv2d_bwd_data for compute bf16x3 with the layer's weights pre-split in the BACKWARD arrangement: bit-identical."""
    dx = out if out is not None else torch.empty((d.N, d.H, d.W, d.C), dtype=torch.float32, device=dy.device)
    with _timed(d, 1):
        check(_lib.load().lmh_conv2d_bwd_data_x3w(ctypes.byref(d), _p(_f32(dy)), _p(w3), _p(kscale), _p(addend), _p(xbits),
                                                  _p(dx), _stream()), 'lmh_conv2d_bwd_data_x3w')
    return dx


def new_winograd_u(C, K, device):
    return torch.empty((_lib.load().lmh_winograd_u_bytes(int(C), int(K)) // 4,), dtype=torch.float32, device=device)


def conv2d_fwd_winograd(d, x, w, scale=None, shift=None, residual=None, out=None, act_bits=None, keep_v=False, u=None):
    """keep_v: the transformed input planes B^T x B are written to a tensor of their own and left on `x` (attribute
    `_lmh_wino_v`): this layer's Winograd weight gradient needs exactly them again (conv2d_bwd_weight_winograd)."""
    lib = _lib.load()
    y = out if out is not None else torch.empty((d.N, d.OH, d.OW, d.K), dtype=torch.float32, device=x.device)
    ws = _workspace(lib.lmh_conv2d_winograd_workspace_bytes(ctypes.byref(d)), x.device, 'winograd')
    # u: transformed weights prepared ahead (winograd_weights_batch); None: produced inside the call
    v = None
    if keep_v:
        v = torch.empty((lib.lmh_conv2d_winograd_v_bytes(ctypes.byref(d)) // 4,), dtype=torch.float32, device=x.device)
    with _timed(d, 0, wino=True):
        check(lib.lmh_conv2d_fwd_winograd(ctypes.byref(d), _p(_f32(x)), _p(_f32(w)), _p(u), _p(scale), _p(shift),
                                          _p(residual), _p(y), _p(act_bits), _p(v), _p(ws), ctypes.c_size_t(ws.numel()),
                                          _stream()), 'lmh_conv2d_fwd_winograd')
    if v is not None:
        x._lmh_wino_v = (_desc_key(d), get_option('wino_m'), x._version, v)
    return y


def conv2d_bwd_data_winograd(d, dy, w, kscale=None, addend=None, out=None, xbits=None, u=None):
    lib = _lib.load()
    dx = out if out is not None else torch.empty((d.N, d.H, d.W, d.C), dtype=torch.float32, device=dy.device)
    ws = _workspace(lib.lmh_conv2d_winograd_workspace_bytes(ctypes.byref(d)), dy.device, 'winograd')
    with _timed(d, 1, wino=True):
        check(lib.lmh_conv2d_bwd_data_winograd(ctypes.byref(d), _p(_f32(dy)), _p(_f32(w)), _p(u), _p(kscale),
                                               _p(addend), _p(xbits), _p(dx), _p(ws), ctypes.c_size_t(ws.numel()),
                                               _stream()), 'lmh_conv2d_bwd_data_winograd')
    return dx


# weight gradient: 16 reduction GEMMs over T tiles instead of 9 over 4T pixels; pays off from 128 channels on
# (128->128 at 128^2: 117 -> 88 us, 256->256: 105 -> 72, RPN: 778 -> 389)
WINOGRAD_WGRAD = os.environ.get('LUMINOTH_AMD_WINOGRAD_WGRAD', '1') == '1'
WINOGRAD_WGRAD_MIN_CK = int(os.environ.get('LUMINOTH_AMD_WINOGRAD_WGRAD_MIN_CK', str(128 * 128)))


def conv2d_bwd_weight_winograd(d, x, dy, out=None, colsum=None):
    """RAW weight gradient (w.r.t. the un-scaled convolution output), like conv2d_bwd_weight.  colsum (K,): WRITTEN
    with the per-channel sums of dy (one tiny reduction over the tiles' pixel sums, no pass over dy).  The transformed
    input planes are taken from `x._lmh_wino_v` when the forward pass left them there for this very x."""
    lib = _lib.load()
    dw = out if out is not None else torch.empty((3, 3, d.C, d.K), dtype=torch.float32, device=x.device)
    ws = _workspace(lib.lmh_conv2d_bwd_weight_winograd_workspace_bytes(ctypes.byref(d)), x.device, 'winograd_w')
    v = None
    kept = getattr(x, '_lmh_wino_v', None)
    if kept is not None:
        dk = _desc_key(d)
        # same geometry (the activation / compute fields do not enter the transform), same tile size, x unmodified
        same = all(a == b for a, b, (f, _) in zip(kept[0], dk, d._fields_) if f not in ('act', 'compute'))
        if same and kept[1] == get_option('wino_m') and kept[2] == x._version:
            v = kept[3]
            keep_alive(v, effective_stream(x.device))
    with _timed(d, 2, wino=True):
        check(lib.lmh_conv2d_bwd_weight_winograd(ctypes.byref(d), _p(_f32(x)), _p(_f32(dy)), _p(dw), _p(v), _p(colsum),
                                                 _p(ws), ctypes.c_size_t(ws.numel()), _stream()),
              'lmh_conv2d_bwd_weight_winograd')
    return dw


def conv2d_fwd(d, x, w, scale=None, shift=None, residual=None, in_sub=None, out=None, act_bits=None, keep_v=False,
               wino_u=None):
    """act_bits (int32 (rows, K/32), optional): WRITTEN with the activation bit mask of y (bit = act'(y) != 0) — what the
    backward pass needs of the activation; the consumer's backward-data epilogue applies it (conv2d_bwd_data xbits)."""
    lib = _lib.load()
    if in_sub is None and _use_winograd(d):
        return conv2d_fwd_winograd(d, x, w, scale, shift, residual, out, act_bits,
                                   keep_v=keep_v and WINOGRAD_WGRAD and d.C * d.K >= WINOGRAD_WGRAD_MIN_CK, u=wino_u)
    y = out if out is not None else torch.empty((d.N, d.OH, d.OW, d.K), dtype=torch.float32, device=x.device)
    with _timed(d, 0):
        check(lib.lmh_conv2d_fwd(ctypes.byref(d), _p(_f32(x)), _p(_f32(w)), _p(scale), _p(shift), _p(residual),
                                 _p(in_sub), _p(y), _p(act_bits), _stream()), 'lmh_conv2d_fwd')
    return y


def act_bits(y, act):
    """The bit mask of an activation tensor on its own (producers that are not a convolution of this library)."""
    K = y.shape[-1]
    rows = y.numel() // K
    bits = new_act_bits(rows, K, y.device)
    check(_lib.load().lmh_act_bits(_p(y), ACT[act], rows, K, _p(bits), _stream()), 'lmh_act_bits')
    return bits


def apply_act_bits(dx, bits):
    """dx <- dx where the bit is set else 0, in place."""
    C = dx.shape[-1]
    check(_lib.load().lmh_apply_act_bits(_p(dx), _p(bits), dx.numel() // C, C, _stream()), 'lmh_apply_act_bits')
    return dx


def conv_fused_act_ok(d):
    """True when both backward kernels of `d` take the fused `yact` / `colsum` operands (fast paths)."""
    lib = _lib.load()
    return d.act != 0 and lib.lmh_conv2d_kernel_id(ctypes.byref(d), 1) < 1000000 and \
        lib.lmh_conv2d_kernel_id(ctypes.byref(d), 2) < 1000000


def conv_fused_colsum_ok(d):
    """True when bwd_weight of `d` can emit the per-channel sums of its dy operand itself (fp32 fast paths); other
    layers take them from the lmh_act_bwd pass that makes g."""
    return bool(_lib.load().lmh_conv2d_bwd_weight_fuses_colsum(ctypes.byref(d)))


def conv_bwd_data_fast(d):
    return _lib.load().lmh_conv2d_kernel_id(ctypes.byref(d), 1) < 1000000


def conv2d_bwd_data(d, dy, w, kscale=None, addend=None, out=None, yact=None, xbits=None, wino_u=None):
    """yact: layer output y -> the kernel applies g = dy*act'(y) on load (fused activation backward).
    xbits (the activation bit mask of the layer input x, written by the forward kernel that produced x): the result is
    dx * act'(x), i.e. the pre-activation gradient of the layer below (epilogue of the fast / Winograd kernels; one
    in-place pass inside the C call for the others)."""
    lib = _lib.load()
    if yact is None and _use_winograd(d):
        return conv2d_bwd_data_winograd(d, dy, w, kscale, addend, out, xbits, u=wino_u)
    dx = out if out is not None else torch.empty((d.N, d.H, d.W, d.C), dtype=torch.float32, device=dy.device)
    with _timed(d, 1):
        check(lib.lmh_conv2d_bwd_data(ctypes.byref(d), _p(_f32(dy)), _p(_f32(w)), _p(kscale), _p(addend), _p(yact),
                                      _p(xbits), _p(dx), _stream()), 'lmh_conv2d_bwd_data')
    return dx


def conv2d_bwd_weight(d, x, dy, out=None, yact=None, colsum=None, defer=None):
    """yact: fused g = dy*act'(y); colsum (K,): WRITTEN with the per-channel sums of g.
    defer: layer key — with TAILS.active the split-K reduction (and the last stage of `colsum`) is queued instead of
    launched; the slabs then live in a workspace of their own (per layer) until TAILS.flush()."""
    lib = _lib.load()
    defer = defer if (defer is not None and TAILS.active and d.K % 4 == 0 and d.K <= 4096) else None
    if yact is None and WINOGRAD and WINOGRAD_WGRAD and (d.compute == 0 or (d.compute == 3 and X3_WINOGRAD_MODE == '3')) and d.R == 3 and \
            d.C * d.K >= WINOGRAD_WGRAD_MIN_CK and winograd_ok(d):
        dw = conv2d_bwd_weight_winograd(d, x, dy, out, colsum=colsum)      # dbeta / dbias from the tiles' pixel sums
        if defer is not None:
            TAILS.entry(defer).update(dw=dw, slabs=0, splits=0)
        return dw
    dw = out if out is not None else torch.empty((d.R, d.S, d.C, d.K), dtype=torch.float32, device=x.device)
    nbytes = lib.lmh_conv2d_bwd_weight_workspace_bytes(ctypes.byref(d))
    ws = _workspace(nbytes, x.device, 'bwd_weight' if defer is None else ('wgslab', defer))
    if defer is not None:
        lib.lmh_tail_defer(1)
    try:
        with _timed(d, 2):
            check(lib.lmh_conv2d_bwd_weight(ctypes.byref(d), _p(_f32(x)), _p(_f32(dy)), _p(yact), _p(dw), _p(colsum),
                                            _p(ws), ctypes.c_size_t(ws.numel()), _stream()), 'lmh_conv2d_bwd_weight')
        if defer is not None:
            slabs, splits, colpart, colrows = _last_plan()
            e = TAILS.entry(defer)
            e.update(dw=dw, slabs=slabs, splits=splits, _ws=ws)
            if colsum is not None and colrows:
                e.update(colpart=colpart, colrows=colrows, colsum=colsum)
    finally:
        if defer is not None:
            lib.lmh_tail_defer(0)
    return dw


ACT_GRAD_FROM_INPUT = ('softplus', 'softsign')   # act_bwd takes the PRE-activation for these (TF's SoftplusGrad / SoftsignGrad)


def act_fwd(z, act, out=None):
    """act(z) for any activation id of ACT (the ones no convolution epilogue fuses: codes 3..9); out=z works in place."""
    assert z.dtype == torch.float32
    y = torch.empty_like(z) if out is None else out
    check(_lib.load().lmh_act_fwd(_p(z), _p(y), ACT[act], z.numel(), _stream()), 'lmh_act_fwd')
    return y


def act_bwd(dy, y, act, want_g=True, colsum=None, defer=None):
    """g = dy * act'(y); colsum (K,) is WRITTEN with the per-channel sums of g (with `defer` + TAILS.active: its
    last stage is queued; the partial rows wait in a per-layer workspace).  For the activations of ACT_GRAD_FROM_INPUT `y`
    is the pre-activation."""
    lib = _lib.load()
    K = dy.shape[-1]
    rows = dy.numel() // K
    g = torch.empty_like(dy) if want_g else None
    ws, nbytes = None, 0
    defer = defer if (defer is not None and colsum is not None and TAILS.active and K % 4 == 0 and K <= 4096) else None
    if colsum is not None:
        nbytes = lib.lmh_act_bwd_workspace_bytes(rows, K)
        ws = _workspace(nbytes, dy.device, 'colsum' if defer is None else ('colpart', defer))
    if defer is not None:
        lib.lmh_tail_defer(1)
    try:
        check(lib.lmh_act_bwd(_p(dy), _p(y), ACT[act], rows, K, _p(g), _p(colsum), _p(ws),
                              ctypes.c_size_t(0 if ws is None else ws.numel()), _stream()), 'lmh_act_bwd')
        if defer is not None:
            _, _, colpart, colrows = _last_plan()
            TAILS.entry(defer).update(colpart=colpart, colrows=colrows, colsum=colsum, _cws=ws)
    finally:
        if defer is not None:
            lib.lmh_tail_defer(0)
    return g


def bn_param_grads(w, dw_raw, dbeta, mean, rstd, scale, out=None):
    lib = _lib.load()
    K = w.shape[-1]
    rsc = w.numel() // K
    dgamma = out if out is not None else torch.empty_like(dbeta)
    ws = _workspace(lib.lmh_bn_param_grads_workspace_bytes(rsc, K), w.device, 'bn_wdot')
    check(lib.lmh_bn_param_grads(_p(w), _p(dw_raw), _p(dbeta), _p(mean), _p(rstd), _p(scale), rsc, K,
                                 _p(dgamma), _p(ws), ctypes.c_size_t(ws.numel()), _stream()),
          'lmh_bn_param_grads')
    return dgamma


# ------------------------------------------------- half-storage convolutions ----
# f16 / bf16 tensors in HBM (include/luminoth_hip.h "Half-STORAGE convolution path", csrc/conv_hs.h): BASELINE configs[4].
_HALF = {'f16': (1, torch.float16), 'bf16': (2, torch.bfloat16)}


def half_type(storage):
    return _HALF[storage]


def half_type_of(t):
    for code, tdt in _HALF.values():
        if t.dtype == tdt:
            return code, tdt
    raise _lib.LuminothHipError('not a half-storage tensor: %s' % t.dtype)


def _half(t, tdt):
    assert t.dtype == tdt, (t.dtype, tdt)
    return t


def conv_hs_ok(d):
    return bool(_lib.load().lmh_conv2d_hs_supported(ctypes.byref(d)))


def hs_fragment_order(b_nq):
    """The layout lmh_half_weights_batch gives a working copy when C % 64 == 0 and K % 64 == 0 (include/luminoth_hip.h,
    lmh_half_weight_job): B (N, Q) -> flat [N/32][Q/64][4 k-steps][lane = 32 * (q % 16 // 8) + n % 32][q % 8] — the B operand
    fragments of v_mfma_f32_32x32x16_*.  Host-side restatement for tests and tools (the product never needs it)."""
    N, Q = b_nq.shape
    assert N % 32 == 0 and Q % 64 == 0, (N, Q)
    return b_nq.reshape(N // 32, 32, Q // 64, 4, 2, 8).permute(0, 2, 3, 4, 1, 5).contiguous().reshape(-1)


def half_weights_batch(jobs, storage):
    """jobs: [(w (R,S,C,K) fp32, kscale (K,) or None, w_fwd half (K*R*S*C elements) or None, w_bwd half or None)]: the
    working copies of every layer in one launch (per 48 layers).  The copies are opaque operands of conv2d_fwd_hs /
    conv2d_bwd_data_hs (fragment order, hs_fragment_order, for the shapes those accept)."""
    if not jobs:
        return
    code, tdt = half_type(storage)
    arr = (_lib.HalfWeightJob * len(jobs))()
    for i, (w, ks, wf, wb) in enumerate(jobs):
        arr[i].w = _f32(w).data_ptr()
        arr[i].kscale = ks.data_ptr() if ks is not None else None
        arr[i].w_fwd = _half(wf, tdt).data_ptr() if wf is not None else None
        arr[i].w_bwd = _half(wb, tdt).data_ptr() if wb is not None else None
        arr[i].RS, arr[i].C, arr[i].K = w.shape[0] * w.shape[1], w.shape[2], w.shape[3]
        plan_keep(w, ks, wf, wb)
    check(_lib.load().lmh_half_weights_batch(arr, len(jobs), code, _stream()), 'lmh_half_weights_batch')


def cast_to_half(x, storage, mul=1.0, bits=None):
    code, tdt = half_type(storage)
    C = x.shape[-1]
    y = torch.empty(x.shape, dtype=tdt, device=x.device)
    check(_lib.load().lmh_cast_to_half(_p(_f32(x)), x.numel() // C, C, float(mul), _p(bits), _p(y), code, _stream()),
          'lmh_cast_to_half')
    return y


def cast_to_f32(x, mul=1.0):
    code, _ = half_type_of(x)
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(_lib.load().lmh_cast_to_f32(_p(x), x.numel(), float(mul), _p(y), code, _stream()), 'lmh_cast_to_f32')
    return y


def conv2d_fwd_hs(d, x, w_fwd, scale=None, shift=None, residual=None, out_f32=False, act_bits=None, out=None):
    code, tdt = half_type_of(x)
    assert code == d.compute, (code, d.compute)
    y = out if out is not None else torch.empty((d.N, d.OH, d.OW, d.K), dtype=torch.float32 if out_f32 else tdt,
                                                device=x.device)
    assert y.dtype == (torch.float32 if out_f32 else tdt) and tuple(y.shape) == (d.N, d.OH, d.OW, d.K)
    with _timed(d, 0):
        check(_lib.load().lmh_conv2d_fwd_hs(ctypes.byref(d), _p(x), _p(_half(w_fwd, tdt)), _p(scale), _p(shift),
                                            _p(None if residual is None else _half(residual, tdt)), _p(y), int(bool(out_f32)),
                                            _p(act_bits), _stream()), 'lmh_conv2d_fwd_hs')
    return y


def conv2d_bwd_data_hs(d, g, w_bwd, addend=None, xbits=None, out_f32=False, mul=1.0):
    """out_f32: the data gradient leaves as an fp32 tensor (mul = 1 / loss scale): a half-storage layer fed by an fp32 one."""
    code, tdt = half_type_of(g)
    assert code == d.compute, (code, d.compute)
    dx = torch.empty((d.N, d.H, d.W, d.C), dtype=torch.float32 if out_f32 else tdt, device=g.device)
    with _timed(d, 1):
        check(_lib.load().lmh_conv2d_bwd_data_hs(ctypes.byref(d), _p(g), _p(_half(w_bwd, tdt)),
                                                 _p(None if addend is None else _half(addend, tdt)), _p(xbits), _p(dx),
                                                 int(bool(out_f32)), float(mul), _stream()), 'lmh_conv2d_bwd_data_hs')
    return dx


def conv2d_bwd_weight_hs(d, x, g, inv_scale, out=None, colsum=None, defer=None):
    """fp32 RAW weight gradient of half tensors x, g (g carries the loss scale 1 / inv_scale); colsum, defer: as
    conv2d_bwd_weight."""
    lib = _lib.load()
    code, tdt = half_type_of(x)
    assert code == d.compute and g.dtype == tdt, (code, d.compute, g.dtype)
    defer = defer if (defer is not None and TAILS.active and d.K % 4 == 0 and d.K <= 4096) else None
    dw = out if out is not None else torch.empty((d.R, d.S, d.C, d.K), dtype=torch.float32, device=x.device)
    nbytes = lib.lmh_conv2d_bwd_weight_workspace_bytes(ctypes.byref(d))
    ws = _workspace(nbytes, x.device, 'bwd_weight' if defer is None else ('wgslab', defer))
    if defer is not None:
        lib.lmh_tail_defer(1)
    try:
        with _timed(d, 2):
            check(lib.lmh_conv2d_bwd_weight_hs(ctypes.byref(d), _p(x), _p(g), float(inv_scale), _p(_f32(dw)), _p(colsum), _p(ws),
                                               ctypes.c_size_t(ws.numel()), _stream()), 'lmh_conv2d_bwd_weight_hs')
        if defer is not None:
            slabs, splits, colpart, colrows = _last_plan()
            e = TAILS.entry(defer)
            e.update(dw=dw, slabs=slabs, splits=splits, _ws=ws)
            if colsum is not None and colrows:
                e.update(colpart=colpart, colrows=colrows, colsum=colsum)
    finally:
        if defer is not None:
            lib.lmh_tail_defer(0)
    return dw


def maxpool_fwd(x, ksize, stride, padding='SAME', storage=None):
    """storage 'f16' / 'bf16': the result is a half tensor (x fp32 or that half type) — half-storage trunk."""
    lib = _lib.load()
    N, H, W, C = x.shape
    if padding == 'SAME':
        OH, pt = same_pads(H, ksize, stride)
        OW, pl = same_pads(W, ksize, stride)
    else:
        OH, OW, pt, pl = (H - ksize) // stride + 1, (W - ksize) // stride + 1, 0, 0
    if storage is not None or x.dtype != torch.float32:
        code, tdt = half_type(storage) if storage is not None else half_type_of(x)
        assert x.dtype in (torch.float32, tdt), (x.dtype, tdt)
        y = torch.empty((N, OH, OW, C), dtype=tdt, device=x.device)
        check(lib.lmh_maxpool_fwd_hs(_p(x), int(x.dtype == torch.float32), N, H, W, C, ksize, stride, pt, pl, OH, OW, _p(y),
                                     code, _stream()), 'lmh_maxpool_fwd_hs')
        return y, (pt, pl, OH, OW)
    y = torch.empty((N, OH, OW, C), dtype=torch.float32, device=x.device)
    check(lib.lmh_maxpool_fwd(_p(x), N, H, W, C, ksize, stride, pt, pl, OH, OW, _p(y), _stream()),
          'lmh_maxpool_fwd')
    return y, (pt, pl, OH, OW)


def resize_bilinear(image, out_h, out_w, flip_lr=False, flip_ud=False):
    """tf.image.resize_images(BILINEAR) (TF 1.x legacy sampling) of one (H,W,C) uint8/float32 image, optionally
    of its left-right / up-down flip."""
    lib = _lib.load()
    assert image.dim() == 3 and image.is_cuda and image.dtype in (torch.uint8, torch.float32), \
        (image.shape, image.dtype, image.device)
    image = image.contiguous()
    H, W, C = image.shape
    out = torch.empty((int(out_h), int(out_w), C), dtype=torch.float32, device=image.device)
    check(lib.lmh_resize_bilinear(_p(image), int(image.dtype == torch.uint8), H, W, C, _p(out), int(out_h),
                                  int(out_w), int(bool(flip_lr)), int(bool(flip_ud)), _stream()),
          'lmh_resize_bilinear')
    return out


def maxpool_bwd(x, y, dy, ksize, stride, geom):
    lib = _lib.load()
    N, H, W, C = x.shape
    pt, pl, OH, OW = geom
    if dy.dtype != torch.float32:           # half storage: only the 1x1 subsample of a bottleneck shortcut has a backward
        if ksize != 1:
            raise NotImplementedError('half-storage max-pool backward: 1x1 subsample only (the 3x3 pool of the ResNet '
                                      'stem is in the frozen prefix)')
        dx = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
        check(lib.lmh_subsample_bwd_hs(_p(dy), N, H, W, C, stride, OH, OW, _p(dx), _stream()), 'lmh_subsample_bwd_hs')
        return dx
    dx = zero_(torch.empty_like(x))
    check(lib.lmh_maxpool_bwd(_p(x), _p(y), _p(dy), N, H, W, C, ksize, stride, pt, pl, OH, OW, _p(dx),
                              _stream()), 'lmh_maxpool_bwd')
    return dx


# ------------------------------------------------------------- proposals ----
def rpn_proposal(cls_score, bbox_pred, anchor_ref_i32, feat_h, feat_w, stride, im_shape,
                 pre_nms_top_n=12000, post_nms_top_n=2000, nms_threshold=0.7, min_prob_threshold=0.0,
                 apply_nms=True, clip_after_nms=False, filter_outside_anchors=False):
    """cls_score (B,N,2), bbox_pred (B,N,4) -> cls_prob (B,N,2), proposals (B,post,4), scores, count (B)."""
    lib = _lib.load()
    B, N, _ = cls_score.shape
    A = anchor_ref_i32.shape[0]
    assert N == feat_h * feat_w * A
    d = RpnProposalDesc(B, feat_h, feat_w, A, stride, float(im_shape[0]), float(im_shape[1]),
                        int(pre_nms_top_n), int(post_nms_top_n), float(nms_threshold),
                        float(min_prob_threshold), int(bool(apply_nms)), int(bool(clip_after_nms)),
                        int(bool(filter_outside_anchors)))
    dev = cls_score.device
    cls_prob = torch.empty_like(cls_score)
    cap = int(post_nms_top_n if apply_nms else pre_nms_top_n)
    proposals = torch.empty((B, cap, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((B, cap), dtype=torch.float32, device=dev)
    count = torch.empty((B,), dtype=torch.int32, device=dev)
    ws = _workspace(lib.lmh_rpn_proposal_workspace_bytes(ctypes.byref(d)), dev, 'rpn_proposal')
    check(lib.lmh_rpn_proposal(ctypes.byref(d), _p(cls_score), _p(bbox_pred), _p(anchor_ref_i32), _p(cls_prob),
                               _p(proposals), _p(scores), _p(count), _p(ws), ctypes.c_size_t(ws.numel()),
                               _stream()), 'lmh_rpn_proposal')
    return cls_prob, proposals, scores, count


def sort_u64(keys):
    lib = _lib.load()
    B, n = keys.shape
    assert keys.dtype == torch.int64 and (n & (n - 1)) == 0
    check(lib.lmh_sort_u64(_p(keys), B, n, _stream()), 'lmh_sort_u64')
    return keys


def nms(boxes, counts, iou_threshold, max_out):
    """boxes (B,K,4) sorted by descending score; counts (B) int32."""
    lib = _lib.load()
    B, K, _ = boxes.shape
    keep_idx = torch.empty((B, max_out), dtype=torch.int32, device=boxes.device)
    keep_count = torch.empty((B,), dtype=torch.int32, device=boxes.device)
    ws = _workspace(lib.lmh_nms_workspace_bytes(B, K), boxes.device, 'nms')
    check(lib.lmh_nms(_p(boxes), _p(counts), B, K, float(iou_threshold), int(max_out), _p(keep_idx),
                      _p(keep_count), _p(ws), ctypes.c_size_t(ws.numel()), _stream()), 'lmh_nms')
    return keep_idx, keep_count


# --------------------------------------------------------------- targets ----
def rpn_target(anchor_ref_i32, feat_h, feat_w, stride, gt, gt_count, seeds, im_shape, allowed_border=0,
               clobber_positives=False, foreground_threshold=0.7, background_threshold_high=0.3,
               foreground_fraction=0.5, minibatch_size=256, want_pre=False, out=None):
    """out: optional (labels, targets, max_ov) tensors to write into (fixed addresses for the cross-step prefetch)."""
    lib = _lib.load()
    B, Gmax, _ = gt.shape
    A = anchor_ref_i32.shape[0]
    N = feat_h * feat_w * A
    d = RpnTargetDesc(B, feat_h, feat_w, A, stride, Gmax, int(im_shape[0]), int(im_shape[1]),
                      int(allowed_border), int(bool(clobber_positives)), float(foreground_threshold),
                      float(background_threshold_high), float(foreground_fraction), int(minibatch_size))
    dev = gt.device
    if out is not None:
        labels, targets, max_ov = out
        assert labels.shape == (B, N) and targets.shape == (B, N, 4) and max_ov.shape == (B, N)
    else:
        labels = torch.empty((B, N), dtype=torch.float32, device=dev)
        targets = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
        max_ov = torch.empty((B, N), dtype=torch.float32, device=dev)
    pre = torch.empty((B, N), dtype=torch.float32, device=dev) if want_pre else None
    ws = _workspace(lib.lmh_rpn_target_workspace_bytes(ctypes.byref(d)), dev, 'rpn_target')
    check(lib.lmh_rpn_target(ctypes.byref(d), _p(anchor_ref_i32), _p(gt), _p(gt_count), _p(seeds), _p(labels),
                             _p(targets), _p(max_ov), _p(pre), _p(ws), ctypes.c_size_t(ws.numel()),
                             _stream()), 'lmh_rpn_target')
    return labels, targets, max_ov, pre


def rcnn_target(proposals, prop_count, gt, gt_count, seeds, minibatch_size=256, foreground_fraction=0.25,
                foreground_threshold=0.5, background_threshold_high=0.5, background_threshold_low=0.0,
                variances=(0.1, 0.2), want_pre=False):
    lib = _lib.load()
    B, Pn, _ = proposals.shape
    Gmax = gt.shape[1]
    v = (1.0, 1.0) if variances is None else variances
    d = RcnnTargetDesc(B, Pn, Gmax, int(minibatch_size), float(foreground_fraction),
                       float(foreground_threshold), float(background_threshold_high),
                       float(background_threshold_low), float(v[0]), float(v[1]))
    dev = proposals.device
    R = int(minibatch_size)
    labels = torch.empty((B, Pn), dtype=torch.float32, device=dev)
    targets = torch.empty((B, Pn, 4), dtype=torch.float32, device=dev)
    pre = torch.empty((B, Pn), dtype=torch.float32, device=dev) if want_pre else None
    rois = torch.empty((B, R, 4), dtype=torch.float32, device=dev)
    roi_labels = torch.empty((B, R), dtype=torch.float32, device=dev)
    roi_targets = torch.empty((B, R, 4), dtype=torch.float32, device=dev)
    roi_count = torch.empty((B,), dtype=torch.int32, device=dev)
    ws = _workspace(lib.lmh_rcnn_target_workspace_bytes(ctypes.byref(d)), dev, 'rcnn_target')
    check(lib.lmh_rcnn_target(ctypes.byref(d), _p(proposals), _p(prop_count), _p(gt), _p(gt_count), _p(seeds),
                              _p(labels), _p(targets), _p(pre), _p(rois), _p(roi_labels), _p(roi_targets),
                              _p(roi_count), _p(ws), ctypes.c_size_t(ws.numel()), _stream()), 'lmh_rcnn_target')
    return dict(labels=labels, bbox_targets=targets, labels_pre=pre, rois=rois, roi_labels=roi_labels,
                roi_targets=roi_targets, roi_count=roi_count)


def rcnn_proposal(proposals, prop_count, bbox_pred, cls_prob, im_shape, num_classes, variances=(0.1, 0.2),
                  class_max_detections=100, class_nms_threshold=0.5, total_max_detections=300,
                  min_prob_threshold=0.5, class_agnostic_boxes=False):
    """proposals (B,R,4), bbox_pred (B,R,4C), cls_prob (B,R,C+1) -> objects (B,T,4), labels, probs, num."""
    lib = _lib.load()
    B, R, _ = proposals.shape
    v = (1.0, 1.0) if variances is None else variances
    d = RcnnProposalDesc(B, R, int(num_classes), float(im_shape[0]), float(im_shape[1]), float(v[0]),
                         float(v[1]), int(class_max_detections), float(class_nms_threshold),
                         int(total_max_detections), float(min_prob_threshold or 0.0),
                         int(bool(class_agnostic_boxes)))
    dev = proposals.device
    T = int(total_max_detections)
    objects = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
    labels = torch.empty((B, T), dtype=torch.int32, device=dev)
    probs = torch.empty((B, T), dtype=torch.float32, device=dev)
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    ws = _workspace(lib.lmh_rcnn_proposal_workspace_bytes(ctypes.byref(d)), dev, 'rcnn_proposal')
    check(lib.lmh_rcnn_proposal(ctypes.byref(d), _p(proposals), _p(prop_count), _p(bbox_pred), _p(cls_prob),
                                _p(objects), _p(labels), _p(probs), _p(num), _p(ws),
                                ctypes.c_size_t(ws.numel()), _stream()), 'lmh_rcnn_proposal')
    return objects, labels, probs, num


def ssd_proposal(anchors, anchor_count, loc_pred, cls_prob, im_shape, num_classes, variances=(0.1, 0.2),
                 class_max_detections=100, class_nms_threshold=0.45, total_max_detections=100,
                 min_prob_threshold=0.5):
    """SSDProposal._build (ssd/proposal.py:41-171) with all five keys of its return dict: anchors (B,N,4),
    loc_pred (B,N,4), cls_prob (B,N,C+1) -> objects (B,T,4), labels, probs, num_objects, raw_proposals (B,N,4) +
    num_raw_proposals, anchors (B,T,4)."""
    lib = _lib.load()
    B, R, _ = anchors.shape
    v = (1.0, 1.0) if variances is None else variances
    d = RcnnProposalDesc(B, R, int(num_classes), float(im_shape[0]), float(im_shape[1]), float(v[0]),
                         float(v[1]), int(class_max_detections), float(class_nms_threshold),
                         int(total_max_detections), float(min_prob_threshold or 0.0), 1)
    dev = anchors.device
    T = int(total_max_detections)
    objects = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
    labels = torch.empty((B, T), dtype=torch.int32, device=dev)
    probs = torch.empty((B, T), dtype=torch.float32, device=dev)
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    raw = torch.empty((B, R, 4), dtype=torch.float32, device=dev)
    raw_count = torch.empty((B,), dtype=torch.int32, device=dev)
    det_anchors = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
    ws = _workspace(lib.lmh_ssd_proposal_workspace_bytes(ctypes.byref(d)), dev, 'rcnn_proposal')
    check(lib.lmh_ssd_proposal(ctypes.byref(d), _p(anchors), _p(anchor_count), _p(loc_pred), _p(cls_prob),
                               _p(objects), _p(labels), _p(probs), _p(num), _p(raw), _p(raw_count),
                               _p(det_anchors), _p(ws), ctypes.c_size_t(ws.numel()), _stream()), 'lmh_ssd_proposal')
    return {'objects': objects, 'labels': labels, 'probs': probs, 'num_objects': num, 'raw_proposals': raw,
            'num_raw_proposals': raw_count, 'anchors': det_anchors}


# ------------------------------------------------------------------- ROI ----
def roi_pool_fwd(feat, rois, roi_count, im_shape, ph=7, pw=7):
    lib = _lib.load()
    B, FH, FW, C = feat.shape
    R = rois.shape[1]
    out = torch.empty((B * R, ph, pw, C), dtype=torch.float32, device=feat.device)
    argmax = torch.empty((B * R, ph, pw, C), dtype=torch.uint8, device=feat.device)
    check(lib.lmh_roi_pool_fwd(_p(feat), _p(rois), _p(roi_count), B, R, FH, FW, C, float(im_shape[0]),
                               float(im_shape[1]), ph, pw, _p(out), _p(argmax), _stream()), 'lmh_roi_pool_fwd')
    return out, argmax


def roi_pool_bwd(dout, argmax, rois, roi_count, feat_shape, im_shape, ph=7, pw=7, out=None, addend=None):
    """addend (feat_shape, optional): added to the result in the same store (the other consumer's gradient of the map)."""
    lib = _lib.load()
    B, FH, FW, C = feat_shape
    R = rois.shape[1]
    dfeat = out if out is not None else torch.empty(feat_shape, dtype=torch.float32, device=dout.device)   # overwritten
    ws = _workspace(lib.lmh_roi_pool_bwd_workspace_bytes(B, R, ph, pw), dout.device, 'roi_bwd')
    check(lib.lmh_roi_pool_bwd(_p(dout), _p(argmax), _p(rois), _p(roi_count), B, R, FH, FW, C,
                               float(im_shape[0]), float(im_shape[1]), ph, pw, _p(addend), _p(dfeat), _p(ws),
                               ctypes.c_size_t(ws.numel()), _stream()), 'lmh_roi_pool_bwd')
    return dfeat


def roi_pool_mean_supported(feat_shape):
    _, FH, FW, C = feat_shape
    return bool(_lib.load().lmh_roi_pool_mean_supported(FH, FW, C))


def roi_pool_mean_fwd(feat, rois, roi_count, im_shape, ph=7, pw=7):
    """reduce_mean over the cells of roi_pool_fwd's output, without that output (see roi.hip)."""
    lib = _lib.load()
    B, FH, FW, C = feat.shape
    R = rois.shape[1]
    mean = torch.empty((B * R, C), dtype=torch.float32, device=feat.device)
    argmax = torch.empty((B * R, ph, pw, C), dtype=torch.uint8, device=feat.device)
    check(lib.lmh_roi_pool_mean_fwd(_p(feat), _p(rois), _p(roi_count), B, R, FH, FW, C, float(im_shape[0]),
                                    float(im_shape[1]), ph, pw, _p(mean), _p(argmax), _stream()), 'lmh_roi_pool_mean_fwd')
    return mean, argmax


def roi_pool_mean_bwd(dmean, argmax, rois, roi_count, feat_shape, im_shape, ph=7, pw=7, addend=None):
    lib = _lib.load()
    B, FH, FW, C = feat_shape
    R = rois.shape[1]
    dfeat = torch.empty(feat_shape, dtype=torch.float32, device=dmean.device)   # overwritten
    ws = _workspace(lib.lmh_roi_pool_bwd_workspace_bytes(B, R, ph, pw), dmean.device, 'roi_bwd')
    check(lib.lmh_roi_pool_mean_bwd(_p(dmean), _p(argmax), _p(rois), _p(roi_count), B, R, FH, FW, C,
                                    float(im_shape[0]), float(im_shape[1]), ph, pw, _p(addend), _p(dfeat), _p(ws),
                                    ctypes.c_size_t(ws.numel()), _stream()), 'lmh_roi_pool_mean_bwd')
    return dfeat


def spatial_mean_fwd(x):
    lib = _lib.load()
    M, S, C = x.shape[0], x.shape[1] * x.shape[2], x.shape[3]
    y = torch.empty((M, C), dtype=torch.float32, device=x.device)
    check(lib.lmh_spatial_mean_fwd(_p(x), M, S, C, _p(y), _stream()), 'lmh_spatial_mean_fwd')
    return y


def spatial_mean_bwd(dy, shape):
    lib = _lib.load()
    M, S, C = shape[0], shape[1] * shape[2], shape[3]
    dx = torch.empty(shape, dtype=torch.float32, device=dy.device)
    check(lib.lmh_spatial_mean_bwd(_p(dy), M, S, C, _p(dx), _stream()), 'lmh_spatial_mean_bwd')
    return dx


# ---------------------------------------------------------------- losses ----
def rpn_loss(cls_score, bbox_pred, labels, bbox_targets, sigma=3.0, w_cls=1.0, w_reg=1.0, want_grad=True):
    lib = _lib.load()
    B, N, _ = cls_score.shape
    dev = cls_score.device
    losses = torch.empty((2,), dtype=torch.float32, device=dev)
    per_image = torch.empty((B, 4), dtype=torch.float32, device=dev)
    d_cls = torch.empty_like(cls_score) if want_grad else None
    d_bbox = torch.empty_like(bbox_pred) if want_grad else None
    check(lib.lmh_rpn_loss(_p(cls_score), _p(bbox_pred), _p(labels), _p(bbox_targets), B, N, float(sigma),
                           float(w_cls), float(w_reg), _p(losses), _p(per_image), _p(d_cls), _p(d_bbox),
                           _stream()), 'lmh_rpn_loss')
    return losses, per_image, d_cls, d_bbox


def rcnn_loss(cls_score, bbox_offsets, labels, targets, num_classes, sigma=1.0, w_cls=1.0, w_reg=1.0,
              want_grad=True):
    lib = _lib.load()
    B, R = labels.shape
    dev = cls_score.device
    losses = torch.empty((2,), dtype=torch.float32, device=dev)
    per_image = torch.empty((B, 4), dtype=torch.float32, device=dev)
    d_cls = torch.empty_like(cls_score) if want_grad else None
    d_off = torch.empty_like(bbox_offsets) if want_grad else None
    check(lib.lmh_rcnn_loss(_p(cls_score), _p(bbox_offsets), _p(labels), _p(targets), B, R, int(num_classes),
                            float(sigma), float(w_cls), float(w_reg), _p(losses), _p(per_image), _p(d_cls),
                            _p(d_off), _stream()), 'lmh_rcnn_loss')
    return losses, per_image, d_cls, d_off


def rcnn_loss_grad(cls_score, bbox_offsets, labels, targets, num_classes, sigma=1.0, w_cls=1.0, w_reg=1.0):
    """The gradients of rcnn_loss alone (lmh_rcnn_loss_grad): the fused train step issues them where the loss sits and the
    reported sums (rcnn_loss(..., want_grad=False)) behind the RCNN backward."""
    lib = _lib.load()
    B, R = labels.shape
    d_cls = torch.empty_like(cls_score)
    d_off = torch.empty_like(bbox_offsets)
    check(lib.lmh_rcnn_loss_grad(_p(cls_score), _p(bbox_offsets), _p(labels), _p(targets), B, R, int(num_classes),
                                 float(sigma), float(w_cls), float(w_reg), _p(d_cls), _p(d_off), _stream()),
          'lmh_rcnn_loss_grad')
    return d_cls, d_off


def softmax(x):
    lib = _lib.load()
    C = x.shape[-1]
    y = torch.empty_like(x)
    check(lib.lmh_softmax(_p(x), x.numel() // C, C, _p(y), _stream()), 'lmh_softmax')
    return y


# ------------------------------------------------------------- optimizer ----
def sgd_momentum(w, g, v, seg_offset, seg_wd, lr, momentum, gscale=1.0):
    lib = _lib.load()
    check(lib.lmh_sgd_momentum(_p(w), _p(g), _p(v), w.numel(), _p(seg_offset), _p(seg_wd), seg_wd.numel(),
                               float(lr), float(momentum), float(gscale), _stream()), 'lmh_sgd_momentum')


def sgd_momentum_range(w, g, v, seg_offset, seg_wd, lo, hi, lr_dev, momentum, gscale=1.0, early=False):
    """The momentum update over [lo, hi) of the flat buffer; `lr_dev`: one-float device tensor (recordable in a launch plan)."""
    lib = _lib.load()
    check(lib.lmh_sgd_momentum_range(_p(w), _p(g), _p(v), w.numel(), int(lo), int(hi), _p(seg_offset), _p(seg_wd),
                                     seg_wd.numel(), _p(lr_dev), float(momentum), float(gscale), int(bool(early)), _stream()),
          'lmh_sgd_momentum_range')


def grad_clip_factors(w, g, seg_offset, seg_wd, gscale, clip_norm, out):
    """out[s] = clip / max(||g*gscale + wd*w||_2 over segment s, clip)  (tf.clip_by_norm per variable)."""
    lib = _lib.load()
    nseg = seg_wd.numel()
    ws = _workspace(lib.lmh_grad_clip_workspace_bytes(nseg), w.device, 'clip')
    check(lib.lmh_grad_clip_factors(_p(w), _p(g), w.numel(), _p(seg_offset), _p(seg_wd), nseg, float(gscale),
                                    float(clip_norm), _p(out), _p(ws), ctypes.c_size_t(ws.numel()), _stream()),
          'lmh_grad_clip_factors')
    return out


def optimizer_step(kind, w, g, slot1, slot2, seg_offset, seg_wd, seg_factor, lr, p1, p2, eps, gscale=1.0, slot3=None):
    """kind 0 momentum, 1 Adam, 2 RMSProp, 3 Nesterov momentum, 4 centered RMSProp (lmh_optimizer_step); seg_factor:
    per-segment clip factors or None."""
    lib = _lib.load()
    check(lib.lmh_optimizer_step(int(kind), _p(w), _p(g), _p(slot1), _p(slot2), _p(slot3), w.numel(), _p(seg_offset),
                                 _p(seg_wd), _p(seg_factor), seg_wd.numel(), float(lr), float(p1), float(p2),
                                 float(eps), float(gscale), _stream()), 'lmh_optimizer_step')


def dropout(x, keep_prob, seed):
    """tf.nn.dropout: x * keep / keep_prob, keep = f(seed, element index) — the same call on dy is the backward."""
    lib = _lib.load()
    y = torch.empty_like(x)
    check(lib.lmh_dropout(_p(x), x.numel(), float(keep_prob), int(seed) & 0xFFFFFFFF, _p(y), _stream()), 'lmh_dropout')
    return y


def l2_reg_loss(w, seg_offset, seg_wd, out=None):
    lib = _lib.load()
    out = out if out is not None else torch.empty((1,), dtype=torch.float32, device=w.device)
    nbytes = lib.lmh_l2_reg_workspace_bytes()
    ws = _workspace(nbytes, w.device, 'l2reg')
    check(lib.lmh_l2_reg_loss(_p(w), w.numel(), _p(seg_offset), _p(seg_wd), seg_wd.numel(), _p(out), _p(ws),
                              ctypes.c_size_t(ws.numel()), _stream()), 'lmh_l2_reg_loss')
    return out


# ------------------------------------------------------------------- SSD ----
def l2norm_scale_fwd(x, gamma, eps=1e-12):
    """tf.nn.l2_normalize(x, axis=-1, epsilon) * gamma (ssd/feature_extractor.py:75-89)."""
    lib = _lib.load()
    C = x.shape[-1]
    y = torch.empty_like(x)
    check(lib.lmh_l2norm_scale_fwd(_p(x), _p(gamma), x.numel() // C, C, float(eps), _p(y), _stream()),
          'lmh_l2norm_scale_fwd')
    return y


def l2norm_scale_bwd(x, dy, gamma, eps=1e-12, dgamma=None):
    lib = _lib.load()
    C = x.shape[-1]
    P = x.numel() // C
    dx = torch.empty_like(x)
    dgamma = dgamma if dgamma is not None else torch.empty((C,), dtype=torch.float32, device=x.device)
    ws = _workspace(lib.lmh_l2norm_scale_bwd_workspace_bytes(P, C), x.device, 'l2norm')
    check(lib.lmh_l2norm_scale_bwd(_p(x), _p(dy), _p(gamma), P, C, float(eps), _p(dx), _p(dgamma), _p(ws),
                                   ctypes.c_size_t(ws.numel()), _stream()), 'lmh_l2norm_scale_bwd')
    return dx, dgamma


def ssd_target(anchors, gt, gt_count, probs, num_classes, foreground_threshold=0.5, background_threshold_high=0.2,
               hard_negative_ratio=3.0, variances=(0.1, 0.2)):
    """anchors (N,4), gt (B,Gmax,5), gt_count (B), probs (B,N,C+1) -> labels (B,N), bbox_targets (B,N,4), max_ov."""
    lib = _lib.load()
    B, N, K = probs.shape
    assert K == num_classes + 1 and anchors.shape == (N, 4)
    v = (1.0, 1.0) if variances is None else variances
    d = SsdTargetDesc(B, N, int(num_classes), gt.shape[1], float(foreground_threshold),
                      float(background_threshold_high), float(hard_negative_ratio), float(v[0]), float(v[1]))
    dev = probs.device
    labels = torch.empty((B, N), dtype=torch.float32, device=dev)
    targets = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
    max_ov = torch.empty((B, N), dtype=torch.float32, device=dev)
    ws = _workspace(lib.lmh_ssd_target_workspace_bytes(ctypes.byref(d)), dev, 'ssd_target')
    check(lib.lmh_ssd_target(ctypes.byref(d), _p(anchors), _p(gt), _p(gt_count), _p(probs), _p(labels), _p(targets),
                             _p(max_ov), _p(ws), ctypes.c_size_t(ws.numel()), _stream()), 'lmh_ssd_target')
    return labels, targets, max_ov


def ssd_loss(cls_pred, loc_pred, labels, targets, num_classes, sigma=3.0, w_loc=1.0, want_grad=True):
    lib = _lib.load()
    B, N, K = cls_pred.shape
    dev = cls_pred.device
    losses = torch.empty((3,), dtype=torch.float32, device=dev)
    per_image = torch.empty((B, 4), dtype=torch.float32, device=dev)
    d_cls = torch.empty_like(cls_pred) if want_grad else None
    d_loc = torch.empty_like(loc_pred) if want_grad else None
    check(lib.lmh_ssd_loss(_p(cls_pred), _p(loc_pred), _p(labels), _p(targets), B, N, int(num_classes), float(sigma),
                           float(w_loc), _p(losses), _p(per_image), _p(d_cls), _p(d_loc), _stream()), 'lmh_ssd_loss')
    return losses, per_image, d_cls, d_loc
