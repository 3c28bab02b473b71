"""Layer executors of the backbone / heads: each node knows how to run its
forward and its hand-derived backward through the HIP kernels
(luminoth_amd/kernels.py).  No torch compute ops are used on activations.

The nodes restate what tf.contrib.slim builds for the reference
(models/base/base_network.py:70-101): conv2d [+ frozen BatchNorm | + bias]
[+ ReLU], conv2d_same, max_pool2d, resnet_v1 bottleneck units.
"""
import os

import torch

from luminoth_amd import kernels as K

FUSE_ACT = os.environ.get('LUMINOTH_AMD_FUSE_ACT', '0') == '1'
# ReLU gradient without a pass of its own (round 3).  Every convolution whose output feeds a trainable layer also writes
# the ACTIVATION BIT MASK of that output (1 bit per element, emitted by the epilogue that holds the values anyway);
# the backward-data kernel of the consumer applies it in ITS epilogue, so the gradient that reaches a layer already is
# g = dy * act'(y): ~36 of the 41 lmh_act_bwd launches of a ResNet-50 step (2.1 GB of traffic, 0.44 ms of kernels plus
# their launch gaps on the critical stream) disappear.  Round 2 tried the same fusion with the fp32 activation itself as
# the mask operand: 64 KB of cold HBM reads per tile in the epilogue, latency-exposed, measured slower than the
# streaming pass.  The bit mask is 2 KB per tile and is requested before the accumulator transpose.
FUSE_MASK = os.environ.get('LUMINOTH_AMD_FUSE_MASK', '1') == '1'
BN_EPS = 1e-5  # slim resnet_arg_scope batch_norm_epsilon (truncated_base_network.py:69-73)
BN_DECAY = 0.997  # slim resnet_arg_scope batch_norm_decay (its default; the reference does not override it)
# Half-STORAGE trunk (BASELINE configs[4], csrc/conv_hs.h): layers with `storage` 'f16' / 'bf16' keep their activations,
# activation gradients and working weight copies as 16-bit tensors.  Activation gradients carry a static loss scale (f16
# has 5 exponent bits: gradients of 1e-7 would flush); the weight-gradient kernels divide it out of their fp32 sums.
HS_LOSS_SCALE = {'f16': 1024.0, 'bf16': 1.0}


# Called by Trunk.backward as hook(nodes, j) once node j's backward (data + weight gradients) is enqueued, and as
# hook(nodes, len(nodes)) before the first node: lets the data-parallel layer start all-reducing finished gradient
# ranges under the rest of the backward pass.
BACKWARD_HOOK = None
# Test instrumentation: when set to a dict, every ConvLayer.forward stores its output under the layer scope, so a
# parity test can hand the kernels' own ReLU decisions to the oracle (tests/e2e_util.py).  None in production.
ACT_TAP = None


class SideStream(object):
    """Second HIP stream for the weight-gradient chain of every layer (bwd_weight -> split-K reduce ->
    BN parameter gradients).  It is independent of the data-gradient chain once g = dy*act'(y) exists,
    so running it beside bwd_data fills the CUs each kernel leaves idle in its prologue / tail.
    `join()` makes the caller's stream wait for everything enqueued here (before all-reduce / update)."""
    enabled = os.environ.get('LUMINOTH_AMD_SIDE_STREAM', '1') != '0'
    # ONE side stream per issuing stream: two MFMA-heavy kernels in flight fill each other's prologues / tails,
    # three (bwd_data + two bwd_weight) thrash each other (measured 10.45 -> 13.8 ms/step on MI355X).
    # The side stream lags the data-gradient stream, so when the main stream finishes the trunk backward it
    # would idle while the side stream drains its backlog: the weight gradients of the LAST `inline_layers`
    # trainable conv layers of a trunk backward are issued on the main stream itself, behind that layer's data
    # gradient (10.43 -> 10.13 ms/step).
    inline_layers = int(os.environ.get('LUMINOTH_AMD_INLINE_LAYERS', '4'))
    layers_left = 0
    cu_mask = os.environ.get('LUMINOTH_AMD_SIDE_CU_MASK', '')      # 'period:keep', e.g. '4:1' = a quarter of the CUs
    _streams = {}      # (device, issuing stream) -> stream (the fused train step issues from two streams)
    # tensors a side stream still reads (x, g of a layer whose weight gradient is queued there): references held until
    # join() instead of three tensor.record_stream calls per layer — nothing is freed, so nothing can be recycled early
    _held = []

    @classmethod
    def get(cls, device):
        key = (device, K._stream_id(device))
        st = cls._streams.get(key)
        if st is None:
            if cls.cu_mask:
                # experiment: the weight-gradient stream confined to a subset of the compute units
                st = K.cu_range_stream(cls.cu_mask, device)
            else:
                st = torch.cuda.Stream(device=device)
            cls._streams[key] = st
        return st

    @classmethod
    def join(cls):
        for st in cls._streams.values():
            K.stream_wait(torch.cuda.current_stream(st.device), st)
        # the joining stream is now ordered behind every side-stream reader, so the allocator may have the tensors back: a
        # freed block is only reused by later allocations of its OWN stream — the stream that just joined, or the proposal
        # stream, which waits for the joining stream before it allocates again (start of the next step)
        cls._held = []


class ConvLayer(object):
    """conv2d (HWIO weights `<scope>/weights`) followed by either a frozen
    BatchNorm (`<scope>/BatchNorm/{gamma,beta,moving_mean,moving_variance}`:
    inference statistics, gamma/beta trainable — base_network.py:84-89) or a
    bias (`<scope>/biases`), an optional residual add and an activation."""

    def __init__(self, scope, cin, cout, ksize, stride=1, rate=1, padding='SAME', act='relu',
                 norm='bn', wd=0.0, init=None, bias_name='biases', weight_name='weights'):
        self.scope, self.cin, self.cout, self.k = scope, cin, cout, ksize
        self.stride, self.rate, self.padding, self.act = stride, rate, padding, act
        self.norm = norm          # 'bn' | 'bias' | None
        self.wd, self.init = wd, init
        self.trainable = True
        # `train_batch_norm: True` while training (base_network.py:82-93): normalise with the statistics of the batch and
        # advance the moving averages (set per call by the base network; csrc/bnorm.hip)
        self.bn_train = False
        self.compute = None       # MFMA operand arithmetic: None = fp32; 'f16' / 'bf16' = mixed precision (conv_half.h)
        self.compute_wgrad = 'same'   # arithmetic of the weight-gradient GEMM alone ('same' = self.compute)
        self.storage = None       # 'f16' / 'bf16': half tensors in HBM (then compute is the same type)
        self.hs_out_f32 = False   # half-storage layer whose OUTPUT is handed on as fp32 (top of the trunk)
        self.hs_in_f32 = False    # half-storage layer fed by an fp32 tensor (casts it; its data gradient leaves as fp32): RPN conv
        self.wh = [None, None]    # working copies of the weights: [K,R,S,C] q(w), [R,S,C,K] q(w * bn_scale)
        self._wh_ready = False
        self.w_name = '%s/%s' % (scope, weight_name)
        self.b_name = '%s/%s' % (scope, bias_name)
        self._desc = {}
        self._fused = {}
        # transformed Winograd weights prepared ahead of the convolution calls (prepare_winograd_weights): [forward, backward]
        self._wino_u = [None, None]
        self._wino_ready = [False, False]
        # bf16x3: the weights split once per step into their three bf16 pieces in MFMA fragment order (prepare_x3_weights):
        # [forward arrangement, backward-data arrangement]
        self._x3w = [None, None]
        self._x3w_ready = [False, False]

    # ---- variable names in TF creation order (drives fine_tune_from) --------
    def var_names(self):
        names = [self.w_name]
        if self.norm == 'bn':
            names += ['%s/BatchNorm/beta' % self.scope, '%s/BatchNorm/gamma' % self.scope]
        elif self.norm == 'bias':
            names += [self.b_name]
        return names

    def bind(self, store, bn_table):
        self.store = store
        self.w = store[self.w_name]
        self.gw = store.grads.get(self.w_name)
        self.scale = self.shift = None
        if self.norm == 'bn':
            self.bn = bn_table.views(self.scope)
            self.scale, self.shift = self.bn['scale'], self.bn['shift']
            self.bn_table = bn_table
            self.bn_vars = tuple(store['%s/BatchNorm/%s' % (self.scope, n)]
                                 for n in ('gamma', 'beta', 'moving_mean', 'moving_variance'))
        elif self.norm == 'bias':
            self.shift = store[self.b_name]
            self.gb = store.grads.get(self.b_name)

    def desc(self, x_shape, wgrad=False):
        compute = self.compute if (not wgrad or self.compute_wgrad == 'same') else self.compute_wgrad
        key = tuple(x_shape) + (compute,)
        d = self._desc.get(key)
        if d is None:
            d = K.conv_desc(x_shape, (self.k, self.k, self.cin, self.cout), self.stride, self.rate,
                            self.padding, self._desc_act(), compute)
            if d.compute == 3 and K.X3_WINOGRAD_MODE == '1':
                # bf16x3 is fp32 arithmetic: layers the Winograd F(2x2,3x3) path takes may run it natively (fp32 GEMMs on
                # 2.25x fewer FLOPs) instead of with bf16x3 GEMMs (mode '3', the default) — DESIGN.md §3.4
                d0 = K.conv_desc(x_shape, (self.k, self.k, self.cin, self.cout), self.stride, self.rate,
                                 self.padding, self._desc_act(), None)
                if K._use_winograd(d0):
                    d = d0
            self._desc[key] = d
        return d

    def _generic_act(self):
        """An activation the convolution epilogues do not fuse (tf.nn.elu, selu, softplus, softsign, sigmoid, tanh,
        leaky_relu: luminoth/utils/vars.py:80-88): the descriptor then says 'none', the activation is applied in place
        behind the convolution and the backward takes g = dy * act'(y) from the output in one pass (no bit mask)."""
        return self.act if self.act not in K.FUSED_ACTS else None

    def _desc_act(self):
        return self.act if self.act in K.FUSED_ACTS else None

    def forward(self, x, residual=None, in_sub=None, want_bits=False, keep_v=False, out=None):
        """want_bits: also return the activation bit mask of y (None when the layer has no activation or a channel
        count that is not a multiple of 32) -> (y, bits).  out: tensor the result is written to (a buffer at a fixed
        address: the frozen trunk prefix computed one step ahead)."""
        if self.bn_train and self.norm == 'bn':
            return self._forward_bn_train(x, residual, in_sub, want_bits, keep_v, out)
        d = self.desc(x.shape)
        bits = None
        if want_bits and FUSE_MASK and K.act_bits_ok(self.cout, self.act):
            bits = K.new_act_bits(d.N * d.OH * d.OW, self.cout, x.device)
        if self.storage is not None and self.hs_in_f32 and x.dtype == torch.float32:
            # fp32 boundary on the input side: one cast pass; the 16-bit copy is kept on x for this layer's weight gradient
            assert in_sub is None and residual is None, self.scope
            xh = K.cast_to_half(x, self.storage)
            x._lmh_half = (x._version, xh)
            if bits is None and FUSE_MASK and K.act_bits_ok(self.cout, self.act):
                bits = K.new_act_bits(d.N * d.OH * d.OW, self.cout, x.device)
            x = xh
        if x.dtype != torch.float32:            # half-storage layer
            assert self.storage is not None and in_sub is None, (self.scope, x.dtype)
            if self._generic_act():
                raise NotImplementedError('%s: activation %r with 16-bit storage (relu / relu6 only)' % (self.scope, self.act))
            if not self._wh_ready:
                prepare_half_weights([self], self.storage)
            y = K.conv2d_fwd_hs(d, x, self.wh[0], self.scale, self.shift, residual, out_f32=self.hs_out_f32, act_bits=bits,
                                out=out)
            if self.hs_in_f32:
                y._lmh_bits = bits               # the backward of this layer masks the incoming fp32 gradient with them
            if ACT_TAP is not None:
                ACT_TAP[self.scope] = y
            return (y, bits) if want_bits else y
        if self._x3w_ready[0] and in_sub is None and not K._use_winograd(d) and K.conv2d_fwd_x3w_ok(d):
            # bf16x3 with this step's pre-split weights (bit-identical to the in-kernel split)
            y = K.conv2d_fwd_x3w(d, x, self._x3w[0], self.scale, self.shift, residual, out=out, act_bits=bits)
        else:
            # a training forward of a trainable Winograd layer keeps B^T x B for its weight gradient (kernels.py)
            y = K.conv2d_fwd(d, x, self.w, self.scale, self.shift, residual, in_sub, out=out, act_bits=bits,
                             keep_v=(want_bits or keep_v) and self.trainable,
                             wino_u=self._wino_u[0] if self._wino_ready[0] else None)
        if self._generic_act():
            y = self._apply_generic_act(y, out)
        if ACT_TAP is not None:
            ACT_TAP[self.scope] = y
        return (y, bits) if want_bits else y

    def _apply_generic_act(self, z, out=None):
        """In place, except where the backward needs the pre-activation (softplus, softsign: kept on the result)."""
        if self.act in K.ACT_GRAD_FROM_INPUT:
            if out is not None and z.data_ptr() == out.data_ptr():
                z = z.clone()
            y = K.act_fwd(z, self.act, out=out)
            y._lmh_z = z
            return y
        return K.act_fwd(z, self.act, out=z)

    def _act_operand(self, y):
        """What K.act_bwd differentiates in: the output, or the kept pre-activation (softplus / softsign)."""
        if self.act in K.ACT_GRAD_FROM_INPUT:
            z = getattr(y, '_lmh_z', None)
            if z is None:
                raise RuntimeError('%s: backward of %s needs the tensor its forward returned' % (self.scope, self.act))
            return z
        return y

    # ---- BatchNorm in training mode (train_batch_norm: True; csrc/bnorm.hip) -------------------------------------------
    def _desc_raw(self, x_shape):
        """The convolution alone: no activation (BatchNorm sits between it and the activation)."""
        key = tuple(x_shape) + ('raw',)
        d = self._desc.get(key)
        if d is None:
            d = self._desc[key] = K.conv_desc(x_shape, (self.k, self.k, self.cin, self.cout), self.stride, self.rate,
                                              self.padding, None, None)
        return d

    def _forward_bn_train(self, x, residual, in_sub, want_bits, keep_v, out):
        if self.storage is not None or x.dtype != torch.float32 or self.compute not in (None, 'f32', 'fp32', 'float32'):
            raise NotImplementedError('%s: train_batch_norm is implemented for fp32 tensors and arithmetic' % self.scope)
        d0 = self._desc_raw(x.shape)
        z = K.conv2d_fwd(d0, x, self.w, None, None, None, in_sub, keep_v=(want_bits or keep_v) and self.trainable)
        gamma, beta, mm, mv = self.bn_vars
        if self._generic_act():       # (slim's BatchNorm layers are the backbone's: always relu)
            raise NotImplementedError('%s: train_batch_norm with activation %r' % (self.scope, self.act))
        y, mean, rstd = K.bn_train_fwd(z, gamma, beta, mm, mv, residual, self.act, eps=BN_EPS, decay=BN_DECAY)
        if out is not None:
            y = K.copy_(out, y)
        self.bn_table.stats_moved = True        # the folded scale / shift of the inference path are stale now
        y._lmh_bn = (z, mean, rstd)
        bits = None
        if want_bits and FUSE_MASK and K.act_bits_ok(self.cout, self.act):
            bits = K.act_bits(y, self.act)
        if ACT_TAP is not None:
            ACT_TAP[self.scope] = y
        return (y, bits) if want_bits else y

    def _backward_bn_train(self, x, y, dy, need_dx, addend, dy_is_g, mask_bits):
        z, mean, rstd = y._lmh_bn
        d0 = self._desc_raw(x.shape)
        g = dy if (dy_is_g or not self.act) else K.act_bwd(dy, y, self.act, want_g=True)
        gamma = self.bn_vars[0]
        if self.trainable:
            dgamma, dbeta = self.bn['ggamma'], self.bn['gbeta']
            SideStream.layers_left -= 1
        else:
            dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        dz = K.bn_train_bwd(g, z, mean, rstd, gamma, dgamma, dbeta)
        # from here on a convolution without normalisation: dz is the gradient of its raw output
        if self.trainable:
            key = self.w_name if (K.TAILS.active and self.cout % 4 == 0 and self.cout <= 4096) else None
            K.conv2d_bwd_weight(d0, x, dz, out=self.gw, colsum=None, defer=key)
        dx = None
        if need_dx:
            dx = K.conv2d_bwd_data(d0, dz, self.w, kscale=None, addend=addend, xbits=mask_bits)
        return dx, g

    def _weight_grads(self, d, x, g, yact, colsum):
        if self.compute_wgrad != 'same':
            d = self.desc(x.shape, wgrad=True)
        key = self.w_name if (K.TAILS.active and self.cout % 4 == 0 and self.cout <= 4096) else None
        if x.dtype != torch.float32:
            K.conv2d_bwd_weight_hs(d, x, g, 1.0 / HS_LOSS_SCALE[self.storage], out=self.gw, colsum=colsum, defer=key)
        else:
            K.conv2d_bwd_weight(d, x, g, out=self.gw, yact=yact, colsum=colsum, defer=key)
        if self.norm == 'bn':
            if key is not None:        # queued: reduction + BN scaling + dgamma + dbeta happen in TAILS.flush()
                e = K.TAILS.entry(key)
                e['bn'] = dict(w=self.w, scale=self.scale, mean=self.bn['mean'], rstd=self.bn['rstd'],
                               dgamma=self.bn['ggamma'])
                e.setdefault('colsum', self.bn['gbeta'])
            else:
                K.bn_param_grads(self.w, self.gw, self.bn['gbeta'], self.bn['mean'], self.bn['rstd'],
                                 self.scale, out=self.bn['ggamma'])

    def _fused_ok(self, d, key):
        ok = self._fused.get(key)
        if ok is None:
            # Fusing act'(y) into the operand loads is supported by the kernels but OFF by default: it doubles
            # the gather traffic of the (tap-repeated) A operand and measured slower than one streaming
            # lmh_act_bwd pass (bwd_data 128x64: 167 us fused vs 75 + 20 us); the column sums stay fused.
            ok = self._fused[key] = (FUSE_ACT and K.conv_fused_act_ok(d), K.conv_fused_colsum_ok(d))
        return ok

    def backward(self, x, y, dy, need_dx=True, addend=None, want_g=False, dy_is_g=False, mask_bits=None):
        """dy: gradient w.r.t. the layer output (after residual add + act).
        Returns (dx or None, g) with g = dy * act'(y) = gradient w.r.t. the pre-activation sum (== gradient
        of the residual branch); g is only materialised when `want_g` (the bottleneck's shortcut needs
        it) or when the fast kernels cannot take it fused — otherwise both backward convolutions apply
        act'(y) while they load dy, and the per-channel sums (dbeta / dbias) come out of bwd_weight.
        dy_is_g: the incoming gradient already is g (the producer applied act'(y) in its epilogue).
        mask_bits: activation bit mask of x (written by the forward kernel of the layer that produced x): the
        returned dx is then dx * act'(x), i.e. THAT layer's g (applied in the bwd_data epilogue)."""
        if getattr(y, '_lmh_bn', None) is not None:
            return self._backward_bn_train(x, y, dy, need_dx, addend, dy_is_g, mask_bits)
        d = self.desc(x.shape)
        boundary = self.storage is not None and self.hs_in_f32 and x.dtype == torch.float32
        if boundary:
            # fp32 tensors on both sides of a half-storage layer: g = q(S * dy * act'(y)) in one cast pass, x from the forward
            assert addend is None and mask_bits is None and dy.dtype == torch.float32, self.scope
            kept = getattr(x, '_lmh_half', None)
            xh = kept[1] if (kept is not None and kept[0] == x._version) else K.cast_to_half(x, self.storage)
            bits = None
            if self.act and not dy_is_g:
                bits = getattr(y, '_lmh_bits', None)
                if bits is None:
                    bits = K.act_bits(y, self.act)
            dy = K.cast_to_half(dy, self.storage, mul=HS_LOSS_SCALE[self.storage], bits=bits)
            x, dy_is_g = xh, True
        hs = x.dtype != torch.float32
        if hs:
            # half storage: the gradient arrives as g (masked by the producer's epilogue or by the trunk's entry cast) in
            # the same 16-bit type, times the loss scale; the channel sums come out of the weight-gradient kernel
            if not (dy_is_g or not self.act) or dy.dtype != x.dtype:
                raise NotImplementedError('%s: half-storage backward needs the masked half gradient (FUSE_MASK on)' % self.scope)
            act_fused, colsum_fused = False, True
        else:
            act_fused, colsum_fused = self._fused_ok(d, tuple(x.shape))
        colsum = None
        if self.trainable:
            if self.norm == 'bn':
                colsum = self.bn['gbeta']
            elif self.norm == 'bias':
                colsum = self.gb
        colsum_in_wgrad = colsum is not None and colsum_fused
        yact = None
        dkey = self.w_name if K.TAILS.active else None
        if dy_is_g or not self.act:
            g = dy
            if colsum is not None and not colsum_in_wgrad:
                K.act_bwd(dy, None, None, want_g=False, colsum=colsum, defer=dkey)
        elif self.act and act_fused and not want_g and not self._generic_act():
            g, yact = dy, y                               # fused: kernels mask on load
        else:
            g = K.act_bwd(dy, self._act_operand(y), self.act, want_g=True, colsum=None if colsum_in_wgrad else colsum, defer=dkey)
        inline = self.trainable and 0 < SideStream.layers_left <= SideStream.inline_layers
        if self.trainable:
            SideStream.layers_left -= 1
        dx = None
        def data_grad():
            if hs:
                if not self._wh_ready:
                    prepare_half_weights([self], self.storage)
                return K.conv2d_bwd_data_hs(d, g, self.wh[1], addend=addend, xbits=mask_bits, out_f32=boundary,
                                            mul=1.0 / HS_LOSS_SCALE[self.storage] if boundary else 1.0)
            if self._x3w_ready[1] and yact is None and not K._use_winograd(d) and K.conv2d_bwd_data_x3w_ok(d):
                return K.conv2d_bwd_data_x3w(d, g, self._x3w[1], kscale=self.scale if self.norm == 'bn' else None,
                                             addend=addend, xbits=mask_bits)
            return K.conv2d_bwd_data(d, g, self.w, kscale=self.scale if self.norm == 'bn' else None,
                                     addend=addend, yact=yact, xbits=mask_bits,
                                     wino_u=self._wino_u[1] if self._wino_ready[1] else None)
        if inline and need_dx:        # tail of the backward: data gradient first, weight gradients behind it
            dx = data_grad()
        if self.trainable:
            cs = colsum if colsum_in_wgrad else None
            if SideStream.enabled and not inline:
                main = torch.cuda.current_stream(x.device)
                side = SideStream.get(x.device)
                K.stream_wait(side, main)               # dy / g are ready once `main` gets here
                with K.launch_on(side):
                    self._weight_grads(d, x, g, yact, cs)
                SideStream._held.append((x, g, yact))   # keep the buffers alive for the side stream (until join)
            else:
                self._weight_grads(d, x, g, yact, cs)
        if need_dx and not inline:
            dx = data_grad()
        return dx, (g if yact is None else None)


def prepare_half_weights(layers, storage):
    """One launch for the 16-bit working copies of every half-storage layer in `layers` (forward copy q(w) laid out
    [K][R][S][C], backward copy q(w * bn_scale) in HWIO) from the fp32 master weights and the current BatchNorm scales:
    the cast pass of the optimizer step.  The copies are used until release_half_weights."""
    jobs = []
    _, tdt = K.half_type(storage)
    for l in layers:
        if l.wh[0] is None:
            l.wh[0] = torch.empty((l.cout, l.k, l.k, l.cin), dtype=tdt, device=l.w.device)
            l.wh[1] = torch.empty((l.k, l.k, l.cin, l.cout), dtype=tdt, device=l.w.device)
        jobs.append((l.w, l.scale if l.norm == 'bn' else None, l.wh[0], l.wh[1]))
    K.half_weights_batch(jobs, storage)
    for l in layers:
        l._wh_ready = True


def release_half_weights(layers):
    for l in layers:
        l._wh_ready = False


def winograd_candidates(layers):
    """The layers whose 3x3 convolution the kernels route through Winograd whatever the spatial size (stride 1, SAME,
    fp32, channel counts the transforms cover, above the routing threshold)."""
    def arithmetic_ok(l):      # fp32, or bf16x3 with the transformed-domain GEMMs on the bf16 pipe (kernels._use_winograd)
        return l.compute in (None, 'f32', 'fp32', 'float32') or (K.COMPUTE.get(l.compute) == 3 and K.X3_WINOGRAD_MODE == '3')
    return [l for l in layers if l.k == 3 and l.stride == 1 and l.rate == 1 and l.padding == 'SAME' and arithmetic_ok(l) and
            l.cin % 32 == 0 and l.cout % 32 == 0 and K.WINOGRAD and l.cin * l.cout >= K.WINOGRAD_MIN_CK]


def prepare_winograd_weights(layers, backward):
    """One launch for the transformed weights of every layer in `layers` (forward or backward set) on the current stream;
    the layers use them until release_winograd_weights.  The weights (and BatchNorm scales) must not change in between:
    the fused train step prepares both sets at its start and releases them before the optimizer update."""
    jobs = []
    b = int(bool(backward))
    for l in layers:
        n = K._lib.load().lmh_winograd_u_bytes(l.cin, l.cout) // 4
        if l._wino_u[b] is None or l._wino_u[b].numel() != n:
            l._wino_u[b] = K.new_winograd_u(l.cin, l.cout, l.w.device)
        jobs.append((l.w, (l.scale if l.norm == 'bn' else None) if backward else None, l._wino_u[b]))
    K.winograd_weights_batch(jobs, backward)
    for l in layers:
        l._wino_ready[b] = True


def x3w_candidates(layers):
    """bf16x3 layers whose weights can be pre-split (C % 32 == 0, K % 32 == 0, fp32 tensors, inference-mode BatchNorm): the
    1x1 layers and the 3x3 / strided ones that are not routed through Winograd (whose GEMMs multiply transformed weights)."""
    wino = set(id(l) for l in winograd_candidates(layers))
    return [l for l in layers if K.COMPUTE.get(l.compute) == 3 and l.storage is None and not l.bn_train and l.k > 0 and
            l.cin % 32 == 0 and l.cout % 32 == 0 and id(l) not in wino]


def prepare_x3_weights(layers, backward):
    """One launch that splits the weights of every layer in `layers` into the three exact bf16 pieces of bf16x3, laid out in
    MFMA fragment order (csrc/conv_x3.h k_x3_split_w; forward or backward-data arrangement), on the current stream.  The
    layers multiply with them until release_x3_weights; the weights must not change in between (the fused train step
    prepares at its start and releases before the optimizer update)."""
    jobs = []
    b = int(bool(backward))
    for l in layers:
        if l._x3w[b] is None:
            l._x3w[b] = K.new_x3_weights(l.k * l.k, l.cin, l.cout, l.w.device, backward=backward)
        jobs.append((l.w, l.k * l.k, l._x3w[b]))
    K.x3_split_weights_batch(jobs, backward)
    for l in layers:
        l._x3w_ready[b] = True


def release_x3_weights(layers):
    for l in layers:
        l._x3w_ready = [False, False]


def release_winograd_weights(layers):
    for l in layers:
        l._wino_ready = [False, False]


class BNTable(object):
    """All BatchNorm vectors of a network, grouped so that scale = gamma*rstd and
    shift = beta - mean*scale are refreshed for EVERY layer with three
    elementwise launches per step (gamma/beta are trained, statistics frozen)."""

    def __init__(self):
        self.layers = []   # (scope, K, trainable)
        self.prefix = {}   # scope -> variable-name prefix ('<scope>/BatchNorm' behind a convolution; '<scope>' for a stand-alone layer)
        self.stats_moved = False     # a training-mode BatchNorm advanced the moving statistics (ConvLayer._forward_bn_train)

    def add(self, scope, k, trainable, prefix=None):
        self.layers.append((scope, k, trainable))
        self.prefix[scope] = prefix or (scope + '/BatchNorm')

    def var(self, scope, kind):
        return '%s/%s' % (self.prefix[scope], kind)

    def register(self, store, ones, zeros):
        # grouped registration keeps each kind contiguous inside the flat buffers
        for scope, k, tr in self.layers:
            store.add(self.var(scope, 'gamma'), (k,), ones, trainable=tr)
        for scope, k, tr in self.layers:
            store.add(self.var(scope, 'beta'), (k,), zeros, trainable=tr)
        for scope, k, tr in self.layers:
            store.add(self.var(scope, 'moving_mean'), (k,), zeros, trainable=False)
        for scope, k, tr in self.layers:
            store.add(self.var(scope, 'moving_variance'), (k,), ones, trainable=False)

    def bind(self, store):
        self.store = store
        dev = store.flat.device
        self._views = {}
        self.groups = []
        for tr in (True, False):
            ls = [(s, k) for s, k, t in self.layers if t == tr]
            if not ls:
                continue
            total = sum(k for _, k in ls)

            def region(kind):
                first = self.var(ls[0][0], kind)
                is_tr, o, _ = store.offsets[first]
                buf = store.flat if is_tr else store.frozen
                return buf[o:o + total], (store.grad[o:o + total] if is_tr else None)
            gamma, ggamma = region('gamma')
            beta, gbeta = region('beta')
            mean = torch.cat([store[self.var(s, 'moving_mean')].reshape(-1) for s, _ in ls])
            var = torch.cat([store[self.var(s, 'moving_variance')].reshape(-1) for s, _ in ls])
            rstd = torch.rsqrt(var + BN_EPS)
            scale = torch.empty(total, dtype=torch.float32, device=dev)
            shift = torch.empty(total, dtype=torch.float32, device=dev)
            grp = dict(gamma=gamma, beta=beta, mean=mean, rstd=rstd, scale=scale, shift=shift, trainable=tr)
            self.groups.append(grp)
            o = 0
            for s, k in ls:
                self._views[s] = dict(
                    scale=scale[o:o + k], shift=shift[o:o + k], mean=mean[o:o + k], rstd=rstd[o:o + k],
                    ggamma=None if ggamma is None else ggamma[o:o + k],
                    gbeta=None if gbeta is None else gbeta[o:o + k])
                o += k
        self.refresh(force=True)

    def reload_statistics(self):
        """Call after loading a checkpoint: moving statistics changed."""
        for grp in self.groups:
            ls = [(s, k) for s, k, t in self.layers if t == grp['trainable']]
            var = torch.cat([self.store[self.var(s, 'moving_variance')].reshape(-1) for s, _ in ls])
            grp['mean'].copy_(torch.cat([self.store[self.var(s, 'moving_mean')].reshape(-1) for s, _ in ls]))
            grp['rstd'].copy_(torch.rsqrt(var + BN_EPS))
        self.refresh(force=True)

    def refresh(self, force=False):
        if self.stats_moved:
            self.stats_moved = False
            return self.reload_statistics()
        for grp in self.groups:
            if grp['trainable'] or force:
                # scale = gamma * rstd; shift = beta - mean * scale: one launch of the library per group
                if grp['gamma'].is_cuda:
                    K.bn_refresh(grp['gamma'], grp['beta'], grp['mean'], grp['rstd'], grp['scale'], grp['shift'])
                else:       # a model laid out on the host (layout / sharding logic in the CPU tests): nothing runs on it
                    torch.mul(grp['gamma'], grp['rstd'], out=grp['scale'])
                    torch.sub(grp['beta'], grp['mean'] * grp['scale'], out=grp['shift'])

    def views(self, scope):
        return self._views[scope]


# ---------------------------------------------------------------- nodes -----
class ConvNode(object):
    def __init__(self, layer, in_sub=None):
        self.layer, self.in_sub = layer, in_sub
        self.layers = [layer]

    def forward(self, x, save, out=None):
        if not save:
            return self.layer.forward(x, in_sub=self.in_sub, out=out), None
        y, bits = self.layer.forward(x, in_sub=self.in_sub, want_bits=True, out=out)
        return y, (x, y, bits)

    out_act = property(lambda self: self.layer.act)

    @staticmethod
    def out_bits(saved):
        return saved[2]

    def out_hw(self, h, w):
        l = self.layer
        d = K.conv_desc((1, h, w, l.cin), (l.k, l.k, l.cin, l.cout), l.stride, l.rate, l.padding, l.act)
        return d.OH, d.OW

    def backward(self, saved, dy, need_dx, dy_is_g=False, mask_bits=None):
        x, y, _ = saved
        dx, _ = self.layer.backward(x, y, dy, need_dx=need_dx, dy_is_g=dy_is_g, mask_bits=mask_bits)
        return dx


class MaxPoolNode(object):
    layers = []
    out_act = None

    def __init__(self, ksize, stride, padding):
        self.k, self.s, self.p = ksize, stride, padding
        self.storage = None       # 'f16' / 'bf16': the result is a half tensor (first node of a half-storage trunk)

    def out_hw(self, h, w):
        if self.p == 'SAME':
            return -(-h // self.s), -(-w // self.s)
        return (h - self.k) // self.s + 1, (w - self.k) // self.s + 1

    def forward(self, x, save, out=None):
        if out is not None:
            raise NotImplementedError('MaxPoolNode: no preallocated output')
        y, geom = K.maxpool_fwd(x, self.k, self.s, self.p, storage=self.storage)
        return y, ((x, y, geom) if save else None)

    @staticmethod
    def out_bits(saved):
        return None

    def backward(self, saved, dy, need_dx, dy_is_g=False, mask_bits=None):
        if not need_dx:
            return None
        x, y, geom = saved
        dx = K.maxpool_bwd(x, y, dy, self.k, self.s, geom)
        return K.apply_act_bits(dx, mask_bits) if mask_bits is not None else dx


class BottleneckNode(object):
    """slim resnet_v1.bottleneck: shortcut (identity | 1x1 max-pool subsample |
    1x1 conv+BN) + [1x1 -> 3x3 conv2d_same(stride, rate) -> 1x1], relu(sum)."""

    def __init__(self, scope, cin, depth, depth_bottleneck, stride, rate, wd, init):
        p = scope + '/bottleneck_v1'
        self.stride = stride
        self.shortcut = None
        if depth != cin:
            self.shortcut = ConvLayer(p + '/shortcut', cin, depth, 1, stride=stride, act=None, wd=wd, init=init)
        self.conv1 = ConvLayer(p + '/conv1', cin, depth_bottleneck, 1, act='relu', wd=wd, init=init)
        self.conv2 = ConvLayer(p + '/conv2', depth_bottleneck, depth_bottleneck, 3, stride=stride, rate=rate,
                               padding='SAME' if stride == 1 else 'SAME_EXPLICIT', act='relu', wd=wd, init=init)
        self.conv3 = ConvLayer(p + '/conv3', depth_bottleneck, depth, 1, act='relu', wd=wd, init=init)
        # TF creation order inside a unit: shortcut, conv1, conv2, conv3
        self.layers = ([self.shortcut] if self.shortcut else []) + [self.conv1, self.conv2, self.conv3]

    def forward(self, x, save, out=None):
        geom = None
        if self.shortcut is not None:
            sc = self.shortcut.forward(x)
        elif self.stride > 1:
            sc, geom = K.maxpool_fwd(x, 1, self.stride, 'VALID')   # resnet_utils.subsample
        else:
            sc = x
        if not save:
            a = self.conv1.forward(x)
            b = self.conv2.forward(a)
            return self.conv3.forward(b, residual=sc, out=out), None
        a, ba = self.conv1.forward(x, want_bits=True)
        b, bb = self.conv2.forward(a, want_bits=True)
        y, by = self.conv3.forward(b, residual=sc, want_bits=True, out=out)
        return y, (x, sc, a, b, y, geom, ba, bb, by)

    @staticmethod
    def out_bits(saved):
        return saved[8]

    out_act = 'relu'

    def out_hw(self, h, w):
        return -(-h // self.stride), -(-w // self.stride)

    def backward(self, saved, dy, need_dx, dy_is_g=False, mask_bits=None):
        """Every bwd_data epilogue applies the activation bit mask of its own input, so the gradient that reaches the
        layer below already is that layer's g: no lmh_act_bwd passes inside the unit (and none for the unit below when
        `mask_bits` — the mask of this unit's input x — is handed in)."""
        x, sc, a, b, y, geom, ba, bb, _ = saved
        d_b, g = self.conv3.backward(b, y, dy, want_g=True, dy_is_g=dy_is_g, mask_bits=bb)
        d_a, _ = self.conv2.backward(a, b, d_b, dy_is_g=bb is not None, mask_bits=ba)
        if self.shortcut is not None:
            d_sc, _ = self.shortcut.backward(x, sc, g, need_dx=need_dx)
        elif self.stride > 1:
            d_sc = K.maxpool_bwd(x, sc, g, 1, self.stride, geom) if need_dx else None
        else:
            d_sc = g
        dx, _ = self.conv1.backward(x, a, d_a, need_dx=need_dx, addend=d_sc, dy_is_g=ba is not None,
                                    mask_bits=mask_bits if need_dx else None)
        return dx


class PreactLayer(object):
    """A BatchNorm + ReLU that does not follow a convolution: `slim.batch_norm(inputs, activation_fn=tf.nn.relu,
    scope='preact')` at the head of every slim resnet_v2 bottleneck (base_network.py:94-101 builds resnet_v2 through
    tf.contrib.slim.nets.resnet_v2).  Variables `<scope>/{beta,gamma,moving_mean,moving_variance}` (no `BatchNorm`
    level).  Inference: y = relu(x * scale + shift) with the folded moving statistics (lmh_bn_apply); training: the
    statistics of the batch (lmh_bn_train_fwd / _bwd) — the reference hands `is_training` to every resnet_v2 BatchNorm."""
    norm, act, k, stride, rate, padding = 'bn', 'relu', 0, 1, 1, 'SAME'
    storage = compute = None
    w_name = b_name = None
    wd = 0.0

    def __init__(self, scope, channels):
        self.scope, self.cin, self.cout = scope, channels, channels
        self.trainable = True
        self.bn_train = False

    def var_names(self):
        return ['%s/beta' % self.scope, '%s/gamma' % self.scope]

    def bind(self, store, bn_table):
        self.store, self.bn_table = store, bn_table
        self.bn = bn_table.views(self.scope)
        self.bn_vars = tuple(store['%s/%s' % (self.scope, n)] for n in ('gamma', 'beta', 'moving_mean', 'moving_variance'))

    def forward(self, x, want_bits=False):
        if x.dtype != torch.float32:
            raise NotImplementedError('%s: fp32 tensors only' % self.scope)
        if self.bn_train:
            gamma, beta, mm, mv = self.bn_vars
            y, mean, rstd = K.bn_train_fwd(x, gamma, beta, mm, mv, None, 'relu', eps=BN_EPS, decay=BN_DECAY)
            y._lmh_bn = (x, mean, rstd)
            self.bn_table.stats_moved = True
        else:
            y = K.bn_apply(x, self.bn['scale'], self.bn['shift'], None, 'relu')
        bits = K.act_bits(y, 'relu') if (want_bits and FUSE_MASK and K.act_bits_ok(self.cout, 'relu')) else None
        if ACT_TAP is not None:
            ACT_TAP[self.scope] = y
        return (y, bits) if want_bits else y

    def backward(self, x, y, g, need_dx, addend=None):
        """g: gradient of the pre-activation (already masked by relu'(y)).  -> dx (+ addend) or None."""
        if not (need_dx or self.trainable):
            return None
        gamma = self.bn_vars[0]
        if self.trainable:
            dgamma, dbeta = self.bn['ggamma'], self.bn['gbeta']
        else:
            dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        kept = getattr(y, '_lmh_bn', None)
        if kept is not None:
            return K.bn_train_bwd(g, kept[0], kept[1], kept[2], gamma, dgamma, dbeta, addend=addend, need_dz=need_dx)
        return K.bn_train_bwd(g, x, self.bn['mean'], self.bn['rstd'], gamma, dgamma, dbeta, addend=addend, frozen=True,
                              need_dz=need_dx)


class PreactBottleneckNode(object):
    """slim resnet_v2.bottleneck: preact = relu(BN(x)); shortcut = subsample(x) | conv1x1(preact) + bias;
    residual = conv1x1(preact) [BN, relu] -> conv3x3 conv2d_same(stride, rate) [BN, relu] -> conv1x1 + bias;
    output = shortcut + residual (no activation: the next unit's preact applies it)."""
    out_act = None

    def __init__(self, scope, cin, depth, depth_bottleneck, stride, rate, wd, init):
        p = scope + '/bottleneck_v2'
        self.stride = stride
        self.preact = PreactLayer(p + '/preact', cin)
        self.shortcut = None
        if depth != cin:
            self.shortcut = ConvLayer(p + '/shortcut', cin, depth, 1, stride=stride, act=None, norm='bias', wd=wd, init=init)
        self.conv1 = ConvLayer(p + '/conv1', cin, depth_bottleneck, 1, act='relu', wd=wd, init=init)
        self.conv2 = ConvLayer(p + '/conv2', depth_bottleneck, depth_bottleneck, 3, stride=stride, rate=rate,
                               padding='SAME' if stride == 1 else 'SAME_EXPLICIT', act='relu', wd=wd, init=init)
        self.conv3 = ConvLayer(p + '/conv3', depth_bottleneck, depth, 1, act=None, norm='bias', wd=wd, init=init)
        # TF creation order inside a unit: preact, shortcut, conv1, conv2, conv3
        self.layers = [self.preact] + ([self.shortcut] if self.shortcut else []) + [self.conv1, self.conv2, self.conv3]

    def out_hw(self, h, w):
        return -(-h // self.stride), -(-w // self.stride)

    @staticmethod
    def out_bits(saved):
        return None

    def forward(self, x, save, out=None):
        geom = None
        pre, pbits = self.preact.forward(x, want_bits=True) if save else (self.preact.forward(x), None)
        if self.shortcut is not None:
            sc = self.shortcut.forward(pre)
        elif self.stride > 1:
            sc, geom = K.maxpool_fwd(x, 1, self.stride, 'VALID')   # resnet_utils.subsample(inputs)
        else:
            sc = x
        if not save:
            a = self.conv1.forward(pre)
            b = self.conv2.forward(a)
            return self.conv3.forward(b, residual=sc, out=out), None
        a, ba = self.conv1.forward(pre, want_bits=True)
        b, bb = self.conv2.forward(a, want_bits=True)
        y = self.conv3.forward(b, residual=sc, out=out)
        return y, (x, pre, pbits, sc, a, b, y, geom, ba, bb)

    def backward(self, saved, dy, need_dx, dy_is_g=False, mask_bits=None):
        x, pre, pbits, sc, a, b, y, geom, ba, bb = saved
        d_b, _ = self.conv3.backward(b, y, dy, dy_is_g=True, mask_bits=bb)          # no activation on the sum: g = dy
        d_a, _ = self.conv2.backward(a, b, d_b, dy_is_g=bb is not None, mask_bits=ba)
        need_pre = need_dx or self.preact.trainable
        d_pre_sc = None
        if self.shortcut is not None:
            d_pre_sc, _ = self.shortcut.backward(pre, sc, dy, need_dx=need_pre, dy_is_g=True)
        g_pre, _ = self.conv1.backward(pre, a, d_a, need_dx=need_pre, addend=d_pre_sc, dy_is_g=ba is not None,
                                       mask_bits=pbits if need_pre else None)
        if not need_pre:
            return None
        if pbits is None:                          # channel count not a multiple of 32: the ReLU gradient as a pass
            g_pre = K.act_bwd(g_pre, pre, 'relu', want_g=True)
        d_x_sc = None                               # the shortcut's share of the gradient of x (identity / subsample)
        if need_dx and self.shortcut is None:
            d_x_sc = K.maxpool_bwd(x, sc, dy, 1, self.stride, geom) if self.stride > 1 else dy
        dx = self.preact.backward(x, pre, g_pre, need_dx, addend=d_x_sc)
        return K.apply_act_bits(dx, mask_bits) if (dx is not None and mask_bits is not None) else dx


class Trunk(object):
    """Ordered nodes; nodes before `first_trainable` run forward-only."""

    def __init__(self, nodes):
        self.nodes = nodes

    def all_layers(self):
        return [l for n in self.nodes for l in n.layers]

    def out_hw(self, h, w):
        for n in self.nodes:
            h, w = n.out_hw(h, w)
        return h, w

    def first_trainable(self):
        for i, n in enumerate(self.nodes):
            if any(l.trainable for l in n.layers):
                return i
        return len(self.nodes)

    def forward(self, x, save_from=None, out=None):
        """out: tensor the LAST node writes its result to."""
        saved = []
        last = len(self.nodes) - 1
        for i, n in enumerate(self.nodes):
            save = save_from is not None and i >= save_from
            x, s = n.forward(x, save, out=out) if (i == last and out is not None) else n.forward(x, save)
            if save:
                saved.append(s)
        return x, saved

    def backward(self, saved, dy, save_from, need_dx_first=False):
        nodes = self.nodes[save_from:]
        dy_is_g = False
        storage = next((l.storage for n in nodes[::-1] for l in n.layers if l.storage), None)
        if storage is not None and dy.dtype == torch.float32:
            # entry of a half-storage trunk: gradient of the fp32 feature map -> g of the top node, times the loss scale
            top = nodes[-1].out_bits(saved[-1]) if FUSE_MASK else None
            if nodes[-1].out_act and top is None:
                raise NotImplementedError('half-storage trunk backward needs the activation bit masks (FUSE_MASK on)')
            dy = K.cast_to_half(dy, storage, mul=HS_LOSS_SCALE[storage], bits=top)
            dy_is_g = True
        hook = BACKWARD_HOOK        # data-parallel gradient buckets (utils/training.py); None on one GPU
        if hook is not None:
            hook(nodes, len(nodes))
        SideStream.layers_left = sum(1 for n in nodes for l in n.layers if l.trainable and isinstance(l, ConvLayer))
        for j in range(len(nodes) - 1, -1, -1):
            need_dx = (j > 0) or need_dx_first
            # node j-1's activation gradient is folded into node j's data gradient (bit mask in the bwd_data epilogue)
            below = nodes[j - 1].out_bits(saved[j - 1]) if (j > 0 and FUSE_MASK) else None
            dy = nodes[j].backward(saved[j], dy, need_dx, dy_is_g=dy_is_g, mask_bits=below)
            dy_is_g = below is not None
            if hook is not None:
                hook(nodes, j)
            elif j > 2:
                K.TAILS.maybe_flush_early()
        SideStream.layers_left = 0
        if dy is not None and dy.dtype != torch.float32:
            dy = K.cast_to_f32(dy, 1.0 / HS_LOSS_SCALE[storage])
        return dy
