"""FasterRCNN — the drop-in model module of the hot path.

Same protocol as the reference Sonnet module (luminoth/models/fasterrcnn/
fasterrcnn.py:12-364):

    model = get_model('fasterrcnn')(config)
    prediction_dict = model(image, gt_boxes=None, is_training=False)
    total_loss = model.loss(prediction_dict)            # or return_all=True
    model.get_trainable_vars(); model.get_base_network_checkpoint_vars(); model.summary

with the same `prediction_dict` keys.  Differences, all
extensions: `image` may be a batch `(B,H,W,3)` (the reference is batch-1:
fasterrcnn.py:101-103) with `gt_boxes` a list / padded `(B,G,5)` tensor, tensors
are torch ROCm tensors, and ragged results are fixed-capacity buffers plus
per-image counts (`num_proposals`, `num_objects`) so that the train step never
synchronises with the host.  For un-batched inference calls the results are
truncated to the reference's exact shapes.
"""
import ctypes
import os
import time

import numpy as np
import torch

from luminoth_amd import _lib
from luminoth_amd import kernels as K
from luminoth_amd.models.base import layers as L
from luminoth_amd.models.base.layers import SideStream
from luminoth_amd.models.base.truncated_base_network import TruncatedBaseNetwork
from luminoth_amd.models.fasterrcnn.rcnn import RCNN
from luminoth_amd.models.fasterrcnn.rpn import RPN
from luminoth_amd.params import ParamStore
from luminoth_amd.utils import rng
from luminoth_amd.utils.anchors import generate_anchors_reference, truncate_reference, all_anchors_numpy



# software pipelining across steps (train_step(next_image=...)); LUMINOTH_AMD_PREFETCH_PREFIX=0 turns it off
PREFETCH_PREFIX = os.environ.get('LUMINOTH_AMD_PREFETCH_PREFIX', '1') != '0'
# RPN backward on the weight-gradient stream (idle until the trunk backward): '1' / '0' force it; default: on for fp32
# tensors (6.95 against 7.06 ms per step), off for the half-storage trunk (4.26 against 4.17 ms) — measured, one box
RPN_BWD_SIDE = os.environ.get('LUMINOTH_AMD_RPN_BWD_SIDE', 'auto')
# where the next batch's frozen prefix runs: 'middle' = main stream while it waits for the RCNN branch (rounds 2-3);
# 'side' / 'aux' = at the START of the step on the weight-gradient / proposal stream, under the trunk forward
PREFIX_AT = os.environ.get('LUMINOTH_AMD_PREFIX_AT', 'middle')
# the RCNN loss VALUES (one block per image, reported only) behind the join instead of in front of the RCNN backward
RCNN_LOSS_LATE = os.environ.get('LUMINOTH_AMD_RCNN_LOSS_LATE', '1') != '0'
# the RPN backward BEHIND the join instead of beside the proposal chain / the next batch's prefix (round 5): its data gradient
# takes the RCNN branch's gradient of the feature map as `addend` (instead of the ROI-pooling backward taking the RPN's)
RPN_BWD_LATE = os.environ.get('LUMINOTH_AMD_RPN_BWD_LATE', '0') != '0'
PREFIX_SPLIT = os.environ.get('LUMINOTH_AMD_PREFIX_SPLIT', '0') != '0'      # stem of the next batch right behind the RPN heads
WINO_BATCH = os.environ.get('LUMINOTH_AMD_WINO_BATCH', '1') != '0'      # transformed Winograd weights of the whole step in two launches
# bf16x3: weights pre-split once per step into MFMA fragment order, B fragments loaded straight from global memory (round 6)
X3_PRESPLIT = os.environ.get('LUMINOTH_AMD_X3_PRESPLIT', '1') != '0'


class FasterRCNN(object):
    def __init__(self, config, name='fasterrcnn', device=None):
        self._config = config
        self._name = name
        self._num_classes = config.model.network.num_classes
        self._with_rcnn = config.model.network.with_rcnn
        self._debug = config.train.debug
        self._seed = config.train.seed
        self._anchor_base_size = config.model.anchors.base_size
        self._anchor_scales = np.array(config.model.anchors.scales)
        self._anchor_ratios = np.array(config.model.anchors.ratios)
        self._anchor_stride = config.model.anchors.stride
        self._anchor_reference = generate_anchors_reference(
            self._anchor_base_size, self._anchor_ratios, self._anchor_scales)
        self._num_anchors = self._anchor_reference.shape[0]
        self._rpn_cls_loss_weight = config.model.loss.rpn_cls_loss_weight
        self._rpn_reg_loss_weight = config.model.loss.rpn_reg_loss_weights
        self._rcnn_cls_loss_weight = config.model.loss.rcnn_cls_loss_weight
        self._rcnn_reg_loss_weight = config.model.loss.rcnn_reg_loss_weights
        self._losses_collections = ['fastercnn_losses']

        if device is None:
            if not torch.cuda.is_available():
                raise _lib.LuminothHipError('FasterRCNN needs a ROCm device (no CPU fallback on the product path)')
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        _lib.load()   # fail loudly, now, if the HIP library is missing

        self.base_network = TruncatedBaseNetwork(config.model.base_network)
        self._rpn = RPN(self._num_anchors, config.model.rpn, self.base_network.feat_channels,
                        debug=self._debug, seed=self._seed, scope=name)
        if self.base_network.compute_dtype not in (None, 'f32', 'fp32', 'float32'):
            self._rpn._rpn.compute = self.base_network.compute_dtype     # the 3x3 RPN conv; the 1x1 heads stay fp32
            rc = self._rpn._rpn
            if self.base_network.storage_dtype in ('f16', 'bf16') and rc.cin % 64 == 0 and rc.cout % 64 == 0:
                # half-storage trunk: the RPN convolution (a quarter of the step's FLOPs) runs on 16-bit operands in HBM
                # too — it casts the fp32 feature map once and hands fp32 tensors on in both directions
                rc.storage = self.base_network.storage_dtype
                rc.hs_in_f32 = rc.hs_out_f32 = True
                self.base_network.extra_hs_layers.append(rc)
        self._rcnn = None
        if self._with_rcnn:
            self._rcnn = RCNN(self._num_classes, config.model.rcnn, self.base_network.tail_channels,
                              debug=self._debug, seed=self._seed, scope=name)
        self.store = ParamStore()
        self.base_network.register(self.store, base_trainable=bool(config.model.base_network.trainable))
        self._rpn.register(self.store)
        if self._rcnn is not None:
            self._rcnn.register(self.store)
        self.store.build(self.device, seed=self._seed)
        self.base_network.bind(self.store)
        self._rpn.bind(self.store)
        if self._rcnn is not None:
            self._rcnn.bind(self.store)
        self._anchor_ref_i32 = torch.tensor(truncate_reference(self._anchor_reference), dtype=torch.int32,
                                            device=self.device)
        self._step = 0
        # launch-plan bookkeeping of the fused step, readable by the driver (a loader whose shapes thrash the four
        # kept states or keep the steps eager is a silent performance cliff otherwise)
        import collections
        self.plan_stats = collections.Counter()
        self._frozen_reg = None

    # ------------------------------------------------------------------ inputs --
    _SEED_BLOCK = 1024

    def _image_seeds(self, B):
        """Per-image RNG seeds of the current step as a device tensor (B,) int32.  Seeds for the next
        _SEED_BLOCK steps are generated and uploaded in one go: a per-step host->device copy from
        pageable memory would re-synchronise the host with the GPU every step."""
        c = getattr(self, '_seed_cache', None)
        if c is None or c[0] != B or not (c[1] <= self._step < c[1] + self._SEED_BLOCK):
            base = self._step
            # data parallel: image b of rank r is image r*B + b of the global batch (distinct subsample streams)
            from luminoth_amd.utils.sharding import rank_world
            b0 = rank_world()[0] * B
            tab = np.array([[rng.image_seed(self._seed, base + s, b0 + b) for b in range(B)]
                            for s in range(self._SEED_BLOCK)], dtype=np.uint32).view(np.int32)
            c = self._seed_cache = (B, base, torch.from_numpy(tab).to(self.device))
        return c[2][self._step - c[1]]

    def _pack_gt(self, gt_boxes, B):
        """-> (gt (B,Gmax,5) fp32 device, gt_count (B) int32 device)."""
        if gt_boxes is None:
            return None, None
        if isinstance(gt_boxes, (tuple, list)) and len(gt_boxes) == 2 and torch.is_tensor(gt_boxes[0]) \
                and gt_boxes[0].dim() == 3:
            return gt_boxes[0].to(self.device, torch.float32).contiguous(), \
                gt_boxes[1].to(self.device, torch.int32).contiguous()
        if torch.is_tensor(gt_boxes) and gt_boxes.dim() == 3:
            cnt = torch.full((B,), gt_boxes.shape[1], dtype=torch.int32, device=self.device)
            return gt_boxes.to(self.device, torch.float32).contiguous(), cnt
        if torch.is_tensor(gt_boxes) or isinstance(gt_boxes, np.ndarray):
            gt_boxes = [gt_boxes]
        gts = [torch.as_tensor(g, dtype=torch.float32).reshape(-1, 5) for g in gt_boxes]
        gmax = max(1, max(g.shape[0] for g in gts))
        packed = torch.zeros((B, gmax, 5), dtype=torch.float32)
        for b, g in enumerate(gts):
            packed[b, :g.shape[0]] = g
        cnt = torch.tensor([g.shape[0] for g in gts], dtype=torch.int32)
        return packed.to(self.device), cnt.to(self.device)

    # ----------------------------------------------------------------- forward --
    def __call__(self, image, gt_boxes=None, is_training=False):
        """image (H,W,3) or (B,H,W,3) fp32 RGB in [0,255]; gt_boxes (G,5) / list / (B,G,5)."""
        image = torch.as_tensor(image)
        unbatched = image.dim() == 3
        if unbatched:
            image = image.unsqueeze(0)
        image = image.to(self.device, torch.float32).contiguous()
        B, H, W, _ = image.shape
        gt, gt_count = self._pack_gt(gt_boxes, B)
        seeds = None
        if gt is not None:
            seeds = self._image_seeds(B)
            if is_training:
                self._step += 1
        # bf16x3: this call's weights pre-split once (csrc/conv_x3.h), forward and backward-data arrangements, for a training
        # call; whatever an earlier call left is dropped first (the weights may have been updated or loaded since)
        xl = self._x3w_layers()
        if xl:
            L.release_x3_weights(xl)
            if is_training and gt is not None and not any(l.bn_train for l in self.base_network.trunk.all_layers()):
                L.prepare_x3_weights(xl, backward=False)
                if [l for l in xl if l.trainable]:
                    L.prepare_x3_weights([l for l in xl if l.trainable], backward=True)
        with torch.set_grad_enabled(bool(is_training)):
            conv_feature_map = self.base_network(image, is_training=is_training)
            im_shape = (H, W)
            rpn_prediction = self._rpn(conv_feature_map, im_shape, self._anchor_ref_i32, self._anchor_stride,
                                       gt_boxes=gt, gt_count=gt_count, seeds=seeds, is_training=is_training)
            prediction_dict = {'rpn_prediction': rpn_prediction}
            if self._debug:
                prediction_dict['image'] = image
                prediction_dict['image_shape'] = im_shape
                prediction_dict['all_anchors'] = torch.from_numpy(all_anchors_numpy(
                    self._anchor_reference, conv_feature_map.shape[1], conv_feature_map.shape[2],
                    self._anchor_stride))
                prediction_dict['anchor_reference'] = torch.from_numpy(self._anchor_reference)
                if gt is not None:
                    prediction_dict['gt_boxes'] = gt
                prediction_dict['conv_feature_map'] = conv_feature_map
            if self._with_rcnn:
                proposals = rpn_prediction['proposals'].detach()        # stop_gradient, fasterrcnn.py:147
                prediction_dict['classification_prediction'] = self._rcnn(
                    conv_feature_map, proposals, rpn_prediction['num_proposals'], im_shape, self.base_network,
                    gt_boxes=gt, gt_count=gt_count, seeds=seeds, is_training=is_training)
        prediction_dict['_batch'] = {'B': B, 'unbatched': unbatched}
        if unbatched and not is_training:
            self._truncate_unbatched(prediction_dict)
        return prediction_dict

    def _truncate_unbatched(self, pd):
        """Reference shapes for a single image: drop the batch dim and the padding (host sync)."""
        rp = pd['rpn_prediction']
        n = int(rp['num_proposals'][0])
        rp['proposals'], rp['scores'] = rp['proposals'][0, :n], rp['scores'][0, :n]
        for k in ('rpn_cls_prob', 'rpn_cls_score', 'rpn_bbox_pred', 'rpn_cls_target', 'rpn_bbox_target'):
            if k in rp:
                rp[k] = rp[k][0]
        cp = pd.get('classification_prediction')
        if cp is not None:
            m = int(cp['num_proposals'][0])
            for k in ('cls_score', 'cls_prob', 'bbox_offsets'):
                cp['rcnn'][k] = cp['rcnn'][k][0, :m]
            if 'objects' in cp:
                d = int(cp['num_objects'][0])
                cp['objects'], cp['labels'], cp['probs'] = cp['objects'][0, :d], cp['labels'][0, :d], cp['probs'][0, :d]
            if 'target' in cp:
                cp['target'] = {k: v[0, :m] for k, v in cp['target'].items()}

    # -------------------------------------------------------------------- loss --
    def regularization_loss(self):
        """tf.losses.get_regularization_loss(): sum_w scale * sum(w^2)/2 over every
        regularised weight, frozen ones included (fasterrcnn.py:223)."""
        st = self.store
        return (K.l2_reg_loss(st.flat, st.seg_offset, st.seg_wd) + self._frozen_reg_tensor())[0]

    def loss(self, prediction_dict, return_all=False):
        """fasterrcnn.py:158-259.  Mutates prediction_dict (adds rpn_loss_dict / rcnn_loss_dict)."""
        rpn_loss_dict = self._rpn.loss(prediction_dict['rpn_prediction'], self._rpn_cls_loss_weight,
                                       self._rpn_reg_loss_weight)
        prediction_dict['rpn_loss_dict'] = rpn_loss_dict
        rcnn_loss_dict = {}
        if self._with_rcnn:
            rcnn_loss_dict = self._rcnn.loss(prediction_dict['classification_prediction'],
                                             self._rcnn_cls_loss_weight, self._rcnn_reg_loss_weight)
            prediction_dict['rcnn_loss_dict'] = rcnn_loss_dict
        items = list(rpn_loss_dict.items()) + list(rcnn_loss_dict.items())
        no_reg_loss = items[0][1]
        for _, t in items[1:]:
            no_reg_loss = no_reg_loss + t
        regularization_loss = self.regularization_loss()
        total_loss = no_reg_loss + regularization_loss
        self._last_losses = dict(items, total_loss=total_loss, no_reg_loss=no_reg_loss,
                                 regularization_loss=regularization_loss)
        if return_all:
            out = {'total_loss': total_loss, 'no_reg_loss': no_reg_loss,
                   'regularization_loss': regularization_loss}
            out.update(items)
            return out
        return total_loss

    def backward(self, total_loss):
        """Gradients of the data loss flow through the HIP backward kernels into the flat
        gradient buffer; the L2 term's gradient (wd*w) is folded into the optimizer kernel."""
        self.store.grad.zero_()
        K.TAILS.begin()      # split-K reductions / BN parameter gradients are queued, finished in two launches
        try:
            total_loss.backward()
            SideStream.join()      # weight-gradient chain runs on a second stream (models/base/layers.py)
            K.TAILS.flush()
        finally:
            K.TAILS.active = False

    # ------------------------------------------------------------ fused step --
    accepts_next_image = True

    def _device_image(self, image):
        image = torch.as_tensor(image)
        if image.dim() == 3:
            image = image.unsqueeze(0)
        return image.to(self.device, torch.float32).contiguous()

    def train_step(self, image, gt_boxes, next_image=None, next_gt=None):
        """forward + loss + backward of ONE train step (train.py:66-91): same arithmetic as
        `__call__(is_training=True)` -> `loss()` -> `backward()`, issued as plain kernel calls (no torch.autograd, no
        framework arithmetic) on three HIP streams (`_step_body`), and — once a shape has been seen — re-issued from a
        recorded launch plan with one host call (luminoth_amd/plan.py).  Returns (total_loss, prediction_dict);
        gradients are complete in `self.store.grad` when the caller's stream reaches this point.

        ALIASING: a replayed step returns the tensors of the recorded one — `total_loss`, the loss scalars and every entry
        of `prediction_dict` are the SAME device buffers each time a plan of that parity runs, rewritten two steps later.
        Read (or `.clone()`) what you want to keep before calling the step after next; `luminoth_amd.train.run` fetches
        the loss right away.  Eager steps (plans off, shapes not yet recorded) return fresh tensors.

        The process-global tail queue is guarded: if anything raises mid-step (an argument check of a kernel, an
        out-of-memory) it is left inactive and empty, not silently swallowing the tails of whatever backward runs next."""
        try:
            if not self._with_rcnn or self._debug:
                pred = self(image, gt_boxes, is_training=True)
                total = self.loss(pred)
                self.backward(total)
                return total, pred
            return self._planned_step(image, gt_boxes, next_image, next_gt)
        except BaseException:
            K.TAILS.abort()
            self._step_state = {}
            from luminoth_amd.utils import training as T_
            if getattr(T_, 'ACTIVE_BUCKETS', None) is not None:      # gradient buckets / early range updates of this step
                T_.ACTIVE_BUCKETS.abort()
            raise
        finally:
            L.release_winograd_weights(self._winograd_layers())
            L.release_x3_weights(self._x3w_layers())

    def _x3w_layers(self):
        xl = getattr(self, '_x3w_layer_list', None)
        if xl is None:
            bn = self.base_network
            layers = bn.trunk.all_layers() + [self._rpn._rpn]
            if getattr(bn, 'tail', None) is not None and getattr(bn, '_use_tail', True):
                layers = layers + bn.tail.all_layers()          # ResNet-101: block4 on the pooled ROIs
            xl = self._x3w_layer_list = L.x3w_candidates(layers) if X3_PRESPLIT else []
        return xl

    def _winograd_layers(self):
        wl = getattr(self, '_wino_layers', None)
        if wl is None:
            layers = self.base_network.trunk.all_layers() + [self._rpn._rpn]
            wl = self._wino_layers = L.winograd_candidates(layers)
        return wl

    # ---- per-shape state of the fused step: buffers at fixed addresses + the recorded plans -------------------------
    @staticmethod
    def _gt_bucket(G):
        """Capacity of the fixed gt buffers for G boxes per image: 8, 16, 32, ... — the number of gt boxes varies from batch
        to batch with real data, and `gt_count` carries the real counts (the kernels never read a padded row), so it is
        not part of what identifies a shape."""
        g = 8
        while g < G:
            g *= 2
        return g

    def _gt_capacity(self, G):
        """The capacity a batch with G boxes per image is given: its bucket, but never less than the largest bucket this model
        has seen.  With real data the box count varies from batch to batch; keyed by each batch's own bucket,
        two batches of one image size landed in different states (8 vs 16 boxes), every switch ran eagerly and past four
        states recorded plans were destroyed.  The capacity only grows — at most four times up to 100 boxes — after which
        every batch of an image size maps to ONE state; `gt_count` carries the real counts and no kernel reads a padded row."""
        cap = max(self._gt_bucket(G), getattr(self, '_gt_cap_seen', 8))
        self._gt_cap_seen = cap
        return cap

    def _state_for(self, B, H, W, G):
        """What one step hands to the next and what the caller hands to a step lives at FIXED addresses, double-buffered
        by step parity p (step n uses slot n % 2 as "current" and fills slot 1 - p for step n + 1): images, gt boxes,
        seeds, the frozen trunk prefix computed one step ahead and the anchor targets computed one step ahead.  A
        recorded step can therefore be replayed: the addresses it reads and writes mean the same thing every time.
        One state per (batch, image size, gt capacity bucket); the 4 most recently USED are kept."""
        main = torch.cuda.current_stream(self.device)
        from luminoth_amd.utils import training as _tr
        G = self._gt_capacity(G)
        # the gradient buckets by GENERATION, not id(): a plan holds closures bound to the buckets object it was recorded
        # with, and CPython may hand a new object the id of a dead one
        key = (B, H, W, G, main.cuda_stream, K.OPTION_VERSION, getattr(_tr.ACTIVE_BUCKETS, 'generation', None),
               L.SideStream.enabled, getattr(self, '_phase_left', 0) > 0)
        states = getattr(self, '_step_state', None)
        if states is None:
            states = self._step_state = {}
        S = states.get(key)
        if S is not None:
            states[key] = states.pop(key)          # most recently used last
            return S
        if len(states) >= 4:          # a data loader with many shapes: keep the plans of the most recently used ones only
            old = states.pop(next(iter(states)))
            self.plan_stats['evicted_states'] += 1
            self.plan_stats['destroyed_plans'] += len(old['plans'])
            for pl in old['plans'].values():
                pl.destroy()
        dev = self.device
        bn = self.base_network
        fh, fw = bn.feature_hw(H, W)
        N = fh * fw * self._num_anchors
        start = bn.trunk.first_trainable()
        f32 = dict(dtype=torch.float32, device=dev)
        S = dict(key=key, n=0, plans={}, seen={}, pf=None,
                 images=[torch.empty((B, H, W, 3), **f32) for _ in range(2)],
                 gt=[torch.zeros((B, G, 5), **f32) for _ in range(2)],
                 cnt=[torch.zeros((B,), dtype=torch.int32, device=dev) for _ in range(2)],
                 seeds=[torch.zeros((B,), dtype=torch.int32, device=dev) for _ in range(2)],
                 tgt=[(torch.empty((B, N), **f32), torch.empty((B, N, 4), **f32), torch.empty((B, N), **f32))
                      for _ in range(2)],
                 prefix=None, start=start)
        if 0 < start:
            ph, pw = L.Trunk(bn.trunk.nodes[:start]).out_hw(H, W)
            ch = None
            for node in bn.trunk.nodes[start - 1::-1]:          # channels of the prefix output: the last convolution in it
                if hasattr(node, 'conv3') or hasattr(node, 'layer'):
                    ch = node.conv3.cout if hasattr(node, 'conv3') else node.layer.cout
                    break
            pdt = torch.float32
            if bn.storage_dtype in ('f16', 'bf16') and start > 1:
                pdt = K.half_type(bn.storage_dtype)[1]
            S['prefix'] = [torch.empty((B, ph, pw, ch), dtype=pdt, device=dev) for _ in range(2)]
        states[key] = S
        return S

    def _planned_step(self, image, gt_boxes, next_image, next_gt):
        from luminoth_amd import plan as P
        image_src = image
        image = torch.as_tensor(image)
        if image.dim() == 3:
            image = image.unsqueeze(0)
        B, H, W, _ = image.shape
        gt, gt_count = self._pack_gt(gt_boxes, B)
        S = self._state_for(B, H, W, gt.shape[1])
        p = S['n'] & 1
        S['n'] += 1
        main = torch.cuda.current_stream(self.device)
        # ---- this step's inputs: already in slot p if the previous step was told about them, else copied there now
        pf, S['pf'] = S['pf'], None
        have_pf = (pf is not None and pf['slot'] == p and pf['image'] is image_src and pf['image_v'] == image_src._version
                   and pf['gt'] is gt_boxes and pf['gt_v'] == self._gt_versions(gt_boxes) and pf['step'] == self._step)
        if not have_pf:
            self._fill_slot(S, p, image, gt, gt_count, B)
        self._step += 1
        # ---- the next step's inputs into the slot that step will read (its prefix / anchor targets are computed inside
        # this step).  Same shape: the other parity of this state.  Another shape (real data: the image size changes from
        # batch to batch): the current slot of THAT shape's state — the look-ahead happens either way; only a step whose
        # next batch has its own shape can be replayed from a plan (a plan bakes in the addresses it writes)
        bn_train = self.base_network.set_bn_mode(True)
        produce = (next_image is not None and next_gt is not None and PREFETCH_PREFIX and torch.is_tensor(next_image)
                   and not bn_train)
        NS, q = S, 1 - p
        if produce:
            nimg = next_image if next_image.dim() == 4 else next_image.unsqueeze(0)
            ngt, ncnt = self._pack_gt(next_gt, nimg.shape[0])
            nB, nH, nW = (int(v) for v in nimg.shape[:3])
            if (nB, nH, nW, self._gt_capacity(ngt.shape[1])) != S['key'][:4]:
                NS = self._state_for(nB, nH, nW, ngt.shape[1])
                q = NS['n'] & 1
        if produce:
            # (self._step already counts this step: these are the NEXT step's seeds)
            self._fill_slot(NS, q, nimg, ngt, ncnt, NS['key'][0])
            NS['pf'] = dict(slot=q, image=next_image, image_v=next_image._version, gt=next_gt,
                            gt_v=self._gt_versions(next_gt), step=self._step)
        variant = (p, bool(have_pf), bool(produce))
        # host-drawn dropout seeds and the in-place moving averages of training-mode BatchNorm are per-step state a
        # recorded plan would freeze / the look-ahead would advance early: those configurations run every step eagerly
        plannable = (P.ENABLED and self._rcnn._dropout_keep_prob in (None, 1, 1.0) and not bn_train and NS is S)
        plan = S['plans'].get(variant) if plannable else None
        self.plan_stats['replayed_steps' if plan is not None else 'eager_steps'] += 1
        if plan is None and P.ENABLED and NS is not S:
            self.plan_stats['eager_because_next_batch_has_another_shape'] += 1
        if plan is not None:
            self._phase_collect(S, p)
            out = plan.run()
            self._last_losses = out[2]
            self._phase_host_done(S, p)
            return out[0], out[1]
        seen = S['seen'].get(variant, 0)
        S['seen'][variant] = seen + 1
        if plannable and seen >= P.WARM_STEPS and self._plan_budget_left():
            plan = P.StepPlan()
            with plan:
                out = self._step_body(S, p, have_pf, produce, B, H, W, NS, q)
                plan.result = out
                plan.keep.append(out)
            S['plans'][variant] = plan
        else:
            out = self._step_body(S, p, have_pf, produce, B, H, W, NS, q)
        self._last_losses = out[2]
        return out[0], out[1]

    @staticmethod
    def _gt_versions(gt):
        """In-place modification counters of the tensors of a gt argument (a loader that refills its buffers between
        announcing a batch and passing it must not be served the copy taken earlier); None where nothing can be versioned
        (numpy arrays, lists of arrays: the identity check is all there is, as for the reference's feed dicts)."""
        if torch.is_tensor(gt):
            return gt._version
        if isinstance(gt, (tuple, list)):
            return tuple(g._version if torch.is_tensor(g) else None for g in gt)
        return None

    def _fill_slot(self, S, slot, image, gt, gt_count, B):
        """The caller's batch into the fixed-address buffers of `slot` (gt rows beyond this batch's Gmax are zeroed: the
        buffers have the bucket's capacity)."""
        S['images'][slot].copy_(image, non_blocking=True)
        G = gt.shape[1]
        buf = S['gt'][slot]
        buf[:, :G].copy_(gt)
        if G < buf.shape[1]:
            buf[:, G:].zero_()
        S['cnt'][slot].copy_(gt_count)
        S['seeds'][slot].copy_(self._image_seeds(B))

    # every tensor whose address entered a recorded launch stays alive with the plan: all activations of a step (~3 GB at
    # config 2).  LUMINOTH_AMD_PLAN_MAX_GB bounds what the plans of one model may pin; past it new variants run eagerly.
    PLAN_MAX_BYTES = int(float(os.environ.get('LUMINOTH_AMD_PLAN_MAX_GB', '64')) * (1 << 30))

    def _plan_budget_left(self):
        from luminoth_amd import plan as P
        pinned = sum(P.pinned_bytes(pl) for S in self._step_state.values() for pl in S['plans'].values())
        return pinned < self.PLAN_MAX_BYTES

    def _step_body(self, S, p, have_pf, produce, B, H, W, NS=None, q=None):
        """The device work of one train step on three HIP streams, from / into the fixed-address buffers of `S`:

            main : trunk fwd -> RPN convs -> RPN targets -> RPN loss -> RPN backward -> [next batch: frozen prefix] -> trunk backward -> tails
            aux  :               '-> proposals (sort + NMS) -> RCNN targets -> ROI pool -> FCs -> RCNN loss
                                     -> RCNN backward -> ROI-pool backward (+ the RPN branch's gradient) --'  [next batch: anchor targets]
            side : every layer's weight-gradient kernel beside that layer's data gradient

        Everything here launches through luminoth_amd.kernels (recordable by a launch plan); cross-stream order is
        made with K.stream_wait only.  -> (total_loss, prediction_dict, losses dict)."""
        image, gt, gt_count, seeds = S['images'][p], S['gt'][p], S['cnt'][p], S['seeds'][p]
        if NS is None:
            NS, q = S, 1 - p
        im_shape = (H, W)
        main = torch.cuda.current_stream(self.device)
        aux = self._aux_stream()
        bn = self.base_network
        rpn, rcnn = self._rpn, self._rcnn
        self._phase_begin(S, p)
        # BatchNorm scale / shift of every layer, 16-bit working weights, Winograd weight transforms of the whole step: the
        # forward set here, the backward set on the (idle) weight-gradient stream while the forward pass runs.  Nothing
        # writes the weights or the BatchNorm scales before the update.
        bn.bn_table.refresh()
        if bn._hs_layers:
            L.prepare_half_weights(bn._hs_layers + bn.extra_hs_layers, bn.storage_dtype)
        # (training-mode BatchNorm: the backward weights are NOT pre-scaled by a frozen BatchNorm scale: transformed per call)
        wl = self._winograd_layers() if (WINO_BATCH and not any(l.bn_train for l in self.base_network.trunk.all_layers())) else []
        # bf16x3: the weights of the layers that multiply them directly, split once for the whole step (csrc/conv_x3.h)
        xl = self._x3w_layers() if not any(l.bn_train for l in self.base_network.trunk.all_layers()) else []
        wino_bwd_side = None
        if wl:
            L.prepare_winograd_weights(wl, backward=False)
        if xl:
            L.prepare_x3_weights(xl, backward=False)
        if wl or xl:
            wino_bwd_side = SideStream.get(self.device)
            K.stream_wait(wino_bwd_side, main)
            with K.launch_on(wino_bwd_side):
                if wl:
                    L.prepare_winograd_weights([l for l in wl if l.trainable], backward=True)
                if [l for l in xl if l.trainable]:
                    L.prepare_x3_weights([l for l in xl if l.trainable], backward=True)
        K.zero_(self.store.grad)
        start = S['start']
        if produce and start > 0 and PREFIX_AT in ('side', 'aux') and SideStream.enabled:
            # HBM-bound (conv1 + block1 on 256-channel fp32 maps) beside the MFMA-bound trunk forward
            early = SideStream.get(self.device) if PREFIX_AT == 'side' else aux
            K.stream_wait(early, main)
            with torch.cuda.stream(early):
                self._prefix_forward(0, start, NS['images'][q], NS['prefix'][q])
                self._mark('early:next_prefix_done')
        from luminoth_amd.utils import training as _tr
        # the aux stream is idle once the RCNN branch is done: weight-gradient tails of the trunk backward are finished
        # there in batches while the MFMA kernels run (not with gradient buckets: those flush on their own stream)
        K.TAILS.begin(early=None if _tr.ACTIVE_BUCKETS is not None else (aux, lambda: list(SideStream._streams.values())))
        fh, fw = bn.feature_hw(H, W)
        # ---- trunk forward: the frozen prefix (conv1 + fixed blocks) of THIS batch was computed during the previous step
        nodes = bn.trunk.nodes
        x0 = image
        if start > 0:
            if have_pf:
                x0 = S['prefix'][p]
            else:
                x0 = self._prefix_forward(0, start, image, S['prefix'][p])
        saved = None
        if start < len(nodes):
            feat, saved = self._sub_trunk(start, len(nodes)).forward(x0, save_from=0)
        else:
            feat = x0
        assert (feat.shape[1], feat.shape[2]) == (fh, fw)
        self._mark('trunk_fwd_done')
        rpn_pred, rpn_ctx = rpn.heads_fwd(feat)
        self._mark('rpn_heads_done')
        # Experiment (LUMINOTH_AMD_PREFIX_SPLIT=1, off): the next batch's stem (conv1 + max-pool) HERE, while only the small
        # kernels of the proposal chain are in flight — its 62 KB of LDS per block do not fit beside the 96-131 KB blocks
        # of the RPN weight gradient / ROI pooling that fill every CU a moment later, and a kernel that arrives then
        # waits until those grids are fully dispatched (the stem starts ~200 us after the kernel in front of it).
        # Measured: fp32 7.01 -> 6.99 ms (noise), f16 4.20 -> 4.30 ms (the stem then delays the RPN backward and slows the
        # proposal chain, which bounds the f16 step) — rejected
        split_prefix = (produce and start > 2 and PREFIX_AT == 'middle' and PREFIX_SPLIT)
        stem_out = None
        if split_prefix:
            stem_out, _ = self._sub_trunk(0, 2).forward(NS['images'][q], save_from=None)
            self._mark('next_stem_done')
        # Host enqueue order matters while the host is not far ahead of the GPU (eager steps): the proposal chain is
        # ONE C call (cheap to enqueue, long to run), so it goes first; then the RPN branch of the main stream; the
        # RCNN part of the aux stream last (it cannot start before the NMS finishes anyway).
        K.stream_wait(aux, main)
        with torch.cuda.stream(aux):
            prop = rpn._proposal(rpn_pred['rpn_cls_score'], rpn_pred['rpn_bbox_pred'], self._anchor_ref_i32, (fh, fw),
                                 self._anchor_stride, im_shape)
            self._mark('aux:proposals_done')
            rcnn_tgt = rcnn.targets(prop['proposals'], prop['num_proposals'], gt, gt_count, seeds)
            self._mark('aux:rcnn_targets_done')
        # ---- main stream: RPN targets -> RPN loss -> RPN backward
        rpn_tgt = {}
        if have_pf:          # computed on the aux stream during the previous step, which the main stream joined at its end
            rpn_tgt['rpn_cls_target'], rpn_tgt['rpn_bbox_target'] = S['tgt'][p][0], S['tgt'][p][1]
        else:
            rpn.targets(rpn_tgt, self._anchor_ref_i32, (fh, fw), self._anchor_stride, gt, gt_count, seeds, im_shape,
                        out=S['tgt'][p])
        rpn_pred.update(rpn_tgt)
        rpn_losses, rpn_g = rpn.loss_and_grads(rpn_pred, self._rpn_cls_loss_weight, self._rpn_reg_loss_weight)
        if wino_bwd_side is not None:
            K.stream_wait(main, wino_bwd_side)      # transformed backward weights (enqueued before the forward pass: long done)
        rpn_bwd_stream = main
        # late: nothing MFMA-bound runs beside the proposal chain and the prefix; the RPN data gradients follow the join
        # (not when the RPN convolution is a half-storage layer with fp32 tensors on both sides: no addend there)
        rpn_late = RPN_BWD_LATE and not (rpn._rpn.storage is not None and rpn._rpn.hs_in_f32)
        d_feat_rpn = None
        if rpn_late:
            pass
        elif SideStream.enabled and (RPN_BWD_SIDE == '1' or (RPN_BWD_SIDE == 'auto' and not bn._hs_layers)):
            # the RPN backward (MFMA-bound) on the weight-gradient stream, which is idle until the trunk backward starts:
            # it then runs beside the next batch's prefix (HBM-bound) on the main stream and the proposal chain
            # (latency-bound) on the aux stream instead of in front of the prefix
            rpn_bwd_stream = SideStream.get(self.device)
            K.stream_wait(rpn_bwd_stream, main)
            with torch.cuda.stream(rpn_bwd_stream):
                SideStream.layers_left = 3          # its weight gradients follow their data gradients on that stream
                d_feat_rpn = rpn.heads_bwd(rpn_ctx, rpn_g[0], rpn_g[1])
                SideStream.layers_left = 0
                self._mark('side:rpn_bwd_done')
        else:
            SideStream.layers_left = 0
            d_feat_rpn = rpn.heads_bwd(rpn_ctx, rpn_g[0], rpn_g[1])
            self._mark('rpn_bwd_done')
        # ---- aux stream: RCNN forward -> RCNN loss -> RCNN backward; the ROI-pooling backward adds the RPN branch's
        # gradient of the feature map in its store (TF: AddN over the two consumers of conv_feature_map)
        with torch.cuda.stream(aux):
            self._mark('aux:rcnn_enqueue')
            cp, rcnn_ctx = rcnn.train_fwd(feat, rcnn_tgt, im_shape, bn)
            if RCNN_LOSS_LATE:
                # gradients only (one grid-wide launch); the reported sums — a one-block-per-image kernel — behind the join
                rcnn_losses, rcnn_g = None, rcnn.loss_grads(cp, self._rcnn_cls_loss_weight, self._rcnn_reg_loss_weight)
            else:
                rcnn_losses, rcnn_g = rcnn.loss_and_grads(cp, self._rcnn_cls_loss_weight, self._rcnn_reg_loss_weight)
            self._mark('aux:rcnn_loss_done')
            d_feat = rcnn.train_bwd(rcnn_ctx, rcnn_g[0], rcnn_g[1], addend=d_feat_rpn,
                                    before_pool_bwd=None if rpn_late else (lambda: K.stream_wait(aux, rpn_bwd_stream)))
            self._mark('aux:rcnn_bwd_done')
        # ---- the main stream has nothing left but to wait for the RCNN branch: the slot for the frozen trunk prefix of
        # the NEXT step's images (conv1 + fixed blocks: nothing this step's update writes)
        if produce and start > 0 and not (PREFIX_AT in ('side', 'aux') and SideStream.enabled):
            if split_prefix:
                self._prefix_forward(2, start, stem_out, NS['prefix'][q])
            else:
                self._prefix_forward(0, start, NS['images'][q], NS['prefix'][q])
            self._mark('next_prefix_done')
        # ---- join (the wait captures the aux stream as of NOW: what is queued there below does not delay the trunk backward)
        K.stream_wait(main, aux)
        self._mark('joined')
        if rpn_late:
            SideStream.layers_left = 0          # weight gradients on the side stream, beside the start of the trunk backward
            d_feat = rpn.heads_bwd(rpn_ctx, rpn_g[0], rpn_g[1], addend=d_feat)
            self._mark('rpn_bwd_done')
        with torch.cuda.stream(aux):
            # the loss scalars are only reported: built on the stream that has nothing else to do
            if rcnn_losses is None:
                rcnn_losses = rcnn.loss_values(cp, self._rcnn_cls_loss_weight, self._rcnn_reg_loss_weight)
            reg = K.l2_reg_loss(self.store.flat, self.store.seg_offset, self.store.seg_wd)
            sums = K.loss_sums([rpn_losses['rpn_cls_loss'], rpn_losses['rpn_reg_loss'],
                                rcnn_losses['rcnn_cls_loss'], rcnn_losses['rcnn_reg_loss']], reg, self._frozen_reg_tensor())
            # ... and the anchor targets of the NEXT batch (they depend on its gt boxes and this model's seeds only)
            if produce:
                nH, nW = NS['key'][1], NS['key'][2]          # the next batch may have its own image size
                rpn.targets({}, self._anchor_ref_i32, bn.feature_hw(nH, nW), self._anchor_stride, NS['gt'][q], NS['cnt'][q],
                            NS['seeds'][q], (nH, nW), out=NS['tgt'][q])
                self._mark('aux:next_targets_done')
        total_loss, no_reg_loss, regularization_loss = sums[0], sums[1], sums[2]
        # ---- trunk backward.  Data parallel: the head gradients (RPN on main / side, RCNN joined from aux) are complete
        # here, so the gradient buckets may start all-reducing under the trunk backward (utils/training.py)
        if saved is not None:
            buckets = _tr.ACTIVE_BUCKETS
            if buckets is not None and buckets.store is self.store:
                buckets.arm(bn.trunk)
            self._sub_trunk(start, len(nodes)).backward(saved, d_feat, 0, need_dx_first=False)
            if buckets is not None:
                buckets.disarm()
        self._mark('trunk_bwd_data_done')
        SideStream.join()
        self._mark('wgrad_stream_joined')
        K.TAILS.flush()          # what is left of the weight-gradient tails (RPN, RCNN, trunk) in two launches
        K.TAILS.active = False
        K.stream_wait(main, aux)     # early tail batches, the loss scalars, the next batch's anchor targets
        K.TAILS.early = None
        self._mark('tails_done')
        rpn_pred.update({k: prop[k] for k in ('rpn_cls_prob', 'proposals', 'scores')})
        rpn_pred['num_proposals'] = prop['num_proposals']
        losses = dict(rpn_losses, total_loss=total_loss, no_reg_loss=no_reg_loss,
                      regularization_loss=regularization_loss, **rcnn_losses)
        pred = {'rpn_prediction': rpn_pred, 'classification_prediction': cp, 'rpn_loss_dict': rpn_losses,
                'rcnn_loss_dict': rcnn_losses, '_batch': {'B': B, 'unbatched': False}}
        return total_loss, pred, losses

    def _prefix_forward(self, lo, hi, x, out):
        """Nodes [lo, hi) of the trunk forward-only into the fixed-address buffer `out`: written by the last node itself
        when it is a convolution, copied there when it is a pooling node (VGG: the frozen prefix ends in pool2)."""
        sub = self._sub_trunk(lo, hi)
        if isinstance(sub.nodes[-1], L.MaxPoolNode):
            y, _ = sub.forward(x, save_from=None)
            return K.copy_(out, y)
        y, _ = sub.forward(x, save_from=None, out=out)
        return y

    def _sub_trunk(self, lo, hi):
        c = getattr(self, '_sub_trunks', None)
        if c is None:
            c = self._sub_trunks = {}
        t = c.get((lo, hi))
        if t is None:
            t = c[(lo, hi)] = L.Trunk(self.base_network.trunk.nodes[lo:hi])
        return t

    def _frozen_reg_tensor(self):
        if self._frozen_reg is None:
            st = self.store
            self._frozen_reg = K.l2_reg_loss(st.frozen, st.frozen_seg_offset, st.frozen_seg_wd)
        return self._frozen_reg

    # -------------------------------------------------------------- diagnostics --
    def record_phases(self, steps):
        """Arm HIP-event marks for the next `steps` train steps (bench.py --phases): the un-profiled timeline of the
        three-stream schedule.  rocprofv3 slows the launch path enough to change where the step waits, so the gaps in a
        kernel trace are not the gaps of the real step; a handful of event records are.  The marks are library events
        recorded through the C ABI, so they are part of a launch plan and are re-recorded at every replay; a slot's marks
        are read back right before the slot is used again (two steps later: the host never waits for the GPU's current
        step)."""
        self._phase_left = steps
        self._phase_sum, self._phase_n, self._phase_next_n = {}, 0, 0

    def _phase_begin(self, S, p):
        self._phase_cur = None
        if not S['key'][-1]:
            return
        self._phase_collect(S, p)
        self._phase_left = max(0, getattr(self, '_phase_left', 0) - 1)
        self._phase_cur = S.setdefault('marks', {}).setdefault(p, [])
        del self._phase_cur[:]
        S.setdefault('pending', {})[p] = True
        self._mark('step_start')

    def _mark(self, name):
        cur = getattr(self, '_phase_cur', None)
        if cur is None:
            return
        from luminoth_amd import plan as P
        if P.recording():
            ev = P.active().new_event()
        else:
            ev = _lib.load().lmh_event_create()
        K.event_record(ev, torch.cuda.current_stream(self.device))
        cur.append((name, ev, time.perf_counter()))

    def _phase_host_done(self, S, p):
        """A replayed step: its marks were re-recorded by the plan; what the host can say is when it was done enqueueing."""
        if S['key'][-1] and S.get('marks', {}).get(p):
            self._phase_left = max(0, getattr(self, '_phase_left', 0) - 1)
            S.setdefault('pending', {})[p] = True
            S.setdefault('host_done', {})[p] = time.perf_counter()

    def _phase_collect(self, S, p):
        """Read the marks slot p recorded two steps ago (synchronises with THAT step only) into the running sums."""
        if not S['key'][-1] or not S.get('pending', {}).get(p):
            if S['key'][-1]:
                S.setdefault('host_t0', {})[p] = time.perf_counter()
            return
        lib = _lib.load()
        marks = S['marks'][p]
        acc = self._phase_sum
        e0, h0 = marks[0][1], marks[0][2]
        replayed = p in S.get('host_done', {})
        for name, ev, host_t in marks[1:]:
            acc[name] = acc.get(name, 0.0) + lib.lmh_event_elapsed_ms(ctypes.c_void_p(e0), ctypes.c_void_p(ev))
            if not replayed:
                # when the HOST enqueued the mark, relative to the step's first enqueue
                acc['host:' + name] = acc.get('host:' + name, 0.0) + (host_t - h0) * 1e3
        if replayed:
            acc['host:step_enqueued'] = acc.get('host:step_enqueued', 0.0) + \
                (S['host_done'].pop(p) - S.get('host_t0', {}).get(p, h0)) * 1e3
        other = S['marks'].get(1 - p)
        if other and S['pending'].get(1 - p):       # the following step's start, recorded by the other slot
            dt = lib.lmh_event_elapsed_ms(ctypes.c_void_p(e0), ctypes.c_void_p(other[0][1]))
            if dt > 0:
                acc['next_step_start'] = acc.get('next_step_start', 0.0) + dt
                self._phase_next_n += 1
        self._phase_n += 1
        S['pending'][p] = False
        S.setdefault('host_t0', {})[p] = time.perf_counter()

    def phase_times(self):
        """-> {mark: mean ms after step_start} over the recorded steps (synchronises), plus 'next_step_start'."""
        torch.cuda.synchronize(self.device)
        for S in getattr(self, '_step_state', {}).values():
            if S['key'][-1]:
                order = sorted(S.get('pending', {}), key=lambda q: (S['n'] - 1 - q) & 1, reverse=True)
                for q in order:          # older slot first, so that it still sees the newer slot's step_start
                    self._phase_collect(S, q)
        n = max(1, getattr(self, '_phase_n', 0))
        res = {k: v / (max(1, self._phase_next_n) if k == 'next_step_start' else n)
               for k, v in getattr(self, '_phase_sum', {}).items()}
        self._phase_cur = None
        self._phase_left = 0
        return res

    _AUX_STREAMS = {}      # device -> stream, shared by every model of the process

    def _aux_stream(self):
        # ONE proposal / RCNN stream per device, not per model: HIP maps streams onto a handful of hardware queues, and
        # a second model's extra streams alias the first one's (measured: the same step at 12.1 instead of 7.1 ms when a
        # second model with its own aux stream ran in the process).  High priority: the chain is a string of small
        # latency-bound launches; its blocks must not queue behind the CU-filling convolution grids of the other streams.
        key = str(self.device)
        st = FasterRCNN._AUX_STREAMS.get(key)
        if st is None:
            spec = os.environ.get('LUMINOTH_AMD_AUX_CU_MASK', '')       # experiment: the chain on compute units of its own
            st = K.cu_range_stream(spec, self.device) if spec else torch.cuda.Stream(device=self.device, priority=-1)
            FasterRCNN._AUX_STREAMS[key] = st
        return st

    # --------------------------------------------------------------- variables --
    @property
    def summary(self):
        """Scalar summaries (the reference merges TensorBoard summaries here: fasterrcnn.py:310-327)."""
        return {k: float(v) for k, v in getattr(self, '_last_losses', {}).items()}

    @property
    def vars_summary(self):
        return {}

    def get_trainable_vars(self):
        """Module variables + the fine-tuned base-network variables (fasterrcnn.py:337-358):
        OrderedDict name -> tensor view (with `.grad` views in `self.store.grads`)."""
        st = self.store
        return {n: st.params[n] for n in st.trainable_names()}

    def get_base_network_checkpoint_vars(self):
        return self.base_network.get_base_network_checkpoint_vars(self.store)

    def get_checkpoint_file(self):
        return self.base_network.get_checkpoint_file()

    def state_dict(self):
        return self.store.state_dict()

    def load_state_dict(self, sd, strict=True):
        self.store.load_state_dict(sd, strict=strict)
        self.base_network.bn_table.reload_statistics()
        self._frozen_reg = None
