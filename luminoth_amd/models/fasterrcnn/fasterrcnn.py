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
WINO_BATCH = os.environ.get('LUMINOTH_AMD_WINO_BATCH', '1') != '0'      # transformed Winograd weights of the whole step in two launches


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
        if self._frozen_reg is None:
            st = self.store
            self._frozen_reg = K.l2_reg_loss(st.frozen, st.frozen_seg_offset, st.frozen_seg_wd)
        st = self.store
        return (K.l2_reg_loss(st.flat, st.seg_offset, st.seg_wd) + self._frozen_reg)[0]

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
        """`_train_step` with the process-global tail queue guarded: if anything raises mid-step (an argument check of a
        kernel, an out-of-memory) the queue is left inactive and empty, not silently swallowing the tails of whatever
        backward runs next."""
        try:
            return self._train_step(image, gt_boxes, next_image=next_image, next_gt=next_gt)
        except BaseException:
            K.TAILS.abort()
            self._prefetch = self._tgt_prefetch = None
            raise
        finally:
            L.release_winograd_weights(self._winograd_layers())

    def _winograd_layers(self):
        wl = getattr(self, '_wino_layers', None)
        if wl is None:
            layers = self.base_network.trunk.all_layers() + [self._rpn._rpn]
            wl = self._wino_layers = L.winograd_candidates(layers)
        return wl

    def _train_step(self, image, gt_boxes, next_image=None, next_gt=None):
        """forward + loss + backward of ONE train step (train.py:66-91), same arithmetic as
        `__call__(is_training=True)` -> `loss()` -> `backward()`, scheduled on two HIP streams:

            main: trunk fwd -> RPN convs -> RPN targets -> RPN loss -> RPN backward ------> trunk backward
            aux :               '-> proposals (sort + NMS) -> RCNN targets -> ROI pool -> FCs
                                    -> RCNN loss -> RCNN backward -> ROI-pool backward ----'

        The proposal chain is a sequence of latency-bound launches that occupy a handful of CUs; it now
        runs beside the RPN backward (the largest MFMA kernels of the step) instead of in front of it.
        Returns (total_loss, prediction_dict); gradients are complete in `self.store.grad` when the
        caller's stream reaches this point."""
        if not self._with_rcnn:
            pred = self(image, gt_boxes, is_training=True)
            total = self.loss(pred)
            self.backward(total)
            return total, pred
        pf, self._prefetch = getattr(self, '_prefetch', None), None
        if pf is not None and pf[0] is image and pf[1] == image._version:
            image = pf[2]            # uploaded (and its frozen trunk prefix computed) during the previous step
        else:
            image = self._device_image(image)
        B, H, W, _ = image.shape
        gt, gt_count = self._pack_gt(gt_boxes, B)
        seeds = self._image_seeds(B)
        pt, self._tgt_prefetch = getattr(self, '_tgt_prefetch', None), None
        if pt is not None and not (pt['src'] is gt_boxes and pt['key'] == (self._step, B, H, W)):
            pt = None            # another batch arrived than the one announced: compute the targets now
        self._step += 1
        im_shape = (H, W)
        main = torch.cuda.current_stream(self.device)
        aux = self._aux_stream()
        self._phase_begin()
        # Winograd weight transforms of the whole step: the forward set in one launch here, the backward set in one launch
        # on the (idle) weight-gradient stream while the forward pass runs — instead of one small launch inside each of
        # the ~20 Winograd convolution calls.  Nothing writes the weights or the BatchNorm scales before the update.
        self.base_network.bn_table.refresh()
        self.base_network._bn_fresh = True
        wl = self._winograd_layers() if WINO_BATCH else []
        wino_bwd_ready = None
        if wl:
            L.prepare_winograd_weights(wl, backward=False)
            side0 = SideStream.get(self.device)
            side0.wait_stream(main)
            with torch.cuda.stream(side0):
                L.prepare_winograd_weights([l for l in wl if l.trainable], backward=True)
                wino_bwd_ready = torch.cuda.Event()
                wino_bwd_ready.record(side0)
        self.store.grad.zero_()
        from luminoth_amd.utils import training as _tr
        # the aux stream is idle once the RCNN branch is done: weight-gradient tails of the trunk backward are finished
        # there in batches while the MFMA kernels run (not with gradient buckets: those flush on their own stream)
        K.TAILS.begin(early=None if _tr.ACTIVE_BUCKETS is not None else (aux, lambda: list(SideStream._streams.values())))
        with torch.enable_grad():
            fh, fw = self.base_network.feature_hw(H, W)
            rpn = self._rpn
            rpn_tgt = {}
            for t in (gt, gt_count, seeds):
                K.keep_alive(t, aux)
            feat = self.base_network(image, is_training=True)
            assert (feat.shape[1], feat.shape[2]) == (fh, fw)
            self._mark('trunk_fwd_done')
            f_rpn = feat.detach().requires_grad_(True)
            f_rcnn = feat.detach().requires_grad_(True)
            rpn_pred = rpn.heads(f_rpn)
            self._mark('rpn_heads_done')
            # Host enqueue order matters while the host is not far ahead of the GPU: the proposal chain is
            # ONE C call (cheap to enqueue, long to run), so it goes first; then the RPN branch of the main
            # stream; the RCNN part of the aux stream last (it cannot start before the NMS finishes anyway).
            # ---- aux stream: proposals
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                prop = rpn._proposal(rpn_pred['rpn_cls_score'].detach(), rpn_pred['rpn_bbox_pred'].detach(),
                                     self._anchor_ref_i32, (fh, fw), self._anchor_stride, im_shape)
                self._mark('aux:proposals_done')
                rcnn_tgt = self._rcnn.targets(prop['proposals'], prop['num_proposals'], gt, gt_count, seeds)
                self._mark('aux:rcnn_targets_done')
            for t in (rpn_pred['rpn_cls_score'], rpn_pred['rpn_bbox_pred'], feat):
                K.keep_alive(t, aux)
            # ---- main stream: RPN targets -> RPN loss -> RPN backward
            if pt is not None:       # anchor targets of this batch were computed on the aux stream during the previous step
                main.wait_event(pt['event'])
                rpn_tgt = pt['tgt']
                for t in rpn_tgt.values():
                    K.keep_alive(t, main)
            else:
                rpn.targets(rpn_tgt, self._anchor_ref_i32, (fh, fw), self._anchor_stride, gt, gt_count, seeds, im_shape)
            rpn_pred.update(rpn_tgt)
            rpn_losses, rpn_g = rpn.loss_and_grads(rpn_pred, self._rpn_cls_loss_weight, self._rpn_reg_loss_weight)
            rpn_loss_done = torch.cuda.Event()
            rpn_loss_done.record(main)
            if wino_bwd_ready is not None:
                main.wait_event(wino_bwd_ready)      # (recorded before the forward pass was even enqueued: long done)
            torch.autograd.backward([rpn_pred['rpn_cls_score'], rpn_pred['rpn_bbox_pred']],
                                    [g.view_as(t) for g, t in zip(rpn_g, (rpn_pred['rpn_cls_score'], rpn_pred['rpn_bbox_pred']))])
            self._mark('rpn_bwd_done')
            # ---- aux stream: RCNN forward -> RCNN loss -> RCNN backward
            with torch.cuda.stream(aux):
                self._mark('aux:rcnn_enqueue')      # later than aux:rcnn_targets_done = the host was not ahead here
                cp = self._rcnn(f_rcnn, prop['proposals'], prop['num_proposals'], im_shape, self.base_network,
                                gt_boxes=gt, gt_count=gt_count, seeds=seeds, is_training=True, targets=rcnn_tgt)
                rcnn_losses, rcnn_g = self._rcnn.loss_and_grads(cp, self._rcnn_cls_loss_weight, self._rcnn_reg_loss_weight)
                self._mark('aux:rcnn_loss_done')
                torch.autograd.backward([cp['rcnn']['cls_score'], cp['rcnn']['bbox_offsets']],
                                        [g.view_as(t) for g, t in zip(rcnn_g, (cp['rcnn']['cls_score'], cp['rcnn']['bbox_offsets']))])
                self._mark('aux:rcnn_bwd_done')
                rcnn_done = torch.cuda.Event()
                rcnn_done.record(aux)          # the join below waits for THIS, not for what the aux stream is given next
                # the loss scalars (sums, L2 regulariser: a handful of tiny launches) are only reported: they are built here,
                # on the stream that has nothing else to do, not in front of the trunk backward
                aux.wait_event(rpn_loss_done)
                no_reg_loss = (rpn_losses['rpn_cls_loss'] + rpn_losses['rpn_reg_loss'] +
                               rcnn_losses['rcnn_cls_loss'] + rcnn_losses['rcnn_reg_loss'])
                regularization_loss = self.regularization_loss()
                total_loss = no_reg_loss + regularization_loss
                # the aux stream is idle from here to the end of the step: the anchor targets of the NEXT batch (they
                # depend on its gt boxes and this model's seeds only, not on any weight) leave the next step's critical path
                if next_gt is not None and next_image is not None and PREFETCH_PREFIX and torch.is_tensor(next_image):
                    nshape = tuple(next_image.shape) if next_image.dim() == 4 else (1,) + tuple(next_image.shape)
                    Bn, Hn, Wn = nshape[0], nshape[1], nshape[2]
                    ngt, ncnt = self._pack_gt(next_gt, Bn)
                    nseeds = self._image_seeds(Bn)             # self._step already counts this step: the next step's seeds
                    ntgt = {}
                    rpn.targets(ntgt, self._anchor_ref_i32, self.base_network.feature_hw(Hn, Wn), self._anchor_stride,
                                ngt, ncnt, nseeds, (Hn, Wn))
                    ev = torch.cuda.Event()
                    ev.record(aux)
                    self._tgt_prefetch = dict(src=next_gt, key=(self._step, Bn, Hn, Wn), tgt=ntgt, event=ev,
                                              keep=(ngt, ncnt, nseeds))
                    self._mark('aux:next_targets_done')
            # ---- the main stream has nothing left but to wait for the RCNN branch (0.3-0.4 ms at config 2): the slot
            # for the frozen trunk prefix of the NEXT step's images (conv1 + block1: nothing this step's update writes)
            if next_image is not None and PREFETCH_PREFIX and torch.is_tensor(next_image):
                nxt = self._device_image(next_image)
                if self.base_network.prefetch_prefix(nxt):
                    self._prefetch = (next_image, next_image._version, nxt)
                self._mark('next_prefix_done')
            # ---- join, trunk backward
            main.wait_event(rcnn_done)
            self._mark('joined')
            K.keep_alive(f_rcnn.grad, main)
            for t in (rpn_losses['rpn_cls_loss'], rpn_losses['rpn_reg_loss']):
                K.keep_alive(t, aux)
            # data parallel: head gradients (RPN on main/side, RCNN joined from aux) are complete here, so the
            # gradient buckets may start all-reducing under the trunk backward (utils/training.py)
            from luminoth_amd.utils import training as _tr
            buckets = _tr.ACTIVE_BUCKETS
            if buckets is not None and buckets.store is self.store:
                buckets.arm(self.base_network.trunk)
            feat.backward(f_rpn.grad + f_rcnn.grad)
            if buckets is not None:
                buckets.disarm()
            self._mark('trunk_bwd_data_done')
        SideStream.join()
        self._mark('wgrad_stream_joined')
        K.TAILS.flush()          # what is left of the weight-gradient tails (RPN, RCNN, trunk) in two launches
        K.TAILS.active = False
        main.wait_stream(aux)        # early tail batches and the loss scalars; long finished by now
        for t in (total_loss, no_reg_loss, regularization_loss, rcnn_losses['rcnn_cls_loss'], rcnn_losses['rcnn_reg_loss']):
            K.keep_alive(t, main)
        K.TAILS.early = None
        self._mark('tails_done')
        rpn_pred.update({k: prop[k] for k in ('rpn_cls_prob', 'proposals', 'scores')})
        rpn_pred['num_proposals'] = prop['num_proposals']
        self._last_losses = dict(rpn_losses, total_loss=total_loss, no_reg_loss=no_reg_loss,
                                 regularization_loss=regularization_loss, **rcnn_losses)
        pred = {'rpn_prediction': rpn_pred, 'classification_prediction': cp, 'rpn_loss_dict': rpn_losses,
                'rcnn_loss_dict': rcnn_losses, '_batch': {'B': B, 'unbatched': False}}
        return total_loss, pred

    # -------------------------------------------------------------- diagnostics --
    def record_phases(self, steps):
        """Arm HIP-event marks for the next `steps` train steps (bench.py --phases): the un-profiled timeline of the
        three-stream schedule.  rocprofv3 slows the launch path enough to make the step host-bound where it forks, so
        the gaps in a kernel trace are not the gaps of the real step; a handful of event records are."""
        self._phase_left = steps
        self._phase_log = []

    def _mark(self, name, stream=None):
        cur = getattr(self, '_phase_cur', None)
        if cur is None:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream if stream is not None else torch.cuda.current_stream(self.device))
        cur.append((name, ev, time.perf_counter()))

    def _phase_begin(self):
        left = getattr(self, '_phase_left', 0)
        self._phase_cur = None
        if left > 0:
            self._phase_left = left - 1
            self._phase_cur = []
            self._phase_log.append(self._phase_cur)
            self._mark('step_start')

    def phase_times(self):
        """-> {mark: mean ms after step_start} over the recorded steps (synchronises), plus 'next_step_start'."""
        torch.cuda.synchronize(self.device)
        log = getattr(self, '_phase_log', [])
        out, n = {}, 0
        for i, marks in enumerate(log):
            t0 = marks[0][1]
            for name, ev, host_t in marks[1:]:
                out[name] = out.get(name, 0.0) + t0.elapsed_time(ev)
                # when the HOST enqueued the mark, relative to the step's first enqueue: a mark whose host time is
                # later than its GPU time was waited for by the GPU (the step is host-bound there)
                out['host:' + name] = out.get('host:' + name, 0.0) + (host_t - marks[0][2]) * 1e3
            if i + 1 < len(log):
                out['next_step_start'] = out.get('next_step_start', 0.0) + t0.elapsed_time(log[i + 1][0][1])
                n += 1
        res = {k: v / (n if k == 'next_step_start' else len(log)) for k, v in out.items()}
        self._phase_log, self._phase_cur = [], None
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
            st = FasterRCNN._AUX_STREAMS[key] = torch.cuda.Stream(device=self.device, priority=-1)
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
