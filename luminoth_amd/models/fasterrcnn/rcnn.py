"""RCNN — region classifier / box refiner (reference:
luminoth/models/fasterrcnn/rcnn.py:14-411): targets + training-batch
compaction, fused ROI crop-pooling, the base network tail (ResNet-101 block4),
spatial mean, optional FC stack, classifier / bbox FC heads, softmax, the final
per-class NMS proposals and the RCNN loss.  Sonnet Linear names:
`<scope>/rcnn/{fc_i,fc_classifier,fc_bbox}/{w,b}`."""
import torch

from luminoth_amd import autograd as A
from luminoth_amd import kernels as K
from luminoth_amd.models.base.layers import ConvLayer
from luminoth_amd.models.fasterrcnn.rcnn_proposal import RCNNProposal
from luminoth_amd.models.fasterrcnn.rcnn_target import RCNNTarget
from luminoth_amd.models.fasterrcnn.roi_pool import ROIPoolingLayer
from luminoth_amd.utils.vars import get_activation_function, get_initializer


class RCNN(object):
    def __init__(self, num_classes, config, feat_channels, debug=False, seed=None, name='rcnn',
                 scope='fasterrcnn'):
        self._num_classes = num_classes
        self._layer_sizes = list(config.layer_sizes or [])
        self._activation = get_activation_function(config.activation_function)
        self._dropout_keep_prob = config.dropout_keep_prob
        self._use_mean = config.use_mean
        self._variances = config.target_normalization_variances
        self._l1_sigma = config.l1_sigma
        self._debug, self._config, self._seed = debug, config, seed
        self._dropout_calls = 0
        wd = float(config.l2_regularization_scale or 0.0)
        p = '%s/%s' % (scope, name)
        roi = config.roi
        in_feat = feat_channels if self._use_mean else feat_channels * roi.pooled_width * roi.pooled_height
        self._layers = []
        for i, size in enumerate(self._layer_sizes):
            self._layers.append(ConvLayer('%s/fc_%d' % (p, i), in_feat, size, 1, padding='VALID',
                                          act=self._activation, norm='bias', wd=wd,
                                          init=get_initializer(config.rcnn_initializer, seed),
                                          weight_name='w', bias_name='b'))
            in_feat = size
        self._classifier_layer = ConvLayer(p + '/fc_classifier', in_feat, num_classes + 1, 1, padding='VALID',
                                           act=None, norm='bias', wd=wd,
                                           init=get_initializer(config.cls_initializer, seed),
                                           weight_name='w', bias_name='b')
        self._bbox_layer = ConvLayer(p + '/fc_bbox', in_feat, num_classes * 4, 1, padding='VALID', act=None,
                                     norm='bias', wd=wd, init=get_initializer(config.bbox_initializer, seed),
                                     weight_name='w', bias_name='b')
        self.layers = self._layers + [self._classifier_layer, self._bbox_layer]
        self._roi_pool = ROIPoolingLayer(config.roi, debug=debug)
        self._rcnn_target = RCNNTarget(num_classes, config.target, variances=self._variances, seed=seed)
        self._rcnn_proposal = RCNNProposal(num_classes, config.proposals, variances=self._variances)

    def register(self, store):
        zeros = lambda shape, gen: torch.zeros(shape)
        for l in self.layers:
            # Sonnet Linear weights are (in, out): stored as the HWIO view (1,1,in,out)
            store.add(l.w_name, (l.cin, l.cout), l.init, trainable=True, wd=l.wd)
            store.add(l.b_name, (l.cout,), zeros, trainable=True)

    def bind(self, store):
        for l in self.layers:
            l.bind(store, None)
            l.w = store[l.w_name].view(1, 1, l.cin, l.cout)
            l.gw = store.grads[l.w_name].view(1, 1, l.cin, l.cout)
        self._anchor = torch.zeros(1, device=store.flat.device, requires_grad=True)

    def _dropout(self, net, is_training):
        """tf.nn.dropout(net, keep_prob=dropout_keep_prob) while training (rcnn.py:196,218); identity at the reference
        default 1.0 and at inference."""
        kp = self._dropout_keep_prob
        if not is_training or kp in (None, 1, 1.0):
            return net
        from luminoth_amd.utils import rng
        self._dropout_calls += 1
        seed = rng.hash_u32(0 if self._seed is None else int(self._seed), 0xD509, self._dropout_calls & 0xFFFFFFFF)
        return A.DropoutFn.apply(net, float(kp), seed)

    def _linear(self, layer, x2d):
        y = A.conv(layer, x2d.reshape(1, 1, x2d.shape[0], x2d.shape[1]), self._anchor)
        return y.reshape(x2d.shape[0], layer.cout)

    def targets(self, proposals, prop_count, gt_boxes, gt_count, seeds):
        """RCNNTarget alone (rcnn.py:139-154): the fused train step enqueues it right behind the proposal chain,
        before the host turns to the RPN branch, so the auxiliary stream does not wait for the interpreter."""
        return self._rcnn_target(proposals, prop_count, gt_boxes, gt_count, seeds)

    def __call__(self, conv_feature_map, proposals, prop_count, im_shape, base_network, gt_boxes=None,
                 gt_count=None, seeds=None, is_training=False, targets=None):
        B = conv_feature_map.shape[0]
        pred = {'_debug': {}}
        if gt_boxes is not None:
            tgt = targets if targets is not None else self._rcnn_target(proposals, prop_count, gt_boxes, gt_count, seeds)
            if is_training:
                # rcnn.py:156-167 keeps only proposals with label >= 0 (<= minibatch_size per image)
                proposals, prop_count = tgt['rois'], tgt['roi_count']
                pred['target'] = {'cls': tgt['roi_labels'], 'bbox_offsets': tgt['roi_targets']}
            else:
                pred['target'] = {'cls': tgt['labels'], 'bbox_offsets': tgt['bbox_targets']}
        R = proposals.shape[1]
        net = None
        if self._use_mean and not base_network.has_tail and not self._debug:
            # no tail between the pooling and the mean: one kernel, the (B*R,ph,pw,C) tensor is never written
            net = self._roi_pool.pooled_mean(proposals, prop_count, conv_feature_map, im_shape)
        if net is None:
            pooled = self._roi_pool(proposals, prop_count, conv_feature_map, im_shape)['roi_pool']   # (B*R,ph,pw,C)
            features = base_network._build_tail(pooled, is_training=is_training)
            if self._use_mean:
                net = A.SpatialMeanFn.apply(features)                        # (B*R, C)
            else:
                net = features.reshape(features.shape[0], -1)
        net = self._dropout(net, is_training)                          # rcnn.py:196
        for layer in self._layers:
            net = self._dropout(self._linear(layer, net), is_training)   # rcnn.py:214-218
        cls_score = self._linear(self._classifier_layer, net)          # (B*R, C+1)
        bbox_offsets = self._linear(self._bbox_layer, net)             # (B*R, 4C)
        cls_prob = K.softmax(cls_score.detach())
        C = self._num_classes
        pred['rcnn'] = {'cls_score': cls_score.reshape(B, R, C + 1), 'cls_prob': cls_prob.reshape(B, R, C + 1),
                        'bbox_offsets': bbox_offsets.reshape(B, R, 4 * C)}
        pred['proposals'] = proposals
        pred['num_proposals'] = prop_count
        if not is_training or self._debug:
            # rcnn.py:232-239 builds this always, but it is off the train_op path
            det = self._rcnn_proposal(proposals, prop_count, pred['rcnn']['bbox_offsets'].detach(),
                                      pred['rcnn']['cls_prob'], im_shape)
            pred['objects'] = det['objects']
            pred['labels'] = det['proposal_label']
            pred['probs'] = det['proposal_label_prob']
            pred['num_objects'] = det['num_objects']
        return pred

    def loss_and_grads(self, prediction_dict, w_cls=1.0, w_reg=1.0):
        """loss() plus d(cls + reg)/d(cls_score, bbox_offsets) from the same kernel launch, outside autograd (see
        RPN.loss_and_grads)."""
        cs, bo = prediction_dict['rcnn']['cls_score'], prediction_dict['rcnn']['bbox_offsets']
        losses, _, d_cls, d_off = K.rcnn_loss(cs.detach().contiguous(), bo.detach().contiguous(),
                                              prediction_dict['target']['cls'], prediction_dict['target']['bbox_offsets'],
                                              self._num_classes, float(self._l1_sigma), float(w_cls), float(w_reg),
                                              want_grad=True)
        return {'rcnn_cls_loss': losses[0], 'rcnn_reg_loss': losses[1]}, (d_cls, d_off)

    def loss(self, prediction_dict, w_cls=1.0, w_reg=1.0):
        """rcnn.py:255-411; batch mean over images; weights per fasterrcnn.py:194-201."""
        losses = A.RcnnLossFn.apply(prediction_dict['rcnn']['cls_score'], prediction_dict['rcnn']['bbox_offsets'],
                                    prediction_dict['target']['cls'], prediction_dict['target']['bbox_offsets'],
                                    self._num_classes, float(self._l1_sigma), float(w_cls), float(w_reg))
        return {'rcnn_cls_loss': losses[0], 'rcnn_reg_loss': losses[1]}
