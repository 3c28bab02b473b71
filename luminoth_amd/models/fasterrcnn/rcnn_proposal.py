"""RCNNProposal — class-specific decode, clip, filter, per-class NMS and global
top-k (reference: luminoth/models/fasterrcnn/rcnn_proposal.py:7-164), as one
batched kernel sequence over (image, class)."""
from luminoth_amd import kernels as K


class RCNNProposal(object):
    def __init__(self, num_classes, config, variances=None, name='rcnn_proposal'):
        self._num_classes = num_classes
        self._variances = variances
        self._class_max_detections = config.class_max_detections
        self._class_nms_threshold = float(config.class_nms_threshold)
        self._total_max_detections = config.total_max_detections
        self._min_prob_threshold = config.min_prob_threshold or 0.0

    def __call__(self, proposals, prop_count, bbox_pred, cls_prob, im_shape):
        """proposals (B,R,4), bbox_pred (B,R,4C), cls_prob (B,R,C+1)."""
        objects, labels, probs, num = K.rcnn_proposal(
            proposals, prop_count, bbox_pred, cls_prob, im_shape, self._num_classes,
            variances=self._variances, class_max_detections=self._class_max_detections,
            class_nms_threshold=self._class_nms_threshold,
            total_max_detections=self._total_max_detections,
            min_prob_threshold=self._min_prob_threshold)
        return {'objects': objects, 'proposal_label': labels, 'proposal_label_prob': probs,
                'num_objects': num}
