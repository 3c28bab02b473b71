"""RPNTarget — anchor labels / regression targets for a batch
(reference: luminoth/models/fasterrcnn/rpn_target.py:8-335)."""
from luminoth_amd import kernels as K


class RPNTarget(object):
    def __init__(self, num_anchors, config, seed=None, name='anchor_target'):
        self._num_anchors = num_anchors
        self._allowed_border = config.allowed_border
        self._clobber_positives = config.clobber_positives
        self._positive_overlap = config.foreground_threshold
        self._negative_overlap = config.background_threshold_high
        self._foreground_fraction = config.foreground_fraction
        self._minibatch_size = config.minibatch_size
        self._seed = seed

    def __call__(self, anchor_ref_i32, feat_hw, stride, gt_boxes, gt_count, seeds, im_shape, out=None):
        """Returns labels (B,N) in {-1,0,1}, bbox_targets (B,N,4), max_overlaps (B,N)."""
        labels, targets, max_ov, _ = K.rpn_target(
            anchor_ref_i32, feat_hw[0], feat_hw[1], stride, gt_boxes, gt_count, seeds, im_shape,
            allowed_border=self._allowed_border, clobber_positives=self._clobber_positives,
            foreground_threshold=self._positive_overlap, background_threshold_high=self._negative_overlap,
            foreground_fraction=self._foreground_fraction, minibatch_size=self._minibatch_size, out=out)
        return labels, targets, max_ov
