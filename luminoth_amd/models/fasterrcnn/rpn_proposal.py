"""RPNProposal — anchors + RPN predictions -> object proposals.

Same configuration keys and result keys as the reference module
(luminoth/models/fasterrcnn/rpn_proposal.py:7-197); the whole chain (softmax,
decode, filters, clip, top-k, NMS, gather) is one C-ABI call, batched."""
from luminoth_amd import kernels as K


class RPNProposal(object):
    def __init__(self, num_anchors, config, debug=False, name='proposal_layer'):
        self._num_anchors = num_anchors
        self._pre_nms_top_n = config.pre_nms_top_n
        self._apply_nms = config.apply_nms
        self._post_nms_top_n = config.post_nms_top_n
        self._nms_threshold = float(config.nms_threshold)
        self._min_size = config.min_size
        self._filter_outside_anchors = config.filter_outside_anchors
        self._clip_after_nms = config.clip_after_nms
        self._min_prob_threshold = float(config.min_prob_threshold)
        self._debug = debug

    def __call__(self, rpn_cls_score, rpn_bbox_pred, anchor_ref_i32, feat_hw, stride, im_shape):
        """rpn_cls_score (B,N,2) logits, rpn_bbox_pred (B,N,4).  Returns dict with
        `rpn_cls_prob` (B,N,2), `proposals` (B,P,4), `scores` (B,P), `num_proposals` (B)."""
        prob, proposals, scores, count = K.rpn_proposal(
            rpn_cls_score, rpn_bbox_pred, anchor_ref_i32, feat_hw[0], feat_hw[1], stride, im_shape,
            pre_nms_top_n=self._pre_nms_top_n, post_nms_top_n=self._post_nms_top_n,
            nms_threshold=self._nms_threshold, min_prob_threshold=self._min_prob_threshold,
            apply_nms=self._apply_nms, clip_after_nms=self._clip_after_nms,
            filter_outside_anchors=self._filter_outside_anchors)
        return {'rpn_cls_prob': prob, 'proposals': proposals, 'scores': scores, 'num_proposals': count}
