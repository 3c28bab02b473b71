"""ROIPoolingLayer (reference: luminoth/models/fasterrcnn/roi_pool.py:9-107):
'crop' mode = crop_and_resize to 2x the pooled size + 2x2 max pool, fused in one
kernel; 'roi_pooling' raises NotImplementedError exactly like the reference."""
from luminoth_amd import autograd as A
from luminoth_amd import kernels as K

CROP = 'crop'
ROI_POOLING = 'roi_pooling'


class ROIPoolingLayer(object):
    def __init__(self, config, debug=False, name='roi_pooling'):
        self._pooling_mode = config.pooling_mode.lower()
        self._pooled_width = config.pooled_width
        self._pooled_height = config.pooled_height
        self._pooled_padding = config.padding
        self._debug = debug

    def __call__(self, roi_proposals, roi_count, conv_feature_map, im_shape):
        if self._pooling_mode == CROP:
            pooled = A.RoiPoolFn.apply(conv_feature_map, roi_proposals, roi_count,
                                       (float(im_shape[0]), float(im_shape[1])),
                                       self._pooled_height, self._pooled_width)
            return {'roi_pool': pooled}
        elif self._pooling_mode == ROI_POOLING:
            raise NotImplementedError()
        raise NotImplementedError('Pooling mode {} does not exist.'.format(self._pooling_mode))

    def pooled_mean(self, roi_proposals, roi_count, conv_feature_map, im_shape):
        """tf.reduce_mean(self(...)['roi_pool'], [1, 2]) in one kernel, or None when the feature map does not fit it
        (the caller then pools and averages separately)."""
        if self._pooling_mode != CROP or not K.roi_pool_mean_supported(conv_feature_map.shape):
            return None
        return A.RoiPoolMeanFn.apply(conv_feature_map, roi_proposals, roi_count,
                                     (float(im_shape[0]), float(im_shape[1])),
                                     self._pooled_height, self._pooled_width)
