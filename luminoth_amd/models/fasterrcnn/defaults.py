"""Default configuration of the Faster R-CNN model: the schema and values of
the reference's luminoth/models/fasterrcnn/base_config.yml (line numbers cited
per group), held as a Python dict because the per-model YAML is looked up next
to the model class in the reference (utils/config.py:60-63)."""

NORMAL = 'random_normal_initializer'


def _normal(stddev):
    return {'_replace': True, 'type': NORMAL, 'mean': 0.0, 'stddev': stddev}


DEFAULTS = {
    'train': {                                           # base_config.yml:1-70
        'debug': False, 'seed': None, 'batch_size': 1, 'job_dir': 'jobs/', 'ignore_scope': None,
        'tf_debug': False, 'run_name': None, 'no_log': False, 'display_every_steps': None,
        'display_every_secs': 300, 'random_shuffle': True, 'save_timeline': False,
        'save_checkpoint_secs': 600, 'checkpoints_max_keep': 1, 'save_summaries_steps': None,
        'save_summaries_secs': 30, 'full_trace': False, 'clip_by_norm': False,
        'learning_rate': {'_replace': True, 'decay_method': None, 'learning_rate': 0.0003},
        'optimizer': {'_replace': True, 'type': 'momentum', 'momentum': 0.9},
        'num_epochs': 1000, 'image_vis': 'train', 'var_vis': None,
    },
    'eval': {'image_vis': 'eval'},
    'dataset': {                                         # base_config.yml:78-98
        'type': 'object_detection', 'dir': 'datasets/voc/tf', 'split': 'train',
        'image_preprocessing': {'min_size': 600, 'max_size': 1024},
        'data_augmentation': [{'flip': {'left_right': True, 'up_down': False, 'prob': 0.5}}],
    },
    'model': {
        'type': 'fasterrcnn',
        'network': {'num_classes': 20, 'with_rcnn': True},   # :122-126
        'batch_norm': False,
        'base_network': {                                # :131-156
            'architecture': 'resnet_v1_101', 'trainable': True, 'weights': None, 'download': True,
            'endpoint': None, 'fine_tune_from': 'block2', 'train_batch_norm': False, 'use_tail': True,
            'freeze_tail': False, 'output_stride': 16, 'arg_scope': {'weight_decay': 0.0005},
        },
        'loss': {'rpn_cls_loss_weight': 1.0, 'rpn_reg_loss_weights': 1.0,   # :158-163
                 'rcnn_cls_loss_weight': 1.0, 'rcnn_reg_loss_weights': 1.0},
        'anchors': {'base_size': 256, 'scales': [0.25, 0.5, 1, 2], 'ratios': [0.5, 1, 2], 'stride': 16},
        'rpn': {                                         # :175-232
            'activation_function': 'relu6', 'l2_regularization_scale': 0.0005, 'l1_sigma': 3.0,
            'num_channels': 512, 'kernel_shape': [3, 3],
            'rpn_initializer': _normal(0.01), 'cls_initializer': _normal(0.01),
            'bbox_initializer': _normal(0.001),
            'proposals': {'pre_nms_top_n': 12000, 'post_nms_top_n': 2000, 'apply_nms': True,
                          'nms_threshold': 0.7, 'min_size': 0, 'clip_after_nms': False,
                          'filter_outside_anchors': False, 'min_prob_threshold': 0.0},
            'target': {'allowed_border': 0, 'clobber_positives': False, 'foreground_threshold': 0.7,
                       'background_threshold_high': 0.3, 'background_threshold_low': 0.0,
                       'foreground_fraction': 0.5, 'minibatch_size': 256, 'random_seed': None},
        },
        'rcnn': {                                        # :234-287
            'layer_sizes': [], 'dropout_keep_prob': 1.0, 'activation_function': 'relu6',
            'l2_regularization_scale': 0.0005, 'l1_sigma': 1.0, 'use_mean': True,
            'target_normalization_variances': [0.1, 0.2],
            'rcnn_initializer': {'_replace': True, 'type': 'variance_scaling_initializer', 'factor': 1.0,
                                 'uniform': True, 'mode': 'FAN_AVG'},
            'cls_initializer': _normal(0.01), 'bbox_initializer': _normal(0.001),
            'roi': {'pooling_mode': 'crop', 'pooled_width': 7, 'pooled_height': 7, 'padding': 'VALID'},
            'proposals': {'class_max_detections': 100, 'class_nms_threshold': 0.5,
                          'total_max_detections': 300, 'min_prob_threshold': 0.5},
            'target': {'foreground_fraction': 0.25, 'minibatch_size': 256, 'foreground_threshold': 0.5,
                       'background_threshold_high': 0.5, 'background_threshold_low': 0.0},
        },
    },
}
