"""Default configuration of the SSD model: schema and values of the reference's
luminoth/models/ssd/base_config.yml (line numbers cited per group)."""

DEFAULTS = {
    'train': {                                           # ssd/base_config.yml:1-66 (key for key: tests/test_defaults.py)
        'debug': True, 'seed': None, 'batch_size': 1, 'job_dir': 'jobs/',
        'tf_debug': False, 'no_log': False, 'display_every_steps': 5000,
        'display_every_secs': None, 'random_shuffle': False, 'save_timeline': False,
        'save_checkpoint_secs': 600, 'checkpoints_max_keep': 1, 'save_summaries_steps': None,
        'save_summaries_secs': 30, 'full_trace': False, 'clip_by_norm': False,
        'learning_rate': {'_replace': True, 'decay_method': None, 'learning_rate': 0.0003},
        'optimizer': {'_replace': True, 'type': 'momentum', 'momentum': 0.5},
        'num_epochs': 10000, 'image_vis': 'debug',
    },
    'eval': {'image_vis': 'eval'},
    'dataset': {                                         # :68-102
        'type': 'tfrecord', 'dir': 'datasets/voc/tf', 'split': 'train',
        'image_preprocessing': {'fixed_height': 300, 'fixed_width': 300},
        'data_augmentation': [
            {'flip': {'left_right': True, 'up_down': False, 'prob': 0.5}},
            {'patch': {'min_height': 30, 'min_width': 30, 'prob': 0.5}},
            {'distortion': {'brightness': {'max_delta': 0.2}, 'hue': {'max_delta': 0.2},
                            'saturation': {'lower': 0.5, 'upper': 1.5}, 'prob': 0.5}},
            {'expand': {'prob': 0.5}},
        ],
    },
    'model': {
        'type': 'ssd',
        'network': {'num_classes': 20},                  # :108
        'base_network': {'architecture': 'truncated_vgg_16', 'trainable': True, 'weights': None,
                         'download': True, 'arg_scope': {'weight_decay': 0.0005}, 'dropout_keep_prob': 1.0},
        'loss': {'localization_loss_weight': 1.0},       # :126
        'anchors': {'anchors_per_point': [4, 6, 6, 6, 4, 4], 'ratios': [1, 0.5, 2, 0.333, 3],
                    'min_scale': 0.1, 'max_scale': 0.88},      # :128-138 (linspace .10-.88)
        'target': {'hard_negative_ratio': 3.0, 'foreground_threshold': 0.5,
                   'background_threshold_high': 0.2, 'background_threshold_low': 0.0},
        'proposals': {'total_max_detections': 100, 'class_max_detections': 100,
                      'class_nms_threshold': 0.45, 'min_prob_threshold': 0.5, 'filter_outside_anchors': True},
        'variances': [0.1, 0.2],                         # :166
    },
}
