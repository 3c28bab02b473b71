"""`PredictorNetwork` — the inference caller of the hot path (reference: luminoth/utils/predicting.py:10-148).

Same protocol: `PredictorNetwork(config).predict_image(image)` -> list of `{'bbox': [x1,y1,x2,y2] ints in the
ORIGINAL image scale, 'label': int or class name, 'prob': round(p, 4)}` sorted by prob.  The TF session / graph /
placeholder machinery is replaced by direct calls: preprocess (device resize) -> model forward (HIP kernels) ->
one device->host copy of the detections -> the reference's host post-processing.
"""
import json
import logging
import os

import numpy as np

from luminoth_amd.datasets import get_dataset
from luminoth_amd.models import get_model

log = logging.getLogger('luminoth_amd')


class PredictorNetwork(object):
    def __init__(self, config):
        self.class_labels = None
        if config.dataset.get('dir'):                                  # predicting.py:22-28
            classes_file = os.path.join(config.dataset.dir, 'classes.json')
            if os.path.exists(classes_file):
                with open(classes_file) as f:
                    self.class_labels = json.load(f)
        config.dataset.data_augmentation = None                        # predicting.py:30-31
        self.config = config
        self.dataset = get_dataset(config.dataset.type)(config)
        self.model = get_model(config.model.type)(config)

        from luminoth_amd.train import checkpoint_dir, restore_latest
        if config.train.get('job_dir'):                                # predicting.py:51-63
            job_dir = checkpoint_dir(config)
            if restore_latest(self.model, job_dir) is None:
                raise ValueError('Could not find checkpoint in {}.'.format(job_dir))
            log.info('Loaded checkpoint.')
        else:                                                          # predicting.py:64-72
            log.warning('Could not load checkpoint. Using initialized model.')
        mtype = config.model.type
        if mtype == 'ssd':
            self._fetch = self._fetch_classification
        elif mtype == 'fasterrcnn':
            with_rcnn = config.model.network.get('with_rcnn', False)
            self._fetch = self._fetch_classification if with_rcnn else self._fetch_rpn
        else:
            raise ValueError("Model type '{}' not supported".format(mtype))

    @staticmethod
    def _fetch_classification(pred):                                   # predicting.py:74-84
        cp = pred['classification_prediction']
        return cp['objects'], cp['labels'], cp['probs']

    @staticmethod
    def _fetch_rpn(pred):                                              # predicting.py:85-93
        rp = pred['rpn_prediction']
        return rp['proposals'], rp['scores'].new_zeros(rp['scores'].shape).int(), rp['scores']

    def predict_image(self, image):
        image_dev, _, meta = self.dataset.preprocess(np.array(image))
        pred = self.model(image_dev, is_training=False)
        objects, labels, probs = self._fetch(pred)
        objects = objects.detach().cpu().numpy().astype(np.float32).reshape(-1, 4)
        labels = labels.detach().cpu().numpy().tolist()
        probs = probs.detach().cpu().numpy().tolist()
        scale_factor = meta['scale_factor']
        predictions = format_predictions(objects, labels, probs, scale_factor, self.class_labels)
        self._last = {'prediction_dict': pred, 'image': image_dev, 'scale_factor': scale_factor}
        return predictions


def format_predictions(objects, labels, probs, scale_factor, class_labels=None):
    """predicting.py:118-148."""
    if class_labels is not None:
        labels = [class_labels[label] for label in labels]
    if isinstance(scale_factor, tuple):
        # (scale_factor_height, scale_factor_width): x by width, y by height
        objects = objects / [scale_factor[1], scale_factor[0], scale_factor[1], scale_factor[0]]
    else:
        objects = objects / scale_factor
    objects = [[int(round(coord)) for coord in obj] for obj in objects.tolist()]
    return sorted([{'bbox': obj, 'label': label, 'prob': round(prob, 4)}
                   for obj, label, prob in zip(objects, labels, probs)],
                  key=lambda x: x['prob'], reverse=True)
