"""Task-level API (reference: luminoth/tasks.py:12-159).

`Detector(config=cfg).predict(images, prob=None, classes=None)` with the reference's semantics: the model's own
probability filter is switched off and the threshold / class filter is applied here.  The checkpoint registry
behind `Detector(checkpoint='accurate')` (tools/checkpoint/, needs network) is out of scope, so `config` is
required."""
from luminoth_amd.utils.predicting import PredictorNetwork


class Detector(object):
    DEFAULT_CHECKPOINT = 'accurate'

    def __init__(self, checkpoint=None, config=None, prob=0.7, classes=None):
        if checkpoint is not None and config is not None:
            raise ValueError('Only one of `checkpoint` or `config` must be specified in order to instantiate '
                             'a Detector.')
        if config is None:
            raise NotImplementedError('the remote checkpoint registry (luminoth/tools/checkpoint) is not hosted: '
                                      'pass `config` (its train.job_dir selects the checkpoint to restore)')
        if config.model.type == 'fasterrcnn':                       # tasks.py:62-65
            config.model.rcnn.proposals.min_prob_threshold = 0.0
        elif config.model.type == 'ssd':
            config.model.proposals.min_prob_threshold = 0.0
        self._network = PredictorNetwork(config)
        self.prob = prob
        self._model_classes = (self._network.class_labels if self._network.class_labels
                               else list(range(config.model.network.num_classes)))
        if classes:
            self.classes = set(classes)
            if not set(self._model_classes).issuperset(self.classes):
                raise ValueError('`classes` must be contained in the detector\'s classes. '
                                 'Available classes are: {}.'.format(self._model_classes))
        else:
            self.classes = set(self._model_classes)

    def predict(self, images, prob=None, classes=None):
        single_image = False
        if not isinstance(images, list):
            if len(images.shape) == 3:
                images = [images]
                single_image = True
        if prob is None:
            prob = self.prob
        classes = self.classes if classes is None else set(classes)
        predictions = []
        for image in images:
            predictions.append([pred for pred in self._network.predict_image(image)
                                if pred['prob'] >= prob and pred['label'] in classes])
        if single_image:
            predictions = predictions[0]
        return predictions
