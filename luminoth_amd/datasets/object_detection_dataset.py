"""`ObjectDetectionDataset.preprocess` — the resize step the inference driver calls
(reference: luminoth/datasets/object_detection_dataset.py:71-83,202-234; base_dataset.py:18-28).

Only the preprocessing entry point is hosted here: reading TFRecord `SequenceExample` files and the training
augmentations are the input pipeline and stay out of scope, so iterating this
dataset raises."""
from luminoth_amd.utils.image import resize_image, resize_image_fixed


class ObjectDetectionDataset(object):
    def __init__(self, config, name='object_detection_dataset', **kwargs):
        ip = config.dataset.image_preprocessing
        self._dataset_dir = config.dataset.get('dir')
        self._image_min_size = ip.get('min_size')
        self._image_max_size = ip.get('max_size')
        self._fixed_resize = 'fixed_height' in ip and 'fixed_width' in ip
        if self._fixed_resize:
            self._image_fixed_height = ip.fixed_height
            self._image_fixed_width = ip.fixed_width
        self._data_augmentation = config.dataset.get('data_augmentation') or []

    def preprocess(self, image, bboxes=None):
        """Returns (image (H',W',3) float32 on the device, bboxes, {'scale_factor', 'applied_augmentations'})."""
        if self._data_augmentation and bboxes is not None:
            raise NotImplementedError('training-time data augmentation is CPU-side input pipeline work '
                                      '(utils/image.py:150-620) and is not hosted')
        if self._fixed_resize:
            resized = resize_image_fixed(image, self._image_fixed_height, self._image_fixed_width, bboxes=bboxes)
        else:
            resized = resize_image(image, bboxes=bboxes, min_size=self._image_min_size,
                                   max_size=self._image_max_size)
        return resized['image'], resized.get('bboxes'), {'scale_factor': resized['scale_factor'],
                                                         'applied_augmentations': []}

    def __iter__(self):
        raise NotImplementedError('TFRecord reading is not hosted (SURVEY.md §8f-3); use dataset.type=synthetic '
                                  'for training')
