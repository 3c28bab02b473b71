"""Dataset registry (reference: luminoth/datasets/datasets.py:4-22 `get_dataset`): `object_detection` reads the
TFRecord `SequenceExample` splits, `tfrecord` is its deprecated alias (still the SSD default: ssd/base_config.yml:69),
`synthetic` is this repo's in-memory generator."""
import logging

from luminoth_amd.datasets.object_detection_dataset import ObjectDetectionDataset
from luminoth_amd.datasets.synthetic import SyntheticObjectDetectionDataset

DATASETS = {'synthetic': SyntheticObjectDetectionDataset, 'object_detection': ObjectDetectionDataset,
            'tfrecord': ObjectDetectionDataset}


def get_dataset(dataset_type):
    dataset_type = dataset_type.lower()
    if dataset_type not in DATASETS:
        raise ValueError('"{}" is not a valid dataset_type'.format(dataset_type))
    if dataset_type == 'tfrecord':
        logging.getLogger('luminoth_amd').warning('Dataset `tfrecord` is deprecated. Use `object_detection` instead.')
    return DATASETS[dataset_type]
