"""Dataset registry (reference: luminoth/datasets/datasets.py:4-15 `get_dataset`).  Only in-memory datasets
live here: reading TFRecord `SequenceExample` files is a later row."""
from luminoth_amd.datasets.object_detection_dataset import ObjectDetectionDataset
from luminoth_amd.datasets.synthetic import SyntheticObjectDetectionDataset

DATASETS = {'synthetic': SyntheticObjectDetectionDataset, 'object_detection': ObjectDetectionDataset}


def get_dataset(dataset_type):
    dataset_type = dataset_type.lower()
    if dataset_type not in DATASETS:
        raise ValueError('"{}" is not a valid dataset_type'.format(dataset_type))
    return DATASETS[dataset_type]
