"""The per-model default configuration dicts are key-for-key and value-for-value the reference's
`luminoth/models/<type>/base_config.yml` (the reference looks its defaults up next to the model class:
utils/config.py:60-63).  The YAML files only exist in the build container (/root/reference is not shipped to the GPU
box), where the driver runs the CPU suite; elsewhere the comparison is skipped."""
import os

import pytest
import yaml

REF = '/root/reference/luminoth/models'


def _diff(ours, ref, path=''):
    out = []
    if isinstance(ours, dict) and isinstance(ref, dict):
        for k in sorted(set(ours) | set(ref)):
            if k not in ours:
                out.append((path + '.' + k, '<missing>', ref[k]))
            elif k not in ref:
                out.append((path + '.' + k, ours[k], '<missing>'))
            else:
                out += _diff(ours[k], ref[k], path + '.' + k)
    elif ours != ref:
        out.append((path, ours, ref))
    return out


@pytest.mark.parametrize('model_type', ['fasterrcnn', 'ssd'])
def test_defaults_equal_reference_base_config(model_type):
    path = os.path.join(REF, model_type, 'base_config.yml')
    if not os.path.exists(path):
        pytest.skip('reference tree not present on this host')
    import importlib
    ours = importlib.import_module('luminoth_amd.models.%s.defaults' % model_type).DEFAULTS
    ref = yaml.safe_load(open(path))
    assert _diff(ours, ref) == []


@pytest.mark.parametrize('model_type', ['fasterrcnn', 'ssd'])
def test_get_config_without_files_is_the_model_default(model_type):
    from luminoth_amd.utils.config import get_config
    cfg = get_config({'model': {'type': model_type}})
    assert cfg.model.type == model_type
    if model_type == 'ssd':
        assert cfg.train.num_epochs == 10000 and cfg.train.random_shuffle is False and cfg.dataset.type == 'tfrecord'
        assert [list(a)[0] for a in cfg.dataset.data_augmentation] == ['flip', 'patch', 'distortion', 'expand']
    else:
        assert cfg.train.num_epochs == 1000 and cfg.model.base_network.architecture == 'resnet_v1_101'
