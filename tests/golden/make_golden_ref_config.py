"""Generates tests/golden/ref_config_golden.json by RUNNING the reference's own config code
(luminoth/utils/config.py:14-232: load_config_files, merge_into, should_replace, types_compatible, parse_override,
parse_config_value, cleanup_config, get_model_config) in the build container.

The module imports TensorFlow (tf.gfile.GFile, tf.logging), easydict and luminoth.models at the top; none of them is
installed here.  Stand-ins: `tf.gfile.GFile` = open, `tf.logging` = no-ops, `luminoth.models.get_model` unused (the
per-model defaults are loaded straight from luminoth/models/<type>/base_config.yml with the module's own
load_config_files), and a minimal EasyDict with the published behaviour of easydict 1.x (dict subclass, attribute
access, nested dicts and dicts inside lists / tuples converted on assignment).  PyYAML >= 6 needs an explicit loader:
`yaml.load(f)` of config.py:37 is given SafeLoader.

/root/reference does not exist on the GPU box, so the outputs are committed as a fixture;
tests/test_ref_config_golden.py replays every scenario through luminoth_amd/utils/config.py.

    python tests/golden/make_golden_ref_config.py
"""
import importlib.util
import json
import os
import sys
import tempfile
import types

import yaml

REF = '/root/reference/luminoth'
HERE = os.path.dirname(os.path.abspath(__file__))


class EasyDict(dict):
    def __init__(self, d=None, **kwargs):
        if d is None:
            d = {}
        if kwargs:
            d.update(**kwargs)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, name, value):
        if isinstance(value, (list, tuple)):
            value = [self.__class__(x) if isinstance(x, dict) else x for x in value]
        elif isinstance(value, dict) and not isinstance(value, self.__class__):
            value = self.__class__(value)
        super(EasyDict, self).__setattr__(name, value)
        super(EasyDict, self).__setitem__(name, value)

    __setitem__ = __setattr__

    def update(self, e=None, **f):
        d = e or dict()
        d.update(f)
        for k in d:
            setattr(self, k, d[k])

    def pop(self, k, d=None):
        if hasattr(self, k):
            delattr(self, k)
        return super(EasyDict, self).pop(k, d)


def load_reference_config_module():
    tf = types.ModuleType('tensorflow')
    tf.gfile = types.SimpleNamespace(GFile=open)
    tf.logging = types.SimpleNamespace(error=lambda *a, **k: None, warn=lambda *a, **k: None, info=lambda *a, **k: None)
    ed = types.ModuleType('easydict')
    ed.EasyDict = EasyDict
    lm = types.ModuleType('luminoth')
    lmm = types.ModuleType('luminoth.models')
    lmm.get_model = lambda t: None
    saved = {k: sys.modules.get(k) for k in ('tensorflow', 'easydict', 'luminoth', 'luminoth.models')}
    sys.modules.update({'tensorflow': tf, 'easydict': ed, 'luminoth': lm, 'luminoth.models': lmm})
    real_load = yaml.load
    yaml.load = lambda f, Loader=None: real_load(f, Loader=Loader or yaml.SafeLoader)
    try:
        spec = importlib.util.spec_from_file_location('ref_config', os.path.join(REF, 'utils/config.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod, (lambda: setattr(yaml, 'load', real_load))


# every scenario: (model type, custom config as a YAML document, list of -o overrides)
SCENARIOS = [
    ('fasterrcnn', 'model:\n  type: fasterrcnn\n', []),
    ('fasterrcnn', 'model:\n  type: fasterrcnn\n  network:\n    num_classes: 80\n  base_network:\n    architecture: resnet_v1_50\n'
                   'train:\n  job_dir: jobs\n  run_name: r50\n  learning_rate:\n    decay_method: piecewise_constant\n'
                   '    boundaries: [250000, 450000]\n    values: [0.0003, 0.0001, 0.00003]\n', []),
    ('fasterrcnn', 'model:\n  type: fasterrcnn\n', ['model.rcnn.proposals.total_max_detections=50', 'train.seed=7',
                                                    'model.rpn.proposals.nms_threshold=0.5', 'train.debug=True',
                                                    'dataset.dir=/data/voc', 'train.job_dir=none']),
    # _replace: the sub-dict replaces the default one instead of merging into it (config.py:93-110)
    ('fasterrcnn', 'model:\n  type: fasterrcnn\n  anchors:\n    _replace: True\n    base_size: 128\n    scales: [0.5, 1, 2]\n'
                   '    ratios: [1]\n    stride: 16\n', []),
    ('fasterrcnn', 'model:\n  type: fasterrcnn\ntrain:\n  optimizer:\n    _replace: True\n    type: adam\n    learning_rate: 0.001\n', []),
    ('fasterrcnn', 'model:\n  type: fasterrcnn\ndataset:\n  data_augmentation:\n    - flip:\n        left_right: False\n'
                   '        up_down: True\n        prob: 0.25\n', ['train.num_epochs=3']),
    ('ssd', 'model:\n  type: ssd\n', []),
    ('ssd', 'model:\n  type: ssd\n  network:\n    num_classes: 5\ntrain:\n  batch_size: 32\n  learning_rate:\n    learning_rate: 0.001\n',
     ['model.proposals.class_nms_threshold=0.3', 'dataset.image_preprocessing.fixed_height=512']),
]
# type errors the merge must raise (config.py:73-90, 120-124)
BAD = [
    ('fasterrcnn', 'model:\n  type: fasterrcnn\n  network:\n    num_classes: twenty\n', []),
    ('fasterrcnn', 'model:\n  type: fasterrcnn\n', ['model.anchors.scales=3']),
    ('fasterrcnn', 'model:\n  type: fasterrcnn\ntrain:\n  learning_rate: 0.1\n', []),
]
OVERRIDE_STRINGS = ['a.b.c=1', 'a.b.d=2.5', 'x=None', 'y=true', 'z=False', 's=hello', 'n=-3', 'e=1e-3', 'p=/tmp/x.y']
VALUES = ['None', 'none', 'True', 'false', '12', '-7', '3.0', '1e5', 'abc', '1,2', '', ' 5']


def main():
    mod, restore = load_reference_config_module()
    out = {'scenarios': [], 'bad': [], 'parse_value': {}, 'parse_override': None}
    try:
        with tempfile.TemporaryDirectory() as tmp:
            def run(model_type, doc, overrides):
                path = os.path.join(tmp, 'custom.yml')
                with open(path, 'w') as f:
                    f.write(doc)
                custom = mod.load_config_files([path])
                base = mod.load_config_files([os.path.join(REF, 'models', model_type, 'base_config.yml')])
                return mod.to_dict(mod.get_model_config(base, custom, overrides))

            for model_type, doc, overrides in SCENARIOS:
                out['scenarios'].append({'model': model_type, 'yaml': doc, 'overrides': overrides,
                                         'config': run(model_type, doc, overrides)})
            for model_type, doc, overrides in BAD:
                try:
                    run(model_type, doc, overrides)
                    raised = None
                except ValueError as e:
                    raised = str(e)
                out['bad'].append({'model': model_type, 'yaml': doc, 'overrides': overrides, 'error': raised})
        for v in VALUES:
            out['parse_value'][v] = mod.parse_config_value(v)
        out['parse_override'] = {'options': OVERRIDE_STRINGS, 'result': mod.parse_override(OVERRIDE_STRINGS)}
        try:
            mod.parse_override(['a=b=c'])
            out['parse_override_bad'] = None
        except ValueError as e:
            out['parse_override_bad'] = str(e)
    finally:
        restore()
    with open(os.path.join(HERE, 'ref_config_golden.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('scenarios %d, bad %s' % (len(out['scenarios']), [bool(b['error']) for b in out['bad']]))


if __name__ == '__main__':
    main()
