"""GPU run of the `lumi eval` row: checkpoint + val.tfrecords -> detections and losses through the
HIP path, metrics equal to the oracle's on the same detections.  Run with `-m gpu`."""
import io
import json
import os

import numpy as np
import pytest

from oracle import eval_metrics as om
from tests.test_gpu_dataset import make_split

pytestmark = pytest.mark.gpu


def _cfg_dict(tmp_path, model):
    d = {'model': {'type': model, 'network': {'num_classes': 5}},
         'dataset': {'type': 'object_detection', 'dir': str(tmp_path / 'data')},
         'train': {'seed': 0, 'job_dir': str(tmp_path / 'job'), 'run_name': 'r'}}
    if model == 'fasterrcnn':
        d['model']['base_network'] = {'architecture': 'resnet_v1_50'}
        d['dataset']['image_preprocessing'] = {'min_size': 128, 'max_size': 256}
    return d


@pytest.mark.parametrize('model_type', ['fasterrcnn', 'ssd'])
def test_evaluate_checkpoint_on_val_split(tmp_path, model_type):
    from luminoth_amd import eval as E
    from luminoth_amd.models import get_model
    from luminoth_amd.train import save_checkpoint
    from luminoth_amd.utils.config import get_config
    make_split(str(tmp_path / 'data'), 5, [(120, 160), (160, 120)], split='val')
    with open(str(tmp_path / 'data' / 'classes.json'), 'w') as f:
        json.dump(['a', 'b', 'c', 'd', 'e'], f)
    cfg = get_config(_cfg_dict(tmp_path, model_type))
    model = get_model(model_type)(cfg)
    if model_type == 'fasterrcnn':
        sd = model.state_dict()
        sd['truncated_base_network/resnet_v1_50/conv1/BatchNorm/moving_variance'].fill_(73.6 ** 2 * 2)
        model.load_state_dict(sd)
    save_checkpoint(model, 3, str(tmp_path / 'job' / 'r'), 5)
    save_checkpoint(model, 9, str(tmp_path / 'job' / 'r'), 5)
    cfgf = tmp_path / 'cfg.json.yml'
    cfgf.write_text(json.dumps(_cfg_dict(tmp_path, model_type)))          # JSON is YAML
    buf = io.StringIO()
    res = E.evaluate([str(cfgf)], dataset_split='val', watch=False, max_detections=20, output=buf)
    assert len(res) == 1 and res[0]['global_step'] == 9 and res[0]['total_evaluated'] == 5
    line = json.loads(buf.getvalue().splitlines()[0])
    assert set(['AP@0.50', 'AP@0.75', 'AP@[0.50:0.95]', 'AR@[0.50:0.95]', 'total_evaluated', 'evaluation_time',
                'val_losses/total_loss']) <= set(line)
    assert np.isfinite(line['val_losses/total_loss'])
    # watch mode from a given step evaluates every newer checkpoint once
    res = E.evaluate([str(cfgf)], dataset_split='val', watch=True, from_global_step=1, max_detections=20,
                     poll_secs=0.01, max_evaluations=2)
    assert [r['global_step'] for r in res] == [3, 9]
    # same detections -> the oracle's metrics
    cfg2 = E.prepare_config(get_config(_cfg_dict(tmp_path, model_type)), 'val', 20)
    from luminoth_amd.datasets import get_dataset
    outs = {}
    model2 = get_model(model_type)(cfg2)
    model2.load_state_dict(model.state_dict())
    r = E.evaluate_once(cfg2, model2, get_dataset('object_detection')(cfg2), outputs=outs)
    assert all(len(s) <= 20 for s in outs['scores']) and sum(len(s) for s in outs['scores']) > 0
    assert all(np.all(np.diff(s) <= 0) for s in outs['scores'])            # detections arrive sorted by prob
    with np.errstate(all='ignore'):
        ap, ar = om.calculate_metrics(outs, 5)
    np.testing.assert_allclose(r['ap_per_class'], ap, rtol=1e-12, atol=1e-15, equal_nan=True)
    np.testing.assert_array_equal(np.asarray(r['ar_per_class']), ar)
