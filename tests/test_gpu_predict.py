"""GPU parity of the `lumi predict` row: device resize bit-exact against the oracle, and
PredictorNetwork / Detector / CLI detections equal to the oracle's post-processing of the kernels' own head outputs
(the discrete NMS chain is pinned on identical inputs, as in tests/test_gpu_model.py).  Run with `-m gpu`."""
import json
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import frcnn as of
from oracle import image as oi
from oracle import ssd as oss

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.mark.parametrize('case', [((37, 53), np.uint8, (80, 71)), ((100, 1024), F, (97, 1000)),
                                  ((480, 640), np.uint8, (600, 800)), ((333, 500), np.uint8, (300, 300)),
                                  ((64, 64), F, (64, 64)), ((1, 1), np.uint8, (5, 7)), ((9, 5), F, (1, 1))])
def test_resize_kernel_bit_exact(case):
    from luminoth_amd import kernels as K
    (h, w), dt, (oh, ow) = case
    rs = np.random.RandomState(h * 31 + w)
    img = rs.randint(0, 256, size=(h, w, 3)).astype(dt) if dt == np.uint8 else (rs.rand(h, w, 3) * 255).astype(F)
    got = K.resize_bilinear(torch.from_numpy(img).cuda(), oh, ow).cpu().numpy()
    np.testing.assert_array_equal(got, oi.resize_bilinear(img, oh, ow))


def test_resize_image_host_vs_oracle():
    from luminoth_amd.utils.image import resize_image, resize_image_fixed
    rs = np.random.RandomState(5)
    img = rs.randint(0, 256, size=(120, 200, 3)).astype(np.uint8)
    boxes = np.array([[3, 4, 100, 110, 2], [0, 0, 199, 119, 7]])
    for mn, mx in [(None, None), (300, 400), (64, 128), (600, 1024)]:
        a, b = resize_image(img, bboxes=boxes, min_size=mn, max_size=mx), oi.resize_image(img, boxes, mn, mx)
        np.testing.assert_array_equal(a['image'].cpu().numpy(), b['image'])
        np.testing.assert_array_equal(a['bboxes'], b['bboxes'])
        assert a['scale_factor'] == b['scale_factor']
    a, b = resize_image_fixed(img, 300, 300, bboxes=boxes), oi.resize_image_fixed(img, 300, 300, boxes)
    np.testing.assert_array_equal(a['image'].cpu().numpy(), b['image'])
    np.testing.assert_array_equal(a['bboxes'], b['bboxes'])
    assert a['scale_factor'] == b['scale_factor']


def frcnn_config(**over):
    from luminoth_amd.utils.config import get_config
    cfg = {'model': {'type': 'fasterrcnn', 'network': {'num_classes': 20},
                     'base_network': {'architecture': 'resnet_v1_50'}},
           'dataset': {'type': 'object_detection', 'dir': None,
                       'image_preprocessing': {'min_size': 256, 'max_size': 400}},
           'train': {'seed': 0, 'job_dir': None}}
    return get_config(cfg, ['%s=%s' % kv for kv in over.items()])


def _condition(model):
    sd = model.state_dict()
    sd['truncated_base_network/resnet_v1_50/conv1/BatchNorm/moving_variance'].fill_(73.6 ** 2 * 2)
    model.load_state_dict(sd)


def _oracle_frcnn_predictions(net, cfg, class_labels=None):
    last = net._last
    pred = last['prediction_dict']
    H, W = last['image'].shape[0], last['image'].shape[1]
    cp, rp = pred['classification_prediction'], pred['rpn_prediction']
    p = cfg.model.rcnn.proposals
    r = of.rcnn_proposal(rp['proposals'].cpu().numpy(), cp['rcnn']['bbox_offsets'].cpu().numpy(),
                         cp['rcnn']['cls_prob'].cpu().numpy(), (H, W), cfg.model.network.num_classes,
                         class_max_detections=p.class_max_detections, class_nms_threshold=p.class_nms_threshold,
                         total_max_detections=p.total_max_detections, min_prob_threshold=p.min_prob_threshold)
    return oi.format_predictions(r['objects'], r['proposal_label'], r['proposal_label_prob'], last['scale_factor'],
                                 class_labels)


def test_predictor_network_fasterrcnn(tmp_path):
    from luminoth_amd.train import save_checkpoint
    from luminoth_amd.utils.predicting import PredictorNetwork
    cfg = frcnn_config(**{'model.rcnn.proposals.min_prob_threshold': 0.0})
    net = PredictorNetwork(cfg)
    _condition(net.model)
    img = np.random.RandomState(11).randint(0, 256, size=(150, 210, 3)).astype(np.uint8)
    preds = net.predict_image(img)
    # preprocessing: 150x210 -> min side 256 => scale 256/150, long side 358 < 400
    want = oi.resize_image(img, min_size=256, max_size=400)
    np.testing.assert_array_equal(net._last['image'].cpu().numpy(), want['image'])
    assert net._last['scale_factor'] == want['scale_factor'] and want['image'].shape[:2] == (256, 358)
    assert len(preds) > 0 and preds == _oracle_frcnn_predictions(net, cfg)
    assert all(0 <= o['bbox'][0] <= o['bbox'][2] <= 211 and 0 <= o['bbox'][1] <= o['bbox'][3] <= 151 for o in preds)
    # default threshold 0.5 filters every near-uniform random-init detection (rcnn_proposal.py:97-102)
    # checkpoint round trip: a second network restored from job_dir gives the same detections
    sd = net.model.state_dict()
    save_checkpoint(net.model, 7, str(tmp_path / 'job'), 1)
    cfg2 = frcnn_config(**{'model.rcnn.proposals.min_prob_threshold': 0.0, 'train.job_dir': str(tmp_path / 'job')})
    (tmp_path / 'ds').mkdir()
    names = ['c%d' % i for i in range(20)]
    (tmp_path / 'ds' / 'classes.json').write_text(json.dumps(names))
    cfg2.dataset.dir = str(tmp_path / 'ds')
    net2 = PredictorNetwork(cfg2)
    for k, v in net2.model.state_dict().items():
        assert torch.equal(v, sd[k]), k
    preds2 = net2.predict_image(img)
    assert [o['bbox'] for o in preds2] == [o['bbox'] for o in preds]
    assert [o['label'] for o in preds2] == [names[o['label']] for o in preds]
    with pytest.raises(ValueError):
        PredictorNetwork(frcnn_config(**{'train.job_dir': str(tmp_path / 'nothing_here')}))


def test_detector_and_rpn_only():
    from luminoth_amd.tasks import Detector
    from luminoth_amd.utils.predicting import PredictorNetwork
    cfg = frcnn_config()
    det = Detector(config=cfg, prob=0.0)
    _condition(det._network.model)
    imgs = [np.random.RandomState(s).randint(0, 256, size=(140, 180, 3)).astype(np.uint8) for s in (1, 2)]
    out = det.predict(imgs)
    assert len(out) == 2 and all(len(o) > 0 for o in out)
    assert out[1] == _oracle_frcnn_predictions(det._network, cfg)
    top = det.predict(imgs[1], prob=out[1][2]['prob'])
    assert top == [o for o in out[1] if o['prob'] >= out[1][2]['prob']]
    # with_rcnn False: proposals + objectness scores, all labels zero (predicting.py:85-93)
    cfg_rpn = frcnn_config(**{'model.network.with_rcnn': False})
    net = PredictorNetwork(cfg_rpn)
    _condition(net.model)
    preds = net.predict_image(imgs[0])
    rp = net._last['prediction_dict']['rpn_prediction']
    n = rp['proposals'].shape[0]
    assert len(preds) == n and n > 0 and all(o['label'] == 0 for o in preds)
    want = oi.format_predictions(rp['proposals'].cpu().numpy(), [0] * n, rp['scores'].cpu().numpy().tolist(),
                                 net._last['scale_factor'])
    assert preds == want


def test_predictor_network_ssd():
    from luminoth_amd.utils.config import get_config
    from luminoth_amd.utils.predicting import PredictorNetwork
    cfg = get_config({'model': {'type': 'ssd', 'network': {'num_classes': 20}, 'proposals': {'min_prob_threshold': 0.0}},
                      'dataset': {'type': 'object_detection', 'dir': None},
                      'train': {'seed': 0, 'debug': True, 'job_dir': None}})
    net = PredictorNetwork(cfg)
    img = np.random.RandomState(3).randint(0, 256, size=(333, 500, 3)).astype(np.uint8)
    preds = net.predict_image(img)
    want = oi.resize_image_fixed(img, 300, 300)
    np.testing.assert_array_equal(net._last['image'].cpu().numpy(), want['image'])
    assert net._last['scale_factor'] == want['scale_factor']
    pd = net._last['prediction_dict']
    p = cfg.model.proposals
    o = oss.ssd_proposal(pd['cls_prob'].cpu().numpy(), pd['loc_pred'].cpu().numpy(), pd['all_anchors'].cpu().numpy(),
                         (300, 300), 20, class_nms_threshold=p.class_nms_threshold,
                         class_max_detections=p.class_max_detections, total_max_detections=p.total_max_detections,
                         min_prob_threshold=p.min_prob_threshold, variances=tuple(cfg.model.variances))
    got_objs = pd['classification_prediction']['objects'].cpu().numpy()
    np.testing.assert_allclose(got_objs, o['objects'], rtol=1e-6, atol=1e-4)
    assert len(preds) == len(o['probs']) > 0
    assert preds == oi.format_predictions(got_objs, o['labels'], o['probs'], want['scale_factor'])


def test_predict_cli(tmp_path):
    from PIL import Image
    rs = np.random.RandomState(9)
    for name in ('one.png', 'two.jpg'):
        Image.fromarray(rs.randint(0, 256, size=(96, 128, 3)).astype(np.uint8)).save(str(tmp_path / name))
    cfgf = tmp_path / 'cfg.yml'
    cfgf.write_text('model:\n  type: fasterrcnn\n  network:\n    num_classes: 5\n  base_network:\n'
                    '    architecture: resnet_v1_50\ndataset:\n  type: object_detection\n  dir: null\n'
                    '  image_preprocessing:\n    min_size: 128\n    max_size: 256\ntrain:\n  seed: 0\n  job_dir: null\n')
    out = tmp_path / 'preds.json'
    r = subprocess.run([sys.executable, '-m', 'luminoth_amd.predict', str(tmp_path), '-c', str(cfgf), '-f', str(out),
                        '--min-prob', '0.0', '--max-detections', '5'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in out.read_text().splitlines()]
    assert sorted(l['file'] for l in lines) == sorted(str(tmp_path / n) for n in ('one.png', 'two.jpg'))
    for l in lines:
        assert 0 < len(l['objects']) <= 5
        assert all(set(o) == {'bbox', 'label', 'prob'} and len(o['bbox']) == 4 for o in l['objects'])
