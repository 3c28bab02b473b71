"""CPU tests of the `lumi predict` row: the oracle's resize restatement against the reference's
own cases (luminoth/utils/image_test.py:118-244), the host logic of the product path against the oracle, and the
driver / Detector plumbing with a stand-in network (no GPU)."""
import json

import numpy as np
import pytest

from oracle import image as oi

F = np.float32

# (shape, min_size, max_size) -> (expected shape, check on scale): image_test.py:118-221
RESIZE_CASES = [
    ((100, 1024), None, None, (100, 1024), lambda s: abs(s - 1.0) < 1e-6),
    ((100, 1024), 0, 2000, (100, 1024), lambda s: abs(s - 1.0) < 1e-6),
    ((100, 1024), None, 1000, (97, 1000), lambda s: int(s * 100) == 97),
    ((100, 1024), 120, None, (120, 1228), lambda s: int(s * 100) == 120),
    ((100, 1024), None, 512, (50, 512), lambda s: abs(s - 0.5) < 1e-6),
    ((100, 1024), 200, None, (200, 2048), lambda s: abs(s - 2.0) < 1e-6),
    ((100, 200), int(100 * 1.1), round(200 / 1.1), (100, 200), lambda s: int(s) == 1),
    ((100, 200), int(100 * 1.6), round(200 / 1.6), (100, 200), lambda s: int(s) == 1),
    ((100, 200), 600, 1000, (600, 1200), lambda s: abs(s - 6.0) < 1e-6),
    ((2000, 600), 600, 1000, (1000, 300), lambda s: abs(s - 0.5) < 1e-6),
]


@pytest.mark.parametrize('case', RESIZE_CASES)
def test_oracle_resize_matches_reference_cases(case):
    (h, w), mn, mx, shape, ok = case
    img = np.random.RandomState(0).randint(0, 255, size=(h, w, 3)).astype(F)
    out = oi.resize_image(img, min_size=mn, max_size=mx)
    assert out['image'].shape == shape + (3,)
    assert ok(out['scale_factor']), out['scale_factor']


@pytest.mark.parametrize('case', RESIZE_CASES)
def test_host_resize_plan_matches_oracle(case):
    from luminoth_amd.utils.image import resize_plan
    (h, w), mn, mx, shape, ok = case
    scale, nh, nw = resize_plan(h, w, mn, mx)
    assert (int(nh), int(nw)) == shape and ok(float(scale))


def test_adjust_bboxes_reference_cases():
    """image_test.py:223-244."""
    from luminoth_amd.utils.image import adjust_bboxes, resize_plan
    img = np.zeros((100, 100, 3), F)
    for boxes, mx, want in [([[0, 0, 10, 10, -1]], 50, [[0, 0, 5, 5, -1]]),
                            ([[10, 10, 90, 90, -1]], 50, [[5, 5, 45, 45, -1]]),
                            ([[0, 0, 99, 99, -1]], 25, [[0, 0, 24, 24, -1]])]:
        out = oi.resize_image(img, bboxes=np.array(boxes), max_size=mx)
        np.testing.assert_array_equal(out['bboxes'], want)
        _, nh, nw = resize_plan(100, 100, None, mx)
        np.testing.assert_array_equal(adjust_bboxes(np.array(boxes), 100, 100, nh, nw), want)


def test_oracle_bilinear_properties():
    rs = np.random.RandomState(1)
    img = rs.randint(0, 256, size=(13, 17, 3)).astype(np.uint8)
    np.testing.assert_array_equal(oi.resize_bilinear(img, 13, 17), img.astype(F))          # identity
    up = oi.resize_bilinear(img, 26, 34)
    np.testing.assert_array_equal(up[::2, ::2], img.astype(F))                              # lerp 0 at even samples
    np.testing.assert_allclose(up[1, 0], (img[0, 0].astype(F) + img[1, 0]) / 2)             # midpoint rows
    np.testing.assert_array_equal(up[25], up[24])                                           # bottom edge clamps
    const = np.full((5, 9, 3), 7, np.uint8)
    np.testing.assert_array_equal(oi.resize_bilinear(const, 11, 4), np.full((11, 4, 3), 7, F))
    fx = oi.resize_image_fixed(img, 300, 300)
    assert fx['image'].shape == (300, 300, 3)
    assert fx['scale_factor'] == (float(F(300) / F(13)), float(F(300) / F(17)))


def test_format_predictions_host_vs_oracle():
    from luminoth_amd.utils.predicting import format_predictions
    rs = np.random.RandomState(2)
    objs = (rs.rand(12, 4) * 600).astype(F)
    labels = rs.randint(0, 5, size=12).tolist()
    probs = rs.rand(12).astype(F).tolist()
    probs[3] = probs[7]                                    # a tie: stable order
    names = ['a', 'b', 'c', 'd', 'e']
    for sf in (0.5859375, 1.0, (1.5, 0.75)):
        for cl in (None, names):
            got = format_predictions(objs.copy(), list(labels), list(probs), sf, cl)
            want = oi.format_predictions(objs.copy(), list(labels), list(probs), sf, cl)
            assert got == want
            assert all(isinstance(c, int) for o in got for c in o['bbox'])
            assert [o['prob'] for o in got] == sorted([o['prob'] for o in got], reverse=True)
    half = format_predictions(np.array([[0.5, 1.5, 2.5, 3.49]], F), [0], [0.123456], 1.0)
    assert half == [{'bbox': [0, 2, 2, 3], 'label': 0, 'prob': 0.1235}]      # Python 3 round-half-even


class _FakeNet(object):
    class_labels = ['cat', 'dog']

    def __init__(self, config):
        self.config = config

    def predict_image(self, image):
        assert image.shape[2] == 3
        return [{'bbox': [0, 0, 5, 5], 'label': 'cat', 'prob': 0.9}, {'bbox': [1, 1, 4, 4], 'label': 'dog', 'prob': 0.6},
                {'bbox': [2, 2, 3, 3], 'label': 'cat', 'prob': 0.2}]


def test_predict_driver_json_lines(tmp_path):
    from luminoth_amd import predict as P
    d = tmp_path / 'imgs'
    d.mkdir()
    np.save(str(d / 'b.npy'), np.zeros((8, 9, 3), np.uint8))
    np.save(str(d / 'a.npy'), np.zeros((8, 9, 3), np.uint8))
    (d / 'notes.txt').write_text('x')
    (d / 'clip.mp4').write_bytes(b'')
    assert P.resolve_files(str(d)) == [str(d / 'a.npy'), str(d / 'b.npy'), str(d / 'clip.mp4')]
    assert P.resolve_files((str(d / 'missing.png'),), echo=lambda m: None) == []
    out = tmp_path / 'out.json'
    msgs = []
    seen = {}

    def net(config):
        seen['cfg'] = config
        return _FakeNet(config)
    res = P.predict((str(d),), [{'model': {'type': 'fasterrcnn'}}], [], str(out), min_prob=0.3, max_detections=7,
                    only_class=['cat'], echo=msgs.append, network_fn=net)
    lines = [json.loads(l) for l in out.read_text().splitlines()]
    assert [l['file'] for l in lines] == [str(d / 'a.npy'), str(d / 'b.npy')]
    assert all([o['label'] for o in l['objects']] == ['cat', 'cat'] for l in lines)
    assert len(res) == 2 and any('video input is not hosted' in m for m in msgs)
    cfg = seen['cfg']
    assert cfg.model.rcnn.proposals.total_max_detections == 7 and cfg.model.rcnn.proposals.min_prob_threshold == 0.3
    assert P.predict((str(d),), [{'model': {'type': 'fasterrcnn'}}], only_class=['a'], ignore_class=['b'],
                     echo=msgs.append, network_fn=net) is None
    ssd = P.apply_detection_limits(__import__('luminoth_amd.utils.config', fromlist=['get_config'])
                                   .get_config({'model': {'type': 'ssd'}}), 0.25, 9)
    assert ssd.model.proposals.total_max_detections == 9 and ssd.model.proposals.min_prob_threshold == 0.25
    assert P.filter_classes(_FakeNet(None).predict_image(np.zeros((1, 1, 3))), ignore_classes=['cat']) == \
        [{'bbox': [1, 1, 4, 4], 'label': 'dog', 'prob': 0.6}]


def test_detector_filters(monkeypatch):
    from luminoth_amd import tasks
    from luminoth_amd.utils.config import get_config
    monkeypatch.setattr(tasks, 'PredictorNetwork', _FakeNet)
    cfg = get_config({'model': {'type': 'fasterrcnn'}})
    det = tasks.Detector(config=cfg)
    assert cfg.model.rcnn.proposals.min_prob_threshold == 0.0          # tasks.py:62-65
    img = np.zeros((4, 4, 3), np.uint8)
    assert [o['prob'] for o in det.predict(img)] == [0.9]               # default prob 0.7, single image -> flat list
    many = det.predict([img, img], prob=0.5)
    assert len(many) == 2 and [o['label'] for o in many[0]] == ['cat', 'dog']
    assert [o['label'] for o in det.predict(img, prob=0.0, classes=['dog'])] == ['dog']
    with pytest.raises(ValueError):
        tasks.Detector(checkpoint='fast', config=cfg)
    with pytest.raises(ValueError):
        tasks.Detector(config=cfg, classes=['zebra'])
    with pytest.raises(NotImplementedError):
        tasks.Detector()
