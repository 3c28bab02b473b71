"""Generates tests/golden/ref_numpy_golden.npz by RUNNING the reference's own numpy code in the build container.

The reference's hot path is TensorFlow, but several stages are plain numpy that merely live in modules which import
TensorFlow / Sonnet at the top.  With inert stand-ins for those imports in `sys.modules` the files load by path and
their numpy functions run unchanged:

  * luminoth/models/ssd/utils.py:5-145  — SSD anchors (S3): generate_anchors_reference, generate_raw_anchors,
    generate_anchors_per_feat_map, adjust_bboxes  + the glue of ssd.py:111-129 (clip_boxes of utils/bbox_transform.py);
  * luminoth/utils/bbox_overlap.py:51-94 — numpy twin of the +1-convention IoU (A7);
  * luminoth/utils/anchors.py:4-52       — generate_anchors_reference (A3; needs `np.int`, removed in numpy >= 1.24);
  * luminoth/utils/test/anchors.py:4-60  — the numpy twin of the anchor grid (A3);
  * luminoth/eval.py:487-653             — calculate_metrics (COCO-style AP / AR of `lumi eval`, row (f)4; its
    `np.linspace(..., num=<float>)` calls need an int-coercing shim on numpy 2).

/root/reference does not exist on the GPU box, so the outputs are committed as a fixture; tests/test_ref_numpy_golden.py
checks oracle/{boxes,ssd,eval_metrics}.py and the product's host twins (luminoth_amd/models/ssd/utils.py,
luminoth_amd/utils/anchors.py, luminoth_amd/eval.py) against it.

    python tests/golden/make_golden_ref_numpy.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = '/root/reference/luminoth'


class _Inert(types.ModuleType):
    """Module stand-in: any attribute is another inert object; calling it returns a pass-through decorator /
    context manager, which is all the module-level code of the reference files needs."""

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Inert(self.__name__ + '.' + name)

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k and not isinstance(a[0], _Inert):
            return a[0]
        return self

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def install_stubs():
    for name in ('tensorflow', 'sonnet', 'easydict', 'click', 'luminoth', 'luminoth.datasets', 'luminoth.models',
                 'luminoth.utils', 'luminoth.utils.config', 'luminoth.utils.image_vis'):
        sys.modules.setdefault(name, _Inert(name))
    if not hasattr(np, 'int'):
        np.int = int            # utils/anchors.py:42-43


def load(relpath, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Shape(object):
    def __init__(self, s):
        self._s = list(s)

    def as_list(self):
        return list(self._s)


class _FeatMap(object):
    def __init__(self, h, w, c):
        self.shape = _Shape((1, h, w, c))


def main():
    install_stubs()
    ssd_utils = load('models/ssd/utils.py', 'ref_ssd_utils')
    bbox_overlap = load('utils/bbox_overlap.py', 'ref_bbox_overlap')
    sys.modules['luminoth.utils.bbox_overlap'] = bbox_overlap        # eval.py imports the numpy twin from here
    anchors = load('utils/anchors.py', 'ref_anchors')
    test_anchors = load('utils/test/anchors.py', 'ref_test_anchors')
    bbox_transform = load('utils/bbox_transform.py', 'ref_bbox_transform')
    ref_eval = load('eval.py', 'ref_eval')
    out = {}
    rs = np.random.RandomState(4321)

    # ---- S3: SSD anchors, default config (ssd/base_config.yml:128-138), 300x300 and a non-square 240x320 input
    from collections import OrderedDict
    ratios = np.array([1, 0.5, 2, 0.333, 3])
    app = [4, 6, 6, 6, 4, 4]
    for tag, (ih, iw), shapes in (('300', (300, 300), [(37, 37), (18, 18), (9, 9), (5, 5), (3, 3), (1, 1)]),
                                  ('240x320', (240, 320), [(29, 39), (14, 19), (7, 10), (4, 5), (2, 3), (1, 1)])):
        fmaps = OrderedDict(('m%d' % i, _FeatMap(h, w, 8)) for i, (h, w) in enumerate(shapes))
        raw = ssd_utils.generate_raw_anchors(fmaps, 0.1, 0.88, ratios, app)
        parts = []
        for i, (name, fm) in enumerate(fmaps.items()):                   # ssd.py:111-129
            h, w = fm.shape.as_list()[1:3]
            scaled = ssd_utils.adjust_bboxes(raw[name], h, w, ih, iw)
            parts.append(bbox_transform.clip_boxes(scaled, (ih, iw)))
            out['ssd_raw_%s_%d' % (tag, i)] = np.asarray(raw[name], dtype=np.float64)
        out['ssd_anchors_%s' % tag] = np.concatenate(parts, axis=0).astype(np.float32)   # tf.convert_to_tensor(float32)
        out['ssd_shapes_%s' % tag] = np.array(shapes)
    scales = np.linspace(0.1, 0.88, 6)
    out['ssd_ref_first'] = ssd_utils.generate_anchors_reference(ratios, scales[0:2], 4, [37, 37])
    out['ssd_ref_last'] = ssd_utils.generate_anchors_reference(ratios, scales[5:7], 4, [1, 1])
    b = rs.rand(64, 4) * 30
    out['adjust_in'] = b
    out['adjust_out'] = ssd_utils.adjust_bboxes(b, 18.0, 19.0, 300.0, 320.0)

    # ---- A7: numpy IoU twin (float32 boxes incl. disjoint, touching, identical and negative-area cases)
    xy = rs.randint(0, 600, size=(96, 2))
    wh = rs.randint(1, 250, size=(96, 2))
    b1 = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    xy = rs.randint(0, 600, size=(40, 2))
    wh = rs.randint(1, 250, size=(40, 2))
    b2 = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    b1[0], b2[0] = [0, 0, 10, 10], [11, 11, 20, 20]           # bbox_overlap_test.py:44-84 cases
    b1[1], b2[1] = b2[5], b2[5]
    b1[2] = [50, 50, 40, 40]
    b1[3], b2[2] = [0, 0, 10, 10], [10, 10, 20, 20]
    out['iou_b1'], out['iou_b2'] = b1, b2
    out['iou'] = bbox_overlap.bbox_overlap(b1, b2)

    # ---- A3: anchor reference + grid
    cfgs = [(256, [0.5, 1, 2], [0.25, 0.5, 1, 2]), (16, [0.5, 1, 2], [8, 16, 32]), (64, [0.333, 1, 3], [0.5, 1])]
    for i, (base, rat, sc) in enumerate(cfgs):
        ref = anchors.generate_anchors_reference(base, np.array(rat), np.array(sc))
        out['anchor_ref_%d' % i] = ref
        out['anchor_cfg_%d' % i] = np.array([base] + rat + sc, dtype=np.float64)
        out['anchor_nr_%d' % i] = np.array([len(rat), len(sc)])
    ref0 = out['anchor_ref_0']
    out['anchor_grid_float'] = test_anchors.generate_anchors(ref0, 16, np.array([7, 9]))
    # the training graph adds the float64 reference to an int32 grid => truncation (fasterrcnn.py:299-302)
    out['anchor_grid_int'] = test_anchors.generate_anchors(ref0.astype(np.int32), 16, np.array([7, 9]))

    # ---- (f)4: calculate_metrics on two random splits (one with already-sorted detections, one unsorted: the
    # reference's flag/score misalignment on unsorted input is part of the behaviour)
    real_linspace = np.linspace
    np.linspace = lambda a, b, num=50, **k: real_linspace(a, b, int(num), **k)
    try:
        for tag, sort_scores in (('sorted', True), ('unsorted', False)):
            C, n_img = 4, 12
            outb = {k: [] for k in ('bboxes', 'classes', 'scores', 'gt_bboxes', 'gt_classes')}
            for i in range(n_img):
                g = rs.randint(0, 5)
                gxy = rs.randint(0, 200, size=(g, 2))
                gwh = rs.randint(20, 120, size=(g, 2))
                gtb = np.concatenate([gxy, gxy + gwh], 1).astype(np.float32)
                gtc = rs.randint(0, C, size=(g,))
                if i % 4 == 3:
                    gtc[:] = 0                                       # some classes have no examples at all
                d = rs.randint(0, 9)
                det = np.zeros((d, 4), np.float32)
                cls = rs.randint(0, C, size=(d,))
                for j in range(d):
                    if g and rs.rand() < 0.7:                        # jittered copy of a ground-truth box
                        src = rs.randint(0, g)
                        det[j] = gtb[src] + rs.randint(-12, 13, size=4)
                        if rs.rand() < 0.8:
                            cls[j] = gtc[src]
                    else:
                        xy = rs.randint(0, 200, size=2)
                        det[j] = np.concatenate([xy, xy + rs.randint(20, 120, size=2)])
                sc = rs.rand(d).astype(np.float32)
                if sort_scores:
                    order = np.argsort(-sc)
                    det, cls, sc = det[order], cls[order], sc[order]
                outb['bboxes'].append(det)
                outb['classes'].append(cls)
                outb['scores'].append(sc)
                outb['gt_bboxes'].append(gtb)
                outb['gt_classes'].append(gtc)
            with np.errstate(divide='ignore', invalid='ignore'):
                ap, ar = ref_eval.calculate_metrics(outb, C)
            out['metrics_%s_ap' % tag], out['metrics_%s_ar' % tag] = ap, ar
            out['metrics_%s_n' % tag] = np.array([n_img, C])
            for k, v in outb.items():
                for i, a in enumerate(v):
                    out['metrics_%s_%s_%d' % (tag, k, i)] = np.asarray(a)
    finally:
        np.linspace = real_linspace

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_numpy_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote %s: %d arrays, %.1f KB' % (path, len(out), os.path.getsize(path) / 1024.0))


if __name__ == '__main__':
    main()
