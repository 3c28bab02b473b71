"""Generates tests/golden/bbox_transform_golden.npz by importing the ONE
TF-free module of the reference's hot path, luminoth/utils/bbox_transform.py
(numpy twin of bbox_transform_tf.py), by file path in the build container.
/root/reference does not exist on the GPU box, so the outputs are committed.

    python tests/golden/make_golden.py
"""
import importlib.util
import os

import numpy as np

REF = '/root/reference/luminoth/utils/bbox_transform.py'
spec = importlib.util.spec_from_file_location('ref_bbox_transform', REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rs = np.random.RandomState(1234)
n = 512
xy = rs.randint(0, 700, size=(n, 2))
wh = rs.randint(1, 300, size=(n, 2))
boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
xy = rs.randint(0, 700, size=(n, 2))
wh = rs.randint(1, 300, size=(n, 2))
gt = np.concatenate([xy, xy + wh], 1).astype(np.float32)
deltas = (rs.randn(n, 4) * 0.3).astype(np.float32)
clip_in = (rs.randn(n, 4) * 400 + 300).astype(np.float32)
clip_shape = np.array([480, 640])

out = dict(
    boxes=boxes, gt=gt, deltas=deltas,
    encode=ref.encode(boxes.astype(np.float64), gt.astype(np.float64)),
    decode=ref.decode(boxes.astype(np.float64), deltas.astype(np.float64)),
    clip_in=clip_in, clip_shape=clip_shape,
    clip_out=ref.clip_boxes(clip_in.copy(), clip_shape).astype(np.float32),
)
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bbox_transform_golden.npz'), **out)
print({k: np.asarray(v).shape for k, v in out.items()})
