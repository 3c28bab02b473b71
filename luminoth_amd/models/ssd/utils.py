"""SSD anchor generation on the host (numpy, once per geometry) — same functions and results as the
reference's luminoth/models/ssd/utils.py:5-145 + the glue in ssd.py:111-129."""
import numpy as np


def adjust_bboxes(bboxes, old_height, old_width, new_height, new_width):
    """utils.py:5-28: rescale boxes from an (old_h, old_w) frame to a (new_h, new_w) frame."""
    sx = np.array([1.0 / old_width, 1.0 / old_height, 1.0 / old_width, 1.0 / old_height])
    nx = np.array([new_width, new_height, new_width, new_height], dtype=np.float64)
    return (bboxes * sx) * nx


def generate_anchors_reference(ratios, scales, num_anchors, feature_map_shape):
    """utils.py:31-61: anchor 0 is the square sqrt(s_i * s_{i+1}) (last map: s * 0.99); anchors 1.. use
    ratios[:num_anchors-1] with h = s/sqrt(r), w = s*sqrt(r); all centred on (0.5, 0.5), feature-map units."""
    fh, fw = feature_map_shape
    h = np.zeros(num_anchors)
    w = np.zeros(num_anchors)
    if len(scales) > 1:
        h[0] = w[0] = np.sqrt(scales[0] * scales[1]) * fh
    else:
        h[0], w[0] = scales[0] * fh * 0.99, scales[0] * fw * 0.99
    r = np.asarray(ratios, dtype=np.float64)[:num_anchors - 1]
    h[1:] = scales[0] / np.sqrt(r) * fh
    w[1:] = scales[0] * np.sqrt(r) * fw
    return np.column_stack([0.5 - w / 2, 0.5 - h / 2, 0.5 + w / 2, 0.5 + h / 2])


def generate_anchors_per_feat_map(feature_map_shape, anchor_reference):
    """utils.py:95-145: reference + integer (x, y) cell shifts, cells row-major, anchors innermost."""
    fh, fw = feature_map_shape
    ys, xs = np.mgrid[0:fh, 0:fw]
    shifts = np.stack([xs.ravel(), ys.ravel(), xs.ravel(), ys.ravel()], axis=1)        # (fh*fw, 4)
    return (shifts[:, None, :] + anchor_reference[None, :, :]).reshape(-1, 4)


def clip_boxes(bboxes, imshape):
    """utils/bbox_transform.py:105-122 (numpy twin): clip to [0, W-1] x [0, H-1], float32."""
    b = np.asarray(bboxes).astype(np.float32)
    hi = np.array([imshape[1] - 1., imshape[0] - 1., imshape[1] - 1., imshape[0] - 1.], dtype=np.float32)
    return np.maximum(np.minimum(b, hi), np.float32(0.))


def generate_raw_anchors(feat_shapes, anchor_min_scale, anchor_max_scale, anchor_ratios, anchors_per_point):
    """utils.py:64-92 on a list of (fh, fw) shapes: list of per-map anchors in feature-map units."""
    scales = np.linspace(anchor_min_scale, anchor_max_scale, len(feat_shapes))
    out = []
    for i, shp in enumerate(feat_shapes):
        ref = generate_anchors_reference(anchor_ratios, scales[i:i + 2], anchors_per_point[i], shp)
        out.append(generate_anchors_per_feat_map(shp, ref))
    return out


def generate_all_anchors(feat_shapes, image_hw, anchor_min_scale, anchor_max_scale, anchor_ratios, anchors_per_point):
    """ssd.py:111-129: scale every map's anchors to the image, clip, concatenate -> (N, 4) float32."""
    raw = generate_raw_anchors(feat_shapes, anchor_min_scale, anchor_max_scale, anchor_ratios, anchors_per_point)
    parts = [clip_boxes(adjust_bboxes(a, shp[0], shp[1], image_hw[0], image_hw[1]), image_hw)
             for a, shp in zip(raw, feat_shapes)]
    return np.concatenate(parts, axis=0).astype(np.float32)
