"""Anchor reference generation (host side, numpy float64).

Same contract as luminoth/utils/anchors.py:4-52: ratio-major / scale-minor
reference boxes centred on 0 with the inclusive-pixel (w-1)/2 half extents, and
ValueError when the base size is too small.  The grid itself is never
materialised on the product path: kernels regenerate anchor n from the
int32-truncated reference (fasterrcnn.py:299-302 quirk) and the stride.
"""
import numpy as np


def generate_anchors_reference(base_size, aspect_ratios, scales):
    ratios = np.repeat(np.asarray(aspect_ratios, dtype=np.float64), len(scales))
    scl = np.tile(np.asarray(scales, dtype=np.float64), len(aspect_ratios))
    root = np.sqrt(ratios)
    half_h = (scl * root * base_size - 1) / 2
    half_w = (scl / root * base_size - 1) / 2
    ref = np.stack([-half_w, -half_h, half_w, half_h], axis=1)
    if ((2 * half_w).astype(np.int64) == 0).any() or ((2 * half_h).astype(np.int64) == 0).any():
        raise ValueError('base_size {} is too small for aspect_ratios and scales.'.format(base_size))
    return ref


def truncate_reference(anchor_reference):
    """float64 reference -> int32 as TF does when adding it to an int32 grid."""
    return np.trunc(anchor_reference).astype(np.int32)


def all_anchors_numpy(anchor_reference, feat_h, feat_w, stride):
    """(feat_h*feat_w*A, 4) int32 grid, for debugging / visualisation only."""
    ref = truncate_reference(anchor_reference)
    ys, xs = np.meshgrid(np.arange(feat_h, dtype=np.int32) * stride,
                         np.arange(feat_w, dtype=np.int32) * stride, indexing='ij')
    shifts = np.stack([xs, ys, xs, ys], axis=-1).reshape(-1, 1, 4)
    return (shifts + ref[None]).reshape(-1, 4).astype(np.int32)
