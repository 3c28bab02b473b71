"""Oracle (test infrastructure only — never imported by the product path): the image resize in front of the
`lumi predict` path, restated in numpy.

  * `resize_image` / `resize_image_fixed` / `adjust_bboxes`: luminoth/utils/image.py:6-35,38-114,117-147.
    Pinned by the reference's own cases (luminoth/utils/image_test.py:118-244 — shapes, scale factors and
    adjusted boxes), reproduced in tests/test_oracle_predict.py.
  * `resize_bilinear`: the third-party kernel behind `tf.image.resize_images(method=BILINEAR)` — TensorFlow 1.x
    (`setup.py` pins no TF version; docs name 1.5+) `core/kernels/resize_bilinear_op.cc`, legacy sampling:
    scale = in/out (align_corners=False), in = i*scale, lower = (int)in, upper = min(lower+1, in-1),
    lerp = in - lower, value = top + (bottom-top)*y_lerp with top/bottom lerped along x.  TensorFlow is absent
    from this image and the reference has no value-level golden vector for it: PARITY UNPINNED for the pixel
    values (shapes and scale factors are pinned).
"""
import numpy as np

F = np.float32


def resize_bilinear(image, out_h, out_w):
    image = np.asarray(image)
    H, W, _ = image.shape
    src = image.astype(F)
    hs, ws = F(H) / F(out_h), F(W) / F(out_w)
    in_y = np.arange(out_h, dtype=F) * hs
    in_x = np.arange(out_w, dtype=F) * ws
    y0, x0 = in_y.astype(np.int64), in_x.astype(np.int64)
    y1, x1 = np.minimum(y0 + 1, H - 1), np.minimum(x0 + 1, W - 1)
    yl = (in_y - y0.astype(F))[:, None, None]
    xl = (in_x - x0.astype(F))[None, :, None]
    tl, tr = src[y0][:, x0], src[y0][:, x1]
    bl, br = src[y1][:, x0], src[y1][:, x1]
    top = tl + (tr - tl) * xl
    bot = bl + (br - bl) * xl
    return (top + (bot - top) * yl).astype(F)


def adjust_bboxes(bboxes, old_height, old_width, new_height, new_width):
    """image.py:6-35: normalise by the old size, scale by the (float) new size, truncate to int32."""
    b = np.asarray(bboxes).astype(F)
    out = np.empty(b.shape, np.int32)
    out[:, 0] = (b[:, 0] / F(old_width) * F(new_width)).astype(np.int32)
    out[:, 1] = (b[:, 1] / F(old_height) * F(new_height)).astype(np.int32)
    out[:, 2] = (b[:, 2] / F(old_width) * F(new_width)).astype(np.int32)
    out[:, 3] = (b[:, 3] / F(old_height) * F(new_height)).astype(np.int32)
    out[:, 4] = b[:, 4].astype(np.int32)
    return out


def resize_image(image, bboxes=None, min_size=None, max_size=None):
    """image.py:38-114."""
    height, width = F(image.shape[0]), F(image.shape[1])
    up = max(F(min_size) / min(height, width), F(1.0)) if min_size is not None else F(1.0)
    down = min(F(max_size) / max(height, width), F(1.0)) if max_size is not None else F(1.0)
    scale = F(up) * F(down)
    new_h, new_w = height * scale, width * scale
    out = {'image': resize_bilinear(image, int(new_h), int(new_w)), 'scale_factor': float(scale)}
    if bboxes is not None:
        out['bboxes'] = adjust_bboxes(bboxes, height, width, new_h, new_w)
    return out


def resize_image_fixed(image, new_height, new_width, bboxes=None):
    """image.py:117-147: scale_factor is the tuple (height factor, width factor)."""
    height, width = F(image.shape[0]), F(image.shape[1])
    out = {'image': resize_bilinear(image, int(new_height), int(new_width)),
           'scale_factor': (float(F(new_height) / height), float(F(new_width) / width))}
    if bboxes is not None:
        out['bboxes'] = adjust_bboxes(bboxes, height, width, new_height, new_width)
    return out


def format_predictions(objects, labels, probs, scale_factor, class_labels=None):
    """utils/predicting.py:112-148: boxes back to the original image scale, int(round()), prob rounded to 4
    places, sorted by prob (descending, stable)."""
    objects = np.array(objects, dtype=np.float32).reshape(-1, 4)
    labels = [int(l) for l in labels]
    if class_labels is not None:
        labels = [class_labels[l] for l in labels]
    if isinstance(scale_factor, tuple):
        objects = objects / np.array([scale_factor[1], scale_factor[0], scale_factor[1], scale_factor[0]])
    else:
        objects = objects / scale_factor
    objs = [[int(round(c)) for c in o] for o in objects.tolist()]
    preds = [{'bbox': o, 'label': l, 'prob': round(float(p), 4)} for o, l, p in zip(objs, labels, probs)]
    return sorted(preds, key=lambda x: x['prob'], reverse=True)


def flip_image(image, bboxes=None, left_right=True, up_down=False):
    """luminoth/utils/image.py:318-370; pinned by image_test.py:278-353 (tests/test_tfrecord.py)."""
    image = np.asarray(image)
    height, width = image.shape[0], image.shape[1]
    if bboxes is not None:
        bboxes = np.asarray(bboxes).astype(np.int32)
    if left_right:
        image = image[:, ::-1]
        if bboxes is not None:
            x_min, y_min, x_max, y_max, label = bboxes.T
            nx = width - x_max - 1
            bboxes = np.stack([nx, y_min, nx + (x_max - x_min), y_max, label], 1)
    if up_down:
        image = image[::-1]
        if bboxes is not None:
            x_min, y_min, x_max, y_max, label = bboxes.T
            ny = height - y_max - 1
            bboxes = np.stack([x_min, ny, x_max, ny + (y_max - y_min), label], 1)
    out = {'image': np.ascontiguousarray(image)}
    if bboxes is not None:
        out['bboxes'] = bboxes.astype(np.int32)
    return out
