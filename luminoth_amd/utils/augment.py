"""Training-time data augmentation of the dataset reader, host side (reference: luminoth/utils/image.py:150-316 `patch_image`,
:373-449 `random_patch`, :452-498 `random_resize`, :501-566 `random_distortion`, :569-620 `expand`; applied by
luminoth/datasets/object_detection_dataset.py:141-200 BEFORE the resize to the network's input size).

In the reference these are CPU-side TensorFlow image ops inside the input queue; here they are numpy on the decoded host
image (the step that follows — the resize to the model's input size, with the flip folded in — stays on the device).  Only
an image that actually draws one of them leaves the uint8 fast path of the reader.  The random draws come from the
dataset's seeded numpy generator in the reference's order of `tf.random_uniform` calls (TF's own stream is not
reproducible); everything after the draws is deterministic and follows the TF 1.x ops the reference calls:

  * `tf.image.resize_images(BILINEAR)`  — legacy sampling src = dst * in/out, no half-pixel centre (same arithmetic as
    `lmh_resize_bilinear` / oracle/image.py);
  * `tf.image.adjust_brightness` (x + delta on float images), `adjust_contrast` ((x - mean_c) * f + mean_c, per-channel
    mean over the image), `adjust_hue` / `adjust_saturation` (RGB -> HSV -> RGB on the raw float values; both
    conversions are scale-free in V, so 0..255 images behave like 0..1 ones).

Quirks kept: `patch_image` compares ONE global mean of all boxes' x coordinates (its `reduce_mean` has no axis) but a
per-box y centre; a patch that would keep no box returns its inputs unchanged; the patch is resized back to the original
size; `random_patch` clamps the minimum patch size to (H - 1, W - 1); `expand` truncates its paddings to int32 and pads
with a constant.
"""
import numpy as np

from luminoth_amd.utils.image import adjust_bboxes

_F = np.float32


def resize_bilinear_host(image, new_height, new_width):
    """tf.image.resize_images(image, (new_height, new_width), BILINEAR) of TF 1.x on a (H, W, C) array -> float32."""
    img = np.asarray(image, dtype=_F)
    H, W = img.shape[0], img.shape[1]
    new_height, new_width = int(new_height), int(new_width)
    if (new_height, new_width) == (H, W):
        return img.copy()
    ys = np.arange(new_height, dtype=_F) * (_F(H) / _F(new_height))
    xs = np.arange(new_width, dtype=_F) * (_F(W) / _F(new_width))
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    y1 = np.minimum(y0 + 1, H - 1)
    x1 = np.minimum(x0 + 1, W - 1)
    ly = (ys - y0.astype(_F))[:, None, None]
    lx = (xs - x0.astype(_F))[None, :, None]
    top = img[y0][:, x0] + (img[y0][:, x1] - img[y0][:, x0]) * lx
    bot = img[y1][:, x0] + (img[y1][:, x1] - img[y1][:, x0]) * lx
    return (top + (bot - top) * ly).astype(_F)


def _clip_to(boxes, height, width):
    """utils/bbox_transform.py clip_boxes: x in [0, width - 1], y in [0, height - 1]."""
    b = np.asarray(boxes, dtype=_F).copy()
    b[:, 0] = np.clip(b[:, 0], 0, width - 1)
    b[:, 2] = np.clip(b[:, 2], 0, width - 1)
    b[:, 1] = np.clip(b[:, 1], 0, height - 1)
    b[:, 3] = np.clip(b[:, 3], 0, height - 1)
    return b


def patch_image(image, bboxes=None, offset_height=0, offset_width=0, target_height=None, target_width=None):
    """image.py:150-316.  -> {'image': float32 (H, W, C)[, 'bboxes': int32 (G', 5)]}"""
    img = np.asarray(image)
    H, W = img.shape[0], img.shape[1]
    if target_height is None:
        target_height = H - offset_height - 1
    if target_width is None:
        target_width = W - offset_width - 1
    patch = img[offset_height:offset_height + target_height, offset_width:offset_width + target_width]
    resized = resize_bilinear_host(patch, H, W)
    if bboxes is None:
        return {'image': resized}
    b = np.asarray(bboxes)
    if np.issubdtype(b.dtype, np.integer):
        # the dataset hands int32 boxes (object_detection_dataset.py queue dtypes) and tf.reduce_mean of an integer
        # tensor is an integer (sum // count): a half-integer centre on the patch border is truncated first
        cx = int(b[:, [0, 2]].astype(np.int64).sum()) // (2 * b.shape[0])       # no axis in the reference: ONE value
        cy = b[:, [1, 3]].astype(np.int64).sum(axis=1) // 2
    else:
        cx = _F(np.mean(b[:, [0, 2]].astype(_F)))                # no axis in the reference: one value for all boxes
        cy = b[:, [1, 3]].astype(_F).mean(axis=1)
    inside = (cx > offset_width) & (cx < target_width + offset_width) & \
             (cy > offset_height) & (cy < target_height + offset_height)
    kept = b[inside]
    if kept.shape[0] < 1:                                        # would lose every box: nothing changes
        return {'image': np.asarray(image), 'bboxes': b}
    moved = kept[:, :4].astype(_F) - np.array([offset_width, offset_height, offset_width, offset_height], _F)
    clipped = _clip_to(moved, patch.shape[0], patch.shape[1]).astype(np.int32)
    boxes = np.concatenate([clipped, kept[:, 4:].astype(np.int32)], axis=1)
    boxes = adjust_bboxes(boxes, patch.shape[0], patch.shape[1], H, W)
    return {'image': resized, 'bboxes': boxes}


def _randint(rng, low, high):
    """tf.random_uniform(shape=[], minval=low, maxval=high, dtype=int32): an integer in [low, high)."""
    return int(rng.randint(int(low), int(high))) if int(high) > int(low) else int(low)


def random_patch(image, bboxes=None, min_height=600, min_width=600, rng=None):
    """image.py:373-449: draws offset_width, offset_height, target_width, target_height in that order."""
    rng = rng or np.random
    img = np.asarray(image)
    H, W = img.shape[0], img.shape[1]
    min_height, min_width = min(int(min_height), H - 1), min(int(min_width), W - 1)
    offset_width = _randint(rng, 0, W - min_width)
    offset_height = _randint(rng, 0, H - min_height)
    target_width = _randint(rng, min_width, W - offset_width)
    target_height = _randint(rng, min_height, H - offset_height)
    return patch_image(img, bboxes, offset_height=offset_height, offset_width=offset_width,
                       target_height=target_height, target_width=target_width)


def random_resize(image, bboxes=None, min_size=600, max_size=980, rng=None):
    """image.py:452-498: a random (height, width), each in [min_size, max_size)."""
    rng = rng or np.random
    img = np.asarray(image)
    H, W = img.shape[0], img.shape[1]
    new_h, new_w = _randint(rng, min_size, max_size), _randint(rng, min_size, max_size)
    out = {'image': resize_bilinear_host(img, new_h, new_w)}
    if bboxes is not None:
        out['bboxes'] = adjust_bboxes(bboxes, H, W, new_h, new_w)
    return out


def _rgb_to_hsv(rgb):
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    v = np.max(rgb, axis=-1)
    mn = np.min(rgb, axis=-1)
    rng_ = v - mn
    s = np.where(v > 0, rng_ / np.where(v > 0, v, 1), 0).astype(_F)
    safe = np.where(rng_ > 0, rng_, 1)
    h = np.where(v == r, (g - b) / safe, np.where(v == g, 2.0 + (b - r) / safe, 4.0 + (r - g) / safe))
    h = np.where(rng_ > 0, h / 6.0, 0.0)
    h = np.where(h < 0, h + 1.0, h).astype(_F)
    return h, s, v.astype(_F)


def _hsv_to_rgb(h, s, v):
    h6 = h * 6.0
    k = np.stack([(5.0 + h6) % 6.0, (3.0 + h6) % 6.0, (1.0 + h6) % 6.0], axis=-1)
    t = np.clip(np.minimum(k, 4.0 - k), 0.0, 1.0)
    return (v[..., None] * (1.0 - s[..., None] * t)).astype(_F)


def random_distortion(image, bboxes=None, brightness=None, contrast=None, hue=None, saturation=None, rng=None):
    """image.py:501-566; draw order brightness, contrast, hue, saturation (only the configured ones)."""
    rng = rng or np.random
    img = np.asarray(image, dtype=_F).copy()
    if brightness is not None:
        d = float(brightness.get('max_delta', 0.3))
        img = img + _F(rng.uniform(-d, d))
    if contrast is not None:
        f = _F(rng.uniform(float(contrast.get('lower', 0.8)), float(contrast.get('upper', 1.2))))
        mean = img.reshape(-1, img.shape[-1]).mean(axis=0, dtype=np.float64).astype(_F)
        img = (img - mean) * f + mean
    if hue is not None:
        d = float(hue.get('max_delta', 0.2))
        h, s, v = _rgb_to_hsv(img)
        h = np.mod(h + _F(rng.uniform(-d, d)), 1.0).astype(_F)
        img = _hsv_to_rgb(h, s, v)
    if saturation is not None:
        f = _F(rng.uniform(float(saturation.get('lower', 0.8)), float(saturation.get('upper', 1.2))))
        h, s, v = _rgb_to_hsv(img)
        img = _hsv_to_rgb(h, np.clip(s * f, 0.0, 1.0).astype(_F), v)
    out = {'image': img.astype(_F)}
    if bboxes is not None:
        out['bboxes'] = np.asarray(bboxes)
    return out


def expand(image, bboxes=None, fill=0, min_ratio=1, max_ratio=4, rng=None):
    """image.py:569-620: zoom out by padding; draws the size multiplier, then pad_left, then pad_top."""
    rng = rng or np.random
    img = np.asarray(image)
    H, W = _F(img.shape[0]), _F(img.shape[1])
    mult = _F(rng.uniform(float(min_ratio), float(max_ratio)))
    new_h, new_w = H * mult, W * mult
    pad_left = _F(rng.uniform(0.0, max(float(new_w - W), 0.0))) if new_w > W else _F(0)
    pad_right = new_w - W - pad_left
    pad_top = _F(rng.uniform(0.0, max(float(new_h - H), 0.0))) if new_h > H else _F(0)
    pad_bottom = new_h - H - pad_top
    pt, pb, pl, pr = (int(v) for v in (pad_top, pad_bottom, pad_left, pad_right))      # tf.to_int32: truncation
    out_img = np.full((img.shape[0] + pt + pb, img.shape[1] + pl + pr, img.shape[2]), fill, dtype=img.dtype)
    out_img[pt:pt + img.shape[0], pl:pl + img.shape[1]] = img
    out = {'image': out_img}
    if bboxes is not None:
        b = np.asarray(bboxes).astype(np.int32).copy()
        b[:, :4] += np.array([pl, pt, pl, pt], np.int32)
        out['bboxes'] = b
    return out


AUGMENTATIONS = {'patch': random_patch, 'resize': random_resize, 'distortion': random_distortion, 'expand': expand}
