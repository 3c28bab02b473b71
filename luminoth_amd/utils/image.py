"""Image resize of the `lumi predict` / dataset-preprocess path (reference: luminoth/utils/image.py:6-147).

Host side computes the scale factor and output size with the reference's float32 arithmetic and int32
truncation; the pixels are resampled on the device by `lmh_resize_bilinear` (TF 1.x legacy bilinear).  The
training-time augmentations of the reference module (patch, resize, distortion, expand: image.py:150-620) are host code
in luminoth_amd/utils/augment.py; flip is folded into the device resize.
"""
import numpy as np
import torch

from luminoth_amd import kernels as K

_F = np.float32


def _device_image(image, device=None):
    if not torch.is_tensor(image):
        image = torch.from_numpy(np.ascontiguousarray(np.asarray(image)))
    if image.dtype not in (torch.uint8, torch.float32):
        image = image.to(torch.float32)
    if not image.is_cuda:
        if device is None:
            if not torch.cuda.is_available():
                from luminoth_amd import _lib
                raise _lib.LuminothHipError('image resize needs a ROCm device (no CPU fallback on the product path)')
            device = torch.device('cuda', torch.cuda.current_device())
        image = image.to(device)
    return image


def adjust_bboxes(bboxes, old_height, old_width, new_height, new_width):
    """image.py:6-35.  bboxes (G,5) [x_min, y_min, x_max, y_max, label] -> int32, truncated toward zero."""
    b = np.asarray(bboxes).astype(_F)
    out = np.empty(b.shape, np.int32)
    for col, (old, new) in enumerate(((old_width, new_width), (old_height, new_height)) * 2):
        out[:, col] = (b[:, col] / _F(old) * _F(new)).astype(np.int32)
    out[:, 4] = b[:, 4].astype(np.int32)
    return out


def resize_plan(height, width, min_size=None, max_size=None):
    """The float32 arithmetic of image.py:59-89: returns (scale_factor, new_height, new_width) as float32 — the
    output size is the int32 truncation of the new sizes, the boxes are scaled by the untruncated ones."""
    height, width = _F(height), _F(width)
    up = max(_F(min_size) / min(height, width), _F(1.0)) if min_size is not None else _F(1.0)
    down = min(_F(max_size) / max(height, width), _F(1.0)) if max_size is not None else _F(1.0)
    scale = _F(up) * _F(down)
    return scale, height * scale, width * scale


def resize_image(image, bboxes=None, min_size=None, max_size=None, device=None, flip_lr=False, flip_ud=False):
    """image.py:38-114: upscale so the short side reaches `min_size`, downscale so the long side fits
    `max_size` (the product of both factors — an image may still end outside either bound)."""
    image = _device_image(image, device)
    height, width = image.shape[0], image.shape[1]
    scale, new_h, new_w = resize_plan(height, width, min_size, max_size)
    out = {'image': K.resize_bilinear(image, int(new_h), int(new_w), flip_lr, flip_ud), 'scale_factor': float(scale)}
    if bboxes is not None:
        out['bboxes'] = adjust_bboxes(bboxes, height, width, new_h, new_w)
    return out


def resize_image_fixed(image, new_height, new_width, bboxes=None, device=None, flip_lr=False, flip_ud=False):
    """image.py:117-147: scale_factor is (height factor, width factor)."""
    image = _device_image(image, device)
    height, width = _F(image.shape[0]), _F(image.shape[1])
    out = {'image': K.resize_bilinear(image, int(new_height), int(new_width), flip_lr, flip_ud),
           'scale_factor': (float(_F(new_height) / height), float(_F(new_width) / width))}
    if bboxes is not None:
        out['bboxes'] = adjust_bboxes(bboxes, height, width, new_height, new_width)
    return out


def flip_bboxes(bboxes, height, width, left_right=True, up_down=False):
    """The box half of image.py:318-370 (`flip_image`): int32 boxes, x' = width - x_max - 1 (same for y)."""
    b = np.asarray(bboxes).astype(np.int32).copy()
    if left_right:
        new_x_min = np.int32(width) - b[:, 2] - 1
        b[:, 2] = new_x_min + (b[:, 2] - b[:, 0])
        b[:, 0] = new_x_min
    if up_down:
        new_y_min = np.int32(height) - b[:, 3] - 1
        b[:, 3] = new_y_min + (b[:, 3] - b[:, 1])
        b[:, 1] = new_y_min
    return b


def flip_image(image, bboxes=None, left_right=True, up_down=False):
    """image.py:318-370 on a numpy array or a torch tensor (H,W,C) — materialised; the dataset iterator folds
    the flip into the resize kernel instead."""
    height, width = image.shape[0], image.shape[1]
    if bboxes is not None:
        bboxes = flip_bboxes(bboxes, height, width, left_right, up_down)
    if torch.is_tensor(image):
        dims = ([1] if left_right else []) + ([0] if up_down else [])
        image = torch.flip(image, dims) if dims else image
    else:
        image = np.asarray(image)
        if left_right:
            image = image[:, ::-1]
        if up_down:
            image = image[::-1]
    out = {'image': image}
    if bboxes is not None:
        out['bboxes'] = bboxes
    return out
