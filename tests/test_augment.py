"""Host-side training augmentation (luminoth_amd/utils/augment.py) against the reference's own test cases
(luminoth/utils/image_test.py:246-276 patch update condition, :355-449 random_patch, :451-498 random_resize, :500-571
random_distortion) and against the TF 1.x op definitions on hand-computable inputs.  TF's random stream is not
reproducible, so — like the reference's tests — the random wrappers are checked through their invariants."""
import colorsys

import numpy as np
import pytest

from luminoth_amd.utils import augment as A

F = np.float32


def _image(h, w, seed=0):
    return np.random.RandomState(seed).randint(0, 256, size=(h, w, 3)).astype(F)


def _boxes(h, w, n, label=3, seed=1):
    rs = np.random.RandomState(seed)
    x1 = rs.randint(0, w - 20, size=n)
    y1 = rs.randint(0, h - 20, size=n)
    x2 = x1 + rs.randint(5, 20, size=n)
    y2 = y1 + rs.randint(5, 20, size=n)
    return np.stack([x1, y1, x2, y2, np.full(n, label)], 1).astype(np.int32)


def test_patch_image_keeps_inputs_when_no_box_centre_is_inside():
    """image_test.py:246-276."""
    image = _image(600, 800)
    bboxes = np.array([(0, 0, 40, 40, 3), (430, 200, 480, 250, 3)], np.int32)
    out = A.patch_image(image, bboxes, offset_height=45, offset_width=45, target_height=100, target_width=200)
    np.testing.assert_array_equal(out['image'], image)
    np.testing.assert_array_equal(out['bboxes'], bboxes)


def test_patch_image_crops_moves_clips_and_rescales_boxes():
    """One box, so the reference's global x-mean IS its centre: patch (50..250, 100..500) of a 300x600 image, resized
    back to 300x600 (x scale 600/400 = 1.5, y scale 300/200 = 1.5, int32 truncation of adjust_bboxes)."""
    image = _image(300, 600)
    bboxes = np.array([(150, 80, 520, 120, 7)], np.int32)
    out = A.patch_image(image, bboxes, offset_height=50, offset_width=100, target_height=200, target_width=400)
    assert out['image'].shape == image.shape and out['image'].dtype == F
    # moved: (50, 30, 420, 70) -> clipped to the 400 x 200 patch: x2 = 399 -> rescaled by 1.5 and truncated
    np.testing.assert_array_equal(out['bboxes'], [[75, 45, int(F(399) / F(400) * F(600)), 105, 7]])
    # pixels: the patch itself, bilinearly stretched; its top-left pixel is exact
    np.testing.assert_array_equal(out['image'][0, 0], image[50, 100])


def test_patch_image_integer_centres_truncate_like_tf_reduce_mean():
    """The dataset's boxes are int32 and tf.reduce_mean of an integer tensor is an integer (image.py:208-228): a box
    whose centre is 10.5 on a patch that starts at 10 has centre 10 -> NOT inside (`greater` is strict); float boxes
    keep the half.  Odd coordinate sums on both axes."""
    image = _image(100, 120)
    # y: (5 + 16) / 2 = 10.5 -> 10, offset_height 10: 10 > 10 fails.  x (one global mean): (31 + 60) // 2 = 45
    bboxes = np.array([(31, 5, 60, 16, 2)], np.int32)
    out = A.patch_image(image, bboxes, offset_height=10, offset_width=20, target_height=60, target_width=80)
    np.testing.assert_array_equal(out['bboxes'], bboxes)                # dropped -> nothing changes
    np.testing.assert_array_equal(out['image'], image)
    out = A.patch_image(image, bboxes.astype(F), offset_height=10, offset_width=20, target_height=60, target_width=80)
    assert out['bboxes'].shape == (1, 5) and not np.array_equal(out['bboxes'], bboxes)      # 10.5 > 10: kept and moved
    # x: global integer mean over ALL boxes' x1, x2: (0 + 41 + 41 + 82) // 4 = 41 (41.0 exactly), offset 41: dropped
    b2 = np.array([(0, 20, 41, 60, 1), (41, 20, 82, 60, 1), (1, 20, 42, 61, 1)], np.int32)   # sum 207 // 6 = 34
    out = A.patch_image(image, b2, offset_height=0, offset_width=34, target_height=90, target_width=70)
    np.testing.assert_array_equal(out['bboxes'], b2)                     # 34 > 34 fails (float mean 34.5 would pass)


@pytest.mark.parametrize('min_hw', [(600, 600), (900, 900)])       # the second: larger than the image, image_test.py:396-434
def test_random_patch_invariants(min_hw):
    im_shape = (800, 600, 3) if min_hw[0] == 600 else (600, 800, 3)
    rng = np.random.RandomState(5)
    for trial in range(20):
        image = _image(*im_shape[:2], seed=trial)
        bboxes = _boxes(im_shape[0], im_shape[1], 5, seed=trial + 100)
        out = A.random_patch(image, bboxes, min_height=min_hw[0], min_width=min_hw[1], rng=rng)
        b, img = out['bboxes'], out['image']
        assert 0 < b.shape[0] <= 5 and (b >= 0).all()
        assert (b[:, [0, 2]] <= img.shape[1]).all() and (b[:, [1, 3]] <= img.shape[0]).all()
        assert img.shape == im_shape
    out = A.random_patch(_image(600, 800), None, rng=rng)              # image only: no 'bboxes' key
    assert 'bboxes' not in out and out['image'].shape == (600, 800, 3)


def test_random_resize_invariants():
    rng = np.random.RandomState(7)
    image, bboxes = _image(600, 800), _boxes(600, 800, 5)
    for _ in range(10):
        out = A.random_resize(image, bboxes, min_size=400, max_size=980, rng=rng)
        h, w = out['image'].shape[:2]
        assert 400 <= h < 980 and 400 <= w < 980 and out['bboxes'].shape == bboxes.shape
        np.testing.assert_array_equal(out['bboxes'][:, 4], bboxes[:, 4])
        assert (out['bboxes'][:, 2] <= w).all() and (out['bboxes'][:, 3] <= h).all()
    assert 'bboxes' not in A.random_resize(image, None, rng=rng)


def test_resize_bilinear_host_is_tf1_legacy_sampling():
    img = np.arange(4 * 6 * 3, dtype=F).reshape(4, 6, 3)
    np.testing.assert_array_equal(A.resize_bilinear_host(img, 4, 6), img)
    up = A.resize_bilinear_host(img, 8, 12)                  # scale 0.5: even outputs are the sources, odd ones midpoints
    np.testing.assert_array_equal(up[::2, ::2], img)
    np.testing.assert_allclose(up[1, 0], (img[0, 0] + img[1, 0]) / 2)
    np.testing.assert_array_equal(up[7, 11], img[3, 5])      # beyond the last source row / column: clamped
    down = A.resize_bilinear_host(img, 2, 3)                 # scale 2: picks every second pixel (no area averaging)
    np.testing.assert_array_equal(down, img[::2, ::2])


def test_hsv_round_trip_matches_colorsys():
    rs = np.random.RandomState(3)
    rgb = rs.rand(50, 3).astype(F)
    h, s, v = A._rgb_to_hsv(rgb)
    ref = np.array([colorsys.rgb_to_hsv(*p) for p in rgb.astype(np.float64)])
    np.testing.assert_allclose(np.stack([h, s, v], 1), ref, atol=2e-6)
    np.testing.assert_allclose(A._hsv_to_rgb(h, s, v), rgb, atol=2e-6)
    np.testing.assert_allclose(A._hsv_to_rgb(h, s, v * 255), rgb * 255, atol=1e-3)      # V is a pure scale


def test_random_distortion_shapes_and_small_changes():
    """image_test.py:500-571."""
    rng = np.random.RandomState(11)
    image, bboxes = _image(60, 90), _boxes(60, 90, 3)
    cfg = dict(brightness={'max_delta': 0.3}, contrast={'lower': 0.8, 'upper': 1.2}, hue={'max_delta': 0.2},
               saturation={'lower': 0.8, 'upper': 1.2})
    out = A.random_distortion(image, bboxes, rng=rng, **cfg)
    assert out['image'].shape == image.shape
    np.testing.assert_array_equal(out['bboxes'], bboxes)
    small = dict(brightness={'max_delta': 1e-5}, hue={'max_delta': 1e-5}, saturation={'lower': 0.99999, 'upper': 1.00001},
                 contrast={'lower': 0.99999, 'upper': 1.00001})
    out = A.random_distortion(image, bboxes, rng=rng, **small)
    np.testing.assert_allclose(out['image'], image, rtol=0.05, atol=0.1)


def test_distortion_ops_follow_the_tf_definitions():
    class Fixed(object):                      # a "generator" that returns the upper end of every range
        def uniform(self, lo, hi):
            return hi
    image = np.array([[[10, 20, 30], [50, 40, 90]]], F)
    out = A.random_distortion(image, brightness={'max_delta': 2.0}, rng=Fixed())['image']
    np.testing.assert_array_equal(out, image + 2)
    out = A.random_distortion(image, contrast={'lower': 0.5, 'upper': 2.0}, rng=Fixed())['image']
    mean = image.reshape(-1, 3).mean(0)
    np.testing.assert_allclose(out, (image - mean) * 2 + mean)
    out = A.random_distortion(image, saturation={'lower': 0.0, 'upper': 0.0}, rng=Fixed())['image']
    np.testing.assert_allclose(out, np.repeat(image.max(-1, keepdims=True), 3, -1))      # s = 0: grey at V
    out = A.random_distortion(image, hue={'max_delta': 1.0 / 3}, rng=Fixed())['image']       # +120 degrees: R->G->B->R
    np.testing.assert_allclose(out, image[..., [2, 0, 1]], atol=1e-4)


def test_expand_pads_and_shifts_boxes():
    rng = np.random.RandomState(2)
    image, bboxes = _image(50, 80).astype(np.uint8), _boxes(50, 80, 4)
    for _ in range(10):
        out = A.expand(image, bboxes, fill=7, min_ratio=1, max_ratio=4, rng=rng)
        img, b = out['image'], out['bboxes']
        assert img.dtype == image.dtype and img.shape[0] >= 50 and img.shape[1] >= 80 and img.shape[0] < 200 + 1
        dx, dy = b[0, 0] - bboxes[0, 0], b[0, 1] - bboxes[0, 1]
        np.testing.assert_array_equal(b[:, :4] - bboxes[:, :4], [[dx, dy, dx, dy]] * 4)
        np.testing.assert_array_equal(img[dy:dy + 50, dx:dx + 80], image)                 # the image sits at the shift
        mask = np.ones(img.shape[:2], bool)
        mask[dy:dy + 50, dx:dx + 80] = False
        assert (img[mask] == 7).all()


def test_dataset_applies_the_ssd_default_augmentation_list_in_order():
    """ssd/base_config.yml:83-102: flip, patch, distortion, expand, each with prob 0.5 — forced to 1 here.  The dataset
    applies the drawn strategies in the configured order on the host image (object_detection_dataset.py:141-200)."""
    from luminoth_amd.datasets.object_detection_dataset import ObjectDetectionDataset
    from luminoth_amd.utils.config import get_config
    cfg = get_config({'model': {'type': 'ssd'}, 'dataset': {'type': 'object_detection', 'dir': '/nonexistent'},
                      'train': {'seed': 3}})
    for entry in cfg.dataset.data_augmentation:
        list(entry.values())[0]['prob'] = 1.0
    ds = ObjectDetectionDataset(cfg)
    plan = ds._augment_decide()
    assert [(t, a) for t, a, _ in plan] == [('flip', True), ('patch', True), ('distortion', True), ('expand', True)]
    image = _image(120, 160).astype(np.uint8)
    bboxes = _boxes(120, 160, 6)
    out_img, out_boxes = ds._augment_host(image, bboxes, plan)
    assert out_img.flags['C_CONTIGUOUS'] and out_img.dtype == F            # distortion made it float
    assert out_img.shape[0] >= 120 and out_img.shape[1] >= 160             # expand only grows
    assert 0 < out_boxes.shape[0] <= 6 and (out_boxes[:, :4] >= 0).all()
    assert (out_boxes[:, 2] <= out_img.shape[1]).all() and (out_boxes[:, 3] <= out_img.shape[0]).all()
    np.testing.assert_array_equal(np.unique(out_boxes[:, 4]), [3])
    # nothing drawn -> the plan is all False and the flip-only fast path of preprocess() is taken
    for entry in cfg.dataset.data_augmentation:
        list(entry.values())[0]['prob'] = 0.0
    assert not any(a for _, a, _ in ObjectDetectionDataset(cfg)._augment_decide())
