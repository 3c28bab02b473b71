"""Host twin of the counter hash in csrc/lmh_common.h (lmh_hash_u32): derives
the per-(step, image) seed handed to the target-sampling kernels.  Replaces the
role of `seed=` in tf.random_shuffle (rpn_target.py:206,243; rcnn_target.py:
172,223)."""

_M = 0xFFFFFFFF


def _fmix32(h):
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & _M
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & _M
    h ^= h >> 16
    return h


def hash_u32(seed, stream, idx):
    h = (seed & _M) ^ ((idx * 0x9E3779B1) & _M)
    h = _fmix32(h)
    h ^= (stream * 0x85EBCA77) & _M
    return _fmix32(h)


def image_seed(seed, step, image):
    seed = 0 if seed is None else int(seed)
    h = hash_u32(seed, 0x51ED, step & _M)
    return hash_u32(h, 0xA11CE, image & _M)
