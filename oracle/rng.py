"""Counter-based hash RNG shared (bit-for-bit) by the oracle and the HIP kernels.

The reference subsamples fg/bg anchors and proposals with `tf.random_shuffle`
(luminoth/models/fasterrcnn/rpn_target.py:206,243; rcnn_target.py:172,223),
whose Philox stream derivation is TF-internal and not reproducible.  Both the
oracle and `luminoth_amd/csrc/lmh_common.h: lmh_hash_u32` instead rank the
candidates of a stream by `(hash(seed, stream, index), index)` and keep the
`k` smallest: a uniformly random k-subset, i.e. the same distribution as
"shuffle and drop the first n-k".
"""
import numpy as np

STREAM_RPN_FG = 0
STREAM_RPN_BG = 1
STREAM_RCNN_FG = 2
STREAM_RCNN_BG = 3
STREAM_SSD = 4
STREAM_DROPOUT = 5

_M32 = np.uint64(0xFFFFFFFF)


def _fmix32(h):
    h = h.astype(np.uint64)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def hash_u32(seed, stream, idx):
    """uint32 hash of (seed, stream, idx); `idx` may be an array."""
    idx = np.asarray(idx, dtype=np.uint64)
    h = (np.uint64(seed & 0xFFFFFFFF) ^ ((idx * np.uint64(0x9E3779B1)) & _M32))
    h = _fmix32(h)
    h ^= (np.uint64(stream) * np.uint64(0x85EBCA77)) & _M32
    h = _fmix32(h)
    return h.astype(np.uint32)


def image_seed(seed, step, image):
    """Per-(step, image) seed handed to the kernels (host side mixes it)."""
    seed = 0 if seed is None else int(seed)
    h = hash_u32(seed, 0x51ED, np.uint64(step & 0xFFFFFFFF))
    h = hash_u32(int(h), 0xA11CE, np.uint64(image & 0xFFFFFFFF))
    return int(h)


def keep_k_smallest(candidates, k, seed, stream):
    """Boolean mask over `candidates` (index array) keeping the k with the
    smallest (hash, index) composite key."""
    candidates = np.asarray(candidates, dtype=np.int64)
    if k >= candidates.shape[0]:
        return np.ones(candidates.shape[0], dtype=bool)
    keys = (hash_u32(seed, stream, candidates).astype(np.uint64) << np.uint64(32)) \
        | candidates.astype(np.uint64)
    order = np.argsort(keys, kind='stable')
    keep = np.zeros(candidates.shape[0], dtype=bool)
    keep[order[:max(k, 0)]] = True
    return keep


def dropout_mask(n, keep_prob, seed):
    """Twin of csrc/elementwise.hip::k_dropout: element i is kept iff hash(seed, STREAM_DROPOUT, i) < keep_prob*2^32."""
    t = float(keep_prob) * 4294967296.0
    thr = 0xFFFFFFFF if t >= 4294967295.0 else int(t)
    return hash_u32(seed, STREAM_DROPOUT, np.arange(n)) < np.uint32(thr)
