"""TFRecord container + `tf.train.SequenceExample` wire format of the reference's datasets.

Records: luminoth/tools/dataset/writers/object_detection_writer.py:123-177 (`_record_to_tf`: context
{width,height,depth: int64; filename,image_raw: bytes}, feature_lists {label,xmin,ymin,xmax,ymax}: one int64
Feature per box), read back by luminoth/datasets/object_detection_dataset.py:40-54,96-100.  Container framing and
CRC-32C run in C (libluminoth_io.so, include/luminoth_io.h); the protobuf messages (tensorflow/core/example/
{example,feature}.proto — third party, absent from the reference tree) are decoded here with a minimal wire-format
reader, checked in tests/test_tfrecord.py against the `protobuf` runtime built from the published schema.
"""
import ctypes
import mmap
import os

import numpy as np

_LIB = None
ERRORS = {-1: 'truncated record', -2: 'corrupted record length', -3: 'corrupted record data'}


class DataLossError(IOError):
    """tf.errors.DataLossError: what tf.TFRecordReader raises on a bad CRC / truncated file."""


def io_lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'csrc', 'libluminoth_io.so')
        if not os.path.exists(path):
            raise OSError('%s is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                          '(or luminoth_amd/csrc/build.sh)' % path)
        lib = ctypes.CDLL(path)
        vp, sz, u64p = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)
        lib.lmh_io_crc32c.restype, lib.lmh_io_crc32c.argtypes = ctypes.c_uint32, [vp, sz]
        lib.lmh_io_masked_crc32c.restype, lib.lmh_io_masked_crc32c.argtypes = ctypes.c_uint32, [vp, sz]
        lib.lmh_io_crc32c_portable.restype, lib.lmh_io_crc32c_portable.argtypes = ctypes.c_uint32, [vp, sz]
        lib.lmh_io_crc32c_hw.restype, lib.lmh_io_crc32c_hw.argtypes = ctypes.c_int, []
        lib.lmh_io_tfrecord_index.restype = ctypes.c_int64
        lib.lmh_io_tfrecord_index.argtypes = [vp, sz, ctypes.c_int, u64p, u64p, sz, u64p]
        lib.lmh_io_tfrecord_frame.restype, lib.lmh_io_tfrecord_frame.argtypes = sz, [vp, ctypes.c_uint64, vp]
        _LIB = lib
    return _LIB


def _addr(buf):
    a = np.frombuffer(buf, dtype=np.uint8)
    return a, a.ctypes.data


def crc32c(data):
    a, p = _addr(data)
    return int(io_lib().lmh_io_crc32c(p, a.size))


def masked_crc32c(data):
    a, p = _addr(data)
    return int(io_lib().lmh_io_masked_crc32c(p, a.size))


def index_records(buf, verify=True):
    """(offsets, lengths) uint64 arrays of every record payload in a .tfrecords file image."""
    lib = io_lib()
    a, p = _addr(buf)
    err = ctypes.c_uint64(0)
    cap = max(16, a.size // 4096)
    while True:
        offs, lens = np.empty(cap, np.uint64), np.empty(cap, np.uint64)
        n = lib.lmh_io_tfrecord_index(p, a.size, int(bool(verify)),
                                      offs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                      lens.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), cap, ctypes.byref(err))
        if n < 0:
            raise DataLossError('%s at byte %d' % (ERRORS.get(int(n), 'error %d' % n), err.value))
        if n <= cap:
            return offs[:n].copy(), lens[:n].copy()
        cap = int(n)


def frame_record(payload):
    payload = bytes(payload)
    out = np.empty(len(payload) + 16, np.uint8)
    src = np.frombuffer(payload, np.uint8) if payload else np.empty(0, np.uint8)
    n = io_lib().lmh_io_tfrecord_frame(src.ctypes.data, len(payload), out.ctypes.data)
    return out[:n].tobytes()


def write_records(path, payloads):
    with open(path, 'wb') as f:
        for p in payloads:
            f.write(frame_record(p))


class TFRecordFile(object):
    """Memory-mapped .tfrecords file: len(), [i] -> payload bytes (zero-copy memoryview until sliced)."""

    def __init__(self, path, verify=True):
        self.path = path
        self._f = open(path, 'rb')
        size = os.fstat(self._f.fileno()).st_size
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ) if size else b''
        self.offsets, self.lengths = index_records(self._mm, verify) if size else (np.empty(0, np.uint64),) * 2

    def __len__(self):
        return int(self.offsets.shape[0])

    def __getitem__(self, i):
        o, l = int(self.offsets[i]), int(self.lengths[i])
        return bytes(self._mm[o:o + l])

    def close(self):
        if self._mm:
            self._mm.close()
        self._f.close()


# ------------------------------------------------------------------ protobuf wire format ----
def _varint(b, i):
    x = s = 0
    while True:
        c = b[i]
        i += 1
        x |= (c & 0x7F) << s
        if c < 0x80:
            return x, i
        s += 7


def _fields(b):
    """Yields (field_number, wire_type, value) of one message; length-delimited values as memoryview slices."""
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 2:
            l, i = _varint(b, i)
            v = b[i:i + l]
            if len(v) != l:
                raise ValueError('truncated protobuf field')
            i += l
        elif wt == 5:
            v, i = bytes(b[i:i + 4]), i + 4
        elif wt == 1:
            v, i = bytes(b[i:i + 8]), i + 8
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield num, wt, v


def _int64(x):
    return x - (1 << 64) if x >= (1 << 63) else x


def _decode_feature(b):
    """Feature { oneof kind { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3; } }"""
    out = []
    for num, wt, v in _fields(b):
        if num == 1:                                            # BytesList { repeated bytes value = 1; }
            out = [bytes(x) for n2, _, x in _fields(v) if n2 == 1]
        elif num == 2:                                          # FloatList { repeated float value = 1 [packed] }
            vals = []
            for n2, wt2, x in _fields(v):
                if n2 == 1:
                    vals.extend(np.frombuffer(bytes(x), '<f4').tolist())
            out = vals
        elif num == 3:                                          # Int64List { repeated int64 value = 1 [packed] }
            vals = []
            for n2, wt2, x in _fields(v):
                if n2 != 1:
                    continue
                if wt2 == 0:
                    vals.append(_int64(x))
                else:
                    j = 0
                    while j < len(x):
                        y, j = _varint(x, j)
                        vals.append(_int64(y))
            out = vals
    return out


def _decode_map(b, value_fn):
    """repeated MapEntry { string key = 1; V value = 2; } carried in field 1 of Features / FeatureLists."""
    out = {}
    for num, _, entry in _fields(b):
        if num != 1:
            continue
        key, val = '', None
        for n2, _, x in _fields(entry):
            if n2 == 1:
                key = bytes(x).decode('utf-8')
            elif n2 == 2:
                val = value_fn(x)
        out[key] = val if val is not None else value_fn(memoryview(b''))
    return out


def decode_sequence_example(data):
    """SequenceExample { Features context = 1; FeatureLists feature_lists = 2; } ->
    (context {name: list}, feature_lists {name: [list per step]})."""
    b = memoryview(data)
    context, lists = {}, {}
    for num, _, v in _fields(b):
        if num == 1:
            context.update(_decode_map(v, _decode_feature))
        elif num == 2:
            lists.update(_decode_map(v, lambda fl: [_decode_feature(x) for n2, _, x in _fields(fl) if n2 == 1]))
    return context, lists


def _enc_varint(x):
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        c = x & 0x7F
        x >>= 7
        if x:
            out.append(c | 0x80)
        else:
            out.append(c)
            return bytes(out)


def _ld(num, payload):
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def _encode_feature(values):
    values = list(values)
    if values and isinstance(values[0], (bytes, bytearray, str)):
        body = b''.join(_ld(1, v.encode('utf-8') if isinstance(v, str) else bytes(v)) for v in values)
        return _ld(1, body)
    if values and isinstance(values[0], float):
        return _ld(2, _ld(1, np.asarray(values, '<f4').tobytes()))
    return _ld(3, _ld(1, b''.join(_enc_varint(int(v)) for v in values)) if values else b'')


def encode_sequence_example(context, feature_lists):
    """Inverse of decode_sequence_example (map entries in sorted key order — deterministic files)."""
    ctx = b''.join(_ld(1, _ld(1, k.encode('utf-8')) + _ld(2, _encode_feature(v))) for k, v in sorted(context.items()))
    fls = b''
    for k, steps in sorted(feature_lists.items()):
        fl = b''.join(_ld(1, _encode_feature(s)) for s in steps)
        fls += _ld(1, _ld(1, k.encode('utf-8')) + _ld(2, fl))
    return _ld(1, ctx) + _ld(2, fls)


def encode_detection_record(image_raw, filename, width, height, gt_boxes, depth=3):
    """The record of object_detection_writer.py:123-177; gt_boxes: dicts with label/xmin/ymin/xmax/ymax."""
    ctx = {'width': [int(width)], 'height': [int(height)], 'depth': [int(depth)], 'filename': [filename],
           'image_raw': [bytes(image_raw)]}
    lists = {k: [[int(b[k])] for b in gt_boxes] for k in ('label', 'xmin', 'ymin', 'xmax', 'ymax')}
    return encode_sequence_example(ctx, lists)


def decode_detection_record(data):
    """-> dict(image_raw, filename, width, height, depth, bboxes (G,5) int32 [xmin, ymin, xmax, ymax, label])
    (object_detection_dataset.py:96-122)."""
    ctx, lists = decode_sequence_example(data)
    cols = []
    for k in ('xmin', 'ymin', 'xmax', 'ymax', 'label'):
        cols.append([s[0] for s in lists.get(k, [])])
    bboxes = np.array(cols, dtype=np.int64).T.reshape(-1, 5).astype(np.int32)
    fn = ctx['filename'][0]
    return {'image_raw': ctx['image_raw'][0], 'filename': fn.decode('utf-8') if isinstance(fn, bytes) else fn,
            'width': int(ctx['width'][0]), 'height': int(ctx['height'][0]), 'depth': int(ctx.get('depth', [3])[0]),
            'bboxes': bboxes}
