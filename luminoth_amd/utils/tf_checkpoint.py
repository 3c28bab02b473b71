"""TensorFlow checkpoint files -> name-keyed numpy arrays (second half).

What the reference loads with `tf.train.Saver.restore`:
  * the slim pretrained base networks (`resnet_v1_50.ckpt`, `resnet_v1_101.ckpt`, `vgg_16.ckpt`, ... —
    luminoth/utils/checkpoint_downloader.py:11-23, restored into the base network by train.py:114-127 through the
    names of `get_base_network_checkpoint_vars`, base_network.py:243-259): the V1 format, ONE table file whose entry
    "" holds `SavedTensorSlices{meta}` and every other entry one `SavedTensorSlices{data: SavedSlice}`;
  * luminoth's own `model.ckpt-N` files (train.py:104-112): the V2 "tensor bundle" — `<prefix>.index`, a table of
    `BundleEntryProto` per variable name (entry "" = `BundleHeaderProto`), plus raw little-endian bytes in
    `<prefix>.data-0000k-of-0000n`.

Both containers are third party and absent from the reference tree (TensorFlow core/lib/io/table — the LevelDB table
format —, core/util/saved_tensor_slice.proto, core/util/tensor_slice_{reader,writer}.cc, core/util/tensor_bundle/*,
core/protobuf/tensor_bundle.proto).  This file restates their published layouts; there is no sample checkpoint in
the reference tree or in this image, so value-level parity with TensorFlow-written files is UNPINNED.  What is
pinned (tests/test_tf_checkpoint.py): the protobuf messages against the `protobuf` runtime built from the published
schemas, the Snappy block decoder against pyarrow's Snappy codec, block checksums through the RFC 3720 CRC-32C, the
table footer magic, and writer -> reader round trips for both formats.
"""
import os
import struct

import numpy as np

from luminoth_amd.datasets.tfrecord import _enc_varint, _fields, _int64, _ld, _varint, masked_crc32c

TABLE_MAGIC = 0xdb4775248b80fb57
BLOCK_TRAILER = 5                       # 1 byte compression type + 4 bytes masked crc32c
NO_COMPRESSION, SNAPPY = 0, 1

# tensorflow/core/framework/types.proto
DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 4: np.dtype('u1'), 5: np.dtype('<i2'),
          6: np.dtype('i1'), 9: np.dtype('<i8'), 10: np.dtype('bool'), 19: np.dtype('<f2')}
DTYPE_ENUM = {v: k for k, v in DTYPES.items()}


class CheckpointError(IOError):
    pass


# ------------------------------------------------------------------------------ snappy ----
def snappy_uncompress(data):
    """Snappy block format (format_description.txt): varint length, then literal / copy elements."""
    data = bytes(data)
    n, i = _varint(data, 0)
    out = bytearray()
    while i < len(data):
        tag = data[i]
        i += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(data[i:i + nb], 'little')
                i += nb
            ln += 1
            out += data[i:i + ln]
            i += ln
            continue
        if kind == 1:                                   # copy, 1-byte offset
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | data[i]
            i += 1
        elif kind == 2:                                 # copy, 2-byte offset
            ln = (tag >> 2) + 1
            off = data[i] | (data[i + 1] << 8)
            i += 2
        else:                                           # copy, 4-byte offset
            ln = (tag >> 2) + 1
            off = int.from_bytes(data[i:i + 4], 'little')
            i += 4
        if off == 0 or off > len(out):
            raise CheckpointError('corrupt snappy stream (offset %d at %d)' % (off, len(out)))
        start = len(out) - off
        if off >= ln:
            out += out[start:start + ln]
        else:                                           # overlapping run
            for k in range(ln):
                out.append(out[start + k])
    if len(out) != n:
        raise CheckpointError('corrupt snappy stream (%d bytes, header says %d)' % (len(out), n))
    return bytes(out)


# ------------------------------------------------------------------------ LevelDB table ----
def _handle(b, i=0):
    off, i = _varint(b, i)
    size, i = _varint(b, i)
    return off, size, i


def _read_block(buf, off, size, verify=True):
    end = off + size
    if end + BLOCK_TRAILER > len(buf):
        raise CheckpointError('table block [%d, %d) runs past the end of the file' % (off, end))
    ctype = buf[end]
    if verify:
        want = struct.unpack('<I', buf[end + 1:end + 5])[0]
        if masked_crc32c(buf[off:end + 1]) != want:
            raise CheckpointError('table block at %d: checksum mismatch' % off)
    body = bytes(buf[off:end])
    if ctype == SNAPPY:
        body = snappy_uncompress(body)
    elif ctype != NO_COMPRESSION:
        raise CheckpointError('table block at %d: unknown compression type %d' % (off, ctype))
    return body


def _block_entries(block):
    """(key, value) pairs of one block (prefix-compressed keys, restart array at the end)."""
    if len(block) < 4:
        raise CheckpointError('table block too small')
    num_restarts = struct.unpack('<I', block[-4:])[0]
    limit = len(block) - 4 - 4 * num_restarts
    if limit < 0:
        raise CheckpointError('table block: bad restart count')
    i, key = 0, b''
    while i < limit:
        shared, i = _varint(block, i)
        non_shared, i = _varint(block, i)
        vlen, i = _varint(block, i)
        key = key[:shared] + block[i:i + non_shared]
        i += non_shared
        yield key, block[i:i + vlen]
        i += vlen


def read_table(buf, verify=True):
    """All (key, value) pairs of a table file image, in key order."""
    buf = memoryview(buf)
    if len(buf) < 48:
        raise CheckpointError('not a table file (shorter than its footer)')
    footer = bytes(buf[len(buf) - 48:])
    if struct.unpack('<Q', footer[40:])[0] != TABLE_MAGIC:
        raise CheckpointError('not a table file (bad magic number)')
    _, _, i = _handle(footer, 0)                        # metaindex (unused)
    ioff, isize, _ = _handle(footer, i)
    out = []
    for _, hv in _block_entries(_read_block(buf, ioff, isize, verify)):
        doff, dsize, _ = _handle(hv, 0)
        out.extend(_block_entries(_read_block(buf, doff, dsize, verify)))
    return out


def write_table(entries, block_size=4096, restart_interval=16, compress=None):
    """Table file image from (key, value) pairs sorted by key — the writer counterpart, used to export weights and
    to exercise the reader.  Blocks are stored uncompressed unless `compress` (bytes -> Snappy block bytes) is
    given."""
    entries = [(bytes(k), bytes(v)) for k, v in entries]
    if any(a[0] >= b[0] for a, b in zip(entries, entries[1:])):
        raise ValueError('table keys must be strictly increasing')
    out = bytearray()

    def emit(block_entries):
        body, restarts, prev = bytearray(), [], b''
        for n, (k, v) in enumerate(block_entries):
            shared = 0
            if n % restart_interval == 0:
                restarts.append(len(body))
            else:
                while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                    shared += 1
            body += _enc_varint(shared) + _enc_varint(len(k) - shared) + _enc_varint(len(v)) + k[shared:] + v
            prev = k
        if not restarts:
            restarts = [0]
        for r in restarts:
            body += struct.pack('<I', r)
        body += struct.pack('<I', len(restarts))
        ctype = NO_COMPRESSION
        if compress is not None:
            body, ctype = bytearray(compress(bytes(body))), SNAPPY
        off = len(out)
        out.extend(body)
        out.append(ctype)
        out.extend(struct.pack('<I', masked_crc32c(bytes(body) + bytes([ctype]))))
        return off, len(body)

    index, cur, cur_bytes = [], [], 0
    for k, v in entries:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 3
        if cur_bytes >= block_size:
            index.append((cur[-1][0], emit(cur)))
            cur, cur_bytes = [], 0
    if cur or not index:
        index.append((cur[-1][0] if cur else b'', emit(cur)))
    meta_off, meta_size = emit([])
    idx_off, idx_size = emit([(k, _enc_varint(o) + _enc_varint(s)) for k, (o, s) in index])
    footer = _enc_varint(meta_off) + _enc_varint(meta_size) + _enc_varint(idx_off) + _enc_varint(idx_size)
    out.extend(footer + b'\0' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC))
    return bytes(out)


# -------------------------------------------------------------------- protobuf messages ----
def _shape(b):
    """TensorShapeProto { repeated Dim dim = 2 { int64 size = 1; string name = 2; } }"""
    dims = []
    for num, _, v in _fields(b):
        if num == 2:
            size = 0
            for n2, _, x in _fields(v):
                if n2 == 1:
                    size = _int64(x)
            dims.append(size)
    return tuple(dims)


def _enc_shape(shape):
    return b''.join(_ld(2, _enc_varint(1 << 3) + _enc_varint(int(d))) for d in shape)


def _slice(b, shape):
    """TensorSliceProto { repeated Extent extent = 1 { int64 start = 1; oneof { int64 length = 2; } } }
    -> tuple of python slices (an extent without a length covers the whole dimension)."""
    out = []
    for num, _, v in _fields(b):
        if num == 1:
            start, length = 0, None
            for n2, _, x in _fields(v):
                if n2 == 1:
                    start = _int64(x)
                elif n2 == 2:
                    length = _int64(x)
            out.append((start, length))
    while len(out) < len(shape):
        out.append((0, None))
    return tuple(slice(s, None if l is None else s + l) for s, l in out[:len(shape)])


def _tensor(b):
    """TensorProto -> ndarray (tensor_content, or the typed *_val lists tensor_slice_writer fills)."""
    dtype, shape, content, vals = 1, (), None, []
    fields = list(_fields(b))
    for num, wt, v in fields:
        if num == 1:
            dtype = v
        elif num == 2:
            shape = _shape(v)
        elif num == 4:
            content = bytes(v)
    if dtype not in DTYPES:
        raise CheckpointError('unsupported tensor dtype enum %d' % dtype)
    dt = DTYPES[dtype]
    if content is not None and len(content):
        arr = np.frombuffer(content, dtype=dt)
    else:
        field = {1: 5, 2: 6, 3: 7, 9: 10, 10: 11, 19: 13, 4: 7, 5: 7, 6: 7}[dtype]
        for num, wt, v in fields:
            if num != field:
                continue
            if wt == 2:                                  # packed
                if dtype == 1:
                    vals.append(np.frombuffer(bytes(v), '<f4'))
                elif dtype == 2:
                    vals.append(np.frombuffer(bytes(v), '<f8'))
                else:
                    j, xs = 0, []
                    while j < len(v):
                        y, j = _varint(v, j)
                        xs.append(_int64(y))
                    vals.append(np.array(xs, dtype=np.int64))
            elif wt == 5:
                vals.append(np.frombuffer(v, '<f4'))
            elif wt == 1:
                vals.append(np.frombuffer(v, '<f8'))
            else:
                vals.append(np.array([_int64(v)], dtype=np.int64))
        arr = np.concatenate(vals) if vals else np.zeros(0, dt)
        if dtype == 19:
            arr = arr.astype(np.uint16).view(np.float16)
        arr = arr.astype(dt)
    n = int(np.prod(shape)) if shape else 1
    if arr.size == 1 and n > 1:
        arr = np.full(n, arr[0], dtype=dt)              # TensorProto's "repeat the last value" shorthand
    if arr.size != n:
        raise CheckpointError('tensor has %d values for shape %s' % (arr.size, shape))
    return arr.reshape(shape)


# ------------------------------------------------------------------------------- V1 ----
def load_v1(path, verify=True):
    """{variable name: ndarray} of a V1 checkpoint (one table file)."""
    with open(path, 'rb') as f:
        entries = read_table(f.read(), verify)
    meta, out = {}, {}
    for key, value in entries:
        for num, _, v in _fields(value):
            if num == 1:                                 # SavedTensorSliceMeta { repeated SavedSliceMeta tensor = 1 }
                for n2, _, t in _fields(v):
                    if n2 != 1:
                        continue
                    name, shape, dtype = '', (), 1
                    for n3, _, x in _fields(t):
                        if n3 == 1:
                            name = bytes(x).decode('utf-8')
                        elif n3 == 2:
                            shape = _shape(x)
                        elif n3 == 3:
                            dtype = x
                    meta[name] = (shape, dtype)
            elif num == 2:                               # SavedSlice { name = 1; TensorSliceProto slice = 2; data = 3 }
                name, sl, data = '', b'', None
                for n2, _, x in _fields(v):
                    if n2 == 1:
                        name = bytes(x).decode('utf-8')
                    elif n2 == 2:
                        sl = x
                    elif n2 == 3:
                        data = _tensor(x)
                if data is None:
                    continue
                shape, dtype = meta.get(name, (data.shape, DTYPE_ENUM.get(data.dtype, 1)))
                if name not in out:
                    out[name] = np.zeros(shape, dtype=DTYPES.get(dtype, data.dtype))
                region = _slice(sl, shape)
                out[name][region] = data.reshape(out[name][region].shape)
    missing = set(meta) - set(out)
    if missing:
        raise CheckpointError('checkpoint lists tensors without data: %s' % sorted(missing)[:5])
    return out


def save_v1(path, tensors):
    """Writes {name: ndarray} as a V1 checkpoint: float data in TensorProto.float_val like tensor_slice_writer."""
    metas, entries = [], []
    for name in sorted(tensors):
        arr = np.asarray(tensors[name], order="C")
        enum = DTYPE_ENUM[arr.dtype]
        nm = name.encode('utf-8')
        extents = b''.join(_ld(1, b'') for _ in arr.shape)               # full extents (no start, no length)
        metas.append(_ld(1, _ld(1, nm) + _ld(2, _enc_shape(arr.shape)) + _enc_varint(3 << 3) + _enc_varint(enum) +
                         _ld(4, extents)))
        tp = _enc_varint(1 << 3) + _enc_varint(enum) + _ld(2, _enc_shape(arr.shape))
        if enum == 1:
            tp += _ld(5, arr.astype('<f4').tobytes())
        else:
            tp += _ld(4, arr.astype(DTYPES[enum]).tobytes())
        # key: OrderedCode(0, name, rank, (start, length) per dim) — any unique, sorted key works for readers that
        # (like TensorFlow's) find tensors through the values; the name keeps them ordered and distinct
        key = b'\x00' + nm + b'\x00\x01' + bytes([len(arr.shape)])
        entries.append((key, _ld(2, _ld(1, nm) + _ld(2, extents) + _ld(3, tp))))
    entries.append((b'', _ld(1, b''.join(metas))))
    with open(path, 'wb') as f:
        f.write(write_table(sorted(entries)))


# ------------------------------------------------------------------------------- V2 ----
def _shard_name(prefix, k, n):
    return '%s.data-%05d-of-%05d' % (prefix, k, n)


def load_v2(prefix, verify=True):
    """{variable name: ndarray} of a V2 tensor bundle (`prefix.index` + `prefix.data-*`)."""
    with open(prefix + '.index', 'rb') as f:
        entries = read_table(f.read(), verify)
    num_shards, out, shards = 1, {}, {}
    for key, value in entries:
        if key == b'':                                   # BundleHeaderProto { num_shards = 1; endianness = 2; }
            for num, _, v in _fields(value):
                if num == 1:
                    num_shards = v
                elif num == 2 and v != 0:
                    raise CheckpointError('big-endian tensor bundles are not supported')
            continue
        dtype, shape, shard, offset, size, crc, sliced = 1, (), 0, 0, 0, None, False
        for num, wt, v in _fields(value):                # BundleEntryProto
            if num == 1:
                dtype = v
            elif num == 2:
                shape = _shape(v)
            elif num == 3:
                shard = v
            elif num == 4:
                offset = _int64(v)
            elif num == 5:
                size = _int64(v)
            elif num == 6:
                crc = struct.unpack('<I', v)[0]
            elif num == 7:
                sliced = True
        if sliced:
            raise CheckpointError('partitioned variable %r: sliced bundle entries are not supported' % key)
        if dtype not in DTYPES:
            raise CheckpointError('unsupported dtype enum %d for %r' % (dtype, key))
        if shard not in shards:
            with open(_shard_name(prefix, shard, num_shards), 'rb') as f:
                shards[shard] = f.read()
        raw = shards[shard][offset:offset + size]
        if len(raw) != size:
            raise CheckpointError('%r: data shard is truncated' % key)
        if verify and crc is not None and masked_crc32c(raw) != crc:
            raise CheckpointError('%r: data checksum mismatch' % key)
        out[key.decode('utf-8')] = np.frombuffer(raw, dtype=DTYPES[dtype]).reshape(shape).copy()
    return out


def save_v2(prefix, tensors):
    data, entries = bytearray(), []
    for name in sorted(tensors):
        arr = np.asarray(tensors[name], order="C")
        enum = DTYPE_ENUM[arr.dtype]
        raw = arr.astype(DTYPES[enum]).tobytes()
        entry = (_enc_varint(1 << 3) + _enc_varint(enum) + _ld(2, _enc_shape(arr.shape)) +
                 _enc_varint(4 << 3) + _enc_varint(len(data)) + _enc_varint(5 << 3) + _enc_varint(len(raw)) +
                 _enc_varint((6 << 3) | 5) + struct.pack('<I', masked_crc32c(raw)))
        entries.append((name.encode('utf-8'), entry))
        data += raw
    header = _enc_varint(1 << 3) + _enc_varint(1) + _ld(3, _enc_varint(1 << 3) + _enc_varint(1))
    with open(_shard_name(prefix, 0, 1), 'wb') as f:
        f.write(bytes(data))
    with open(prefix + '.index', 'wb') as f:
        f.write(write_table([(b'', header)] + entries))


# ----------------------------------------------------------------------------- front ----
def load_checkpoint(path, verify=True):
    """V2 when `<path>.index` exists (path is the prefix), else a V1 table file."""
    if os.path.exists(path + '.index'):
        return load_v2(path, verify)
    if path.endswith('.index') and os.path.exists(path):
        return load_v2(path[:-len('.index')], verify)
    if not os.path.exists(path):
        raise CheckpointError('checkpoint %r not found' % path)
    return load_v1(path, verify)


def restore_base_network(model, path, strict=True):
    """train.py:114-127: loads the variables of `path` into the tensors `get_base_network_checkpoint_vars()` maps
    them to (checkpoint name = variable name without the module scope).  Returns the list of restored names."""
    values = load_checkpoint(path)
    var_map = model.get_base_network_checkpoint_vars()
    import torch
    restored, missing = [], []
    for name, target in var_map.items():
        if name not in values:
            missing.append(name)
            continue
        arr = values[name]
        if tuple(arr.shape) != tuple(target.shape):
            raise CheckpointError('%s: checkpoint shape %s, variable shape %s' % (name, arr.shape, tuple(target.shape)))
        target.copy_(torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(target.device))
        restored.append(name)
    # derived state: BatchNorm scale/shift tables of the frozen statistics, cached L2 value of the frozen variables
    if restored and hasattr(model, 'load_state_dict') and hasattr(model, 'state_dict'):
        model.load_state_dict(model.state_dict())
    if strict and missing:
        raise CheckpointError('checkpoint %s lacks %d base-network variables, e.g. %s' % (path, len(missing), missing[:3]))
    return restored
