"""CPU tests of the on-disk dataset format row: CRC-32C known answers (RFC 3720 B.4), TFRecord
framing / corruption detection through the C library, the SequenceExample wire codec against the `protobuf` runtime
built from the published tensorflow/core/example schema, and the augmentation plumbing against the reference's own
dataset tests (luminoth/datasets/object_detection_dataset_test.py:51-88, luminoth/utils/image_test.py:278-353)."""
import struct

import numpy as np
import pytest

from luminoth_amd.datasets import tfrecord as T
from oracle import image as oi

# RFC 3720 appendix B.4 test vectors for CRC-32C
CRC_VECTORS = [(b'\x00' * 32, 0x8A9136AA), (b'\xff' * 32, 0x62A8AB43), (bytes(range(32)), 0x46DD794E),
               (bytes(range(31, -1, -1)), 0x113FDB5C), (b'123456789', 0xE3069283), (b'', 0x00000000)]


@pytest.mark.parametrize('data,want', CRC_VECTORS)
def test_crc32c_known_answers_both_paths(data, want):
    lib = T.io_lib()
    assert T.crc32c(data) == want
    a = np.frombuffer(data, np.uint8) if data else np.empty(0, np.uint8)
    assert lib.lmh_io_crc32c_portable(a.ctypes.data, a.size) == want


def test_crc32c_paths_agree_on_unaligned_random_buffers():
    lib = T.io_lib()
    rs = np.random.RandomState(0)
    big = rs.randint(0, 256, size=70001).astype(np.uint8)
    for off, n in [(0, 70001), (1, 7), (3, 64), (5, 4099), (7, 1), (2, 0), (9, 65521)]:
        v = big[off:off + n]
        assert lib.lmh_io_crc32c(v.ctypes.data, n) == lib.lmh_io_crc32c_portable(v.ctypes.data, n)


def test_masked_crc_and_framing_layout():
    payload = b'hello tfrecord'
    c = T.crc32c(payload)
    assert T.masked_crc32c(payload) == (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF
    rec = T.frame_record(payload)
    assert len(rec) == 16 + len(payload)
    length, lcrc = struct.unpack('<QI', rec[:12])
    assert length == len(payload) and lcrc == T.masked_crc32c(rec[:8])
    assert rec[12:12 + length] == payload
    assert struct.unpack('<I', rec[12 + length:])[0] == T.masked_crc32c(payload)


def test_index_and_corruption_detection(tmp_path):
    rs = np.random.RandomState(1)
    payloads = [rs.bytes(n) for n in (0, 1, 17, 4096, 100000)] + [b'x']
    path = str(tmp_path / 'a.tfrecords')
    T.write_records(path, payloads)
    f = T.TFRecordFile(path)
    assert len(f) == len(payloads) and [f[i] for i in range(len(f))] == payloads
    f.close()
    blob = bytearray(open(path, 'rb').read())
    offs, lens = T.index_records(bytes(blob))
    assert lens.tolist() == [len(p) for p in payloads]
    bad = bytearray(blob)
    bad[int(offs[3]) + 100] ^= 0x40                                   # payload bit flip
    with pytest.raises(T.DataLossError, match='corrupted record data'):
        T.index_records(bytes(bad))
    assert T.index_records(bytes(bad), verify=False)[1].tolist() == lens.tolist()
    bad = bytearray(blob)
    bad[int(offs[2]) - 12] ^= 0x01                                    # length header bit flip
    with pytest.raises(T.DataLossError, match='corrupted record length'):
        T.index_records(bytes(bad))
    for cut in (5, 13, len(blob) - 1, len(blob) - 3):
        with pytest.raises(T.DataLossError, match='truncated'):
            T.index_records(bytes(blob[:cut]))
    assert T.index_records(b'')[0].shape == (0,)
    empty = str(tmp_path / 'empty.tfrecords')
    open(empty, 'wb').close()
    assert len(T.TFRecordFile(empty)) == 0


def _schema():
    """tensorflow/core/example/{feature,example}.proto (public schema) built at run time for the protobuf runtime."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name='lmh_example.proto', package='tensorflow', syntax='proto3')
    T_ = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m

    def field(m, name, num, typ, label=T_.LABEL_OPTIONAL, type_name=None, packed=None, oneof=None):
        f = m.field.add(name=name, number=num, type=typ, label=label)
        if type_name:
            f.type_name = type_name
        if packed is not None:
            f.options.packed = packed
        if oneof is not None:
            f.oneof_index = oneof
        return f

    def map_field(m, name, num, value_type):
        e = m.nested_type.add(name=''.join(p.capitalize() for p in name.split('_')) + 'Entry')
        e.options.map_entry = True
        field(e, 'key', 1, T_.TYPE_STRING)
        field(e, 'value', 2, T_.TYPE_MESSAGE, type_name=value_type)
        field(m, name, num, T_.TYPE_MESSAGE, T_.LABEL_REPEATED, '.tensorflow.%s.%s' % (m.name, e.name))

    field(msg('BytesList'), 'value', 1, T_.TYPE_BYTES, T_.LABEL_REPEATED)
    field(msg('FloatList'), 'value', 1, T_.TYPE_FLOAT, T_.LABEL_REPEATED, packed=True)
    field(msg('Int64List'), 'value', 1, T_.TYPE_INT64, T_.LABEL_REPEATED, packed=True)
    feat = msg('Feature')
    feat.oneof_decl.add(name='kind')
    field(feat, 'bytes_list', 1, T_.TYPE_MESSAGE, type_name='.tensorflow.BytesList', oneof=0)
    field(feat, 'float_list', 2, T_.TYPE_MESSAGE, type_name='.tensorflow.FloatList', oneof=0)
    field(feat, 'int64_list', 3, T_.TYPE_MESSAGE, type_name='.tensorflow.Int64List', oneof=0)
    map_field(msg('Features'), 'feature', 1, '.tensorflow.Feature')
    field(msg('FeatureList'), 'feature', 1, T_.TYPE_MESSAGE, T_.LABEL_REPEATED, '.tensorflow.Feature')
    map_field(msg('FeatureLists'), 'feature_list', 1, '.tensorflow.FeatureList')
    se = msg('SequenceExample')
    field(se, 'context', 1, T_.TYPE_MESSAGE, type_name='.tensorflow.Features')
    field(se, 'feature_lists', 2, T_.TYPE_MESSAGE, type_name='.tensorflow.FeatureLists')
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName('tensorflow.SequenceExample'))


BOXES = [dict(label=3, xmin=1, ymin=2, xmax=300, ymax=40), dict(label=0, xmin=5, ymin=6, xmax=7, ymax=8),
         dict(label=19, xmin=0, ymin=0, xmax=1023, ymax=767)]


def test_decoder_reads_what_the_protobuf_runtime_writes():
    SE = _schema()
    se = SE()
    se.context.feature['width'].int64_list.value.append(1024)
    se.context.feature['height'].int64_list.value.append(768)
    se.context.feature['depth'].int64_list.value.append(3)
    se.context.feature['filename'].bytes_list.value.append(b'img_0001.jpg')
    se.context.feature['image_raw'].bytes_list.value.append(bytes(range(256)) * 9)
    for b in BOXES:                                  # object_detection_writer.py:140-158: one Feature per box
        for k, v in b.items():
            se.feature_lists.feature_list[k].feature.add().int64_list.value.append(v)
    se.feature_lists.feature_list['neg'].feature.add().int64_list.value.extend([-1, -(1 << 40), 1 << 62])
    se.context.feature['floats'].float_list.value.extend([0.5, -2.25])
    data = se.SerializeToString()
    ctx, lists = T.decode_sequence_example(data)
    assert ctx['width'] == [1024] and ctx['filename'] == [b'img_0001.jpg'] and ctx['floats'] == [0.5, -2.25]
    assert ctx['image_raw'] == [bytes(range(256)) * 9]
    assert lists['neg'] == [[-1, -(1 << 40), 1 << 62]]
    rec = T.decode_detection_record(data)
    assert rec['filename'] == 'img_0001.jpg' and (rec['width'], rec['height'], rec['depth']) == (1024, 768, 3)
    np.testing.assert_array_equal(rec['bboxes'], [[b['xmin'], b['ymin'], b['xmax'], b['ymax'], b['label']] for b in BOXES])


def test_protobuf_runtime_reads_what_the_encoder_writes():
    SE = _schema()
    data = T.encode_detection_record(b'\x89PNG....', 'a/b.png', 640, 480, BOXES)
    se = SE()
    se.ParseFromString(data)
    assert se.context.feature['width'].int64_list.value[:] == [640]
    assert se.context.feature['height'].int64_list.value[:] == [480]
    assert se.context.feature['filename'].bytes_list.value[:] == [b'a/b.png']
    assert se.context.feature['image_raw'].bytes_list.value[:] == [b'\x89PNG....']
    for k in ('label', 'xmin', 'ymin', 'xmax', 'ymax'):
        got = [f.int64_list.value[0] for f in se.feature_lists.feature_list[k].feature]
        assert got == [b[k] for b in BOXES]
    # canonical re-serialisation parses back identically through the decoder
    assert T.decode_detection_record(se.SerializeToString())['bboxes'].tolist() == \
        T.decode_detection_record(data)['bboxes'].tolist()
    empty = T.decode_detection_record(T.encode_detection_record(b'', 'e', 1, 1, []))
    assert empty['bboxes'].shape == (0, 5)


# ------------------------------------------------------------------------- augmentation ----
def _dataset(augment, seed=None):
    from luminoth_amd.datasets.object_detection_dataset import ObjectDetectionDataset
    from luminoth_amd.utils.config import Config as EasyDict
    cfg = EasyDict({'dataset': {'dir': '', 'split': 'train', 'image_preprocessing': {'min_size': 600, 'max_size': 1024},
                                'data_augmentation': augment},
                    'train': {'num_epochs': 1, 'batch_size': 1, 'random_shuffle': False, 'seed': seed}})
    return ObjectDetectionDataset(cfg)


def test_sorted_augmentation():
    """object_detection_dataset_test.py:51-70."""
    image = np.random.RandomState(0).randint(0, 255, size=(600, 800, 3))
    bboxes = np.array([[10, 10, 26, 28, 1], [10, 10, 20, 22, 1], [10, 11, 20, 21, 1], [19, 30, 31, 33, 1]])
    _, _, aug = _dataset([{'flip': {'prob': 0}}, {'flip': {'prob': 1}}])._augment(image, bboxes)
    assert aug == [{'flip': False}, {'flip': True}]
    _, _, aug = _dataset([{'flip': {'prob': 1}}, {'flip': {'prob': 0}}])._augment(image, bboxes)
    assert aug == [{'flip': True}, {'flip': False}]


def test_identity_augmentation():
    """object_detection_dataset_test.py:72-88: flipping twice returns the image and the boxes."""
    image = np.random.RandomState(1).randint(0, 255, size=(600, 800, 3))
    bboxes = np.array([[10, 10, 26, 28, 1], [19, 30, 31, 33, 1]])
    ds = _dataset([{'flip': {'prob': 1}}, {'flip': {'prob': 1}}])
    image_aug, bboxes_aug, aug = ds._augment(image, bboxes)
    assert aug == [{'flip': True}, {'flip': True}]
    np.testing.assert_array_equal(image, image_aug)
    np.testing.assert_array_equal(bboxes, bboxes_aug)
    lr, ud, folded = ds._fold_flips(ds._augment_decide(), bboxes, 600, 800)
    assert (lr, ud) == (False, False)
    np.testing.assert_array_equal(folded, bboxes)


def test_flip_reference_cases_host_and_oracle():
    """image_test.py:278-353."""
    from luminoth_amd.utils.image import flip_image
    rs = np.random.RandomState(2)
    image = rs.randint(0, 255, size=(100, 100, 3))
    for fn in (flip_image, oi.flip_image):
        out = fn(image, left_right=False)
        np.testing.assert_array_equal(out['image'], image)
        out = fn(image, left_right=True)
        np.testing.assert_array_equal(out['image'][:, 0], image[:, -1])
        np.testing.assert_array_equal(out['image'][:, 1], image[:, -2])
        full = np.array([[0, 0, 99, 99, -1]])
        for kw in (dict(left_right=True), dict(left_right=False, up_down=True), dict(left_right=True, up_down=True)):
            np.testing.assert_array_equal(fn(image, full, **kw)['bboxes'], full)
        out = fn(image, np.array([[0, 0, 10, 10, -1]]), left_right=True, up_down=True)
        np.testing.assert_array_equal(out['bboxes'], [[89, 89, 99, 99, -1]])
        a = fn(image, np.array([[25, 14, 63, 41, -1]], np.float32), left_right=True, up_down=True)['bboxes']
        b = fn(image, np.array([[25, 14, 63, 41, -1]], np.int32), left_right=True, up_down=True)['bboxes']
        np.testing.assert_array_equal(a, b)
    boxes = np.stack([rs.randint(0, 40, 10), rs.randint(0, 40, 10), rs.randint(50, 100, 10), rs.randint(50, 100, 10),
                      rs.randint(0, 5, 10)], 1)
    for kw in (dict(left_right=True), dict(left_right=False, up_down=True), dict(left_right=True, up_down=True)):
        np.testing.assert_array_equal(flip_image(image, boxes, **kw)['bboxes'], oi.flip_image(image, boxes, **kw)['bboxes'])
        np.testing.assert_array_equal(flip_image(image, boxes, **kw)['image'], oi.flip_image(image, boxes, **kw)['image'])


def test_invalid_strategies_are_skipped():
    """object_detection_dataset.py:163-175: an unknown strategy is ignored with a warning; the known ones run in order
    (a distortion with no sub-option configured changes nothing but the dtype)."""
    ds = _dataset([{'distortion': {'prob': 1.0}}, {'bogus': {}}, {'flip': {'prob': 1.0, 'left_right': False, 'up_down': True}}])
    image = np.arange(5 * 4 * 3).reshape(5, 4, 3)
    out, _, aug = ds._augment(image, None)
    assert aug == [{'distortion': True}, {'flip': True}]
    np.testing.assert_array_equal(out, image[::-1])
    with pytest.raises(ValueError):
        _dataset([{'flip': {}, 'patch': {}}])._augment(image, None)


def test_missing_split_raises():
    from luminoth_amd.datasets.object_detection_dataset import InvalidDataDirectory
    ds = _dataset([])
    ds._dataset_dir = '/nonexistent_dir_for_test'
    with pytest.raises(InvalidDataDirectory):
        iter(ds).__next__()
