/* luminoth_io.h — host-side C ABI of the dataset reader: the TFRecord container that
 * `lumi dataset transform` writes (luminoth/tools/dataset/writers/object_detection_writer.py:61-98, through
 * tf.python_io.TFRecordWriter) and `tf.TFRecordReader` reads (luminoth/datasets/base_dataset.py:46-47).
 *
 * The container is third party (TensorFlow core/lib/io/record_{reader,writer}.cc, core/lib/hash/crc32c.h):
 *   record := uint64le length | uint32le masked_crc32c(length bytes) | data[length] | uint32le masked_crc32c(data)
 *   masked(c) := ((c >> 15) | (c << 17)) + 0xa282ead8          (mod 2^32)
 * CRC-32C (Castagnoli, reflected 0x82F63B78, init/xorout 0xFFFFFFFF) pinned by RFC 3720 B.4 known answers
 * (tests/test_tfrecord.py).  Plain pointers and sizes; no allocation; thread-safe. */
#ifndef LUMINOTH_IO_H_
#define LUMINOTH_IO_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LMH_IO_ERR_TRUNCATED (-1)  /* file ends inside a record */
#define LMH_IO_ERR_LENGTH_CRC (-2) /* corrupted length header */
#define LMH_IO_ERR_DATA_CRC (-3)   /* corrupted payload */

uint32_t lmh_io_crc32c(const void* data, size_t n);
uint32_t lmh_io_masked_crc32c(const void* data, size_t n);
/* The table (slicing-by-8) path regardless of CPU support — exported so both paths can be pinned. */
uint32_t lmh_io_crc32c_portable(const void* data, size_t n);
/* 1 when the SSE4.2 crc32 instruction path is in use, 0 for the slicing-by-8 tables. */
int lmh_io_crc32c_hw(void);

/* Scans a whole .tfrecords file image.  Writes up to `capacity` (payload offset, payload length) pairs and
 * returns the total number of records (call again with a larger capacity if it exceeds it), or a negative
 * LMH_IO_ERR_* with *err_offset = byte offset of the offending record.  verify: 0 = length CRCs only,
 * 1 = payload CRCs too (what tf.TFRecordReader does). */
int64_t lmh_io_tfrecord_index(const void* buf, size_t n, int verify, uint64_t* offsets, uint64_t* lengths,
                              size_t capacity, uint64_t* err_offset);

/* Frames one payload: writes 16 + n bytes to out (caller-sized) and returns that count. */
size_t lmh_io_tfrecord_frame(const void* data, uint64_t n, void* out);

#ifdef __cplusplus
}
#endif
#endif
