/* Host-side dataset IO (include/luminoth_io.h): CRC-32C and TFRecord framing.  Plain C, no dependencies. */
#include "../../include/luminoth_io.h"

#include <string.h>

#if defined(__x86_64__)
#include <nmmintrin.h>
#endif

static uint32_t g_tab[8][256];
static int g_tab_ready = 0;

static void build_tables(void) {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
    g_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_tab[t][i] = (g_tab[t - 1][i] >> 8) ^ g_tab[0][g_tab[t - 1][i] & 0xFFu];
  __atomic_store_n(&g_tab_ready, 1, __ATOMIC_RELEASE);
}

static uint32_t crc_sw(uint32_t c, const uint8_t* p, size_t n) {
  if (!__atomic_load_n(&g_tab_ready, __ATOMIC_ACQUIRE)) build_tables(); /* idempotent: racing builders agree */
  while (n && ((uintptr_t)p & 7u)) {
    c = (c >> 8) ^ g_tab[0][(c ^ *p++) & 0xFFu];
    --n;
  }
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= c;
    c = g_tab[7][w & 0xFF] ^ g_tab[6][(w >> 8) & 0xFF] ^ g_tab[5][(w >> 16) & 0xFF] ^ g_tab[4][(w >> 24) & 0xFF] ^
        g_tab[3][(w >> 32) & 0xFF] ^ g_tab[2][(w >> 40) & 0xFF] ^ g_tab[1][(w >> 48) & 0xFF] ^ g_tab[0][w >> 56];
    p += 8;
    n -= 8;
  }
  while (n--) c = (c >> 8) ^ g_tab[0][(c ^ *p++) & 0xFFu];
  return c;
}

#if defined(__x86_64__)
__attribute__((target("sse4.2"))) static uint32_t crc_hw(uint32_t c, const uint8_t* p, size_t n) {
  uint64_t c64 = c;
  while (n && ((uintptr_t)p & 7u)) {
    c64 = _mm_crc32_u8((uint32_t)c64, *p++);
    --n;
  }
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    c64 = _mm_crc32_u64(c64, w);
    p += 8;
    n -= 8;
  }
  while (n--) c64 = _mm_crc32_u8((uint32_t)c64, *p++);
  return (uint32_t)c64;
}
#endif

int lmh_io_crc32c_hw(void) {
#if defined(__x86_64__)
  return __builtin_cpu_supports("sse4.2") ? 1 : 0;
#else
  return 0;
#endif
}

uint32_t lmh_io_crc32c_portable(const void* data, size_t n) {
  return crc_sw(0xFFFFFFFFu, (const uint8_t*)data, n) ^ 0xFFFFFFFFu;
}

uint32_t lmh_io_crc32c(const void* data, size_t n) {
  const uint8_t* p = (const uint8_t*)data;
#if defined(__x86_64__)
  if (lmh_io_crc32c_hw()) return crc_hw(0xFFFFFFFFu, p, n) ^ 0xFFFFFFFFu;
#endif
  return crc_sw(0xFFFFFFFFu, p, n) ^ 0xFFFFFFFFu;
}

uint32_t lmh_io_masked_crc32c(const void* data, size_t n) {
  const uint32_t c = lmh_io_crc32c(data, n);
  return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}

static uint32_t rd32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }
static void wr32(uint8_t* p, uint32_t v) {
  p[0] = (uint8_t)v, p[1] = (uint8_t)(v >> 8), p[2] = (uint8_t)(v >> 16), p[3] = (uint8_t)(v >> 24);
}

int64_t lmh_io_tfrecord_index(const void* buf, size_t n, int verify, uint64_t* offsets, uint64_t* lengths,
                              size_t capacity, uint64_t* err_offset) {
  const uint8_t* b = (const uint8_t*)buf;
  size_t pos = 0;
  int64_t count = 0;
  while (pos < n) {
    if (err_offset) *err_offset = pos;
    if (n - pos < 12) return LMH_IO_ERR_TRUNCATED;
    const uint64_t len = rd64(b + pos);
    if (lmh_io_masked_crc32c(b + pos, 8) != rd32(b + pos + 8)) return LMH_IO_ERR_LENGTH_CRC;
    if (len > n - pos - 12 || n - pos - 12 - len < 4) return LMH_IO_ERR_TRUNCATED;
    const uint8_t* data = b + pos + 12;
    if (verify && lmh_io_masked_crc32c(data, (size_t)len) != rd32(data + len)) return LMH_IO_ERR_DATA_CRC;
    if ((size_t)count < capacity) {
      if (offsets) offsets[count] = pos + 12;
      if (lengths) lengths[count] = len;
    }
    ++count;
    pos += 12 + (size_t)len + 4;
  }
  return count;
}

size_t lmh_io_tfrecord_frame(const void* data, uint64_t n, void* out) {
  uint8_t* o = (uint8_t*)out;
  for (int i = 0; i < 8; ++i) o[i] = (uint8_t)(n >> (8 * i));
  wr32(o + 8, lmh_io_masked_crc32c(o, 8));
  memcpy(o + 12, data, (size_t)n);
  wr32(o + 12 + n, lmh_io_masked_crc32c(data, (size_t)n));
  return (size_t)n + 16;
}
