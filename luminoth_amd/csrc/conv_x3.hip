// Host side of the software-pipelined bf16x3 convolution kernels (conv_x3.h): launch functions called by the dispatch in
// conv.hip / conv_winograd.h.  A translation unit of its own (the kernels are large, fully unrolled instruction streams).
#define lmh_zero_page lmh_zero_page_x3      // (no relocatable device code: every translation unit owns its padding page)
#include "conv_common.h"
__device__ __attribute__((aligned(64))) float lmh_zero_page[16];
#include "conv_x3.h"
#include "conv_x3_api.h"
int lmh_opt(const char* name);

// gbatch > 0: `gbatch` stacked problems (k_x3_fwd<..., true>)
int lmh_x3_fwd_launch(const lmh_conv_desc* d, const float* x, const float* w, const float* scale, const float* shift,
                      const float* residual, float* y, uint32_t* act_bits, int gbatch, int bm, int bn, int pipe, hipStream_t st) {
  const int sg = lmh_opt("x3_stagger");
  const lmh_fastdiv dvw = lmh_make_fastdiv((uint32_t)d->OW), dvh = lmh_make_fastdiv((uint32_t)d->OH);
  const int64_t M = (int64_t)d->N * d->OH * d->OW;
  const int grid = (int)(((M + bm - 1) / bm) * ((d->K + bn - 1) / bn)) * (gbatch > 0 ? gbatch : 1);
#define X3_FWD(BM_, BN_)                                                                                              \
  do {                                                                                                                \
    if (gbatch > 0 && pipe) lmh_launch((k_x3_fwd<BM_, BN_, true, 1>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, \
                                       shift, residual, y, gbatch, act_bits, sg, dvw, dvh);                                         \
    else if (gbatch > 0) lmh_launch((k_x3_fwd<BM_, BN_, true, 0>), dim3(grid), dim3(256), 0, st, *d, x, w, scale,    \
                                    shift, residual, y, gbatch, act_bits, sg, dvw, dvh);                                            \
    else if (pipe) lmh_launch((k_x3_fwd<BM_, BN_, false, 1>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, shift,  \
                              residual, y, 1, act_bits, sg, dvw, dvh);                                                              \
    else lmh_launch((k_x3_fwd<BM_, BN_, false, 0>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, shift, residual,  \
                    y, 1, act_bits, sg, dvw, dvh);                                                                                  \
  } while (0)
  if (bm == 128 && bn == 128) X3_FWD(128, 128);
  else if (bm == 128 && bn == 64) X3_FWD(128, 64);
  else if (bm == 64 && bn == 64) X3_FWD(64, 64);
  else { lmh_set_error("lmh_x3_fwd_launch: no %d x %d tile", bm, bn); return LMH_ERR_INVALID; }
#undef X3_FWD
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

int lmh_x3_bwd_data_launch(const lmh_conv_desc* d, const float* dy, const float* w, const float* kscale,
                           const float* addend, const uint32_t* xbits, float* dx, int bm, int bn, int pipe, hipStream_t st) {
  const int sg = lmh_opt("x3_stagger");
  const lmh_fastdiv dvw = lmh_make_fastdiv((uint32_t)d->W), dvh = lmh_make_fastdiv((uint32_t)d->H);
  const int64_t M = (int64_t)d->N * d->H * d->W;
  const int grid = (int)(((M + bm - 1) / bm) * ((d->C + bn - 1) / bn));
#define X3_BD(BM_, BN_)                                                                                               \
  do {                                                                                                                \
    if (pipe) lmh_launch((k_x3_bwd_data<BM_, BN_, 1>), dim3(grid), dim3(256), 0, st, *d, dy, w, kscale, addend, xbits, dx, sg, dvw, dvh); \
    else lmh_launch((k_x3_bwd_data<BM_, BN_, 0>), dim3(grid), dim3(256), 0, st, *d, dy, w, kscale, addend, xbits, dx, sg, dvw, dvh); \
  } while (0)
  if (bm == 128 && bn == 128) X3_BD(128, 128);
  else if (bm == 128 && bn == 64) X3_BD(128, 64);
  else if (bm == 64 && bn == 64) X3_BD(64, 64);
  else { lmh_set_error("lmh_x3_bwd_data_launch: no %d x %d tile", bm, bn); return LMH_ERR_INVALID; }
#undef X3_BD
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// source pixel == output pixel for every tap of the launch: the PLAIN instantiation of k_x3_bwd_weight (x3_wg_plain = 0: the general
// decode); also what the profile name of the launch says (conv.hip)
bool lmh_x3_bwd_weight_plain(const lmh_conv_desc* d, bool gb, int pipe) {
  return !pipe && lmh_opt("x3_wg_plain") != 0 && d->stride == 1 && d->pad_top == 0 && d->pad_left == 0 && d->OH == d->H &&
         d->OW == d->W && (gb || d->R * d->S == 1);
}

int lmh_x3_bwd_weight_launch(const lmh_conv_desc* d, const float* x, const float* g, float* out, int kt_per_split,
                             int tiles_x, int tiles_y, int splits, float* colpart, bool gb, int bm, int bn,
                             int pipe, hipStream_t st) {
  const lmh_fastdiv dvw = lmh_make_fastdiv((uint32_t)d->OW), dvh = lmh_make_fastdiv((uint32_t)d->OH);
  const int nblk = tiles_x * tiles_y * splits;
  const int sg = lmh_opt("x3_stagger");
  const bool plain = lmh_x3_bwd_weight_plain(d, gb, pipe);
#define X3_BW(BM_, BN_)                                                                                               \
  do {                                                                                                                \
    if (gb && pipe) lmh_launch((k_x3_bwd_weight<BM_, BN_, true, 1>), dim3(nblk), dim3(256), 0, st, *d, x, g, out,     \
                               kt_per_split, dvw, dvh, tiles_x, tiles_y, splits, (float*)nullptr, sg);                    \
    else if (gb && plain) lmh_launch((k_x3_bwd_weight<BM_, BN_, true, 0, true>), dim3(nblk), dim3(256), 0, st, *d, x, \
                                     g, out, kt_per_split, dvw, dvh, tiles_x, tiles_y, splits, (float*)nullptr, sg);      \
    else if (gb) lmh_launch((k_x3_bwd_weight<BM_, BN_, true, 0>), dim3(nblk), dim3(256), 0, st, *d, x, g, out,        \
                            kt_per_split, dvw, dvh, tiles_x, tiles_y, splits, (float*)nullptr, sg);                       \
    else if (pipe) lmh_launch((k_x3_bwd_weight<BM_, BN_, false, 1>), dim3(nblk), dim3(256), 0, st, *d, x, g, out,     \
                              kt_per_split, dvw, dvh, tiles_x, tiles_y, splits, colpart, sg);                             \
    else if (plain) lmh_launch((k_x3_bwd_weight<BM_, BN_, false, 0, true>), dim3(nblk), dim3(256), 0, st, *d, x, g,   \
                               out, kt_per_split, dvw, dvh, tiles_x, tiles_y, splits, colpart, sg);                       \
    else lmh_launch((k_x3_bwd_weight<BM_, BN_, false, 0>), dim3(nblk), dim3(256), 0, st, *d, x, g, out, kt_per_split, \
                    dvw, dvh, tiles_x, tiles_y, splits, colpart, sg);                                                     \
  } while (0)
  if (bm == 128 && bn == 128) X3_BW(128, 128);
  else if (bm == 128 && bn == 64) X3_BW(128, 64);
  else if (bm == 64 && bn == 128) X3_BW(64, 128);
  else if (bm == 64 && bn == 64) X3_BW(64, 64);
  else { lmh_set_error("lmh_x3_bwd_weight_launch: no %d x %d tile", bm, bn); return LMH_ERR_INVALID; }
#undef X3_BW
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// ---- pre-split weights (conv_x3.h: k_x3_split_w / k_x3_fwd_ws) ---------------------------------------------------------
size_t lmh_x3_w3_bytes(int rs, int c, int k, int fwd) {
  const int ncols = fwd ? k : c;
  const size_t stages = fwd ? (size_t)rs * c / BK : (size_t)rs * (k / BK);
  return stages * x3_w3_stage(ncols) * sizeof(uint4);
}

int lmh_x3_split_launch(const float* const* w, void* const* out, const int* rs, const int* c, const int* k, int n, int fwd,
                        hipStream_t st) {
  for (int j0 = 0; j0 < n; j0 += X3_SPLIT_MAX) {
    x3_split_batch b;
    memset(&b, 0, sizeof(b));
    b.n = (n - j0) < X3_SPLIT_MAX ? (n - j0) : X3_SPLIT_MAX;
    b.fwd = fwd;
    int blocks = 0;
    for (int j = 0; j < b.n; ++j) {
      const int i = j0 + j;
      if ((c[i] % 32) != 0 || (k[i] % 32) != 0) { lmh_set_error("lmh_x3_split_weights: C %% 32 == 0 and K %% 32 == 0 needed"); return LMH_ERR_UNSUPPORTED; }
      b.w[j] = w[i]; b.out[j] = reinterpret_cast<uint4*>(out[i]);
      b.RS[j] = rs[i]; b.C[j] = c[i]; b.K[j] = k[i];
      b.first_block[j] = blocks;
      const size_t threads = lmh_x3_w3_bytes(rs[i], c[i], k[i], fwd) / sizeof(uint4) / 3;
      blocks += (int)((threads + 255) / 256);
    }
    b.first_block[b.n] = blocks;
    if (blocks > 0) lmh_launch(k_x3_split_w, dim3(blocks), dim3(256), 0, st, b);
  }
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

int lmh_x3_fwd_ws_launch(const lmh_conv_desc* d, const float* x, const void* w3, const float* scale, const float* shift,
                         const float* residual, float* y, uint32_t* act_bits, int gbatch, int bm, int bn, hipStream_t st) {
  const lmh_fastdiv dvw = lmh_make_fastdiv((uint32_t)d->OW), dvh = lmh_make_fastdiv((uint32_t)d->OH);
  const int64_t M = (int64_t)d->N * d->OH * d->OW;
  const int grid = (int)(((M + bm - 1) / bm) * ((d->K + bn - 1) / bn)) * (gbatch > 0 ? gbatch : 1);
  const uint4* w3p = reinterpret_cast<const uint4*>(w3);
#define X3_FWS(BM_, BN_)                                                                                              \
  do {                                                                                                                \
    if (gbatch > 0) lmh_launch((k_x3_fwd_ws<BM_, BN_, true>), dim3(grid), dim3(256), 0, st, *d, x, w3p, scale, shift, \
                               residual, y, gbatch, act_bits, 0, dvw, dvh);                                                      \
    else lmh_launch((k_x3_fwd_ws<BM_, BN_, false>), dim3(grid), dim3(256), 0, st, *d, x, w3p, scale, shift, residual, \
                    y, 1, act_bits, 0, dvw, dvh);                                                                                \
  } while (0)
  if (bm == 128 && bn == 128) X3_FWS(128, 128);
  else if (bm == 128 && bn == 64) X3_FWS(128, 64);
  else if (bm == 64 && bn == 64) X3_FWS(64, 64);
  else { lmh_set_error("lmh_x3_fwd_ws_launch: no %d x %d tile", bm, bn); return LMH_ERR_INVALID; }
#undef X3_FWS
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

int lmh_x3_bwd_data_ws_launch(const lmh_conv_desc* d, const float* dy, const void* w3, const float* kscale,
                              const float* addend, const uint32_t* xbits, float* dx, int bm, int bn, hipStream_t st) {
  const lmh_fastdiv dvw = lmh_make_fastdiv((uint32_t)d->W), dvh = lmh_make_fastdiv((uint32_t)d->H);
  const int64_t M = (int64_t)d->N * d->H * d->W;
  const int grid = (int)(((M + bm - 1) / bm) * ((d->C + bn - 1) / bn));
  const uint4* w3p = reinterpret_cast<const uint4*>(w3);
#define X3_BDS(BM_, BN_) lmh_launch((k_x3_bwd_data_ws<BM_, BN_>), dim3(grid), dim3(256), 0, st, *d, dy, w3p, kscale, addend, xbits, dx, 0, dvw, dvh)
  if (bm == 128 && bn == 128) X3_BDS(128, 128);
  else if (bm == 128 && bn == 64) X3_BDS(128, 64);
  else if (bm == 64 && bn == 64) X3_BDS(64, 64);
  else { lmh_set_error("lmh_x3_bwd_data_ws_launch: no %d x %d tile", bm, bn); return LMH_ERR_INVALID; }
#undef X3_BDS
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

