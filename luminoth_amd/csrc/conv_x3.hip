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
  const int64_t M = (int64_t)d->N * d->OH * d->OW;
  const int grid = (int)(((M + bm - 1) / bm) * ((d->K + bn - 1) / bn)) * (gbatch > 0 ? gbatch : 1);
#define X3_FWD(BM_, BN_)                                                                                              \
  do {                                                                                                                \
    if (gbatch > 0 && pipe) lmh_launch((k_x3_fwd<BM_, BN_, true, 1>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, \
                                       shift, residual, y, gbatch, act_bits, sg);                                         \
    else if (gbatch > 0) lmh_launch((k_x3_fwd<BM_, BN_, true, 0>), dim3(grid), dim3(256), 0, st, *d, x, w, scale,    \
                                    shift, residual, y, gbatch, act_bits, sg);                                            \
    else if (pipe) lmh_launch((k_x3_fwd<BM_, BN_, false, 1>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, shift,  \
                              residual, y, 1, act_bits, sg);                                                              \
    else lmh_launch((k_x3_fwd<BM_, BN_, false, 0>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, shift, residual,  \
                    y, 1, act_bits, sg);                                                                                  \
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
  const int64_t M = (int64_t)d->N * d->H * d->W;
  const int grid = (int)(((M + bm - 1) / bm) * ((d->C + bn - 1) / bn));
#define X3_BD(BM_, BN_)                                                                                               \
  do {                                                                                                                \
    if (pipe) lmh_launch((k_x3_bwd_data<BM_, BN_, 1>), dim3(grid), dim3(256), 0, st, *d, dy, w, kscale, addend, xbits, dx, sg); \
    else lmh_launch((k_x3_bwd_data<BM_, BN_, 0>), dim3(grid), dim3(256), 0, st, *d, dy, w, kscale, addend, xbits, dx, sg); \
  } while (0)
  if (bm == 128 && bn == 128) X3_BD(128, 128);
  else if (bm == 128 && bn == 64) X3_BD(128, 64);
  else if (bm == 64 && bn == 64) X3_BD(64, 64);
  else { lmh_set_error("lmh_x3_bwd_data_launch: no %d x %d tile", bm, bn); return LMH_ERR_INVALID; }
#undef X3_BD
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

int lmh_x3_bwd_weight_launch(const lmh_conv_desc* d, const float* x, const float* g, float* out, int kt_per_split,
                             int tiles_x, int tiles_y, int splits, float* colpart, bool gb, int bm, int bn,
                             int pipe, hipStream_t st) {
  const lmh_fastdiv dvw = lmh_make_fastdiv((uint32_t)d->OW), dvh = lmh_make_fastdiv((uint32_t)d->OH);
  const int nblk = tiles_x * tiles_y * splits;
  const int sg = lmh_opt("x3_stagger");
#define X3_BW(BM_, BN_)                                                                                               \
  do {                                                                                                                \
    if (gb && pipe) lmh_launch((k_x3_bwd_weight<BM_, BN_, true, 1>), dim3(nblk), dim3(256), 0, st, *d, x, g, out,     \
                               kt_per_split, dvw, dvh, tiles_x, tiles_y, splits, (float*)nullptr, sg);                    \
    else if (gb) lmh_launch((k_x3_bwd_weight<BM_, BN_, true, 0>), dim3(nblk), dim3(256), 0, st, *d, x, g, out,        \
                            kt_per_split, dvw, dvh, tiles_x, tiles_y, splits, (float*)nullptr, sg);                       \
    else if (pipe) lmh_launch((k_x3_bwd_weight<BM_, BN_, false, 1>), dim3(nblk), dim3(256), 0, st, *d, x, g, out,     \
                              kt_per_split, dvw, dvh, tiles_x, tiles_y, splits, colpart, sg);                             \
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
