// NHWC fp32 implicit-GEMM convolution family on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Replaces the TF conv2d / slim conv2d_same (+ frozen BN, bias, ReLU/ReLU6,
// residual add) kernels the reference reaches through tf.contrib.slim and
// Sonnet: forward, backward-data and backward-weight.
//
// Design (MI355X): fp32 in / fp32 accumulate is the parity dtype (north_star:
// 1e-4 vs the TF-CPU fp32 path).  gfx950 has no xf32, so the matrix pipe runs
// at the fp32 rate: 32x32x2 = 64 cycles/issue/SIMD, 157 TFLOP/s chip peak.  One
// MFMA covers 64 cycles, so the kernels are matrix-pipe bound as long as the
// per-stage non-MFMA instruction stream stays short:
//   * 256-thread workgroups (4 waves, 2x2), block tiles 128x128 / 128x64 /
//     64x64 (picked so the grid is >= ~2 waves of the 256 CUs), BK = 32;
//   * the gather addresses are NOT recomputed per K-stage: every thread keeps
//     a pointer per tile row that advances by a constant (re-derived only when
//     the filter tap (r,s) changes); out-of-image rows point at a zero page
//     with stride 0, so the loads are branch-free;
//   * LDS is double buffered: global -> registers for stage t+1 is issued
//     before the MFMAs of stage t, written to the other buffer after them,
//     ONE barrier per stage;
//   * operands whose GEMM-K axis is contiguous in memory sit in LDS as
//     [row][BK+4] and are read with conflict-free ds_read_b128 (4 k-steps per
//     read, K order permuted identically for A and B); K-major operands sit as
//     [BK][cols] and are read with conflict-free ds_read_b32;
//   * blockIdx -> tile mapping is XCD-aware (each XCD walks a contiguous
//     range of tiles so operand panels stay in its private L2).
// Shapes that break the alignment rules (C % 32, K % 4 ...) go through the
// predicated kernels in conv_generic.h.
#include <stdarg.h>
#include <stdlib.h>
#include <atomic>
#include "conv_common.h"
#include "conv_generic.h"

__device__ __attribute__((aligned(64))) float lmh_zero_page[16];  // zero-initialised: padding source

#include "conv_fast.h"
#include "conv_pp.h"
#include "conv_wgrad1x1.h"
#include "conv_half.h"
#include "conv_x3_api.h"
#define X3W_MAX_JOBS 96

// ============================================================================
// host dispatch
// ============================================================================
static int check_desc(const lmh_conv_desc* d) {
  LMH_CHECK_ARG(d != nullptr);
  LMH_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0);
  LMH_CHECK_ARG(d->OH > 0 && d->OW > 0 && d->stride > 0 && d->dilation > 0);
  LMH_CHECK_ARG(d->act >= 0 && d->act <= 2);
  LMH_CHECK_ARG(d->compute >= 0 && d->compute <= 3);
  LMH_CHECK_ARG((int64_t)d->N * d->OH * d->OW < (1ll << 31) && (int64_t)d->N * d->H * d->W < (1ll << 31));
  return LMH_OK;
}

// ---- per-launch profiling hook (bench.py roofline leg) -------------------------------------------------------
// lmh_conv2d_profile_next(e0, e1) arms the NEXT convolution entry point called from this thread: the two HIP events
// are recorded on the launch stream immediately before / after its MFMA kernel (the implicit-GEMM kernel itself —
// not the split-K reduce or the Winograd transforms around it), and lmh_conv2d_profile_last() then returns that
// kernel's name as rocprofv3 prints it plus the FLOPs the launch executed.
static thread_local hipEvent_t g_prof_e0 = nullptr, g_prof_e1 = nullptr;
static thread_local char g_prof_name[96] = "";
static thread_local double g_prof_flops = 0.0, g_prof_bytes = 0.0;
extern "C" int lmh_conv2d_profile_next(void* ev_start, void* ev_stop) {
  g_prof_e0 = (hipEvent_t)ev_start;
  g_prof_e1 = (hipEvent_t)ev_stop;
  g_prof_name[0] = 0;
  g_prof_flops = 0.0;
  g_prof_bytes = 0.0;
  return LMH_OK;
}
// compulsory HBM bytes of the profiled launch: every operand tensor read once, the result written once (fp32)
extern "C" double lmh_conv2d_profile_last_bytes(void) { return g_prof_bytes; }
extern "C" const char* lmh_conv2d_profile_last(double* flops) {
  if (flops) *flops = g_prof_flops;
  return g_prof_name;
}
static thread_local double g_prof_pending_bytes = 0.0;      // set by the entry point before its launch
static inline void prof_begin(hipStream_t st) {
  if (g_prof_e0) (void)hipEventRecord(g_prof_e0, st);
}
static inline void prof_end(hipStream_t st, double flops, const char* fmt, ...) {
  if (!g_prof_e0) return;
  (void)hipEventRecord(g_prof_e1, st);
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_prof_name, sizeof(g_prof_name), fmt, ap);
  va_end(ap);
  g_prof_flops = flops;
  g_prof_bytes = g_prof_pending_bytes;
  g_prof_e0 = g_prof_e1 = nullptr;
}
static inline double desc_flops(const lmh_conv_desc* d) {
  return 2.0 * d->N * d->OH * d->OW * (double)d->K * d->R * d->S * d->C;
}
static inline double desc_bytes(const lmh_conv_desc* d) {
  return 4.0 * ((double)d->N * d->H * d->W * d->C + (double)d->R * d->S * d->C * d->K + (double)d->N * d->OH * d->OW * d->K);
}

// Tuning override (diagnostics only: scripts/bench_conv.py sweeps tile shapes / split counts with it).
// (experiment, conv_common.h) units of ~1 us by which co-resident blocks of the fast forward / backward-data kernels are staggered
extern "C" int lmh_conv_set_stagger(int units) {
#ifdef LMH_PROBES
  LMH_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stagger), &units, sizeof(int)));
  return LMH_OK;
#else
  if (units == 0) return LMH_OK;
  lmh_set_error("lmh_conv_set_stagger: probes are not compiled in (LMH_PROBES=1 bash build.sh)");
  return LMH_ERR_UNSUPPORTED;
#endif
}
static thread_local int g_force_bm = 0, g_force_bn = 0, g_force_splits = 0;      // (per thread, like the options)
extern "C" void lmh_conv2d_force_config(int bm, int bn, int splits) {
  g_force_bm = bm; g_force_bn = bn; g_force_splits = splits;
}

// Block tile: minimise (tile rounds over the 256 CUs) x (tile area), i.e. the matrix-pipe time of the
// busiest CU; ties go to the larger tile (less L2 traffic).  Fitted to scripts/sweep_conv.py on MI355X.
static void pick_tile(int64_t M, int64_t Ncols, int* bm, int* bn, int slots = 256) {
  if (g_force_bm && g_force_bn) { *bm = g_force_bm; *bn = g_force_bn; return; }
  static const int cand[3][2] = {{128, 128}, {128, 64}, {64, 64}};
  int64_t best = -1;
  for (int i = 0; i < 3; ++i) {
    const int64_t tiles = ((M + cand[i][0] - 1) / cand[i][0]) * ((Ncols + cand[i][1] - 1) / cand[i][1]);
    const int64_t cost = ((tiles + slots - 1) / slots) * cand[i][0] * cand[i][1];
    if (best < 0 || cost < best) { best = cost; *bm = cand[i][0]; *bn = cand[i][1]; }
  }
}
int lmh_opt(const char* name);   // api.hip: the lmh_set_option registry (this library reads no environment variable)

static bool stem_fast(const lmh_conv_desc* d) {
  return d->R == 7 && d->S == 7 && d->C == 3 && d->K == 64 && d->stride == 2 && d->dilation == 1;
}
static bool fwd_fast(const lmh_conv_desc* d) { return (d->C % BK) == 0 && (d->K & 3) == 0; }
static bool bwd_data_fast(const lmh_conv_desc* d) { return (d->K % BK) == 0 && (d->C & 3) == 0; }
// Stride-2 3x3 backward data walks the pixels parity class by parity class (k_conv_bwd_data): the classes carry
// 1, 2, 2 and 4 taps, so with one 128-row tile per CU the launch lasts as long as its 4-tap tiles.  64x64 tiles
// (several per CU, scheduled dynamically) even that out (block2 unit4: 91 us before tap skipping, 56.7 us with
// 128x128 tiles, 55.3 us with 64x64 — scripts/bench_conv.py "3x3/2").
static void bwd_data_parity_tile(const lmh_conv_desc* d, int64_t M, int* bm, int* bn) {
  const int on = lmh_opt("bd_parity_small");
  if (!on || !bwd_data_fast(d) || g_force_bm) return;
  const bool par = d->stride == 2 && d->dilation == 1 && d->R * d->S > 1 && d->R <= 3 && d->S <= 3 &&
                   !(d->H & 1) && !(d->W & 1) && ((M >> 2) % 64) == 0;
  if (par && (M / *bm) * ((d->C + *bn - 1) / *bn) <= 512) { *bm = 64; *bn = 64; }
}
static bool bwd_weight_fast(const lmh_conv_desc* d) { return (d->C & 3) == 0 && (d->K & 3) == 0; }
// Static loss scaling inside the half-precision backward kernels: the gradient operand is multiplied by 2^10 before
// it is rounded to f16 (5 exponent bits: activations gradients of 1e-7 would flush) and the fp32 accumulators by
// 2^-10 afterwards — exact in fp32.  bf16 has fp32's exponent range and needs none.
static float half_gscale(const lmh_conv_desc* d) { return d->compute == 1 ? 1024.f : 1.f; }
// prefetch depth of the half-precision kernels (register sets of staged tiles).  Measured on MI355X (COCO-shape R50
// step, f16): one set 6.82 ms/step, two sets 8.02 — the second set pushes the 128x128 kernels past 256 VGPRs (one wave
// per SIMD instead of two), which costs more than the deeper prefetch buys.  LMH_HALF_PF=2 keeps the variant reachable.
#define half_pf lmh_opt("half_pf")     // 1, 2: register sets; 3, 4: warp-specialised 512-thread blocks (one / two sets)
// bf16x3 pipeline per pass: 0 = one LDS buffer, two 256-thread blocks per CU; 1 / 2 = double-buffered LDS, one block per
// CU, one / two register sets of prefetched tiles; 3 / 4 = warp-specialised 512-thread block (waves 4-7 split and stage
// tile t+1 while waves 0-3 multiply tile t).  Measured per layer on MI355X (scripts/bench_conv.py, ResNet-50 shapes, sums
// over the 19 layers): forward 1.42 / 1.48 / 1.57 / 1.32 ms for 0 / 1 / 2 / 3 (native fp32 MFMA 1.59), backward data 1.28 /
// 1.36 / 1.42 / 1.45 (1.59), weight gradient 1.69 / 1.98 / 2.00 / 1.78 (1.95) — but INSIDE the train step, where the
// other streams fill a CU's second block slot, the two-blocks-per-CU pipeline wins everywhere: whole step 7.19 ms with
// 0 for every pass against 7.29 (forward 3), 7.46 (stacked Winograd GEMMs 3), 7.60 (forward + stacked 1).
// LMH_X3_PF forces one value for all passes.
#define x3_tile_pick lmh_opt("x3_tile_slots")   // tile of the bf16x3 fwd / bwd_data kernels by pick_tile(slots); 0: half_tile
// schedule of the round-6 bf16x3 kernels (conv_x3.h): 1 = software-pipelined (needs an even stage count), 0 = phase by phase
// x3_pipe: 0 = phase by phase; 1 = the software-pipelined variant wherever the stage count is even; N >= 2 = only for
// reductions of at least N stages (the RPN 3x3 convolution's transformed-domain GEMMs have 32)
static int x3_pipe(int stages) {
  const int o = lmh_opt("x3_pipe");
  if (o <= 0 || (stages & 1) != 0) return 0;
  return (o == 1 || stages >= o) ? 1 : 0;
}
static int x3_pf(const char* pass) { const int a = lmh_opt("x3_pf"); return a >= 0 ? a : lmh_opt(pass); }
#define x3_pf_fwd x3_pf("x3_pf_fwd")
#define x3_pf_gb x3_pf("x3_pf_gb")     // stacked Winograd GEMMs (forward kernel)
#define x3_pf_bd x3_pf("x3_pf_bd")
#define x3_pf_bw x3_pf("x3_pf_bw")
// Tile of the half-precision kernels: they are bound by the staging path (bytes per MFMA), not by matrix-pipe rounds, so
// the largest tile the problem fills wins (128x128 moves half the bytes per FLOP of 64x64) as long as the grid still
// covers the chip once.
static void half_tile(int64_t M, int64_t Ncols, int* bm, int* bn) {
  if (g_force_bm && g_force_bn) { *bm = g_force_bm; *bn = g_force_bn; return; }
  *bm = 128;
  *bn = Ncols > 64 ? 128 : 64;
  if (((M + 127) / 128) * ((Ncols + *bn - 1) / *bn) < 128) { *bm = 64; *bn = 64; }
}

int lmh_act_bits_impl(const float* y, int act, int64_t rows, int K, uint32_t* bits, hipStream_t st);   // elementwise.hip
int lmh_apply_act_bits_impl(float* dx, const uint32_t* bits, int64_t rows, int C, hipStream_t st);

static int conv2d_fwd_launch(const lmh_conv_desc* d, const float* x, const float* w, const float* scale,
                             const float* shift, const float* residual, const float* in_sub, float* y,
                             uint32_t* act_bits, bool* bits_done, lmh_stream_t stream);

extern "C" int lmh_conv2d_fwd(const lmh_conv_desc* d, const float* x, const float* w, const float* scale,
                              const float* shift, const float* residual, const float* in_sub, float* y,
                              uint32_t* act_bits, lmh_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  LMH_CHECK_ARG(act_bits == nullptr || (d->act != 0 && (d->K & 31) == 0));
  bool bits_done = false;
  rc = conv2d_fwd_launch(d, x, w, scale, shift, residual, in_sub, y, act_bits, &bits_done, stream);
  if (rc) return rc;
  if (act_bits && !bits_done)     // kernels without the fused epilogue (stem, predicated, half precision): one small pass
    return lmh_act_bits_impl(y, d->act, (int64_t)d->N * d->OH * d->OW, d->K, act_bits, (hipStream_t)stream);
  return LMH_OK;
}

// ---- bf16x3 with PRE-SPLIT weights (conv_x3.h, round 6): the weight matrix is split once per step into the MFMA fragment
// order (lmh_x3_split_weights_batch) and the forward kernel loads its B fragments straight from global memory
extern "C" size_t lmh_x3_weights_bytes(int rs, int c, int k, int backward) {
  if (rs <= 0 || c <= 0 || k <= 0 || (c % 32) != 0 || (k % 32) != 0) return 0;
  return lmh_x3_w3_bytes(rs, c, k, backward ? 0 : 1);
}
extern "C" int lmh_x3_split_weights_batch(const lmh_x3_weight_job* jobs, int n, int backward, lmh_stream_t stream) {
  LMH_CHECK_ARG(jobs && n > 0);
  const float* w[X3W_MAX_JOBS]; void* out[X3W_MAX_JOBS]; int rs[X3W_MAX_JOBS], c[X3W_MAX_JOBS], k[X3W_MAX_JOBS];
  for (int j0 = 0; j0 < n; j0 += X3W_MAX_JOBS) {
    const int m = (n - j0) < X3W_MAX_JOBS ? (n - j0) : X3W_MAX_JOBS;
    for (int j = 0; j < m; ++j) {
      const lmh_x3_weight_job& q = jobs[j0 + j];
      LMH_CHECK_ARG(q.w && q.out && q.rs > 0 && q.C > 0 && q.K > 0 && (((uintptr_t)q.out) & 15) == 0);
      w[j] = q.w; out[j] = q.out; rs[j] = q.rs; c[j] = q.C; k[j] = q.K;
    }
    const int rc = lmh_x3_split_launch(w, out, rs, c, k, m, backward ? 0 : 1, (hipStream_t)stream);
    if (rc) return rc;
  }
  return LMH_OK;
}
static bool x3w_fwd_ok(const lmh_conv_desc* d) {
  return d->compute == 3 && fwd_fast(d) && (d->K % 32) == 0 && !stem_fast(d);
}
extern "C" int lmh_conv2d_fwd_x3w_supported(const lmh_conv_desc* d) { return d && check_desc(d) == LMH_OK && x3w_fwd_ok(d) ? 1 : 0; }
extern "C" int lmh_conv2d_fwd_x3w(const lmh_conv_desc* d, const float* x, const void* w3, const float* scale,
                                  const float* shift, const float* residual, float* y, uint32_t* act_bits,
                                  lmh_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  LMH_CHECK_ARG(x && w3 && y && (((uintptr_t)w3) & 15) == 0);
  LMH_CHECK_ARG(act_bits == nullptr || (d->act != 0 && (d->K & 31) == 0));
  if (!x3w_fwd_ok(d)) { lmh_set_error("lmh_conv2d_fwd_x3w: needs compute bf16x3, C %% 32 == 0, K %% 32 == 0"); return LMH_ERR_UNSUPPORTED; }
  g_prof_pending_bytes = desc_bytes(d) + (residual ? 4.0 * d->N * d->OH * d->OW * (double)d->K : 0.0);
  const int64_t M = (int64_t)d->N * d->OH * d->OW;
  int bm, bn;
  if (x3_tile_pick) pick_tile(M, d->K, &bm, &bn, x3_tile_pick); else half_tile(M, d->K, &bm, &bn);
  hipStream_t st = (hipStream_t)stream;
  prof_begin(st);
  rc = lmh_x3_fwd_ws_launch(d, x, w3, scale, shift, residual, y, act_bits, 0, bm, bn, st);
  prof_end(st, desc_flops(d), "k_x3_fwd_ws<%d, %d, false>", bm, bn);
  return rc;
}

static int conv2d_fwd_launch(const lmh_conv_desc* d, const float* x, const float* w, const float* scale,
                             const float* shift, const float* residual, const float* in_sub, float* y,
                             uint32_t* act_bits, bool* bits_done, lmh_stream_t stream) {
  int rc = 0;
  LMH_CHECK_ARG(x && w && y);
  // (the residual / addend is an operand tensor like the others: read once)
  g_prof_pending_bytes = desc_bytes(d) + (residual ? 4.0 * d->N * d->OH * d->OW * (double)d->K : 0.0);
  const int64_t M = (int64_t)d->N * d->OH * d->OW;
  const bool fast = fwd_fast(d);
  LMH_CHECK_ARG((d->C % BK) != 0 || in_sub == nullptr);
  int bm, bn;
  pick_tile(M, d->K, &bm, &bn);
  hipStream_t st = (hipStream_t)stream;
  if (stem_fast(d) && residual == nullptr && !g_force_bm && ((uintptr_t)w & 15) == 0) {   // ResNet conv1: dedicated persistent kernel (reads w as float4)
    const int tiles_h = (d->OH + STEM_TH - 1) / STEM_TH, tiles_w = (d->OW + STEM_TW - 1) / STEM_TW;
    const int ntiles = d->N * tiles_h * tiles_w;
    const int grid1 = ntiles < 512 ? ntiles : 512;   // 2 resident blocks per CU (61 KB LDS each)
    prof_begin(st);
    lmh_launch(k_conv_stem7x7s2<0>, dim3(grid1), dim3(256), 0, st, *d, x, w, scale, shift, in_sub, y, tiles_h,
                       tiles_w);
    prof_end(st, desc_flops(d), "k_conv_stem7x7s2<0>");
    LMH_CHECK_LAUNCH();
    return LMH_OK;
  }
  const int grid = (int)(((M + bm - 1) / bm) * ((d->K + bn - 1) / bn));
  if (d->compute && fast && in_sub == nullptr) {          // f16 / bf16 operands, fp32 accumulate (conv_half.h)
    if (d->compute == 3 && x3_tile_pick) pick_tile(M, d->K, &bm, &bn, x3_tile_pick); else half_tile(M, d->K, &bm, &bn);
    if (d->compute == 3 && lmh_opt("x3_new") ) {      // round-6 bf16x3 kernel, bit mask fused (conv_x3.h)
      prof_begin(st);
      const int pipe = x3_pipe(d->R * d->S * (d->C / BK));
      rc = lmh_x3_fwd_launch(d, x, w, scale, shift, residual, y, act_bits, 0, bm, bn, pipe, st);
      prof_end(st, desc_flops(d), "k_x3_fwd<%d, %d, false, %d>", bm, bn, pipe);   // as rocprofv3 prints it
      *bits_done = true;
      return rc;
    }
    const int grid = (int)(((M + bm - 1) / bm) * ((d->K + bn - 1) / bn));
#define LAUNCH_FWD_H(DT_, BM_, BN_)                                                                       \
    do { if (half_pf == 4) lmh_launch((k_conv_fwd_h<DT_, BM_, BN_, 4>), dim3(grid), dim3(512), 0, st, *d, x, w, scale, shift, residual, y, 1); \
         else if (half_pf == 3) lmh_launch((k_conv_fwd_h<DT_, BM_, BN_, 3>), dim3(grid), dim3(512), 0, st, *d, x, w, scale, shift, residual, y, 1); \
         else if (half_pf == 2) lmh_launch((k_conv_fwd_h<DT_, BM_, BN_, 2>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, shift, residual, y, 1); \
         else lmh_launch((k_conv_fwd_h<DT_, BM_, BN_, 1>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, shift, residual, y, 1); } while (0)
#define LAUNCH_FWD_HT(BM_, BN_)                                                                           \
    do { if (d->compute == 1) LAUNCH_FWD_H(1, BM_, BN_); else if (d->compute == 2) LAUNCH_FWD_H(2, BM_, BN_);                \
         else if (x3_pf_fwd == 4) lmh_launch((k_conv_fwd_h<3, BM_, BN_, 4>), dim3(grid), dim3(512), 0, st, *d, x, w, scale, shift, residual, y, 1); \
         else if (x3_pf_fwd == 3) lmh_launch((k_conv_fwd_h<3, BM_, BN_, 3>), dim3(grid), dim3(512), 0, st, *d, x, w, scale, shift, residual, y, 1); \
         else if (x3_pf_fwd == 0) lmh_launch((k_conv_fwd_h<3, BM_, BN_, 0>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, shift, residual, y, 1); \
         else if (x3_pf_fwd == 1) lmh_launch((k_conv_fwd_h<3, BM_, BN_, 1>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, shift, residual, y, 1); \
         else lmh_launch((k_conv_fwd_h<3, BM_, BN_, 2>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, shift, residual, y, 1); } while (0)
    prof_begin(st);
    if (bm == 128 && bn == 128) LAUNCH_FWD_HT(128, 128);
    else if (bm == 128) LAUNCH_FWD_HT(128, 64);
    else LAUNCH_FWD_HT(64, 64);
#undef LAUNCH_FWD_HT
#undef LAUNCH_FWD_H
    prof_end(st, desc_flops(d), "k_conv_fwd_h<%d, %d, %d>", d->compute, bm, bn);
    LMH_CHECK_LAUNCH();
    return LMH_OK;
  }
  if (d->R == 1 && d->S == 1 && d->stride == 1 && (d->C % 128) == 0 && d->C >= 512 && d->C <= 4096 && d->K <= 512 &&
      d->act == 0 && !in_sub && M <= 4096 && ((M + 63) / 64) * ((d->K + 63) / 64) < 128 && !g_force_bm &&
      lmh_opt("head_gemm")) {      // (fewer than 128 tiles of 64 x 64: the tiled kernels would leave half the chip empty)
    // Linear head on few rows (the RCNN classifier / box regressor over 512 ROIs): 32x32 tiles, the reduction split over
    // the four waves of a block (conv_generic.h k_head_fwd); fp32 whatever `compute` says, like the skinny kernels
    const int tiles = (int)(((M + 31) / 32) * ((d->K + 31) / 32));
    prof_begin(st);
    lmh_launch(k_head_fwd, dim3(tiles), dim3(256), 0, st, x, w, scale, shift, residual, y, (int)M, d->C, d->K, d->act);
    prof_end(st, desc_flops(d), "k_head_fwd");
    *bits_done = false;
    LMH_CHECK_LAUNCH();
    return LMH_OK;
  }
  if (!fast && d->R == 1 && d->S == 1 && d->stride == 1 && d->K <= 128 && (d->C & 3) == 0 &&
      !in_sub && M <= 4096 && (size_t)d->C * 8 <= 64 * 1024) {
    // skinny Linear / 1x1 head (e.g. the 81-wide RCNN classifier): vector-ALU kernel out of LDS (conv_generic.h); shapes
    // off the MFMA fast paths run fp32 whatever `compute` says (like the predicated kernels they replace)
    prof_begin(st);
    lmh_launch(k_skinny_fwd, dim3((unsigned)((M + 1) / 2)), dim3(256), (unsigned)((size_t)d->C * 8), st, x, w, scale, shift,
               residual, y, (int)M, d->C, d->K, d->act);
    prof_end(st, desc_flops(d), "k_skinny_fwd");
    *bits_done = false;
    LMH_CHECK_LAUNCH();
    return LMH_OK;
  }
  if (fast && d->R == 1 && d->S == 1 && d->stride == 1 && d->pad_top == 0 && d->pad_left == 0 && d->OH == d->H &&
      d->OW == d->W && !in_sub && d->C >= 128 && !g_force_bm && lmh_opt("conv_pp")) {
    // persistent software-pipelined kernel (conv_pp.h): one block per compute unit walks >= 2 tiles of 128 x 128; taken when
    // the tiles divide evenly enough over the chip (the tiled kernel keeps the layers with fewer than two tiles per CU)
    // compute units of the CURRENT device (cached per device id; a stream created with a CU mask sees fewer: the balance
    // heuristic below then over-estimates and the kernel is merely slower than the tiled one, never wrong)
    static std::atomic<int> ncu_dev[16];
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    ncu = ncu_dev[dev].load(std::memory_order_relaxed);
    if (!ncu) {
      if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
      ncu_dev[dev].store(ncu, std::memory_order_relaxed);
    }
    const int tiles_m = (int)((M + 127) / 128), tiles_n = (d->K + 127) / 128;
    const int64_t ntiles = (int64_t)tiles_m * tiles_n;
    const int nsub = (int)((ntiles + ncu - 1) / ncu);
    // (measured, scripts/r5_epilogue_decomp.py: 4- and 8-stage tiles gain 6-15 %; at 16 stages per tile the epilogue is a
    // small share and two independent blocks per CU are as good: those layers stay with the tiled kernel)
    if (nsub >= 2 && ntiles * 10 >= (int64_t)nsub * ncu * 9 && (d->K % 128) == 0 && d->C <= 256 && (d->C >= 160 || d->C == 128) &&
        (M + 128) * (int64_t)d->K * 4 < ((int64_t)1 << 31) && (act_bits == nullptr || (d->K % 32) == 0)) {
      const int nblk = (int)((ntiles + nsub - 1) / nsub);
      prof_begin(st);
#define LAUNCH_PP(CC4_, BITS_)                                                                                    \
      lmh_launch((k_conv1x1_pp<CC4_, BITS_>), dim3(nblk), dim3(512), 0, st, (int)M, d->C, d->K, d->act, x, w, scale, \
                 shift, residual, y, act_bits, (int)ntiles, tiles_n, nsub)
      if (d->C == 128) { if (act_bits) LAUNCH_PP(true, true); else LAUNCH_PP(true, false); }
      else { if (act_bits) LAUNCH_PP(false, true); else LAUNCH_PP(false, false); }
#undef LAUNCH_PP
      prof_end(st, desc_flops(d), "k_conv1x1_pp<%s, %s>", d->C == 128 ? "true" : "false", act_bits ? "true" : "false");   // as rocprofv3 prints it
      *bits_done = true;
      LMH_CHECK_LAUNCH();
      return LMH_OK;
    }
  }
#define LAUNCH_FWD(BM_, BN_)                                                                              \
  do {                                                                                                    \
    if (fast)                                                                                             \
      lmh_launch((k_conv_fwd<BM_, BN_>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, shift,    \
                         residual, y, 1, act_bits);                                                       \
    else if ((d->C % BK) != 0)                                                                            \
      lmh_launch((k_conv_fwd_gen<BM_, BN_, true>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, \
                         shift, residual, in_sub, y);                                                     \
    else                                                                                                  \
      lmh_launch((k_conv_fwd_gen<BM_, BN_, false>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, \
                         shift, residual, in_sub, y);                                                     \
  } while (0)
  prof_begin(st);
  if (bm == 128 && bn == 128) LAUNCH_FWD(128, 128);
  else if (bm == 128) LAUNCH_FWD(128, 64);
  else LAUNCH_FWD(64, 64);
#undef LAUNCH_FWD
  if (fast) prof_end(st, desc_flops(d), "k_conv_fwd<%d, %d, false>", bm, bn);
  else prof_end(st, desc_flops(d), "k_conv_fwd_gen<%d, %d, %s>", bm, bn, (d->C % BK) != 0 ? "true" : "false");
  *bits_done = fast;
  (void)rc;
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

static bool x3w_bwd_data_ok(const lmh_conv_desc* d) {
  return d->compute == 3 && bwd_data_fast(d) && (d->C % 32) == 0 && (d->K % 32) == 0;
}
extern "C" int lmh_conv2d_bwd_data_x3w_supported(const lmh_conv_desc* d) { return d && check_desc(d) == LMH_OK && x3w_bwd_data_ok(d) ? 1 : 0; }
// lmh_conv2d_bwd_data for compute bf16x3 with `w3` from a BACKWARD split of the layer's w (lmh_x3_split_weights_batch);
// kscale / addend / xbits as there; bit-identical
extern "C" int lmh_conv2d_bwd_data_x3w(const lmh_conv_desc* d, const float* dy, const void* w3, const float* kscale,
                                       const float* addend, const uint32_t* xbits, float* dx, lmh_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  LMH_CHECK_ARG(dy && w3 && dx && (((uintptr_t)w3) & 15) == 0);
  LMH_CHECK_ARG(xbits == nullptr || (d->C & 31) == 0);
  if (!x3w_bwd_data_ok(d)) { lmh_set_error("lmh_conv2d_bwd_data_x3w: needs compute bf16x3, C %% 32 == 0, K %% 32 == 0"); return LMH_ERR_UNSUPPORTED; }
  g_prof_pending_bytes = desc_bytes(d) + (addend ? 4.0 * d->N * d->H * d->W * (double)d->C : 0.0);
  const int64_t M = (int64_t)d->N * d->H * d->W;
  int bm, bn;
  if (x3_tile_pick) pick_tile(M, d->C, &bm, &bn, x3_tile_pick); else half_tile(M, d->C, &bm, &bn);
  hipStream_t st = (hipStream_t)stream;
  prof_begin(st);
  rc = lmh_x3_bwd_data_ws_launch(d, dy, w3, kscale, addend, xbits, dx, bm, bn, st);
  prof_end(st, desc_flops(d), "k_x3_bwd_data_ws<%d, %d>", bm, bn);
  return rc;
}

static int conv2d_bwd_data_launch(const lmh_conv_desc* d, const float* dy, const float* w, const float* kscale,
                                  const float* addend, const float* yact, const uint32_t* xbits, bool* bits_done,
                                  float* dx, lmh_stream_t stream);

extern "C" int lmh_conv2d_bwd_data(const lmh_conv_desc* d, const float* dy, const float* w,
                                   const float* kscale, const float* addend, const float* yact,
                                   const uint32_t* xbits, float* dx, lmh_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  LMH_CHECK_ARG(xbits == nullptr || (d->C & 31) == 0);
  bool bits_done = false;
  rc = conv2d_bwd_data_launch(d, dy, w, kscale, addend, yact, xbits, &bits_done, dx, stream);
  if (rc) return rc;
  if (xbits && !bits_done)        // predicated / half-precision kernels: the mask as one in-place pass over dx
    return lmh_apply_act_bits_impl(dx, xbits, (int64_t)d->N * d->H * d->W, d->C, (hipStream_t)stream);
  return LMH_OK;
}

static int conv2d_bwd_data_launch(const lmh_conv_desc* d, const float* dy, const float* w, const float* kscale,
                                  const float* addend, const float* yact, const uint32_t* xbits, bool* bits_done,
                                  float* dx, lmh_stream_t stream) {
  LMH_CHECK_ARG(dy && w && dx);
  g_prof_pending_bytes = desc_bytes(d) + (addend ? 4.0 * d->N * d->H * d->W * (double)d->C : 0.0);
  LMH_CHECK_ARG(yact == nullptr || (bwd_data_fast(d) && d->act != 0));   // fused act'(y) only on the fast path
  const int64_t M = (int64_t)d->N * d->H * d->W;
  const bool fast = bwd_data_fast(d);
  int bm, bn;
  const int bd_slots = lmh_opt("bd_slots");
  pick_tile(M, d->C, &bm, &bn, bd_slots);
  bwd_data_parity_tile(d, M, &bm, &bn);
  hipStream_t st = (hipStream_t)stream;
  if (d->compute && fast && !yact) {
    if (d->compute == 3 && x3_tile_pick) pick_tile(M, d->C, &bm, &bn, x3_tile_pick); else half_tile(M, d->C, &bm, &bn);   // (no parity classes here)
    if (d->compute == 3 && lmh_opt("x3_new") ) {      // round-6 bf16x3 kernel, input mask fused (conv_x3.h)
      prof_begin(st);
      const int pipe = x3_pipe(d->R * d->S * (d->K / BK));
      const int rc3 = lmh_x3_bwd_data_launch(d, dy, w, kscale, addend, xbits, dx, bm, bn, pipe, st);
      prof_end(st, desc_flops(d), "k_x3_bwd_data<%d, %d, %d>", bm, bn, pipe);
      *bits_done = true;
      return rc3;
    }
    const int gridh = (int)(((M + bm - 1) / bm) * ((d->C + bn - 1) / bn));
    const float gs = half_gscale(d);
#define LAUNCH_BD_H(DT_, BM_, BN_)                                                                        \
    do { if (half_pf == 4) lmh_launch((k_conv_bwd_data_h<DT_, BM_, BN_, 4>), dim3(gridh), dim3(512), 0, st, *d, dy, w, kscale, addend, gs, dx); \
         else if (half_pf == 3) lmh_launch((k_conv_bwd_data_h<DT_, BM_, BN_, 3>), dim3(gridh), dim3(512), 0, st, *d, dy, w, kscale, addend, gs, dx); \
         else lmh_launch((k_conv_bwd_data_h<DT_, BM_, BN_, 1>), dim3(gridh), dim3(256), 0, st, *d, dy, w, kscale, addend, gs, dx); } while (0)
#define LAUNCH_BD_HT(BM_, BN_)                                                                            \
    do { if (d->compute == 1) LAUNCH_BD_H(1, BM_, BN_); else if (d->compute == 2) LAUNCH_BD_H(2, BM_, BN_);                  \
         else if (x3_pf_bd == 4) lmh_launch((k_conv_bwd_data_h<3, BM_, BN_, 4>), dim3(gridh), dim3(512), 0, st, *d, dy, w, kscale, addend, gs, dx); \
         else if (x3_pf_bd == 3) lmh_launch((k_conv_bwd_data_h<3, BM_, BN_, 3>), dim3(gridh), dim3(512), 0, st, *d, dy, w, kscale, addend, gs, dx); \
         else if (x3_pf_bd == 0) lmh_launch((k_conv_bwd_data_h<3, BM_, BN_, 0>), dim3(gridh), dim3(256), 0, st, *d, dy, w, kscale, addend, gs, dx); \
         else if (x3_pf_bd == 1) lmh_launch((k_conv_bwd_data_h<3, BM_, BN_, 1>), dim3(gridh), dim3(256), 0, st, *d, dy, w, kscale, addend, gs, dx); \
         else lmh_launch((k_conv_bwd_data_h<3, BM_, BN_, 2>), dim3(gridh), dim3(256), 0, st, *d, dy, w, kscale, addend, gs, dx); } while (0)
    prof_begin(st);
    if (bm == 128 && bn == 128) LAUNCH_BD_HT(128, 128);
    else if (bm == 128) LAUNCH_BD_HT(128, 64);
    else LAUNCH_BD_HT(64, 64);
#undef LAUNCH_BD_HT
#undef LAUNCH_BD_H
    prof_end(st, desc_flops(d), "k_conv_bwd_data_h<%d, %d, %d>", d->compute, bm, bn);
    LMH_CHECK_LAUNCH();
    return LMH_OK;
  }
  if (!fast && !yact && d->R == 1 && d->S == 1 && d->stride == 1 && d->K <= 64 && (d->K & 3) == 0 &&
      (d->C & 3) == 0) {
    // skinny 1x1 head (24 / 48 output channels of the RPN): vector-ALU kernel out of LDS, addend + bit mask fused
    static bool attr = false;
    const unsigned shmem = (unsigned)(((size_t)d->K * SKB_CH + (size_t)SKB_PIX * d->K) * sizeof(float));
    if (!attr) {
      LMH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_skinny_bwd_data),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      attr = true;
    }
    const int tiles = (int)(((M + SKB_PIX - 1) / SKB_PIX) * ((d->C + SKB_CH - 1) / SKB_CH));
    prof_begin(st);
    lmh_launch(k_skinny_bwd_data, dim3(tiles), dim3(256), shmem, st, dy, w, kscale, addend, xbits, dx, (int)M, d->C, d->K);
    prof_end(st, desc_flops(d), "k_skinny_bwd_data");
    *bits_done = true;
    LMH_CHECK_LAUNCH();
    return LMH_OK;
  }
  const int grid = (int)(((M + bm - 1) / bm) * ((d->C + bn - 1) / bn));
#define LAUNCH_BD(BM_, BN_)                                                                                 \
  do {                                                                                                      \
    if (fast && yact)                                                                                       \
      lmh_launch((k_conv_bwd_data<BM_, BN_, true>), dim3(grid), dim3(256), 0, st, *d, dy, w, kscale, \
                         addend, yact, xbits, dx);                                                          \
    else if (fast)                                                                                          \
      lmh_launch((k_conv_bwd_data<BM_, BN_, false>), dim3(grid), dim3(256), 0, st, *d, dy, w, kscale, \
                         addend, yact, xbits, dx);                                                          \
    else                                                                                                    \
      lmh_launch((k_conv_bwd_data_gen<BM_, BN_>), dim3(grid), dim3(256), 0, st, *d, dy, w, kscale,  \
                         addend, dx);                                                                       \
  } while (0)
  prof_begin(st);
  if (bm == 128 && bn == 128) LAUNCH_BD(128, 128);
  else if (bm == 128) LAUNCH_BD(128, 64);
  else LAUNCH_BD(64, 64);
#undef LAUNCH_BD
  if (fast) prof_end(st, desc_flops(d), "k_conv_bwd_data<%d, %d, %s>", bm, bn, yact ? "true" : "false");
  else prof_end(st, desc_flops(d), "k_conv_bwd_data_gen<%d, %d>", bm, bn);
  *bits_done = fast;
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// Backward-weight plan: output tiles x split-K over the pixels.  128x128 tiles when there are >= 128 of
// them, else 64x64 (4x the tiles, fewer splits => less partial-slab traffic); the split count is the
// smallest one (>= 16 stages per block) that fills the 512 resident-block slots to >= 90 %.
static void bwd_weight_plan(const lmh_conv_desc* d, int* bm, int* bn, int* splits, int* kt_per_split) {
  const int64_t t128 = (int64_t)d->R * d->S * ((d->C + 127) / 128) * ((d->K + 127) / 128);
  // the 36 stacked reduction GEMMs of a Winograd weight gradient (dilation 0: conv_winograd.h) are not split, so 128x128
  // tiles only pay once two of them per CU exist (block3 256x256: 144 tiles, 70.0 us at 128x128 against 57.5 at 64x64;
  // RPN 1024x512: 1152 tiles, 254 against 242 — scripts/sweep_wino_tiles.py)
  const int64_t need128 = d->dilation == 0 ? 2048 : 128;
  if (t128 >= need128 && d->C >= 128 && d->K >= 128) { *bm = 128; *bn = 128; }
  else { *bm = 64; *bn = 64; }
  // half-precision operands: the kernel is bound by its staging path, 128x128 tiles halve the bytes per FLOP
  if (d->compute && d->C >= 128 && d->K >= 128) { *bm = 128; *bn = 128; }
  if (g_force_bm && g_force_bn) { *bm = g_force_bm; *bn = g_force_bn; }
  const int64_t tiles = (int64_t)d->R * d->S * ((d->C + *bm - 1) / *bm) * ((d->K + *bn - 1) / *bn);
  const int64_t P = (int64_t)d->N * d->OH * d->OW;
  const int KT = (int)((P + BK - 1) / BK);
  int max_split = KT / 16 > 0 ? KT / 16 : 1;
  if (d->compute) max_split = KT / 8 > 0 ? KT / 8 : 1;
  if (max_split > (d->compute ? 128 : 64)) max_split = d->compute ? 128 : 64;
  // bf16x3 (conv_x3.h, two 61 KB blocks per CU): filling 256 slots — half the splits, half the slab traffic — measured 5.38 ms per
  // step against 5.54 with 512 (round 6, one box; 128: 5.77, 192: 5.47, 384: 5.48)
  const int bw_slots = d->compute == 3 ? lmh_opt("x3_bw_slots") : lmh_opt("bw_slots");
  int want = 1;
  double best_eff = -1.0;
  for (int s = 1; s <= max_split; ++s) {
    const double rounds = (double)(tiles * s) / (double)bw_slots;
    const double eff = rounds / (double)(int64_t)(rounds + 0.999999);
    if (eff >= 0.9) { want = s; best_eff = eff; break; }
    if (eff > best_eff + 1e-9) { best_eff = eff; want = s; }
  }
  if (g_force_splits) want = g_force_splits < KT ? g_force_splits : KT;
  *kt_per_split = (KT + want - 1) / want;
  *splits = (KT + *kt_per_split - 1) / *kt_per_split;
}

// ---- 1x1 / stride-1 weight gradient through the direct-to-LDS GEMM kernel (conv_wgrad1x1.h) -------------------
// variant: 0 automatic, -1 never (the register-staged k_conv_bwd_weight), 2..4 = forced LDS ring depth
static thread_local int g_wg_variant = 0;
extern "C" void lmh_conv2d_force_wgrad_variant(int v) { g_wg_variant = v; }
static bool wgrad_1x1_ok(const lmh_conv_desc* d) {
  const int on = lmh_opt("wgrad_glds");
  return on && g_wg_variant >= 0 && d->compute == 0 && d->R == 1 && d->S == 1 && d->stride == 1 && d->pad_top == 0 &&
         d->pad_left == 0 && d->OH == d->H && d->OW == d->W && (d->C & 3) == 0 && (d->K & 3) == 0 &&
         d->C >= 32 && d->K >= 32;
}
// 64x64 tiles: the partial-slab traffic of a split-K weight gradient is 2 * 4 B * (blocks) * BM * BN whatever the
// layer, so small tiles win on these skinny outputs once the per-stage instruction overhead is gone (glds).
// Splits: fill `slots` resident blocks (2 per CU) with >= 8 stages per block.
static void wgrad_1x1_plan(const lmh_conv_desc* d, int* bm, int* bn, int* nbuf, int* splits, int* kt_per_split) {
  *bm = 64; *bn = 64;
  if (g_force_bm && g_force_bn) { *bm = g_force_bm; *bn = g_force_bn; }
  *nbuf = (g_wg_variant >= 2 && g_wg_variant <= 4) ? g_wg_variant : 4;
  if (*bm == 128 && *bn == 128 && *nbuf > 4) *nbuf = 4;
  const int64_t tiles = (int64_t)((d->C + *bm - 1) / *bm) * ((d->K + *bn - 1) / *bn);
  const int64_t P = (int64_t)d->N * d->OH * d->OW;
  const int KT = (int)((P + BK - 1) / BK);
  const int slots = lmh_opt("wg_slots");
  int want = (int)(slots / tiles);
  if (want < 1) want = 1;
  const int max_split = KT / 8 > 0 ? KT / 8 : 1;
  if (want > max_split) want = max_split;
  if (g_force_splits) want = g_force_splits < KT ? g_force_splits : KT;
  *kt_per_split = (KT + want - 1) / want;
  *splits = (KT + *kt_per_split - 1) / *kt_per_split;
}

extern "C" int lmh_conv2d_bwd_weight_fuses_colsum(const lmh_conv_desc* d) {
  return d && bwd_weight_fast(d) && (d->compute == 0 || d->compute == 3) ? 1 : 0;
}

extern "C" int lmh_conv2d_kernel_id(const lmh_conv_desc* d, int op) {
  if (!d) return -1;
  int bm = 0, bn = 0;
  if (op == 0) {
    if (stem_fast(d) && !g_force_bm) return 7007;   // k_conv_stem7x7s2
    pick_tile((int64_t)d->N * d->OH * d->OW, d->K, &bm, &bn);
    return bm * 1000 + bn + (fwd_fast(d) ? 0 : 1000000);
  }
  if (op == 1) {
    pick_tile((int64_t)d->N * d->H * d->W, d->C, &bm, &bn, lmh_opt("bd_slots"));
    bwd_data_parity_tile(d, (int64_t)d->N * d->H * d->W, &bm, &bn);
    return bm * 1000 + bn + (bwd_data_fast(d) ? 0 : 1000000);
  }
  int splits, kps;
  if (wgrad_1x1_ok(d)) {
    int nbuf;
    wgrad_1x1_plan(d, &bm, &bn, &nbuf, &splits, &kps);
    return bm * 1000 + bn;
  }
  bwd_weight_plan(d, &bm, &bn, &splits, &kps);
  return bm * 1000 + bn + (bwd_weight_fast(d) ? 0 : 1000000);
}

static bool wgrad_hs_plan(const lmh_conv_desc* d, int* bm, int* bn, int* splits, int* kt_per_split);

extern "C" size_t lmh_conv2d_bwd_weight_workspace_bytes(const lmh_conv_desc* d) {
  if (!d) return 0;
  int bm, bn, splits, kps;
  if (wgrad_1x1_ok(d)) {
    int nbuf;
    wgrad_1x1_plan(d, &bm, &bn, &nbuf, &splits, &kps);
    // slabs; the column sums of this path are NOT fused (the caller takes them from lmh_act_bwd), but a caller that
    // passes `colsum` anyway is served by the register-staged kernel, whose plan may need more room
    int bm2, bn2, s2, k2;
    bwd_weight_plan(d, &bm2, &bn2, &s2, &k2);
    if (s2 > splits) splits = s2;
    const size_t slabs = splits <= 1 ? 256 : lmh_align_up((size_t)splits * d->C * d->K * sizeof(float), 256);
    return slabs + lmh_align_up((size_t)splits * BK * d->K * sizeof(float), 256);   // <= splits * 32 partial column rows
  }
  bwd_weight_plan(d, &bm, &bn, &splits, &kps);
  {
    int hb, hn, hs_splits, hk;
    if (wgrad_hs_plan(d, &hb, &hn, &hs_splits, &hk) && hs_splits > splits) splits = hs_splits;   // half-storage entry point
  }
  // split-K slabs, then [splits][K] column-sum partials (fused dbeta / dbias)
  const size_t slabs = splits <= 1 ? 256 : lmh_align_up((size_t)splits * d->R * d->S * d->C * d->K * sizeof(float), 256);
  return slabs + lmh_align_up((size_t)splits * d->K * sizeof(float), 256);
}

static int bwd_weight_launch(const lmh_conv_desc* d, const float* x, const float* dy, const float* yact,
                             float* dw, float* colsum, void* ws, size_t ws_bytes, hipStream_t stream, bool gb);
// stacked (gb) launches on behalf of the Winograd weight gradient: with `want` set a split reduction is NOT reduced here —
// the slabs and their count are reported and the caller's next kernel (k_wino4_dw) adds them on load
struct lmh_gb_slabs { bool want; const float* slabs; int splits; };
static thread_local lmh_gb_slabs g_gb_slabs = {false, nullptr, 0};

extern "C" int lmh_conv2d_bwd_weight(const lmh_conv_desc* d, const float* x, const float* dy, const float* yact,
                                     float* dw, float* colsum, void* ws, size_t ws_bytes,
                                     lmh_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  return bwd_weight_launch(d, x, dy, yact, dw, colsum, ws, ws_bytes, (hipStream_t)stream, false);
}

// gb: stacked independent GEMMs (see k_conv_bwd_weight<..., GB>); fast path only, no fused operands.
static int bwd_weight_launch(const lmh_conv_desc* d, const float* x, const float* dy, const float* yact,
                             float* dw, float* colsum, void* ws, size_t ws_bytes, hipStream_t stream, bool gb) {
  LMH_CHECK_ARG(x && dy && dw);
  LMH_CHECK_ARG(!gb || (bwd_weight_fast(d) && !yact && !colsum));
  g_prof_pending_bytes = gb ? 4.0 * d->R * d->S * ((double)d->H * d->C + (double)d->H * d->K + (double)d->C * d->K)
                            : desc_bytes(d);
  int bm, bn, splits, kps;
  if (!gb && !yact && wgrad_1x1_ok(d)) {       // pure TN GEMM: direct-to-LDS kernel
    int nbuf;
    wgrad_1x1_plan(d, &bm, &bn, &nbuf, &splits, &kps);
    if (ws_bytes < lmh_conv2d_bwd_weight_workspace_bytes(d) || (splits > 1 && !ws)) {
      lmh_set_error("lmh_conv2d_bwd_weight: workspace too small");
      return LMH_ERR_WORKSPACE;
    }
    const int P = d->N * d->OH * d->OW;
    const int tc = (d->C + bm - 1) / bm, tk = (d->K + bn - 1) / bn;
    float* o = splits > 1 ? reinterpret_cast<float*>(ws) : dw;
    // [splits * min(tc, 32)][K] partial column sums of g behind the slabs (fused dbeta / dbias)
    const int crows = splits * (tc < BK ? tc : BK);
    const size_t slab_b = splits > 1 ? lmh_align_up((size_t)splits * d->C * d->K * sizeof(float), 256) : 256;
    float* cpart1 = colsum ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + slab_b) : nullptr;
    const dim3 grid1(tc * tk * splits);
#define LAUNCH_WG1(BM_, BN_, NB_)                                                                              \
    lmh_launch((k_wgrad_1x1<BM_, BN_, NB_>), grid1, dim3(256), 0, stream, x, dy, o, P, d->C, d->K, kps, tc, \
                       tk, splits, cpart1)
#define LAUNCH_WG1_T(BM_, BN_)                                                                                 \
    do { if (nbuf == 2) LAUNCH_WG1(BM_, BN_, 2); else if (nbuf == 3) LAUNCH_WG1(BM_, BN_, 3); else LAUNCH_WG1(BM_, BN_, 4); } while (0)
    prof_begin(stream);
    if (bm == 128 && bn == 128) LAUNCH_WG1_T(128, 128);
    else if (bm == 128) LAUNCH_WG1_T(128, 64);
    else if (bn == 128) LAUNCH_WG1_T(64, 128);
    else LAUNCH_WG1_T(64, 64);
#undef LAUNCH_WG1_T
#undef LAUNCH_WG1
    prof_end(stream, desc_flops(d), "k_wgrad_1x1<%d, %d, %d>", bm, bn, nbuf);
    if (g_lmh_defer_tail) {
      g_lmh_last_plan.slabs = splits > 1 ? reinterpret_cast<const float*>(ws) : nullptr;
      g_lmh_last_plan.splits = splits > 1 ? splits : 0;
      g_lmh_last_plan.colpart = cpart1;
      g_lmh_last_plan.colrows = cpart1 ? crows : 0;
    } else if (splits > 1 || colsum) {
      const int64_t n = splits > 1 ? (int64_t)d->C * d->K : 0;
      const int nb_slab = n > 0 ? (int)((n / 4 + 255) / 256 + 1) : 0;
      const int nb_col = colsum ? (d->K + 31) / 32 : 0;
      lmh_launch(k_splitk_reduce, dim3(nb_slab + nb_col), dim3(256), 0, stream,
                         reinterpret_cast<const float*>(ws), n, splits, dw, (const float*)cpart1, colsum, d->K, nb_slab,
                         crows);
    }
    LMH_CHECK_LAUNCH();
    return LMH_OK;
  }
  bwd_weight_plan(d, &bm, &bn, &splits, &kps);
  if (ws_bytes < lmh_conv2d_bwd_weight_workspace_bytes(d) || (splits > 1 && !ws)) {
    lmh_set_error("lmh_conv2d_bwd_weight: workspace too small");
    return LMH_ERR_WORKSPACE;
  }
  LMH_CHECK_ARG((yact == nullptr && colsum == nullptr) || bwd_weight_fast(d));   // fused paths: fast kernels only
  LMH_CHECK_ARG(yact == nullptr || d->act != 0);
  hipStream_t st = (hipStream_t)stream;
  const size_t slab_bytes = splits > 1 ? lmh_align_up((size_t)splits * d->R * d->S * d->C * d->K * sizeof(float), 256) : 256;
  float* out = splits > 1 ? reinterpret_cast<float*>(ws) : dw;
  float* cpart = colsum ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + slab_bytes) : nullptr;
  const bool fast = bwd_weight_fast(d);
  dim3 grid(d->R * d->S * ((d->C + bm - 1) / bm), (d->K + bn - 1) / bn, splits);
  const lmh_fastdiv dvw = lmh_make_fastdiv((uint32_t)d->OW), dvh = lmh_make_fastdiv((uint32_t)d->OH);
  if (d->compute == 3 && fast && gb && !yact && !colsum) {      // Winograd weight gradient, 16 stacked GEMMs in bf16x3
    const int nblk = (int)(grid.x * grid.y * grid.z);
    if (lmh_opt("x3_new")) {
      prof_begin(st);
      const int rc3 = lmh_x3_bwd_weight_launch(d, x, dy, out, kps, (int)grid.x, (int)grid.y, (int)grid.z, nullptr, true, bm, bn, x3_pipe(2), st);
      prof_end(st, desc_flops(d), "k_x3_bwd_weight<%d, %d, true, %d, %s>", bm, bn, x3_pipe(2), lmh_x3_bwd_weight_plain(d, true, x3_pipe(2)) ? "true" : "false");
      if (rc3) return rc3;
      if (splits > 1 && g_gb_slabs.want) {
        g_gb_slabs.slabs = reinterpret_cast<const float*>(ws);
        g_gb_slabs.splits = splits;
      } else if (splits > 1) {
        const int64_t n = (int64_t)d->R * d->S * d->C * d->K;
        const int nb_slab = (int)((n / 4 + 255) / 256 + 1);
        lmh_launch(k_splitk_reduce, dim3(nb_slab), dim3(256), 0, st, reinterpret_cast<const float*>(ws), n,
                   splits, dw, (const float*)nullptr, (float*)nullptr, d->K, nb_slab, 0);
      }
      LMH_CHECK_LAUNCH();
      return LMH_OK;
    }
#define LAUNCH_BW_G(BM_, BN_)                                                                              \
    lmh_launch((k_conv_bwd_weight_h<3, BM_, BN_, 0, true>), dim3(nblk), dim3(256), 0, st, *d, x, dy, out, kps, dvw, \
                       dvh, 1.f, (int)grid.x, (int)grid.y, (int)grid.z, (float*)nullptr)
    prof_begin(st);
    if (bm == 128 && bn == 128) LAUNCH_BW_G(128, 128);
    else if (bm == 128) LAUNCH_BW_G(128, 64);
    else if (bn == 128) LAUNCH_BW_G(64, 128);
    else LAUNCH_BW_G(64, 64);
#undef LAUNCH_BW_G
    prof_end(st, desc_flops(d), "k_conv_bwd_weight_h<3, %d, %d, GB>", bm, bn);
    if (splits > 1) {
      const int64_t n = (int64_t)d->R * d->S * d->C * d->K;
      const int nb_slab = (int)((n / 4 + 255) / 256 + 1);
      lmh_launch(k_splitk_reduce, dim3(nb_slab), dim3(256), 0, st, reinterpret_cast<const float*>(ws), n,
                         splits, dw, (const float*)nullptr, (float*)nullptr, d->K, nb_slab, 0);
    }
    LMH_CHECK_LAUNCH();
    return LMH_OK;
  }
  if (d->compute && fast && !gb && !yact && (!colsum || d->compute == 3)) {
    const float gs = half_gscale(d);
    float* cpart_h = (colsum && d->compute == 3) ? cpart : nullptr;      // fused channel sums: bf16x3 only (exact)
    const int nblk = (int)(grid.x * grid.y * grid.z);
#define LAUNCH_BW_H(DT_, BM_, BN_)                                                                         \
    do { if (half_pf == 4) lmh_launch((k_conv_bwd_weight_h<DT_, BM_, BN_, 4>), dim3(nblk), dim3(512), 0, st, *d, x, dy, out, kps, dvw,   \
                       dvh, gs, (int)grid.x, (int)grid.y, (int)grid.z, cpart_h);                                    \
         else if (half_pf == 3) lmh_launch((k_conv_bwd_weight_h<DT_, BM_, BN_, 3>), dim3(nblk), dim3(512), 0, st, *d, x, dy, out, kps, dvw,   \
                       dvh, gs, (int)grid.x, (int)grid.y, (int)grid.z, cpart_h);                                    \
         else if (half_pf == 2) lmh_launch((k_conv_bwd_weight_h<DT_, BM_, BN_, 2>), dim3(nblk), dim3(256), 0, st, *d, x, dy, out, kps, dvw,   \
                       dvh, gs, (int)grid.x, (int)grid.y, (int)grid.z, cpart_h);                                    \
         else lmh_launch((k_conv_bwd_weight_h<DT_, BM_, BN_, 1>), dim3(nblk), dim3(256), 0, st, *d, x, dy, out, kps, dvw,   \
                       dvh, gs, (int)grid.x, (int)grid.y, (int)grid.z, cpart_h); } while (0)
#define LAUNCH_BW_HT(BM_, BN_)                                                                             \
    do { if (d->compute == 1) LAUNCH_BW_H(1, BM_, BN_); else if (d->compute == 2) LAUNCH_BW_H(2, BM_, BN_);                  \
         else if (x3_pf_bw == 4) lmh_launch((k_conv_bwd_weight_h<3, BM_, BN_, 4>), dim3(nblk), dim3(512), 0, st, *d, x, dy, out, kps, dvw, \
                       dvh, gs, (int)grid.x, (int)grid.y, (int)grid.z, cpart_h);                                    \
         else if (x3_pf_bw == 3) lmh_launch((k_conv_bwd_weight_h<3, BM_, BN_, 3>), dim3(nblk), dim3(512), 0, st, *d, x, dy, out, kps, dvw, \
                       dvh, gs, (int)grid.x, (int)grid.y, (int)grid.z, cpart_h);                                    \
         else if (x3_pf_bw == 0) lmh_launch((k_conv_bwd_weight_h<3, BM_, BN_, 0>), dim3(nblk), dim3(256), 0, st, *d, x, dy, out, kps, dvw, \
                       dvh, gs, (int)grid.x, (int)grid.y, (int)grid.z, cpart_h);                                    \
         else if (x3_pf_bw == 1) lmh_launch((k_conv_bwd_weight_h<3, BM_, BN_, 1>), dim3(nblk), dim3(256), 0, st, *d, x, dy, out, kps, dvw, \
                       dvh, gs, (int)grid.x, (int)grid.y, (int)grid.z, cpart_h);                                    \
         else lmh_launch((k_conv_bwd_weight_h<3, BM_, BN_, 2>), dim3(nblk), dim3(256), 0, st, *d, x, dy, out, kps, dvw, \
                       dvh, gs, (int)grid.x, (int)grid.y, (int)grid.z, cpart_h); } while (0)
    prof_begin(st);
    if (d->compute == 3 && lmh_opt("x3_new")) {
      const int rc3 = lmh_x3_bwd_weight_launch(d, x, dy, out, kps, (int)grid.x, (int)grid.y, (int)grid.z, cpart_h, false, bm, bn, x3_pipe(2), st);
      if (rc3) return rc3;
    }
    else if (bm == 128 && bn == 128) LAUNCH_BW_HT(128, 128);
    else if (bm == 128) LAUNCH_BW_HT(128, 64);
    else if (bn == 128) LAUNCH_BW_HT(64, 128);
    else LAUNCH_BW_HT(64, 64);
#undef LAUNCH_BW_HT
#undef LAUNCH_BW_H
    if (d->compute == 3 && lmh_opt("x3_new")) prof_end(st, desc_flops(d), "k_x3_bwd_weight<%d, %d, false, %d, %s>", bm, bn, x3_pipe(2), lmh_x3_bwd_weight_plain(d, false, x3_pipe(2)) ? "true" : "false");
    else prof_end(st, desc_flops(d), "k_conv_bwd_weight_h<%d, %d, %d>", d->compute, bm, bn);
    if (g_lmh_defer_tail) {
      g_lmh_last_plan.slabs = splits > 1 ? reinterpret_cast<const float*>(ws) : nullptr;
      g_lmh_last_plan.splits = splits > 1 ? splits : 0;
      g_lmh_last_plan.colpart = cpart_h;
      g_lmh_last_plan.colrows = cpart_h ? splits : 0;
    } else if (splits > 1 || cpart_h) {
      const int64_t n = splits > 1 ? (int64_t)d->R * d->S * d->C * d->K : 0;
      const int nb_slab = n > 0 ? (int)((n / 4 + 255) / 256 + 1) : 0;
      const int nb_col = cpart_h ? (d->K + 31) / 32 : 0;
      lmh_launch(k_splitk_reduce, dim3(nb_slab + nb_col), dim3(256), 0, st, reinterpret_cast<const float*>(ws), n,
                         splits, dw, (const float*)cpart_h, cpart_h ? colsum : (float*)nullptr, d->K, nb_slab, splits);
    }
    LMH_CHECK_LAUNCH();
    return LMH_OK;
  }
#define LAUNCH_BW(BM_, BN_)                                                                              \
  do {                                                                                                   \
    if (gb)                                                                                              \
      lmh_launch((k_conv_bwd_weight<BM_, BN_, false, true>), dim3(grid.x * grid.y * grid.z), dim3(256), 0, st, \
                         *d, x, dy, out, kps, dvw, dvh, yact, cpart, (int)grid.x, (int)grid.y, (int)grid.z); \
    else if (fast && yact)                                                                               \
      lmh_launch((k_conv_bwd_weight<BM_, BN_, true>), dim3(grid.x * grid.y * grid.z), dim3(256), 0, st, \
                         *d, x, dy, out, kps, dvw, dvh, yact, cpart, (int)grid.x, (int)grid.y, (int)grid.z); \
    else if (fast)                                                                                       \
      lmh_launch((k_conv_bwd_weight<BM_, BN_, false>), dim3(grid.x * grid.y * grid.z), dim3(256), 0, st, \
                         *d, x, dy, out, kps, dvw, dvh, yact, cpart, (int)grid.x, (int)grid.y, (int)grid.z); \
    else                                                                                                 \
      lmh_launch((k_conv_bwd_weight_gen<BM_, BN_>), grid, dim3(256), 0, st, *d, x, dy, out, kps); \
  } while (0)
  prof_begin(st);
  if (bm == 128 && bn == 128) LAUNCH_BW(128, 128);
  else if (bm == 128) LAUNCH_BW(128, 64);
  else if (bn == 128) LAUNCH_BW(64, 128);
  else LAUNCH_BW(64, 64);
#undef LAUNCH_BW
  if (fast) prof_end(st, desc_flops(d), "k_conv_bwd_weight<%d, %d, %s, %s>", bm, bn, (yact && !gb) ? "true" : "false",
                     gb ? "true" : "false");
  else prof_end(st, desc_flops(d), "k_conv_bwd_weight_gen<%d, %d>", bm, bn);
  if (g_lmh_defer_tail && !gb) {
    g_lmh_last_plan.slabs = splits > 1 ? reinterpret_cast<const float*>(ws) : nullptr;
    g_lmh_last_plan.splits = splits > 1 ? splits : 0;
    g_lmh_last_plan.colpart = cpart;                 // [splits][K] partial sums of g (fused dbeta / dbias), or NULL
    g_lmh_last_plan.colrows = cpart ? splits : 0;
  } else if (gb && splits > 1 && g_gb_slabs.want) {
    g_gb_slabs.slabs = reinterpret_cast<const float*>(ws);
    g_gb_slabs.splits = splits;
  } else if (splits > 1 || colsum) {
    const int64_t n = splits > 1 ? (int64_t)d->R * d->S * d->C * d->K : 0;
    const int nb_slab = n > 0 ? (int)((n / 4 + 255) / 256 + 1) : 0;
    const int nb_col = colsum ? (d->K + 31) / 32 : 0;
    lmh_launch(k_splitk_reduce, dim3(nb_slab + nb_col), dim3(256), 0, st, reinterpret_cast<const float*>(ws),
                       n, splits, dw, cpart, colsum, d->K, nb_slab, splits);
  }
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

#include "conv_winograd.h"

// ============================================================================
// half-storage entry points (conv_hs.h): f16 / bf16 tensors in HBM
// ============================================================================
#include "conv_hs.h"

static bool hs_ok(const lmh_conv_desc* d) {
  return (d->compute == 1 || d->compute == 2) && (d->C % HS_BK) == 0 && (d->K % HS_BK) == 0;
}
extern "C" int lmh_conv2d_hs_supported(const lmh_conv_desc* d) { return d && check_desc(d) == LMH_OK && hs_ok(d) ? 1 : 0; }

// bytes of a half-storage launch: every operand read once, the result written once, 2 bytes per element
static inline double hs_bytes(const lmh_conv_desc* d) {
  return 2.0 * ((double)d->N * d->H * d->W * d->C + (double)d->R * d->S * d->C * d->K + (double)d->N * d->OH * d->OW * d->K);
}

template <bool BWD>
static int hs_launch(const lmh_conv_desc* d, const void* A, const void* B, const hs_epilogue& e, hipStream_t st) {
  const int64_t M = BWD ? (int64_t)d->N * d->H * d->W : (int64_t)d->N * d->OH * d->OW;
  const int NC = BWD ? d->C : d->K;
  int bm, bn;
  const bool bg = lmh_opt("hs_bg") != 0;
  const int KT = d->R * d->S * ((BWD ? d->K : d->C) / HS_BK);
  // (round 6: a six-deep ring — 144 KB, five stages in flight, one block per CU — for the launches of about one tile per CU was
  // built, bit-identical, and LOST inside the step: forward 0.85 -> 0.99 ms of the f16 step, profiles/r06_ab.md; removed)
  // B through LDS (hs_bg = 0, rounds 3-5): 128 x 64 once it still gives every CU a block (fewer, longer tiles: the RPN 3x3
  // backward 161 -> 141 us, block3 256->1024 forward 16.2 -> 14.5 us), else 64 x 64; 128 x 128 (one block per CU) never won.
  // B straight from global memory (hs_bg = 1): a stage's LDS-DMA traffic is the A rows alone and 64 x 64 blocks (24 KB of LDS,
  // 88 VGPRs: five per CU) win every trunk layer (scripts/bench_conv_hs.py --sweep, profiles/r06_ab.md); the long reductions
  // with tiles to spare (RPN 3x3: 144 stages) keep 128 x 64
  if (g_force_bm && g_force_bn) { bm = g_force_bm; bn = g_force_bn; }
  else if (bg) {
    if (KT >= 64 && ((M + 127) / 128) * ((NC + 63) / 64) >= 256) { bm = 128; bn = 64; }
    else { bm = 64; bn = 64; }
  }
  else if (((M + 127) / 128) * ((NC + 63) / 64) >= 256) { bm = 128; bn = 64; }
  else { bm = 64; bn = 64; }
  if (bm == 256) bn = 128;
  if (bm == 64 && bn > 64 && !bg) bn = 64;            // the wide 64-row tiles exist for the register-B kernels only
  if (bm == 64 && bn > 128) bn = 128;
  const int grid = (int)(((M + bm - 1) / bm) * ((NC + bn - 1) / bn));
#define LAUNCH_HS(DT_, BM_, BN_, BG_)                                                                        \
  lmh_launch((k_conv_hs<DT_, BM_, BN_, BWD, BG_>), dim3(grid), dim3(256), 0, st, *d,                    \
                     reinterpret_cast<const HT<DT_>::T*>(A), reinterpret_cast<const HT<DT_>::T*>(B), e)
#define LAUNCH_HS_T(BM_, BN_)                                                                                \
  do {                                                                                                       \
    if (d->compute == 1) { if (bg) LAUNCH_HS(1, BM_, BN_, true); else LAUNCH_HS(1, BM_, BN_, false); }       \
    else { if (bg) LAUNCH_HS(2, BM_, BN_, true); else LAUNCH_HS(2, BM_, BN_, false); }                       \
  } while (0)
#define LAUNCH_HS_BG(BM_, BN_) do { if (d->compute == 1) LAUNCH_HS(1, BM_, BN_, true); else LAUNCH_HS(2, BM_, BN_, true); } while (0)
  prof_begin(st);
  if (bm == 256) LAUNCH_HS_T(256, 128);
  else if (bm == 128 && bn == 128) LAUNCH_HS_T(128, 128);
  else if (bm == 128) { bn = 64; LAUNCH_HS_T(128, 64); }
  else if (bn == 128) LAUNCH_HS_BG(64, 128);
  else { bn = 64; LAUNCH_HS_T(64, 64); }
#undef LAUNCH_HS_BG
#undef LAUNCH_HS_T
#undef LAUNCH_HS
  prof_end(st, desc_flops(d), "k_conv_hs<%d, %d, %d, %s, %s>", d->compute, bm, bn, BWD ? "true" : "false", bg ? "true" : "false");
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

extern "C" int lmh_conv2d_fwd_hs(const lmh_conv_desc* d, const void* x, const void* w_fwd, const float* scale,
                                 const float* shift, const void* residual, void* y, int y_is_f32, uint32_t* act_bits,
                                 lmh_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  LMH_CHECK_ARG(x && w_fwd && y);
  if (!hs_ok(d)) { lmh_set_error("lmh_conv2d_fwd_hs: needs compute f16 / bf16, C %% 64 == 0, K %% 64 == 0"); return LMH_ERR_UNSUPPORTED; }
  LMH_CHECK_ARG(act_bits == nullptr || d->act != 0);
  LMH_CHECK_ARG((((uintptr_t)scale | (uintptr_t)shift) & 15) == 0);      // read as float4 by the epilogue
  g_prof_pending_bytes = hs_bytes(d) + (residual ? 2.0 * d->N * d->OH * d->OW * (double)d->K : 0.0) + (y_is_f32 ? 2.0 * d->N * d->OH * d->OW * (double)d->K : 0.0);
  hs_epilogue e = {scale, shift, residual, nullptr, act_bits, y, y_is_f32, 1.f};
  return hs_launch<false>(d, x, w_fwd, e, (hipStream_t)stream);
}

extern "C" int lmh_conv2d_bwd_data_hs(const lmh_conv_desc* d, const void* g, const void* w_bwd, const void* addend,
                                      const uint32_t* xbits, void* dx, int dx_is_f32, float mul, lmh_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  LMH_CHECK_ARG(g && w_bwd && dx);
  if (!hs_ok(d)) { lmh_set_error("lmh_conv2d_bwd_data_hs: needs compute f16 / bf16, C %% 64 == 0, K %% 64 == 0"); return LMH_ERR_UNSUPPORTED; }
  g_prof_pending_bytes = hs_bytes(d) + (addend ? 2.0 * d->N * d->H * d->W * (double)d->C : 0.0) + (dx_is_f32 ? 2.0 * d->N * d->H * d->W * (double)d->C : 0.0);
  hs_epilogue e = {nullptr, nullptr, addend, xbits, nullptr, dx, dx_is_f32, mul};
  return hs_launch<true>(d, g, w_bwd, e, (hipStream_t)stream);
}

// Plan of k_wgrad_hs_tr (stages of 64 pixels): 64 x 64 tiles, as many pixel splits as fill the resident-block slots once with
// >= 4 stages per block.  Rounds 3-5 took 128 x 128 tiles when both channel counts reach it (half the L2 traffic per FLOP): with
// the tiles going through registers (RS = 4) the small tiles are as fast or faster on every layer alone (RPN 3x3 240 -> 173 us, block2
// 3x3/2 18 -> 11 us, the 1x1 layers within 1 us; scripts/bench_conv_hs.py --sweep); for the 1x1 layers a block carries a quarter of the split-K slab
// bytes (the slabs are splits x |dW| fp32 bytes written here and read again by the tail — for a fixed number
// of blocks that is blocks x tile area), and the f16 step gains 1 % (3.20 -> 3.17 ms).  hs_wg_tile = 128: the old rule.
static bool wgrad_hs_plan(const lmh_conv_desc* d, int* bm, int* bn, int* splits, int* kt_per_split) {
  if (!((d->compute == 1 || d->compute == 2) && (d->C % 64) == 0 && (d->K % 64) == 0)) return false;
  const bool gather = d->R * d->S > 1 || d->stride > 1;
  const int wt = lmh_opt("hs_wg_tile");
  // (the gathered layers keep 128 x 128: alone the RPN 3x3 is faster in 64 x 64 blocks, 173 against 240 us, but its 2304 blocks at
  // three per CU then sit on every CU beside the proposal chain — the chain ends 0.2 ms later and the f16 step is 3.28 instead
  // of 3.17 ms, scripts/ab.sh on one box)
  *bm = *bn = ((wt == 128 || gather) && d->C >= 128 && d->K >= 128) ? 128 : 64;
  if (g_force_bm && g_force_bn) *bm = *bn = (g_force_bm >= 128 && g_force_bn >= 128) ? 128 : 64;
  const int64_t tiles = (int64_t)d->R * d->S * ((d->C + *bm - 1) / *bm) * ((d->K + *bn - 1) / *bn);
  const int64_t P = (int64_t)d->N * d->OH * d->OW;
  const int KT = (int)((P + HSW_BK - 1) / HSW_BK);
  // 1x1: fill the resident-block slots once.  Gathered (3x3) layers re-read x once per tap and g once per tap and channel
  // tile, mostly out of the Infinity Cache: they run best cut into ~2000 short blocks (measured, scripts/bench_conv_hs.py:
  // RPN 3x3 1024->512 386 us unsplit, 250 us with 8 splits; block3 3x3 51 -> 42 us)
  const int slots = gather ? 2048 : ((*bm == 128) ? 256 : 512);
  int want = (int)((slots + tiles - 1) / tiles);
  const int max_split = KT / 4 > 0 ? (KT / 4 < 64 ? KT / 4 : 64) : 1;
  if (want > max_split) want = max_split;
  // every split writes an fp32 slab of the whole gradient, and the tail reads it again: keep the slabs within `hs_slab_cap` x the
  // operand bytes (default 2; RPN 3x3: 18.9 MB per split against 25.8 MB of operands -> 2 splits; k_tail_reduce fetched 554 MB per
  // launch before this cap, profiles/r03_f16hs_pmc_traffic.json)
  const double operand_bytes = 2.0 * (double)P * (d->C + d->K);
  const double slab_bytes = 4.0 * d->R * d->S * (double)d->C * d->K;
  const int capf = lmh_opt("hs_slab_cap");          // 0: no cap
  const int cap = (int)((double)capf * operand_bytes / slab_bytes);
  if (capf > 0 && want > cap) want = cap;
  if (want < 1) want = 1;
  if (g_force_splits) want = g_force_splits < KT ? g_force_splits : KT;
  *kt_per_split = (KT + want - 1) / want;
  *splits = (KT + *kt_per_split - 1) / *kt_per_split;
  return true;
}

extern "C" int lmh_conv2d_bwd_weight_hs(const lmh_conv_desc* d, const void* x, const void* g, float inv_scale, float* dw,
                                        float* colsum, void* ws, size_t ws_bytes, lmh_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  LMH_CHECK_ARG(x && g && dw);
  if (!hs_ok(d)) { lmh_set_error("lmh_conv2d_bwd_weight_hs: needs compute f16 / bf16, C %% 64 == 0, K %% 64 == 0"); return LMH_ERR_UNSUPPORTED; }
  int bm, bn, splits, kps;
  wgrad_hs_plan(d, &bm, &bn, &splits, &kps);
  if (ws_bytes < lmh_conv2d_bwd_weight_workspace_bytes(d) || ((splits > 1 || colsum) && !ws)) {
    lmh_set_error("lmh_conv2d_bwd_weight_hs: workspace too small");
    return LMH_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  g_prof_pending_bytes = 2.0 * ((double)d->N * d->H * d->W * d->C + (double)d->N * d->OH * d->OW * d->K) +
                         4.0 * d->R * d->S * d->C * d->K;
  const size_t slab_bytes = splits > 1 ? lmh_align_up((size_t)splits * d->R * d->S * d->C * d->K * sizeof(float), 256) : 256;
  float* out = splits > 1 ? reinterpret_cast<float*>(ws) : dw;
  float* cpart = colsum ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + slab_bytes) : nullptr;
  const int tx = d->R * d->S * ((d->C + bm - 1) / bm), ty = (d->K + bn - 1) / bn;
  const lmh_fastdiv dvw = lmh_make_fastdiv((uint32_t)d->OW), dvh = lmh_make_fastdiv((uint32_t)d->OH);
  // 1x1 / stride 1: source pixel == output pixel, no decode in the loader
  const bool gather = !(d->R == 1 && d->S == 1 && d->stride == 1 && d->pad_top == 0 && d->pad_left == 0 && d->OH == d->H &&
                        d->OW == d->W);
  const bool rs = lmh_opt("hs_wg_rs") != 0;      // tiles through registers (k_wgrad_hs_tr RS = 4) / LDS-DMA instructions
#define LAUNCH_BW_TR(DT_, BM_, G_, RS_)                                                                      \
  lmh_launch((k_wgrad_hs_tr<DT_, BM_, BM_, G_, RS_>), dim3(tx * ty * splits), dim3(256), 0, st, *d,       \
                     reinterpret_cast<const HT<DT_>::T*>(x), reinterpret_cast<const HT<DT_>::T*>(g), out, kps, dvw, dvh, \
                     inv_scale, tx, ty, splits, cpart)
#define LAUNCH_BW_TR_G(DT_, BM_, G_) do { if (rs) LAUNCH_BW_TR(DT_, BM_, G_, 4); else LAUNCH_BW_TR(DT_, BM_, G_, 0); } while (0)
#define LAUNCH_BW_TR_T(BM_)                                                                                  \
  do {                                                                                                       \
    if (d->compute == 1) { if (gather) LAUNCH_BW_TR_G(1, BM_, true); else LAUNCH_BW_TR_G(1, BM_, false); }    \
    else { if (gather) LAUNCH_BW_TR_G(2, BM_, true); else LAUNCH_BW_TR_G(2, BM_, false); }                    \
  } while (0)
  prof_begin(st);
  if (bm == 128) LAUNCH_BW_TR_T(128); else LAUNCH_BW_TR_T(64);
  prof_end(st, desc_flops(d), "k_wgrad_hs_tr<%d, %d, %d, %s, %d>", d->compute, bm, bn, gather ? "true" : "false", rs ? 4 : 0);
#undef LAUNCH_BW_TR_T
#undef LAUNCH_BW_TR_G
#undef LAUNCH_BW_TR
  if (g_lmh_defer_tail) {
    g_lmh_last_plan.slabs = splits > 1 ? reinterpret_cast<const float*>(ws) : nullptr;
    g_lmh_last_plan.splits = splits > 1 ? splits : 0;
    g_lmh_last_plan.colpart = cpart;
    g_lmh_last_plan.colrows = cpart ? splits : 0;
  } else if (splits > 1 || cpart) {
    const int64_t n = splits > 1 ? (int64_t)d->R * d->S * d->C * d->K : 0;
    const int nb_slab = n > 0 ? (int)((n / 4 + 255) / 256 + 1) : 0;
    const int nb_col = cpart ? (d->K + 31) / 32 : 0;
    lmh_launch(k_splitk_reduce, dim3(nb_slab + nb_col), dim3(256), 0, st, reinterpret_cast<const float*>(ws), n,
                       splits, dw, (const float*)cpart, colsum, d->K, nb_slab, splits);
  }
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
