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
#include "conv_common.h"
#include "conv_generic.h"

__device__ __attribute__((aligned(64))) float lmh_zero_page[16];  // zero-initialised: padding source

// ============================================================================
// forward (fast path: C % 32 == 0, K % 4 == 0)
// ============================================================================
template <int BM, int BN>
__global__ void __launch_bounds__(256)
k_conv_fwd(lmh_conv_desc d, const float* __restrict__ x, const float* __restrict__ w,
           const float* __restrict__ scale, const float* __restrict__ shift,
           const float* __restrict__ residual, float* __restrict__ y) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AJ = BM / 32, BJ = BN / 32;
  __shared__ __attribute__((aligned(16))) float As[2][BM * LDK];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = d.N * d.OH * d.OW, K = d.K, C = d.C;
  const int tiles_n = (K + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int CC = C / BK, KT = d.R * d.S * CC;

  const int kq = tid & 7;
  int a_n[AJ], a_ih0[AJ], a_iw0[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int p = m0 + (tid >> 3) + 32 * j;
    if (p < M) {
      const int ow = p % d.OW, t = p / d.OW;
      a_n[j] = t / d.OH;
      a_ih0[j] = (t % d.OH) * d.stride - d.pad_top;
      a_iw0[j] = ow * d.stride - d.pad_left;
    } else { a_n[j] = -1; a_ih0[j] = 0; a_iw0[j] = 0; }
  }
  const float* pa[AJ];
  int inca[AJ];
  auto setup_rs = [&](int rs) {
    const int r = rs / d.S, s = rs - r * d.S;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int ih = a_ih0[j] + r * d.dilation, iw = a_iw0[j] + s * d.dilation;
      const bool ok = a_n[j] >= 0 && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
      pa[j] = ok ? x + ((size_t)(a_n[j] * d.H + ih) * d.W + iw) * C + 4 * kq : lmh_zero_page;
      inca[j] = ok ? BK : 0;
    }
  };
  constexpr int BROW_T = BN / 4, BROW_STEP = 256 / BROW_T;
  const int bx4 = tid % BROW_T, bk = tid / BROW_T;
  const bool b_ok = (n0 + 4 * bx4) < K;
  const float* pb[BJ];
  const size_t incb = b_ok ? (size_t)BK * K : 0;
#pragma unroll
  for (int j = 0; j < BJ; ++j)
    pb[j] = b_ok ? w + (size_t)(bk + BROW_STEP * j) * K + n0 + 4 * bx4 : lmh_zero_page;

  float4 ra[AJ], rb[BJ];
  auto load_tile = [&]() {
#pragma unroll
    for (int j = 0; j < AJ; ++j) { ra[j] = *reinterpret_cast<const float4*>(pa[j]); pa[j] += inca[j]; }
#pragma unroll
    for (int j = 0; j < BJ; ++j) { rb[j] = *reinterpret_cast<const float4*>(pb[j]); pb[j] += incb; }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      *reinterpret_cast<float4*>(&As[buf][((tid >> 3) + 32 * j) * LDK + 4 * kq]) = ra[j];
#pragma unroll
    for (int j = 0; j < BJ; ++j)
      *reinterpret_cast<float4*>(&Bs[buf][(bk + BROW_STEP * j) * BN + 4 * bx4]) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[tm][tn][i] = 0.f;

  int rs = 0, cc = 0;
  setup_rs(0);
  load_tile();
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < KT;
    if (more) {
      if (++cc == CC) { cc = 0; setup_rs(++rs); }
      load_tile();
    }
    mfma_stage<TM, TN, true, false, LDK, BN>(As[cur], Bs[cur], acc, wm * (BM / 2), wn * (BN / 2), lane);
    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }

  const int col_l = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wn * (BN / 2) + tn * 32 + col_l;
    if (col >= K) continue;
    const float sc = scale ? scale[col] : 1.f;
    const float sh = shift ? shift[col] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = m0 + wm * (BM / 2) + tm * 32 + (i & 3) + 8 * (i >> 2) + rbase;
        if (row < M) {
          float v = acc[tm][tn][i];
          if (scale) v = v * sc;
          v = v + sh;
          if (residual) v += residual[(size_t)row * K + col];
          y[(size_t)row * K + col] = apply_act(v, d.act);
        }
      }
    }
  }
}

// ============================================================================
// backward data (fast path: K % 32 == 0)
// dx[p, c] = sum_{r,s,k} dy[opix(p,r,s), k] * kscale[k] * w[r,s,c,k]   (+ addend)
// ============================================================================
template <int BM, int BN>
__global__ void __launch_bounds__(256)
k_conv_bwd_data(lmh_conv_desc d, const float* __restrict__ dy, const float* __restrict__ w,
                const float* __restrict__ kscale, const float* __restrict__ addend,
                float* __restrict__ dx) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AJ = BM / 32, BJ = BN / 32;
  __shared__ __attribute__((aligned(16))) float As[2][BM * LDK];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = d.N * d.H * d.W, K = d.K, C = d.C;
  const int KC = K / BK, KT = d.R * d.S * KC;
  const int tiles_n = (C + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int kq = tid & 7;
  int a_n[AJ], a_h[AJ], a_w[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int p = m0 + (tid >> 3) + 32 * j;
    if (p < M) {
      const int ww = p % d.W, t = p / d.W;
      a_w[j] = ww + d.pad_left;
      a_h[j] = (t % d.H) + d.pad_top;
      a_n[j] = t / d.H;
    } else { a_n[j] = -1; a_h[j] = 0; a_w[j] = 0; }
  }
  const float* pa[AJ];
  int inca[AJ];
  const float* pb[BJ];
  int incb[BJ];
  auto setup_rs = [&](int rs) {
    const int r = rs / d.S, s = rs - r * d.S;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int th = a_h[j] - r * d.dilation, tw = a_w[j] - s * d.dilation;
      int oh = th, ow = tw;
      bool ok = a_n[j] >= 0 && th >= 0 && tw >= 0;
      if (d.stride > 1) {
        oh = th / d.stride; ow = tw / d.stride;
        ok = ok && (oh * d.stride == th) && (ow * d.stride == tw);
      }
      ok = ok && oh < d.OH && ow < d.OW;
      pa[j] = ok ? dy + ((size_t)(a_n[j] * d.OH + oh) * d.OW + ow) * K + 4 * kq : lmh_zero_page;
      inca[j] = ok ? BK : 0;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int c = n0 + (tid >> 3) + 32 * j;
      const bool ok = c < C;
      pb[j] = ok ? w + ((size_t)rs * C + c) * K + 4 * kq : lmh_zero_page;
      incb[j] = ok ? BK : 0;
    }
  };
  float4 ra[AJ], rb[BJ];
  int kcur = 0;  // k offset of the tile being loaded (for kscale)
  auto load_tile = [&]() {
    float4 ks = make_float4(1.f, 1.f, 1.f, 1.f);
    if (kscale) ks = *reinterpret_cast<const float4*>(kscale + kcur + 4 * kq);
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      float4 v = *reinterpret_cast<const float4*>(pa[j]);
      pa[j] += inca[j];
      v.x *= ks.x; v.y *= ks.y; v.z *= ks.z; v.w *= ks.w;
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) { rb[j] = *reinterpret_cast<const float4*>(pb[j]); pb[j] += incb[j]; }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      *reinterpret_cast<float4*>(&As[buf][((tid >> 3) + 32 * j) * LDK + 4 * kq]) = ra[j];
#pragma unroll
    for (int j = 0; j < BJ; ++j)
      *reinterpret_cast<float4*>(&Bs[buf][((tid >> 3) + 32 * j) * LDK + 4 * kq]) = rb[j];
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[tm][tn][i] = 0.f;
  int rs = 0, kc = 0;
  setup_rs(0);
  load_tile();
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < KT;
    if (more) {
      if (++kc == KC) { kc = 0; setup_rs(++rs); }
      kcur = kc * BK;
      load_tile();
    }
    mfma_stage<TM, TN, true, true, LDK, LDK>(As[cur], Bs[cur], acc, wm * (BM / 2), wn * (BN / 2), lane);
    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }
  const int col_l = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wn * (BN / 2) + tn * 32 + col_l;
    if (col >= C) continue;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = m0 + wm * (BM / 2) + tm * 32 + (i & 3) + 8 * (i >> 2) + rbase;
        if (row < M) {
          float v = acc[tm][tn][i];
          if (addend) v += addend[(size_t)row * C + col];
          dx[(size_t)row * C + col] = v;
        }
      }
  }
}

// ============================================================================
// backward weight (fast path: C % 4 == 0, K % 4 == 0)
// dw[rs, c, k] = sum_p x[pix(p,r,s), c] * dy[p, k];  split over gridDim.z
// ============================================================================
template <int BM, int BN>
__global__ void __launch_bounds__(256)
k_conv_bwd_weight(lmh_conv_desc d, const float* __restrict__ x, const float* __restrict__ dy,
                  float* __restrict__ out, int kt_per_split) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AJ = BM / 32, BJ = BN / 32;
  __shared__ __attribute__((aligned(16))) float As[2][BK * BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int P = d.N * d.OH * d.OW, K = d.K, C = d.C;
  const int tiles_c = (C + BM - 1) / BM;
  const int rs = blockIdx.x / tiles_c, m0 = (blockIdx.x % tiles_c) * BM;
  const int n0 = blockIdx.y * BN;
  const int r = rs / d.S, s = rs - r * d.S;
  const int KT_all = (P + BK - 1) / BK;
  const int kt_begin = blockIdx.z * kt_per_split;
  const int kt_end = min(KT_all, kt_begin + kt_per_split);
  constexpr int AROW_T = BM / 4, AROW_STEP = 256 / AROW_T;
  constexpr int BROW_T = BN / 4, BROW_STEP = 256 / BROW_T;
  const int ax4 = tid % AROW_T, ak = tid / AROW_T;
  const int bx4 = tid % BROW_T, bk = tid / BROW_T;
  const bool a_col_ok = (m0 + 4 * ax4) < C, b_col_ok = (n0 + 4 * bx4) < K;
  // incremental pixel decode for the A rows of this thread
  int pn[AJ], poh[AJ], pow_[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int p = kt_begin * BK + ak + AROW_STEP * j;
    pow_[j] = p % d.OW;
    const int t = p / d.OW;
    poh[j] = t % d.OH;
    pn[j] = t / d.OH;
  }
  int bp[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) bp[j] = kt_begin * BK + bk + BROW_STEP * j;
  const float* xb = x + m0 + 4 * ax4;
  const float* dyb = dy + n0 + 4 * bx4;
  float4 ra[AJ], rb[BJ];
  auto load_tile = [&]() {
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int ih = poh[j] * d.stride - d.pad_top + r * d.dilation;
      const int iw = pow_[j] * d.stride - d.pad_left + s * d.dilation;
      const bool ok = a_col_ok && pn[j] < d.N && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
      const float* p = ok ? xb + ((size_t)(pn[j] * d.H + ih) * d.W + iw) * C : lmh_zero_page;
      ra[j] = *reinterpret_cast<const float4*>(p);
      pow_[j] += BK;
      while (pow_[j] >= d.OW) { pow_[j] -= d.OW; ++poh[j]; }
      while (poh[j] >= d.OH) { poh[j] -= d.OH; ++pn[j]; }
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const bool ok = b_col_ok && bp[j] < P;
      const float* p = ok ? dyb + (size_t)bp[j] * K : lmh_zero_page;
      rb[j] = *reinterpret_cast<const float4*>(p);
      bp[j] += BK;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      *reinterpret_cast<float4*>(&As[buf][(ak + AROW_STEP * j) * BM + 4 * ax4]) = ra[j];
#pragma unroll
    for (int j = 0; j < BJ; ++j)
      *reinterpret_cast<float4*>(&Bs[buf][(bk + BROW_STEP * j) * BN + 4 * bx4]) = rb[j];
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[tm][tn][i] = 0.f;
  if (kt_begin < kt_end) {
    load_tile();
    store_tile(0);
  }
  __syncthreads();
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (kt - kt_begin) & 1;
    const bool more = kt + 1 < kt_end;
    if (more) load_tile();
    mfma_stage<TM, TN, false, false, BM, BN>(As[cur], Bs[cur], acc, wm * (BM / 2), wn * (BN / 2), lane);
    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }
  float* o = out + (size_t)blockIdx.z * ((size_t)d.R * d.S * C * K) + (size_t)rs * C * K;
  const int col_l = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + wn * (BN / 2) + tn * 32 + col_l;
    if (col >= K) continue;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = m0 + wm * (BM / 2) + tm * 32 + (i & 3) + 8 * (i >> 2) + rbase;
        if (row < C) o[(size_t)row * K + col] = acc[tm][tn][i];
      }
  }
}

// ============================================================================
// host dispatch
// ============================================================================
static int check_desc(const lmh_conv_desc* d) {
  LMH_CHECK_ARG(d != nullptr);
  LMH_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0);
  LMH_CHECK_ARG(d->OH > 0 && d->OW > 0 && d->stride > 0 && d->dilation > 0);
  LMH_CHECK_ARG(d->act >= 0 && d->act <= 2);
  LMH_CHECK_ARG((int64_t)d->N * d->OH * d->OW < (1ll << 31) && (int64_t)d->N * d->H * d->W < (1ll << 31));
  return LMH_OK;
}

// pick the block tile so the grid covers the chip at least ~2x when possible
static void pick_tile(int64_t M, int64_t Ncols, int* bm, int* bn) {
  const int64_t t128 = ((M + 127) / 128) * ((Ncols + 127) / 128);
  if (Ncols > 64 && t128 >= 512) { *bm = 128; *bn = 128; return; }
  const int64_t t12864 = ((M + 127) / 128) * ((Ncols + 63) / 64);
  if (t12864 >= 512) { *bm = 128; *bn = 64; return; }
  *bm = 64; *bn = 64;
}

static bool fwd_fast(const lmh_conv_desc* d) { return (d->C % BK) == 0 && (d->K & 3) == 0; }
static bool bwd_data_fast(const lmh_conv_desc* d) { return (d->K % BK) == 0; }
static bool bwd_weight_fast(const lmh_conv_desc* d) { return (d->C & 3) == 0 && (d->K & 3) == 0; }

extern "C" int lmh_conv2d_fwd(const lmh_conv_desc* d, const float* x, const float* w, const float* scale,
                              const float* shift, const float* residual, const float* in_sub, float* y,
                              lmh_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  LMH_CHECK_ARG(x && w && y);
  const int64_t M = (int64_t)d->N * d->OH * d->OW;
  const bool fast = fwd_fast(d);
  LMH_CHECK_ARG((d->C % BK) != 0 || in_sub == nullptr);
  int bm, bn;
  pick_tile(M, d->K, &bm, &bn);
  hipStream_t st = (hipStream_t)stream;
  const int grid = (int)(((M + bm - 1) / bm) * ((d->K + bn - 1) / bn));
#define LAUNCH_FWD(BM_, BN_)                                                                              \
  do {                                                                                                    \
    if (fast)                                                                                             \
      hipLaunchKernelGGL((k_conv_fwd<BM_, BN_>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, shift,    \
                         residual, y);                                                                    \
    else if ((d->C % BK) != 0)                                                                            \
      hipLaunchKernelGGL((k_conv_fwd_gen<BM_, BN_, true>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, \
                         shift, residual, in_sub, y);                                                     \
    else                                                                                                  \
      hipLaunchKernelGGL((k_conv_fwd_gen<BM_, BN_, false>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, \
                         shift, residual, in_sub, y);                                                     \
  } while (0)
  if (bm == 128 && bn == 128) LAUNCH_FWD(128, 128);
  else if (bm == 128) LAUNCH_FWD(128, 64);
  else LAUNCH_FWD(64, 64);
#undef LAUNCH_FWD
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

extern "C" int lmh_conv2d_bwd_data(const lmh_conv_desc* d, const float* dy, const float* w,
                                   const float* kscale, const float* addend, float* dx,
                                   lmh_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  LMH_CHECK_ARG(dy && w && dx);
  LMH_CHECK_ARG(d->R * d->S == 1 || (d->K % BK) == 0);
  const int64_t M = (int64_t)d->N * d->H * d->W;
  const bool fast = bwd_data_fast(d);
  int bm, bn;
  pick_tile(M, d->C, &bm, &bn);
  hipStream_t st = (hipStream_t)stream;
  const int grid = (int)(((M + bm - 1) / bm) * ((d->C + bn - 1) / bn));
#define LAUNCH_BD(BM_, BN_)                                                                                 \
  do {                                                                                                      \
    if (fast)                                                                                               \
      hipLaunchKernelGGL((k_conv_bwd_data<BM_, BN_>), dim3(grid), dim3(256), 0, st, *d, dy, w, kscale,      \
                         addend, dx);                                                                       \
    else                                                                                                    \
      hipLaunchKernelGGL((k_conv_bwd_data_gen<BM_, BN_>), dim3(grid), dim3(256), 0, st, *d, dy, w, kscale,  \
                         addend, dx);                                                                       \
  } while (0)
  if (bm == 128 && bn == 128) LAUNCH_BD(128, 128);
  else if (bm == 128) LAUNCH_BD(128, 64);
  else LAUNCH_BD(64, 64);
#undef LAUNCH_BD
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

static void bwd_weight_plan(const lmh_conv_desc* d, int* bm, int* bn, int* splits, int* kt_per_split) {
  *bm = (d->C >= 128) ? 128 : 64;
  *bn = (d->K >= 128) ? 128 : 64;
  const int64_t tiles = (int64_t)d->R * d->S * ((d->C + *bm - 1) / *bm) * ((d->K + *bn - 1) / *bn);
  const int64_t P = (int64_t)d->N * d->OH * d->OW;
  const int KT = (int)((P + BK - 1) / BK);
  int64_t want = (768 + tiles - 1) / tiles;  // ~3 blocks per CU
  int64_t max_split = KT / 8 > 0 ? KT / 8 : 1;  // >= 8 K-steps per block
  if (want > max_split) want = max_split;
  if (want < 1) want = 1;
  *kt_per_split = (int)((KT + want - 1) / want);
  *splits = (KT + *kt_per_split - 1) / *kt_per_split;
}

extern "C" int lmh_conv2d_kernel_id(const lmh_conv_desc* d, int op) {
  if (!d) return -1;
  int bm = 0, bn = 0;
  if (op == 0) {
    pick_tile((int64_t)d->N * d->OH * d->OW, d->K, &bm, &bn);
    return bm * 1000 + bn + (fwd_fast(d) ? 0 : 1000000);
  }
  if (op == 1) {
    pick_tile((int64_t)d->N * d->H * d->W, d->C, &bm, &bn);
    return bm * 1000 + bn + (bwd_data_fast(d) ? 0 : 1000000);
  }
  int splits, kps;
  bwd_weight_plan(d, &bm, &bn, &splits, &kps);
  return bm * 1000 + bn + (bwd_weight_fast(d) ? 0 : 1000000);
}

extern "C" size_t lmh_conv2d_bwd_weight_workspace_bytes(const lmh_conv_desc* d) {
  if (!d) return 0;
  int bm, bn, splits, kps;
  bwd_weight_plan(d, &bm, &bn, &splits, &kps);
  if (splits <= 1) return 256;
  return lmh_align_up((size_t)splits * d->R * d->S * d->C * d->K * sizeof(float), 256);
}

extern "C" int lmh_conv2d_bwd_weight(const lmh_conv_desc* d, const float* x, const float* dy, float* dw,
                                     void* ws, size_t ws_bytes, lmh_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  LMH_CHECK_ARG(x && dy && dw);
  LMH_CHECK_ARG((d->C & 3) == 0);
  int bm, bn, splits, kps;
  bwd_weight_plan(d, &bm, &bn, &splits, &kps);
  if (ws_bytes < lmh_conv2d_bwd_weight_workspace_bytes(d) || (splits > 1 && !ws)) {
    lmh_set_error("lmh_conv2d_bwd_weight: workspace too small");
    return LMH_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* out = splits > 1 ? reinterpret_cast<float*>(ws) : dw;
  const bool fast = bwd_weight_fast(d);
  dim3 grid(d->R * d->S * ((d->C + bm - 1) / bm), (d->K + bn - 1) / bn, splits);
#define LAUNCH_BW(BM_, BN_)                                                                              \
  do {                                                                                                   \
    if (fast)                                                                                            \
      hipLaunchKernelGGL((k_conv_bwd_weight<BM_, BN_>), grid, dim3(256), 0, st, *d, x, dy, out, kps);    \
    else                                                                                                 \
      hipLaunchKernelGGL((k_conv_bwd_weight_gen<BM_, BN_>), grid, dim3(256), 0, st, *d, x, dy, out, kps); \
  } while (0)
  if (bm == 128 && bn == 128) LAUNCH_BW(128, 128);
  else if (bm == 128) LAUNCH_BW(128, 64);
  else if (bn == 128) LAUNCH_BW(64, 128);
  else LAUNCH_BW(64, 64);
#undef LAUNCH_BW
  if (splits > 1) {
    const int64_t n = (int64_t)d->R * d->S * d->C * d->K;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n / 4 + 255) / 256 + 1)), dim3(256), 0, st,
                       reinterpret_cast<const float*>(ws), n, splits, dw);
  }
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
