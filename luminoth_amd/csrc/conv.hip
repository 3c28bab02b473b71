// NHWC fp32 implicit-GEMM convolution family on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Replaces the TF conv2d / slim conv2d_same (+ frozen BN, bias, ReLU/ReLU6,
// residual add) kernels the reference reaches through tf.contrib.slim and
// Sonnet: forward, backward-data and backward-weight.
//
// Design (MI355X): fp32 in / fp32 accumulate is the parity dtype (north_star:
// 1e-4 vs the TF-CPU fp32 path).  gfx950 has no xf32, so the matrix pipe runs
// at the fp32 rate: 32x32x2 = 64 cycles/issue/SIMD, 157 TFLOP/s chip peak.  One
// MFMA therefore covers ~64 cycles of LDS/VMEM work and the kernel is
// matrix-pipe bound once the tile loop is software pipelined:
//   * 256-thread workgroups (4 waves, 2x2), block tiles 128x128 / 128x64 /
//     64x64 (picked so the grid is >= ~2 waves of the 256 CUs), BK = 32;
//   * global -> registers prefetch of tile t+1 is issued before the MFMAs of
//     tile t (HBM/L2 latency hides under 4096 matrix cycles per stage);
//   * operands whose GEMM-K axis is contiguous in memory (activations in the
//     forward/backward-data gathers, weights in backward-data) sit in LDS as
//     [row][BK+4] and are read with conflict-free ds_read_b128 (4 k-steps per
//     read, K order permuted identically for A and B); K-major operands
//     (HWIO weights in forward, both operands in backward-weight) sit as
//     [BK][cols] and are read with conflict-free ds_read_b32;
//   * blockIdx -> tile mapping is XCD-aware (each XCD walks a contiguous
//     range of tiles so operand panels stay in its private L2).
#include "lmh_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 32
#define LDK (BK + 4)  // row stride (floats) of K-contiguous LDS tiles: 9*m mod 16 slots, conflict-free b128

// bijective XCD remap: hardware places block b on XCD b % 8
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

// ---- MFMA stage: A tile (BM x 32), B tile (32 x BN) from LDS -----------------
template <int TM, int TN, bool A_KC, bool B_KC, int LDA, int LDB>
__device__ __forceinline__ void mfma_stage(const float* __restrict__ As, const float* __restrict__ Bs,
                                           f32x16 (&acc)[TM][TN], int a_off, int b_off, int lane) {
  const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int kb = 16 * h + 4 * g;
    float a[TM][4], b[TN][4];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      if (A_KC) {
        const float4 v = *reinterpret_cast<const float4*>(&As[(a_off + tm * 32 + l31) * LDA + kb]);
        a[tm][0] = v.x; a[tm][1] = v.y; a[tm][2] = v.z; a[tm][3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[tm][i] = As[(kb + i) * LDA + a_off + tm * 32 + l31];
      }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      if (B_KC) {
        const float4 v = *reinterpret_cast<const float4*>(&Bs[(b_off + tn * 32 + l31) * LDB + kb]);
        b[tn][0] = v.x; b[tn][1] = v.y; b[tn][2] = v.z; b[tn][3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) b[tn][i] = Bs[(kb + i) * LDB + b_off + tn * 32 + l31];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][i], b[tn][i], acc[tm][tn], 0, 0, 0);
  }
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return fminf(fmaxf(v, 0.f), 6.f);
  return v;
}

// ============================================================================
// forward:  y[p, k] = act( sum_{r,s,c} x[pix(p,r,s), c] * w[r,s,c,k] * scale[k] + shift[k] + res[p,k] )
// GEMM M = N*OH*OW, N = K, Kg = R*S*C.   A: gather, K-contiguous.  B: HWIO, K-major.
// ============================================================================
template <int BM, int BN, bool GENERIC_A>
__global__ void __launch_bounds__(256)
k_conv_fwd(lmh_conv_desc d, const float* __restrict__ x, const float* __restrict__ w,
           const float* __restrict__ scale, const float* __restrict__ shift,
           const float* __restrict__ residual, const float* __restrict__ in_sub, float* __restrict__ y) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AJ = BM / 32;  // A float4 per thread per stage
  constexpr int BJ = BN / 32;  // B float4 per thread per stage
  __shared__ __attribute__((aligned(16))) float As[BM * LDK];
  __shared__ __attribute__((aligned(16))) float Bs[BK * BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = d.N * d.OH * d.OW, Kg = d.R * d.S * d.C, K = d.K;
  const int tiles_n = (K + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

  // per-thread A rows
  const int kq = tid & 7;
  int a_n[AJ], a_ih0[AJ], a_iw0[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int p = m0 + (tid >> 3) + 32 * j;
    if (p < M) {
      const int ow = p % d.OW, t = p / d.OW;
      const int oh = t % d.OH;
      a_n[j] = t / d.OH;
      a_ih0[j] = oh * d.stride - d.pad_top;
      a_iw0[j] = ow * d.stride - d.pad_left;
    } else {
      a_n[j] = -1; a_ih0[j] = 0; a_iw0[j] = 0;
    }
  }
  // per-thread B slots
  constexpr int BROW_T = BN / 4;         // threads per K-major row
  constexpr int BROW_STEP = 256 / BROW_T;  // rows per pass
  const int bx4 = tid % BROW_T, bk = tid / BROW_T;
  const bool vecB = (K & 3) == 0;

  float4 ra[AJ], rb[BJ];
  auto load_tile = [&](int kt) {
    const int kg0 = kt * BK;
    if (!GENERIC_A) {
      const int rs = kg0 / d.C, c0 = kg0 - rs * d.C;
      const int r = rs / d.S, s = rs - r * d.S;
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int ih = a_ih0[j] + r * d.dilation, iw = a_iw0[j] + s * d.dilation;
        const bool ok = a_n[j] >= 0 && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W && kg0 < Kg;
        ra[j] = ok ? *reinterpret_cast<const float4*>(
                         x + ((size_t)(a_n[j] * d.H + ih) * d.W + iw) * d.C + c0 + 4 * kq)
                   : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kg = kg0 + 4 * kq + e;
          float val = 0.f;
          if (kg < Kg && a_n[j] >= 0) {
            const int rs = kg / d.C, c = kg - rs * d.C;
            const int r = rs / d.S, s = rs - r * d.S;
            const int ih = a_ih0[j] + r * d.dilation, iw = a_iw0[j] + s * d.dilation;
            if ((unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W) {
              val = x[((size_t)(a_n[j] * d.H + ih) * d.W + iw) * d.C + c];
              if (in_sub) val -= in_sub[c];
            }
          }
          v[e] = val;
        }
        ra[j] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int kg = kg0 + bk + BROW_STEP * j;
      const int n = n0 + 4 * bx4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kg < Kg) {
        const float* wp = w + (size_t)kg * K + n;
        if (vecB && n + 3 < K) {
          v = *reinterpret_cast<const float4*>(wp);
        } else {
          if (n < K) v.x = wp[0];
          if (n + 1 < K) v.y = wp[1];
          if (n + 2 < K) v.z = wp[2];
          if (n + 3 < K) v.w = wp[3];
        }
      }
      rb[j] = v;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      *reinterpret_cast<float4*>(&As[((tid >> 3) + 32 * j) * LDK + 4 * kq]) = ra[j];
#pragma unroll
    for (int j = 0; j < BJ; ++j)
      *reinterpret_cast<float4*>(&Bs[(bk + BROW_STEP * j) * BN + 4 * bx4]) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[tm][tn][i] = 0.f;

  const int KT = (Kg + BK - 1) / BK;
  load_tile(0);
  for (int kt = 0; kt < KT; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < KT) load_tile(kt + 1);
    mfma_stage<TM, TN, true, false, LDK, BN>(As, Bs, acc, wm * (BM / 2), wn * (BN / 2), lane);
    __syncthreads();
  }

  // epilogue
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
// backward data: dx[p, c] = sum_{r,s,k} dy[opix(p,r,s), k] * kscale[k] * w[r,s,c,k]   (+ addend)
// GEMM M = N*H*W, N = C, Kg = R*S*K.  A: dy gather (K-contiguous).  B: w[rs][c][k] (K-contiguous).
// ============================================================================
template <int BM, int BN>
__global__ void __launch_bounds__(256)
k_conv_bwd_data(lmh_conv_desc d, const float* __restrict__ dy, const float* __restrict__ w,
                const float* __restrict__ kscale, const float* __restrict__ addend,
                float* __restrict__ dx) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AJ = BM / 32, BJ = BN / 32;
  __shared__ __attribute__((aligned(16))) float As[BM * LDK];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = d.N * d.H * d.W, K = d.K, C = d.C;
  const int KTk = (K + BK - 1) / BK;  // k-tiles per (r,s)
  const int KT = d.R * d.S * KTk;
  const int tiles_n = (C + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int kq = tid & 7;
  const bool vecK = (K & 3) == 0;
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
  float4 ra[AJ], rb[BJ];
  auto load_tile = [&](int kt) {
    const int rs = kt / KTk, k0 = (kt - rs * KTk) * BK + 4 * kq;
    const int r = rs / d.S, s = rs - r * d.S;
    const bool kok = k0 < K;
    float4 ks = make_float4(1.f, 1.f, 1.f, 1.f);
    if (kscale && kok) {
      ks.x = kscale[k0];
      if (k0 + 1 < K) ks.y = kscale[k0 + 1];
      if (k0 + 2 < K) ks.z = kscale[k0 + 2];
      if (k0 + 3 < K) ks.w = kscale[k0 + 3];
    }
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int th = a_h[j] - r * d.dilation, tw = a_w[j] - s * d.dilation;
      int oh = th, ow = tw;
      bool ok = a_n[j] >= 0 && kok && th >= 0 && tw >= 0;
      if (d.stride > 1) {
        oh = th / d.stride; ow = tw / d.stride;
        ok = ok && (oh * d.stride == th) && (ow * d.stride == tw);
      }
      ok = ok && oh < d.OH && ow < d.OW;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        const float* src = dy + ((size_t)(a_n[j] * d.OH + oh) * d.OW + ow) * K + k0;
        if (vecK) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          v.x = src[0];
          if (k0 + 1 < K) v.y = src[1];
          if (k0 + 2 < K) v.z = src[2];
          if (k0 + 3 < K) v.w = src[3];
        }
        v.x *= ks.x; v.y *= ks.y; v.z *= ks.z; v.w *= ks.w;
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int c = n0 + (tid >> 3) + 32 * j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < C && kok) {
        const float* src = w + ((size_t)rs * C + c) * K + k0;
        if (vecK) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          v.x = src[0];
          if (k0 + 1 < K) v.y = src[1];
          if (k0 + 2 < K) v.z = src[2];
          if (k0 + 3 < K) v.w = src[3];
        }
      }
      rb[j] = v;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      *reinterpret_cast<float4*>(&As[((tid >> 3) + 32 * j) * LDK + 4 * kq]) = ra[j];
#pragma unroll
    for (int j = 0; j < BJ; ++j)
      *reinterpret_cast<float4*>(&Bs[((tid >> 3) + 32 * j) * LDK + 4 * kq]) = rb[j];
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[tm][tn][i] = 0.f;
  load_tile(0);
  for (int kt = 0; kt < KT; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < KT) load_tile(kt + 1);
    mfma_stage<TM, TN, true, true, LDK, LDK>(As, Bs, acc, wm * (BM / 2), wn * (BN / 2), lane);
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
// backward weight: dw[rs, c, k] = sum_p x[pix(p,r,s), c] * dy[p, k]
// GEMM (per r,s) M = C, N = K, Kg = N*OH*OW (split over gridDim.z).  Both K-major.
// ============================================================================
template <int BM, int BN>
__global__ void __launch_bounds__(256)
k_conv_bwd_weight(lmh_conv_desc d, const float* __restrict__ x, const float* __restrict__ dy,
                  float* __restrict__ out, int kt_per_split) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AJ = BM / 32, BJ = BN / 32;
  __shared__ __attribute__((aligned(16))) float As[BK * BM];
  __shared__ __attribute__((aligned(16))) float Bs[BK * BN];
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
  const bool vecK = (K & 3) == 0;
  float4 ra[AJ], rb[BJ];
  auto load_tile = [&](int kt) {
    const int p0 = kt * BK;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int p = p0 + ak + AROW_STEP * j;
      const int c = m0 + 4 * ax4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p < P && c < C) {
        const int ow = p % d.OW, t = p / d.OW;
        const int oh = t % d.OH, n = t / d.OH;
        const int ih = oh * d.stride - d.pad_top + r * d.dilation;
        const int iw = ow * d.stride - d.pad_left + s * d.dilation;
        if ((unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W)
          v = *reinterpret_cast<const float4*>(x + ((size_t)(n * d.H + ih) * d.W + iw) * C + c);
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int p = p0 + bk + BROW_STEP * j;
      const int n = n0 + 4 * bx4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p < P && n < K) {
        const float* src = dy + (size_t)p * K + n;
        if (vecK && n + 3 < K) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          v.x = src[0];
          if (n + 1 < K) v.y = src[1];
          if (n + 2 < K) v.z = src[2];
          if (n + 3 < K) v.w = src[3];
        }
      }
      rb[j] = v;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      *reinterpret_cast<float4*>(&As[(ak + AROW_STEP * j) * BM + 4 * ax4]) = ra[j];
#pragma unroll
    for (int j = 0; j < BJ; ++j)
      *reinterpret_cast<float4*>(&Bs[(bk + BROW_STEP * j) * BN + 4 * bx4]) = rb[j];
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[tm][tn][i] = 0.f;
  if (kt_begin < kt_end) load_tile(kt_begin);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < kt_end) load_tile(kt + 1);
    mfma_stage<TM, TN, false, false, BM, BN>(As, Bs, acc, wm * (BM / 2), wn * (BN / 2), lane);
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

// deterministic split-K reduction: dw[i] = sum_s part[s][i]
__global__ void __launch_bounds__(256)
k_splitk_reduce(const float* __restrict__ part, int64_t n, int splits, float* __restrict__ out) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 3 < n) {
    float4 a = *reinterpret_cast<const float4*>(part + i);
    for (int s = 1; s < splits; ++s) {
      const float4 b = *reinterpret_cast<const float4*>(part + (size_t)s * n + i);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    *reinterpret_cast<float4*>(out + i) = a;
  } else {
    for (int64_t e = i; e < n; ++e) {
      float a = part[e];
      for (int s = 1; s < splits; ++s) a += part[(size_t)s * n + e];
      out[e] = a;
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

extern "C" int lmh_conv2d_fwd(const lmh_conv_desc* d, const float* x, const float* w, const float* scale,
                              const float* shift, const float* residual, const float* in_sub, float* y,
                              lmh_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  LMH_CHECK_ARG(x && w && y);
  const int64_t M = (int64_t)d->N * d->OH * d->OW;
  const bool generic = (d->C % BK) != 0;
  LMH_CHECK_ARG(generic || in_sub == nullptr);
  int bm, bn;
  pick_tile(M, d->K, &bm, &bn);
  hipStream_t st = (hipStream_t)stream;
  const int grid = (int)(((M + bm - 1) / bm) * ((d->K + bn - 1) / bn));
#define LAUNCH_FWD(BM_, BN_)                                                                         \
  do {                                                                                               \
    if (generic)                                                                                     \
      hipLaunchKernelGGL((k_conv_fwd<BM_, BN_, true>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, \
                         shift, residual, in_sub, y);                                                \
    else                                                                                             \
      hipLaunchKernelGGL((k_conv_fwd<BM_, BN_, false>), dim3(grid), dim3(256), 0, st, *d, x, w, scale, \
                         shift, residual, in_sub, y);                                                \
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
  int bm, bn;
  pick_tile(M, d->C, &bm, &bn);
  hipStream_t st = (hipStream_t)stream;
  const int grid = (int)(((M + bm - 1) / bm) * ((d->C + bn - 1) / bn));
  if (bm == 128 && bn == 128)
    hipLaunchKernelGGL((k_conv_bwd_data<128, 128>), dim3(grid), dim3(256), 0, st, *d, dy, w, kscale, addend, dx);
  else if (bm == 128)
    hipLaunchKernelGGL((k_conv_bwd_data<128, 64>), dim3(grid), dim3(256), 0, st, *d, dy, w, kscale, addend, dx);
  else
    hipLaunchKernelGGL((k_conv_bwd_data<64, 64>), dim3(grid), dim3(256), 0, st, *d, dy, w, kscale, addend, dx);
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
    return bm * 1000 + bn + ((d->C % BK) != 0 ? 1000000 : 0);
  }
  if (op == 1) {
    pick_tile((int64_t)d->N * d->H * d->W, d->C, &bm, &bn);
    return bm * 1000 + bn;
  }
  int splits, kps;
  bwd_weight_plan(d, &bm, &bn, &splits, &kps);
  return bm * 1000 + bn;
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
  dim3 grid(d->R * d->S * ((d->C + bm - 1) / bm), (d->K + bn - 1) / bn, splits);
  if (bm == 128 && bn == 128)
    hipLaunchKernelGGL((k_conv_bwd_weight<128, 128>), grid, dim3(256), 0, st, *d, x, dy, out, kps);
  else if (bm == 128)
    hipLaunchKernelGGL((k_conv_bwd_weight<128, 64>), grid, dim3(256), 0, st, *d, x, dy, out, kps);
  else if (bn == 128)
    hipLaunchKernelGGL((k_conv_bwd_weight<64, 128>), grid, dim3(256), 0, st, *d, x, dy, out, kps);
  else
    hipLaunchKernelGGL((k_conv_bwd_weight<64, 64>), grid, dim3(256), 0, st, *d, x, dy, out, kps);
  if (splits > 1) {
    const int64_t n = (int64_t)d->R * d->S * d->C * d->K;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n / 4 + 255) / 256 + 1)), dim3(256), 0, st,
                       reinterpret_cast<const float*>(ws), n, splits, dw);
  }
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// ============================================================================
// elementwise helpers
// ============================================================================
// ---------------------------------------------------------------------------
// g = dy * act'(y) and per-channel column sums (dbeta / dbias), two stages,
// deterministic: every block reduces its row slab into LDS and writes ONE
// partial row [K]; k_colsum_finish adds the partial rows per column.
// ---------------------------------------------------------------------------
#define ACT_MAX_K 4096
template <bool VEC>
__global__ void __launch_bounds__(256)
k_act_bwd(const float* __restrict__ dy, const float* __restrict__ y, int act, int64_t rows, int K,
          float* __restrict__ g, float* __restrict__ partial, int rows_per_block) {
  __shared__ float scol[ACT_MAX_K];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  const float hi = (act == 2) ? 6.f : INFINITY;
  if (partial) {
    for (int c = threadIdx.x; c < K; c += 256) scol[c] = 0.f;
    __syncthreads();
  }
  if (VEC) {
    const int K4 = K >> 2;
    const int tpr = min(K4, 256), rstep = 256 / tpr;
    const int c4 = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
    if (rsub < rstep) {
      for (int cc = c4; cc < K4; cc += tpr) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t r = r0 + rsub; r < r1; r += rstep) {
          const size_t o = (size_t)r * K + 4 * cc;
          float4 d4 = *reinterpret_cast<const float4*>(dy + o);
          if (act) {
            const float4 y4 = *reinterpret_cast<const float4*>(y + o);
            d4.x = (y4.x > 0.f && y4.x < hi) ? d4.x : 0.f;
            d4.y = (y4.y > 0.f && y4.y < hi) ? d4.y : 0.f;
            d4.z = (y4.z > 0.f && y4.z < hi) ? d4.z : 0.f;
            d4.w = (y4.w > 0.f && y4.w < hi) ? d4.w : 0.f;
          }
          if (g) *reinterpret_cast<float4*>(g + o) = d4;
          s.x += d4.x; s.y += d4.y; s.z += d4.z; s.w += d4.w;
        }
        if (partial) {
          atomicAdd(&scol[4 * cc + 0], s.x);
          atomicAdd(&scol[4 * cc + 1], s.y);
          atomicAdd(&scol[4 * cc + 2], s.z);
          atomicAdd(&scol[4 * cc + 3], s.w);
        }
      }
    }
  } else {
    for (int c = threadIdx.x; c < K; c += 256) {
      float sacc = 0.f;
      for (int64_t r = r0; r < r1; ++r) {
        const size_t o = (size_t)r * K + c;
        float d = dy[o];
        if (act) { const float yv = y[o]; d = (yv > 0.f && yv < hi) ? d : 0.f; }
        if (g) g[o] = d;
        sacc += d;
      }
      if (partial) scol[c] = sacc;
    }
  }
  if (partial) {
    __syncthreads();
    for (int c = threadIdx.x; c < K; c += 256) partial[(size_t)blockIdx.x * K + c] = scol[c];
  }
}

// out[c] = sum_b partial[b][c]  (sequential over b: deterministic)
__global__ void __launch_bounds__(256)
k_colsum_finish(const float* __restrict__ partial, int nb, int K, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= K) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = 0;
  for (; b + 3 < nb; b += 4) {
    s0 += partial[(size_t)b * K + c];
    s1 += partial[(size_t)(b + 1) * K + c];
    s2 += partial[(size_t)(b + 2) * K + c];
    s3 += partial[(size_t)(b + 3) * K + c];
  }
  for (; b < nb; ++b) s0 += partial[(size_t)b * K + c];
  out[c] = (s0 + s1) + (s2 + s3);
}

static int act_bwd_blocks(int64_t rows, int K, int* rpb_out) {
  int rpb = (int)((rows + 511) / 512);
  const int k4 = (K & 3) ? K : (K >> 2);
  const int rstep = k4 >= 256 ? 1 : 256 / k4;
  if (rpb < 4 * rstep) rpb = 4 * rstep;
  *rpb_out = rpb;
  return (int)((rows + rpb - 1) / rpb);
}

extern "C" size_t lmh_act_bwd_workspace_bytes(int64_t rows, int K) {
  int rpb;
  const int nb = act_bwd_blocks(rows, K, &rpb);
  return lmh_align_up((size_t)nb * K * sizeof(float), 256);
}

extern "C" int lmh_act_bwd(const float* dy, const float* y, int act, int64_t rows, int K, float* g,
                           float* colsum, void* ws, size_t ws_bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(dy && rows > 0 && K > 0 && K <= ACT_MAX_K);
  LMH_CHECK_ARG(act == 0 || y != nullptr);
  LMH_CHECK_ARG(g || colsum);
  int rpb;
  const int nb = act_bwd_blocks(rows, K, &rpb);
  float* partial = nullptr;
  if (colsum) {
    if (!ws || ws_bytes < lmh_act_bwd_workspace_bytes(rows, K)) {
      lmh_set_error("lmh_act_bwd: workspace too small");
      return LMH_ERR_WORKSPACE;
    }
    partial = reinterpret_cast<float*>(ws);
  }
  hipStream_t st = (hipStream_t)stream;
  if ((K & 3) != 0)
    hipLaunchKernelGGL((k_act_bwd<false>), dim3(nb), dim3(256), 0, st, dy, y, act, rows, K, g, partial, rpb);
  else
    hipLaunchKernelGGL((k_act_bwd<true>), dim3(nb), dim3(256), 0, st, dy, y, act, rows, K, g, partial, rpb);
  if (colsum)
    hipLaunchKernelGGL(k_colsum_finish, dim3((K + 255) / 256), dim3(256), 0, st, partial, nb, K, colsum);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// BN (frozen) parameter gradients from the raw weight gradient, two stages:
//   partial[b][k] = sum_{i in slab b} w[i,k]*dw_raw[i,k];  dw[i,k] = dw_raw[i,k]*scale[k]
//   dgamma[k] = rstd[k]*(sum_b partial[b][k] - mean[k]*dbeta[k])
__global__ void __launch_bounds__(256)
k_bn_wdot(const float* __restrict__ w, float* __restrict__ dw, const float* __restrict__ scale, int64_t rsc,
          int K, float* __restrict__ partial, int rows_per_block) {
  __shared__ float scol[ACT_MAX_K];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(rsc, r0 + rows_per_block);
  for (int c = threadIdx.x; c < K; c += 256) scol[c] = 0.f;
  __syncthreads();
  const int K4 = K >> 2;
  const int tpr = min(K4, 256), rstep = 256 / tpr;
  const int c4 = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
  if (rsub < rstep) {
    for (int cc = c4; cc < K4; cc += tpr) {
      const float4 sc = *reinterpret_cast<const float4*>(scale + 4 * cc);
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int64_t r = r0 + rsub; r < r1; r += rstep) {
        const size_t o = (size_t)r * K + 4 * cc;
        const float4 wv = *reinterpret_cast<const float4*>(w + o);
        float4 d = *reinterpret_cast<const float4*>(dw + o);
        s.x += wv.x * d.x; s.y += wv.y * d.y; s.z += wv.z * d.z; s.w += wv.w * d.w;
        d.x *= sc.x; d.y *= sc.y; d.z *= sc.z; d.w *= sc.w;
        *reinterpret_cast<float4*>(dw + o) = d;
      }
      atomicAdd(&scol[4 * cc + 0], s.x);
      atomicAdd(&scol[4 * cc + 1], s.y);
      atomicAdd(&scol[4 * cc + 2], s.z);
      atomicAdd(&scol[4 * cc + 3], s.w);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < K; c += 256) partial[(size_t)blockIdx.x * K + c] = scol[c];
}

__global__ void __launch_bounds__(256)
k_bn_finish(const float* __restrict__ partial, int nb, int K, const float* __restrict__ dbeta,
            const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dgamma) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= K) return;
  float s = 0.f;
  for (int b = 0; b < nb; ++b) s += partial[(size_t)b * K + c];
  dgamma[c] = rstd[c] * (s - mean[c] * dbeta[c]);
}

static int bn_blocks(int64_t rsc, int K, int* rpb_out) {
  int rpb = (int)((rsc + 255) / 256);
  const int k4 = K >> 2;
  const int rstep = k4 >= 256 ? 1 : 256 / k4;
  if (rpb < 2 * rstep) rpb = 2 * rstep;
  *rpb_out = rpb;
  return (int)((rsc + rpb - 1) / rpb);
}

extern "C" size_t lmh_bn_param_grads_workspace_bytes(int64_t rsc, int K) {
  int rpb;
  return lmh_align_up((size_t)bn_blocks(rsc, K, &rpb) * K * sizeof(float), 256);
}

extern "C" int lmh_bn_param_grads(const float* w, float* dw_raw_inout, const float* dbeta,
                                  const float* mean, const float* rstd, const float* scale, int64_t rsc,
                                  int K, float* dgamma, void* ws, size_t ws_bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(w && dw_raw_inout && dbeta && mean && rstd && scale && dgamma && rsc > 0 && K > 0);
  LMH_CHECK_ARG((K & 3) == 0 && K <= ACT_MAX_K);
  if (!ws || ws_bytes < lmh_bn_param_grads_workspace_bytes(rsc, K)) {
    lmh_set_error("lmh_bn_param_grads: workspace too small");
    return LMH_ERR_WORKSPACE;
  }
  int rpb;
  const int nb = bn_blocks(rsc, K, &rpb);
  hipStream_t st = (hipStream_t)stream;
  float* partial = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(k_bn_wdot, dim3(nb), dim3(256), 0, st, w, dw_raw_inout, scale, rsc, K, partial, rpb);
  hipLaunchKernelGGL(k_bn_finish, dim3((K + 255) / 256), dim3(256), 0, st, partial, nb, K, dbeta, mean, rstd,
                     dgamma);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// ---- max pool (NHWC) --------------------------------------------------------
__global__ void __launch_bounds__(256)
k_maxpool_fwd(const float* __restrict__ x, int N, int H, int W, int C, int ks, int stride, int pt, int pl,
              int OH, int OW, float* __restrict__ y) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)N * OH * OW * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    int64_t t = i / C4;
    const int ow = (int)(t % OW); t /= OW;
    const int oh = (int)(t % OH);
    const int n = (int)(t / OH);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int r = 0; r < ks; ++r) {
      const int ih = oh * stride - pt + r;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int s = 0; s < ks; ++s) {
        const int iw = ow * stride - pl + s;
        if ((unsigned)iw >= (unsigned)W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)(n * H + ih) * W + iw) * C + 4 * c4);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    *reinterpret_cast<float4*>(y + (size_t)i * 4) = m;
  }
}

// dx must be zeroed by the caller; gradient goes to the first max in (r,s) scan order.
__global__ void __launch_bounds__(256)
k_maxpool_bwd(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy, int N,
              int H, int W, int C, int ks, int stride, int pt, int pl, int OH, int OW,
              float* __restrict__ dx) {
  const int64_t total = (int64_t)N * OH * OW * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t t = i / C;
    const int ow = (int)(t % OW); t /= OW;
    const int oh = (int)(t % OH);
    const int n = (int)(t / OH);
    const float yv = y[i], g = dy[i];
    bool done = false;
    for (int r = 0; r < ks && !done; ++r) {
      const int ih = oh * stride - pt + r;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int s = 0; s < ks && !done; ++s) {
        const int iw = ow * stride - pl + s;
        if ((unsigned)iw >= (unsigned)W) continue;
        const size_t o = ((size_t)(n * H + ih) * W + iw) * C + c;
        if (x[o] == yv) { unsafeAtomicAdd(dx + o, g); done = true; }
      }
    }
  }
}

extern "C" int lmh_maxpool_fwd(const float* x, int N, int H, int W, int C, int ksize, int stride,
                               int pad_top, int pad_left, int OH, int OW, float* y, lmh_stream_t stream) {
  LMH_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && ksize > 0 && stride > 0);
  const int64_t total = (int64_t)N * OH * OW * (C / 4);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(k_maxpool_fwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, ksize,
                     stride, pad_top, pad_left, OH, OW, y);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

extern "C" int lmh_maxpool_bwd(const float* x, const float* y, const float* dy, int N, int H, int W, int C,
                               int ksize, int stride, int pad_top, int pad_left, int OH, int OW, float* dx,
                               lmh_stream_t stream) {
  LMH_CHECK_ARG(x && y && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && ksize > 0 && stride > 0);
  const int64_t total = (int64_t)N * OH * OW * C;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(k_maxpool_bwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, dy, N, H, W, C,
                     ksize, stride, pad_top, pad_left, OH, OW, dx);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
