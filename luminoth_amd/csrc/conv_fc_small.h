// Skinny fully-connected layers (Sonnet Linear of the RCNN head, luminoth/models/fasterrcnn/rcnn.py:73-98, as 1x1
// convolutions over M = ROIs "pixels"): few rows (M <= 4096) and an output width the MFMA fast paths do not take
// (K % 4 != 0: fc_classifier with num_classes + 1 = 81 columns).  The predicated implicit-GEMM kernels run such a layer as
// 16 tiles of 32 serial single-buffered stages (70 us forward for 85 MFLOP, on the proposal / RCNN chain that bounds the
// middle of the train step); these three VALU kernels spread it over the chip instead.  fp32 FMA chains in a fixed order
// (deterministic); results agree with the MFMA kernels to fp32 summation order.
#pragma once
#include "conv_common.h"

#define FCS_ROWS 8

// y[m][k] = act( sum_c x[m][c] w[c][k] * scale[k] + shift[k] + residual[m][k] );  block = FCS_ROWS rows, x rows in LDS
__global__ void __launch_bounds__(256)
k_fc_small_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
               const float* __restrict__ shift, const float* __restrict__ residual, int M, int C, int K, int act,
               float* __restrict__ y) {
  extern __shared__ float xs[];                       // [FCS_ROWS][C]
  const int m0 = blockIdx.x * FCS_ROWS;
  for (int i = threadIdx.x; i < FCS_ROWS * (C >> 2); i += 256) {
    const int r = i / (C >> 2), c4 = i - r * (C >> 2);
    const f32x4 v = (m0 + r < M) ? *reinterpret_cast<const f32x4*>(x + (size_t)(m0 + r) * C + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(&xs[r * C + 4 * c4]) = v;
  }
  __syncthreads();
  const int kl = threadIdx.x & 31, r = threadIdx.x >> 5;      // 8 rows x 32 columns per pass
  const float lo = act ? 0.f : -INFINITY, hi = (act == 2) ? 6.f : INFINITY;
  for (int k0 = 0; k0 < K; k0 += 32) {
    const int k = k0 + kl;
    if (k < K && m0 + r < M) {
      const float* xr = xs + r * C;
      const float* wk = w + k;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;           // four independent chains, fixed order
      for (int c = 0; c < C; c += 4) {
        a0 = fmaf(xr[c], wk[(size_t)c * K], a0);
        a1 = fmaf(xr[c + 1], wk[(size_t)(c + 1) * K], a1);
        a2 = fmaf(xr[c + 2], wk[(size_t)(c + 2) * K], a2);
        a3 = fmaf(xr[c + 3], wk[(size_t)(c + 3) * K], a3);
      }
      float v = (a0 + a1) + (a2 + a3);
      if (scale) v *= scale[k];
      if (shift) v += shift[k];
      if (residual) v += residual[(size_t)(m0 + r) * K + k];
      y[(size_t)(m0 + r) * K + k] = fminf(fmaxf(v, lo), hi);
    }
  }
}

// dx[m][c] = sum_k g[m][k] * kscale[k] * w[c][k] + addend[m][c];  block = FCS_ROWS rows (g rows in LDS), thread = column c
__global__ void __launch_bounds__(256)
k_fc_small_bwd_data(const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ kscale,
                    const float* __restrict__ addend, int M, int C, int K, float* __restrict__ dx) {
  extern __shared__ float gs[];                       // [FCS_ROWS][K]
  const int m0 = blockIdx.x * FCS_ROWS;
  for (int i = threadIdx.x; i < FCS_ROWS * K; i += 256) {
    const int r = i / K, k = i - r * K;
    gs[i] = (m0 + r < M) ? g[(size_t)(m0 + r) * K + k] * (kscale ? kscale[k] : 1.f) : 0.f;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const float* wc = w + (size_t)c * K;
    float acc[FCS_ROWS];
#pragma unroll
    for (int r = 0; r < FCS_ROWS; ++r) acc[r] = 0.f;
    for (int k = 0; k < K; ++k) {
      const float wv = wc[k];
#pragma unroll
      for (int r = 0; r < FCS_ROWS; ++r) acc[r] = fmaf(gs[r * K + k], wv, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < FCS_ROWS; ++r)
      if (m0 + r < M) {
        const size_t o = (size_t)(m0 + r) * C + c;
        dx[o] = acc[r] + (addend ? addend[o] : 0.f);
      }
  }
}

// dw[c][k] = sum_m x[m][c] g[m][k]  (raw);  colsum[k] = sum_m g[m][k].  thread = (c, k), k fastest; the blocks with
// blockIdx.y == gridDim.y - 1 take the column sums.
__global__ void __launch_bounds__(256)
k_fc_small_bwd_weight(const float* __restrict__ x, const float* __restrict__ g, int M, int C, int K,
                      float* __restrict__ dw, float* __restrict__ colsum) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.y == 1) {
    if (colsum && idx < K) {
      float a = 0.f;
      for (int m = 0; m < M; ++m) a += g[(size_t)m * K + idx];
      colsum[idx] = a;
    }
    return;
  }
  if (idx >= (int64_t)C * K) return;
  const int c = (int)(idx / K), k = (int)(idx - (int64_t)c * K);
  float a0 = 0.f, a1 = 0.f;
  int m = 0;
  for (; m + 1 < M; m += 2) {
    a0 = fmaf(x[(size_t)m * C + c], g[(size_t)m * K + k], a0);
    a1 = fmaf(x[(size_t)(m + 1) * C + c], g[(size_t)(m + 1) * K + k], a1);
  }
  if (m < M) a0 = fmaf(x[(size_t)m * C + c], g[(size_t)m * K + k], a0);
  dw[idx] = a0 + a1;
}

static bool fc_small_ok(const lmh_conv_desc* d) {
  const int64_t M = (int64_t)d->N * d->H * d->W;
  return d->compute == 0 && d->R == 1 && d->S == 1 && d->stride == 1 && d->dilation == 1 && d->pad_top == 0 &&
         d->pad_left == 0 && d->OH == d->H && d->OW == d->W && M <= 4096 && d->K <= 128 && (d->K & 3) != 0 &&
         (d->C & 3) == 0 && d->C <= 4096;
}
