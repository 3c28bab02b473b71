// Winograd F(2x2, 3x3) convolution for the wide stride-1 3x3 layers (RPN 1024->512, VGG conv4/conv5): the 9-tap
// implicit GEMM becomes 16 element-wise GEMMs over 4x4 transformed tiles — 2.25x fewer MFMA FLOPs — between two
// HBM-bound transform passes.  Y = A^T [ (G g G^T) (.) (B^T d B) ] A  (Lavin & Gray 2015, correlation form = TF
// conv2d).  Used for forward (x, w[r][s][c][k]) and for backward data, which is the same correlation of dy with
// w'[r][s][k][c] = w[2-r][2-s][c][k] * kscale[k].
//
// Workspace layout (floats): U [16][Cg][Kg] | V [16][T][Cg] | Mo [16][T][Kg], T = N*ceil(H/2)*ceil(W/2) tiles,
// (Cg, Kg) = (C, K) forward, (K, C) backward data; odd H / W use ceil(H/2) x ceil(W/2) tiles.  Included by conv.hip (needs lmh_zero_page in the same TU).
#pragma once
#include "colsum_common.h"

// ---- weights: U[t][cg][kg], t = 4*a + b -----------------------------------------------------------------------
// forward: thread = (c, k4): float4 loads along k.
__global__ void __launch_bounds__(256)
k_wino_weight_fwd(const float* __restrict__ w, int C, int K, float* __restrict__ U) {
  const int K4 = K >> 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= C * K4) return;
  const int c = idx / K4, k4 = idx - c * K4;
  f32x4 g[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s)
      g[r][s] = *reinterpret_cast<const f32x4*>(w + ((size_t)(r * 3 + s) * C + c) * K + 4 * k4);
  // Gg: rows [g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2]
  f32x4 t[4][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    t[0][s] = g[0][s];
    t[1][s] = (g[0][s] + g[1][s] + g[2][s]) * 0.5f;
    t[2][s] = (g[0][s] - g[1][s] + g[2][s]) * 0.5f;
    t[3][s] = g[2][s];
  }
  const size_t plane = (size_t)C * K;
  float* o = U + (size_t)c * K + 4 * k4;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    *reinterpret_cast<f32x4*>(o + (size_t)(4 * a + 0) * plane) = t[a][0];
    *reinterpret_cast<f32x4*>(o + (size_t)(4 * a + 1) * plane) = (t[a][0] + t[a][1] + t[a][2]) * 0.5f;
    *reinterpret_cast<f32x4*>(o + (size_t)(4 * a + 2) * plane) = (t[a][0] - t[a][1] + t[a][2]) * 0.5f;
    *reinterpret_cast<f32x4*>(o + (size_t)(4 * a + 3) * plane) = t[a][2];
  }
}

// backward data: U'[t][k][c] from w[2-r][2-s][c][k] * kscale[k]; thread = (k, c4): four k-strided loads per tap
// (33 MB once per step), float4 stores along c.
__global__ void __launch_bounds__(256)
k_wino_weight_bwd(const float* __restrict__ w, const float* __restrict__ kscale, int C, int K,
                  float* __restrict__ U) {
  const int C4 = C >> 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= K * C4) return;
  const int c4 = idx % C4, k = idx / C4;     // consecutive threads: consecutive c4 (stores coalesce along c)
  const float ks = kscale ? kscale[k] : 1.f;
  f32x4 g[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float* p = w + ((size_t)((2 - r) * 3 + (2 - s)) * C + 4 * c4) * K + k;
      g[r][s] = f32x4{p[0], p[K], p[2 * (size_t)K], p[3 * (size_t)K]} * ks;
    }
  f32x4 t[4][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    t[0][s] = g[0][s];
    t[1][s] = (g[0][s] + g[1][s] + g[2][s]) * 0.5f;
    t[2][s] = (g[0][s] - g[1][s] + g[2][s]) * 0.5f;
    t[3][s] = g[2][s];
  }
  const size_t plane = (size_t)K * C;
  float* o = U + (size_t)k * C + 4 * c4;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    *reinterpret_cast<f32x4*>(o + (size_t)(4 * a + 0) * plane) = t[a][0];
    *reinterpret_cast<f32x4*>(o + (size_t)(4 * a + 1) * plane) = (t[a][0] + t[a][1] + t[a][2]) * 0.5f;
    *reinterpret_cast<f32x4*>(o + (size_t)(4 * a + 2) * plane) = (t[a][0] - t[a][1] + t[a][2]) * 0.5f;
    *reinterpret_cast<f32x4*>(o + (size_t)(4 * a + 3) * plane) = t[a][2];
  }
}

// ---- input: V[t][tile][c] = B^T d B of the 4x4 patch at (2i-1, 2j-1), zero outside the image ------------------
__global__ void __launch_bounds__(256)
k_wino_input(const float* __restrict__ x, int N, int H, int W, int C, float* __restrict__ V) {
  const int C4 = C >> 2, th = (H + 1) >> 1, tw = (W + 1) >> 1;   // odd sizes: the last tile row / column is half empty
  const int T = N * th * tw;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)T * C4) return;
  const int c4 = (int)(idx % C4), tile = (int)(idx / C4);
  const int j = tile % tw, t1 = tile / tw, i = t1 % th, n = t1 / th;
  f32x4 d[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int h = 2 * i - 1 + a;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int ww = 2 * j - 1 + b;
      const bool ok = (unsigned)h < (unsigned)H && (unsigned)ww < (unsigned)W;
      d[a][b] = ok ? *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + h) * W + ww) * C + 4 * c4)
                   : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // B^T d: rows [d0-d2, d1+d2, d2-d1, d1-d3]
  f32x4 t[4][4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    t[0][b] = d[0][b] - d[2][b];
    t[1][b] = d[1][b] + d[2][b];
    t[2][b] = d[2][b] - d[1][b];
    t[3][b] = d[1][b] - d[3][b];
  }
  const size_t plane = (size_t)T * C;
  float* o = V + (size_t)tile * C + 4 * c4;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    *reinterpret_cast<f32x4*>(o + (size_t)(4 * a + 0) * plane) = t[a][0] - t[a][2];
    *reinterpret_cast<f32x4*>(o + (size_t)(4 * a + 1) * plane) = t[a][1] + t[a][2];
    *reinterpret_cast<f32x4*>(o + (size_t)(4 * a + 2) * plane) = t[a][2] - t[a][1];
    *reinterpret_cast<f32x4*>(o + (size_t)(4 * a + 3) * plane) = t[a][1] - t[a][3];
  }
}

// ---- output: y = act( (A^T m A) * scale + shift + extra ), 2x2 pixels per tile ----------------------------------
__global__ void __launch_bounds__(256)
k_wino_output(const float* __restrict__ Mo, int N, int H, int W, int K, const float* __restrict__ scale,
              const float* __restrict__ shift, const float* __restrict__ extra, float act_lo, float act_hi,
              float* __restrict__ y, uint32_t* __restrict__ bits_out, const uint32_t* __restrict__ bits_in) {
  // bits_out (forward): activation bit mask of y, [pixel][K/32] words (see k_conv_fwd); bits_in (backward data): the
  // mask of the layer input — the stored value is dx * act'(x).  8 adjacent lanes = the 32 channels of one word.
  const int K4 = K >> 2, th = (H + 1) >> 1, tw = (W + 1) >> 1;
  const int T = N * th * tw;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)T * K4) return;
  const int k4 = (int)(idx % K4), tile = (int)(idx / K4);
  const int j = tile % tw, t1 = tile / tw, i = t1 % th, n = t1 / th;
  const size_t plane = (size_t)T * K;
  const float* src = Mo + (size_t)tile * K + 4 * k4;
  f32x4 m[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) m[a][b] = *reinterpret_cast<const f32x4*>(src + (size_t)(4 * a + b) * plane);
  // A^T m: rows [m0+m1+m2, m1-m2-m3]
  f32x4 t[2][4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    t[0][b] = m[0][b] + m[1][b] + m[2][b];
    t[1][b] = m[1][b] - m[2][b] - m[3][b];
  }
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (scale) sc = *reinterpret_cast<const f32x4*>(scale + 4 * k4);
  if (shift) sh = *reinterpret_cast<const f32x4*>(shift + 4 * k4);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    f32x4 o2[2] = {t[a][0] + t[a][1] + t[a][2], t[a][1] - t[a][2] - t[a][3]};
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (2 * i + a >= H || 2 * j + b >= W) continue;
      const size_t off = ((size_t)(n * H + 2 * i + a) * W + 2 * j + b) * K + 4 * k4;
      f32x4 v = o2[b] * sc + sh;
      if (extra) v += *reinterpret_cast<const f32x4*>(extra + off);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], act_lo), act_hi);
      const size_t word = (off >> 5);            // off = pixel * K + 4 * k4  ->  pixel * (K / 32) + k4 / 8
      if (bits_in) {
        const unsigned nib = bits_in[word] >> (4 * (k4 & 7));
        v.x = (nib & 1u) ? v.x : 0.f;
        v.y = (nib & 2u) ? v.y : 0.f;
        v.z = (nib & 4u) ? v.z : 0.f;
        v.w = (nib & 8u) ? v.w : 0.f;
      }
      *reinterpret_cast<f32x4*>(y + off) = v;
      if (bits_out) {                            // K % 32 == 0: the 8 lanes of a word share tile and pixel
        unsigned nib = ((v.x > 0.f && v.x < act_hi) ? 1u : 0u) | ((v.y > 0.f && v.y < act_hi) ? 2u : 0u) |
                       ((v.z > 0.f && v.z < act_hi) ? 4u : 0u) | ((v.w > 0.f && v.w < act_hi) ? 8u : 0u);
        nib <<= 4 * (k4 & 7);
        nib |= __shfl_xor(nib, 1);
        nib |= __shfl_xor(nib, 2);
        nib |= __shfl_xor(nib, 4);
        if ((k4 & 7) == 0) bits_out[word] = nib;
      }
    }
  }
}

// ---- weight gradient: dU[t] = V[t]^T dM[t] with dM = A dY A^T per 2x2 gradient tile, dw = G^T dU G --------------
__global__ void __launch_bounds__(256)
k_wino_dy(const float* __restrict__ g, int N, int H, int W, int K, float* __restrict__ dM) {
  const int K4 = K >> 2, th = (H + 1) >> 1, tw = (W + 1) >> 1;
  const int T = N * th * tw;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)T * K4) return;
  const int k4 = (int)(idx % K4), tile = (int)(idx / K4);
  const int j = tile % tw, t1 = tile / tw, i = t1 % th, n = t1 / th;
  f32x4 y[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const bool ok = 2 * i + a < H && 2 * j + b < W;
      y[a][b] = ok ? *reinterpret_cast<const f32x4*>(g + ((size_t)(n * H + 2 * i + a) * W + 2 * j + b) * K + 4 * k4)
                   : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  // A = [[1,0],[1,1],[1,-1],[0,-1]]: rows of A y
  f32x4 t[4][2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    t[0][b] = y[0][b];
    t[1][b] = y[0][b] + y[1][b];
    t[2][b] = y[0][b] - y[1][b];
    t[3][b] = -y[1][b];
  }
  const size_t plane = (size_t)T * K;
  float* o = dM + (size_t)tile * K + 4 * k4;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    *reinterpret_cast<f32x4*>(o + (size_t)(4 * a + 0) * plane) = t[a][0];
    *reinterpret_cast<f32x4*>(o + (size_t)(4 * a + 1) * plane) = t[a][0] + t[a][1];
    *reinterpret_cast<f32x4*>(o + (size_t)(4 * a + 2) * plane) = t[a][0] - t[a][1];
    *reinterpret_cast<f32x4*>(o + (size_t)(4 * a + 3) * plane) = -t[a][1];
  }
}

__global__ void __launch_bounds__(256)
k_wino_dw(const float* __restrict__ dU, int C, int K, float* __restrict__ dw) {
  const int K4 = K >> 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= C * K4) return;
  const int c = idx / K4, k4 = idx - c * K4;
  const size_t plane = (size_t)C * K;
  const float* src = dU + (size_t)c * K + 4 * k4;
  f32x4 u[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) u[a][b] = *reinterpret_cast<const f32x4*>(src + (size_t)(4 * a + b) * plane);
  // G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
  f32x4 t[3][4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    t[0][b] = u[0][b] + (u[1][b] + u[2][b]) * 0.5f;
    t[1][b] = (u[1][b] - u[2][b]) * 0.5f;
    t[2][b] = (u[1][b] + u[2][b]) * 0.5f + u[3][b];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float* o = dw + ((size_t)(r * 3) * C + c) * K + 4 * k4;
    *reinterpret_cast<f32x4*>(o) = t[r][0] + (t[r][1] + t[r][2]) * 0.5f;
    *reinterpret_cast<f32x4*>(o + plane) = (t[r][1] - t[r][2]) * 0.5f;
    *reinterpret_cast<f32x4*>(o + 2 * plane) = (t[r][1] + t[r][2]) * 0.5f + t[r][3];
  }
}

// ================================================================================================================
// F(4x4, 3x3) (round 3): 36 multiplies per 16 outputs — 4x fewer than the direct form, 1.78x fewer than F(2x2,3x3) —
// and SMALLER transformed planes (36 per 4x4 pixels = 2.25x the activation instead of 4x).  Interpolation points
// 0, +-1, +-2, inf (Lavin & Gray, table of F(4x4,3x3)).  fp32 round-off against a float64 direct convolution on
// post-ReLU activations with a 1024-channel reduction: max 1.8e-5 / rms 1.3e-6 of the output scale (F(2x2): 9e-7 /
// 2e-7; direct fp32: 3e-7 / 6e-8) — inside north_star's 1e-4; pinned by tests/test_gpu_kernels.py.
// Same structure as above: weight / input transforms -> 36 stacked GEMMs (one grid) -> output transform with the fused
// epilogue (scale / shift / residual / activation / activation bit mask).  One thread = one tile x TWO channels
// (a 6x6 patch of float2 is 72 registers; float4 would spill).
// ================================================================================================================
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <class V> __device__ __forceinline__ void w4_bt(const V (&d)[6], V (&t)[6]) {       // B^T d
  t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  t[1] = -4.f * d[1] - 4.f * d[2] + d[3] + d[4];
  t[2] = 4.f * d[1] - 4.f * d[2] - d[3] + d[4];
  t[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
  t[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
  t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
template <class V> __device__ __forceinline__ void w4_at(const V (&m)[6], V (&y)[4]) {       // A^T m
  y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
  y[1] = m[1] - m[2] + 2.f * m[3] - 2.f * m[4];
  y[2] = m[1] + m[2] + 4.f * m[3] + 4.f * m[4];
  y[3] = m[1] - m[2] + 8.f * m[3] - 8.f * m[4] + m[5];
}
template <class V> __device__ __forceinline__ void w4_a(const V (&y)[4], V (&m)[6]) {        // A y  (A = (A^T)^T)
  m[0] = y[0];
  m[1] = y[0] + y[1] + y[2] + y[3];
  m[2] = y[0] - y[1] + y[2] - y[3];
  m[3] = y[0] + 2.f * y[1] + 4.f * y[2] + 8.f * y[3];
  m[4] = y[0] - 2.f * y[1] + 4.f * y[2] - 8.f * y[3];
  m[5] = y[3];
}
template <class V> __device__ __forceinline__ void w4_g(const V (&g)[3], V (&u)[6]) {        // G g
  u[0] = g[0] * 0.25f;
  u[1] = (g[0] + g[1] + g[2]) * (-1.f / 6.f);
  u[2] = (g[0] - g[1] + g[2]) * (-1.f / 6.f);
  u[3] = g[0] * (1.f / 24.f) + g[1] * (1.f / 12.f) + g[2] * (1.f / 6.f);
  u[4] = g[0] * (1.f / 24.f) - g[1] * (1.f / 12.f) + g[2] * (1.f / 6.f);
  u[5] = g[2];
}
template <class V> __device__ __forceinline__ void w4_gt(const V (&u)[6], V (&w)[3]) {       // G^T u
  w[0] = u[0] * 0.25f - (u[1] + u[2]) * (1.f / 6.f) + (u[3] + u[4]) * (1.f / 24.f);
  w[1] = (u[2] - u[1]) * (1.f / 6.f) + (u[3] - u[4]) * (1.f / 12.f);
  w[2] = -(u[1] + u[2]) * (1.f / 6.f) + (u[3] + u[4]) * (1.f / 6.f) + u[5];
}

// forward weights: U[6a+b][c][k] = (G g G^T)[a][b];  thread = (c, k2)
__device__ __forceinline__ void wino4_weight_fwd_body(const float* __restrict__ w, int C, int K, float* __restrict__ U,
                                                      int idx) {
  const int K2 = K >> 1;
  if (idx >= C * K2) return;
  const int c = idx / K2, k2 = idx - c * K2;
  f32x2 t[6][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    f32x2 g[3], u[6];
#pragma unroll
    for (int r = 0; r < 3; ++r) g[r] = *reinterpret_cast<const f32x2*>(w + ((size_t)(r * 3 + s) * C + c) * K + 2 * k2);
    w4_g(g, u);
#pragma unroll
    for (int a = 0; a < 6; ++a) t[a][s] = u[a];
  }
  const size_t plane = (size_t)C * K;
  float* o = U + (size_t)c * K + 2 * k2;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    f32x2 u[6];
    w4_g(t[a], u);
#pragma unroll
    for (int b = 0; b < 6; ++b) *reinterpret_cast<f32x2*>(o + (size_t)(6 * a + b) * plane) = u[b];
  }
}

__global__ void __launch_bounds__(256)
k_wino4_weight_fwd(const float* __restrict__ w, int C, int K, float* __restrict__ U) {
  wino4_weight_fwd_body(w, C, K, U, blockIdx.x * blockDim.x + threadIdx.x);
}

// backward-data weights: U'[6a+b][k][c] from w[2-r][2-s][c][k] * kscale[k];  thread = (k, c2)
__device__ __forceinline__ void wino4_weight_bwd_body(const float* __restrict__ w, const float* __restrict__ kscale, int C,
                                                      int K, float* __restrict__ U, int idx) {
  const int C2 = C >> 1;
  if (idx >= K * C2) return;
  const int c2 = idx % C2, k = idx / C2;
  const float ks = kscale ? kscale[k] : 1.f;
  f32x2 t[6][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    f32x2 g[3], u[6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float* p = w + ((size_t)((2 - r) * 3 + (2 - s)) * C + 2 * c2) * K + k;
      g[r] = f32x2{p[0], p[K]} * ks;
    }
    w4_g(g, u);
#pragma unroll
    for (int a = 0; a < 6; ++a) t[a][s] = u[a];
  }
  const size_t plane = (size_t)K * C;
  float* o = U + (size_t)k * C + 2 * c2;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    f32x2 u[6];
    w4_g(t[a], u);
#pragma unroll
    for (int b = 0; b < 6; ++b) *reinterpret_cast<f32x2*>(o + (size_t)(6 * a + b) * plane) = u[b];
  }
}

__global__ void __launch_bounds__(256)
k_wino4_weight_bwd(const float* __restrict__ w, const float* __restrict__ kscale, int C, int K, float* __restrict__ U) {
  wino4_weight_bwd_body(w, kscale, C, K, U, blockIdx.x * blockDim.x + threadIdx.x);
}

// Every F(4x4,3x3) layer's weight transform of a step in ONE launch (the transforms depend on nothing but the weights and,
// backward, the BatchNorm scale): the train step issues the forward set before the trunk and the backward set on the idle
// weight-gradient stream during the forward pass, instead of 2 x 10 small launches inside the convolution calls.
#define WINO_BATCH_MAX 32
struct wino_weight_batch {
  const float* w[WINO_BATCH_MAX];
  const float* kscale[WINO_BATCH_MAX];
  float* u[WINO_BATCH_MAX];
  int32_t C[WINO_BATCH_MAX], K[WINO_BATCH_MAX];
  int32_t first_block[WINO_BATCH_MAX + 1];
  int32_t n, backward;
};
__global__ void __launch_bounds__(256)
k_wino4_weight_batch(wino_weight_batch b) {
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.first_block[j + 1]) ++j;
  const int idx = ((int)blockIdx.x - b.first_block[j]) * 256 + threadIdx.x;
  if (b.backward) wino4_weight_bwd_body(b.w[j], b.kscale[j], b.C[j], b.K[j], b.u[j], idx);
  else wino4_weight_fwd_body(b.w[j], b.C[j], b.K[j], b.u[j], idx);
}

// input: V[6a+b][tile][c] = (B^T d B)[a][b], d = the 6x6 patch at (4i-1, 4j-1), zero outside the image.
// All 36 loads are unconditional (out-of-image positions read the zero page) and issued before the first use: with the
// `ok ? load : 0` form the compiler put every column's six loads behind a branch and a full s_waitcnt — seven serial
// round trips of memory latency per thread in a kernel that has one wave per SIMD to hide them (round 4; ISA check).
__global__ void __launch_bounds__(256)
k_wino4_input(const float* __restrict__ x, int N, int H, int W, int C, float* __restrict__ V) {
  const int C2 = C >> 1, th = (H + 3) >> 2, tw = (W + 3) >> 2;
  const int T = N * th * tw;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)T * C2) return;
  const int c2 = (int)(idx % C2), tile = (int)(idx / C2);
  const int j = tile % tw, t1 = tile / tw, i = t1 % th, n = t1 / th;
  f32x2 d[6][6];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const int h = 4 * i - 1 + a;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const int ww = 4 * j - 1 + b;
      const bool ok = (unsigned)h < (unsigned)H && (unsigned)ww < (unsigned)W;
      const float* p = ok ? x + ((size_t)(n * H + h) * W + ww) * C + 2 * c2 : lmh_zero_page;
      d[a][b] = *reinterpret_cast<const f32x2*>(p);
    }
  }
  f32x2 t[6][6];
#pragma unroll
  for (int b = 0; b < 6; ++b) {            // column b of the patch -> column b of B^T d
    f32x2 col[6], u[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) col[a] = d[a][b];
    w4_bt(col, u);
#pragma unroll
    for (int a = 0; a < 6; ++a) t[a][b] = u[a];
  }
  const size_t plane = (size_t)T * C;
  float* o = V + (size_t)tile * C + 2 * c2;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    f32x2 u[6];
    w4_bt(t[a], u);                        // (t B)[a][b] = sum_b' t[a][b'] B^T[b][b']
#pragma unroll
    for (int b = 0; b < 6; ++b) *reinterpret_cast<f32x2*>(o + (size_t)(6 * a + b) * plane) = u[b];
  }
}

// output: y = act( (A^T m A) * scale + shift + extra ), 4x4 pixels per tile (+ activation bit masks, see k_conv_fwd).
// EXTRA / BITS_IN: the residual (forward) or addend (backward data) rows and the input-mask words of the 16 pixels are
// requested together with the 36 planes, before anything is used (pixels outside the image read the zero page): as
// conditional loads inside the pixel loop each of them was a load + s_waitcnt vmcnt(0) of its own — up to 32 serial
// round trips per thread in the backward-data direction (round 4; ISA check).
template <bool EXTRA, bool BITS_IN>
__global__ void __launch_bounds__(256)
k_wino4_output(const float* __restrict__ Mo, int N, int H, int W, int K, const float* __restrict__ scale,
               const float* __restrict__ shift, const float* __restrict__ extra, float act_lo, float act_hi,
               float* __restrict__ y, uint32_t* __restrict__ bits_out, const uint32_t* __restrict__ bits_in) {
  const int K2 = K >> 1, th = (H + 3) >> 2, tw = (W + 3) >> 2;
  const int T = N * th * tw;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)T * K2) return;
  const int k2 = (int)(idx % K2), tile = (int)(idx / K2);
  const int j = tile % tw, t1 = tile / tw, i = t1 % th, n = t1 / th;
  const size_t plane = (size_t)T * K;
  const float* src = Mo + (size_t)tile * K + 2 * k2;
  f32x2 m[6][6];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) m[a][b] = *reinterpret_cast<const f32x2*>(src + (size_t)(6 * a + b) * plane);
  f32x2 sc = {1.f, 1.f}, sh = {0.f, 0.f};
  if (scale) sc = *reinterpret_cast<const f32x2*>(scale + 2 * k2);
  if (shift) sh = *reinterpret_cast<const f32x2*>(shift + 2 * k2);
  f32x2 ex[EXTRA ? 4 : 1][EXTRA ? 4 : 1];
  uint32_t bw[BITS_IN ? 4 : 1][BITS_IN ? 4 : 1];
  if (EXTRA || BITS_IN) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const bool in = 4 * i + a < H && 4 * j + b < W;
        const size_t off = ((size_t)(n * H + 4 * i + a) * W + 4 * j + b) * K + 2 * k2;
        if (EXTRA) ex[a][b] = *reinterpret_cast<const f32x2*>(in ? extra + off : lmh_zero_page);
        if (BITS_IN) bw[a][b] = *(in ? bits_in + (off >> 5) : reinterpret_cast<const uint32_t*>(lmh_zero_page));
      }
  }
  f32x2 t[4][6];
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    f32x2 col[6], u[4];
#pragma unroll
    for (int a = 0; a < 6; ++a) col[a] = m[a][b];
    w4_at(col, u);
#pragma unroll
    for (int a = 0; a < 4; ++a) t[a][b] = u[a];
  }
  // everything loaded above is awaited HERE, unconditionally and in front of the first store: loads and stores count on
  // the same vmcnt, so a wait the compiler places inside a conditional pixel block below (it must assume the loads still
  // pending on the path that skipped the previous block) is a vmcnt(0) that also waits for the previous pixel's store —
  // fifteen serial store round trips per thread (round 4; ISA check, tools/isa_mixed_vm_waits.py)
  asm volatile("" ::"v"(sc), "v"(sh));
  if (EXTRA) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) asm volatile("" ::"v"(ex[a][b]));
  }
  if (BITS_IN) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) asm volatile("" ::"v"(bw[a][b]));
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    f32x2 o4[4];
    w4_at(t[a], o4);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (4 * i + a >= H || 4 * j + b >= W) continue;     // uniform over the 16 lanes of a mask word (same tile)
      const size_t off = ((size_t)(n * H + 4 * i + a) * W + 4 * j + b) * K + 2 * k2;
      f32x2 v = o4[b] * sc + sh;
      if (EXTRA) v += ex[a][b];
      v.x = fminf(fmaxf(v.x, act_lo), act_hi);
      v.y = fminf(fmaxf(v.y, act_lo), act_hi);
      const size_t word = off >> 5;
      if (BITS_IN) {
        const unsigned two = bw[a][b] >> (2 * (k2 & 15));
        v.x = (two & 1u) ? v.x : 0.f;
        v.y = (two & 2u) ? v.y : 0.f;
      }
      *reinterpret_cast<f32x2*>(y + off) = v;
      if (bits_out) {
        unsigned two = ((v.x > 0.f && v.x < act_hi) ? 1u : 0u) | ((v.y > 0.f && v.y < act_hi) ? 2u : 0u);
        two <<= 2 * (k2 & 15);
        two |= __shfl_xor(two, 1);
        two |= __shfl_xor(two, 2);
        two |= __shfl_xor(two, 4);
        two |= __shfl_xor(two, 8);
        if ((k2 & 15) == 0) bits_out[word] = two;
      }
    }
  }
}

// weight gradient: dM[6a+b][tile][k] = (A dY A^T)[a][b] of the 4x4 gradient tile (zero outside the image); loads
// unconditional and up front, like k_wino4_input
__global__ void __launch_bounds__(256)
k_wino4_dy(const float* __restrict__ g, int N, int H, int W, int K, float* __restrict__ dM) {
  const int K2 = K >> 1, th = (H + 3) >> 2, tw = (W + 3) >> 2;
  const int T = N * th * tw;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)T * K2) return;
  const int k2 = (int)(idx % K2), tile = (int)(idx / K2);
  const int j = tile % tw, t1 = tile / tw, i = t1 % th, n = t1 / th;
  f32x2 yv[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const bool ok = 4 * i + a < H && 4 * j + b < W;
      const float* p = ok ? g + ((size_t)(n * H + 4 * i + a) * W + 4 * j + b) * K + 2 * k2 : lmh_zero_page;
      yv[a][b] = *reinterpret_cast<const f32x2*>(p);
    }
  f32x2 t[6][4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    f32x2 col[4], u[6];
#pragma unroll
    for (int a = 0; a < 4; ++a) col[a] = yv[a][b];
    w4_a(col, u);
#pragma unroll
    for (int a = 0; a < 6; ++a) t[a][b] = u[a];
  }
  const size_t plane = (size_t)T * K;
  float* o = dM + (size_t)tile * K + 2 * k2;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    f32x2 u[6];
    w4_a(t[a], u);
#pragma unroll
    for (int b = 0; b < 6; ++b) *reinterpret_cast<f32x2*>(o + (size_t)(6 * a + b) * plane) = u[b];
  }
}

// dw[r][s][c][k] = (G^T dU G)[r][s];  thread = (c, k2).  Round 6: two launches of the weight gradient ride along —
//   * dU may arrive as `splits` split-K slabs (stride slab_n floats): they are added on load in slab order, the order of
//     k_splitk_reduce (bit-identical), instead of in a launch of their own;
//   * blocks past nb_dw add up the tiles' pixel sums (plane (1,1) of dM: cs_rows, cs_nb rows of K) into cs_out — the
//     per-channel sums of dy (dbeta / dbias), k_colsum_finish's code (colsum_common.h).
__global__ void __launch_bounds__(256)
k_wino4_dw(const float* __restrict__ dU, int C, int K, float* __restrict__ dw, int splits, size_t slab_n,
           const float* __restrict__ cs_rows, int cs_nb, float* __restrict__ cs_out, int nb_dw) {
  if ((int)blockIdx.x >= nb_dw) {
    colsum_finish_block(cs_rows, cs_nb, K, cs_out, (int)blockIdx.x - nb_dw);
    return;
  }
  const int K2 = K >> 1;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= C * K2) return;
  const int c = idx / K2, k2 = idx - c * K2;
  const size_t plane = (size_t)C * K;
  const float* src = dU + (size_t)c * K + 2 * k2;
  f32x2 t[3][6];
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    f32x2 u[6], v[3];
#pragma unroll
    for (int a = 0; a < 6; ++a) u[a] = *reinterpret_cast<const f32x2*>(src + (size_t)(6 * a + b) * plane);
    for (int sp = 1; sp < splits; ++sp) {
#pragma unroll
      for (int a = 0; a < 6; ++a) u[a] += *reinterpret_cast<const f32x2*>(src + (size_t)sp * slab_n + (size_t)(6 * a + b) * plane);
    }
    w4_gt(u, v);
#pragma unroll
    for (int r = 0; r < 3; ++r) t[r][b] = v[r];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    f32x2 v[3];
    w4_gt(t[r], v);
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2)
      *reinterpret_cast<f32x2*>(dw + ((size_t)(r * 3 + s2) * C + c) * K + 2 * k2) = v[s2];
  }
}

// ---- host -----------------------------------------------------------------------------------------------------
static bool wino_ok(const lmh_conv_desc* d) {
  return (d->compute == 0 || d->compute == 3) && d->R == 3 && d->S == 3 && d->stride == 1 && d->dilation == 1 && d->pad_top == 1 &&
         d->pad_left == 1 && d->OH == d->H && d->OW == d->W && (d->C % BK) == 0 && (d->K % BK) == 0;
}

extern "C" int lmh_conv2d_winograd_ok(const lmh_conv_desc* d) { return d && wino_ok(d) ? 1 : 0; }

int lmh_opt(const char* name);
// output tile of the Winograd path: 4 = F(4x4,3x3) (default), 2 = F(2x2,3x3) (lmh_set_option("wino_m", 2))
static int wino_mo() { return lmh_opt("wino_m") == 2 ? 2 : 4; }
static size_t wino_tiles(const lmh_conv_desc* d, int mo) {
  return (size_t)d->N * ((d->H + mo - 1) / mo) * ((d->W + mo - 1) / mo);
}

extern "C" size_t lmh_conv2d_winograd_workspace_bytes(const lmh_conv_desc* d) {
  if (!d || !wino_ok(d)) return 0;
  // sized for whichever tile needs more (F(2x2): 16 planes over T2 tiles; F(4x4): 36 planes over T4 ~ T2/4 tiles)
  size_t best = 0;
  for (int mo = 2; mo <= 4; mo += 2) {
    const size_t T = wino_tiles(d, mo), P2 = (size_t)(mo + 2) * (mo + 2);
    const size_t b = P2 * sizeof(float) * ((size_t)d->C * d->K + T * d->C + T * d->K);
    if (b > best) best = b;
  }
  return best;
}

// Cg = reduction channels, Kg = output channels of this direction.
static int wino_run(const lmh_conv_desc* d, int mo, const float* in, int Cg, int Kg, const float* U, float* V, float* Mo,
                    const float* scale, const float* shift, const float* extra, float act_lo, float act_hi,
                    float* out, uint32_t* bits_out, const uint32_t* bits_in, hipStream_t st) {
  const int T = (int)wino_tiles(d, mo), P2 = (mo + 2) * (mo + 2);
  if (mo == 4) {
    const int64_t n = (int64_t)T * (Cg / 2);
    lmh_launch(k_wino4_input, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, d->N, d->H, d->W, Cg, V);
  } else {
    const int64_t n = (int64_t)T * (Cg / 4);
    lmh_launch(k_wino_input, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, d->N, d->H, d->W, Cg, V);
  }
  // P2 GEMMs [T x Cg] x [Cg x Kg] as ONE grid of the forward kernel (a 1x1 convolution over T "pixels")
  lmh_conv_desc g = *d;
  g.N = 1; g.H = T; g.W = 1; g.OH = T; g.OW = 1; g.C = Cg; g.K = Kg; g.R = 1; g.S = 1;
  g.stride = 1; g.dilation = 1; g.pad_top = 0; g.pad_left = 0; g.act = 0; g.compute = 0;
  int bm, bn;
  pick_tile((int64_t)T * P2, Kg, &bm, &bn);
  const int grid = P2 * ((T + bm - 1) / bm) * ((Kg + bn - 1) / bn);
  const bool x3 = d->compute == 3;        // bf16x3: the stacked GEMMs on the bf16 matrix pipe (conv_half.h), transforms unchanged
#define LAUNCH_WG(BM_, BN_)                                                                           \
  do {                                                                                                \
    if (x3 && lmh_opt("x3_new"))                                                                      \
      (void)lmh_x3_fwd_launch(&g, (const float*)V, U, nullptr, nullptr, nullptr, Mo, nullptr, P2, BM_, BN_, x3_pipe(Cg / BK), st); \
    else if (x3 && x3_pf_gb == 3)                                                                     \
      lmh_launch((k_conv_fwd_h<3, BM_, BN_, 3, true>), dim3(grid), dim3(512), 0, st, g, (const float*)V, U, \
                         (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, Mo, P2);  \
    else if (x3)                                                                                      \
      lmh_launch((k_conv_fwd_h<3, BM_, BN_, 0, true>), dim3(grid), dim3(256), 0, st, g, (const float*)V, U, \
                         (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, Mo, P2);  \
    else                                                                                              \
      lmh_launch((k_conv_fwd<BM_, BN_, true>), dim3(grid), dim3(256), 0, st, g, (const float*)V, U,  \
                         (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, Mo, P2,   \
                         (uint32_t*)nullptr);                                                         \
  } while (0)
  g_prof_pending_bytes = (double)P2 * 4.0 * ((double)T * Cg + (double)Cg * Kg + (double)T * Kg);
  prof_begin(st);
  if (bm == 128 && bn == 128) LAUNCH_WG(128, 128);
  else if (bm == 128) LAUNCH_WG(128, 64);
  else LAUNCH_WG(64, 64);
#undef LAUNCH_WG
  if (x3 && lmh_opt("x3_new")) prof_end(st, (double)P2 * 2.0 * T * (double)Cg * Kg, "k_x3_fwd<%d, %d, true, %d>", bm, bn, x3_pipe(Cg / BK));
  else if (x3) prof_end(st, (double)P2 * 2.0 * T * (double)Cg * Kg, "k_conv_fwd_h<3, %d, %d, GB>", bm, bn);
  else prof_end(st, (double)P2 * 2.0 * T * (double)Cg * Kg, "k_conv_fwd<%d, %d, true>", bm, bn);
  if (mo == 4) {
    const int64_t n = (int64_t)T * (Kg / 2);
#define LAUNCH_WO(EX_, BI_)                                                                              \
    lmh_launch((k_wino4_output<EX_, BI_>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)Mo, \
               d->N, d->H, d->W, Kg, scale, shift, extra, act_lo, act_hi, out, bits_out, bits_in)
    if (extra && bits_in) LAUNCH_WO(true, true);
    else if (extra) LAUNCH_WO(true, false);
    else if (bits_in) LAUNCH_WO(false, true);
    else LAUNCH_WO(false, false);
#undef LAUNCH_WO
  } else {
    const int64_t n = (int64_t)T * (Kg / 4);
    lmh_launch(k_wino_output, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)Mo, d->N,
                       d->H, d->W, Kg, scale, shift, extra, act_lo, act_hi, out, bits_out, bits_in);
  }
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// bytes of the transformed INPUT planes V of the forward pass under the current "wino_m" (a caller that keeps them —
// lmh_conv2d_fwd_winograd v_keep — hands them back to lmh_conv2d_bwd_weight_winograd, which needs exactly B^T x B again)
extern "C" size_t lmh_conv2d_winograd_v_bytes(const lmh_conv_desc* d) {
  if (!d || !wino_ok(d)) return 0;
  const int mo = wino_mo();
  return (size_t)(mo + 2) * (mo + 2) * wino_tiles(d, mo) * d->C * sizeof(float);
}

static int wino_carve(const lmh_conv_desc* d, int mo, void* ws, size_t ws_bytes, int Cg, int Kg, float** U, float** V,
                      float** Mo) {
  if (!ws || ws_bytes < lmh_conv2d_winograd_workspace_bytes(d)) {
    lmh_set_error("winograd: workspace %zu < %zu", ws_bytes, lmh_conv2d_winograd_workspace_bytes(d));
    return LMH_ERR_WORKSPACE;
  }
  const size_t T = wino_tiles(d, mo), P2 = (size_t)(mo + 2) * (mo + 2);
  *U = reinterpret_cast<float*>(ws);
  *V = *U + P2 * (size_t)d->C * d->K;
  *Mo = *V + P2 * T * Cg;
  (void)Kg;
  return LMH_OK;
}

static void wino_weights(const lmh_conv_desc* d, int mo, const float* w, const float* kscale, int backward, float* u,
                         hipStream_t st) {
  if (mo == 4) {
    if (backward) {
      const int n = d->K * (d->C / 2);
      lmh_launch(k_wino4_weight_bwd, dim3((n + 255) / 256), dim3(256), 0, st, w, kscale, d->C, d->K, u);
    } else {
      const int n = d->C * (d->K / 2);
      lmh_launch(k_wino4_weight_fwd, dim3((n + 255) / 256), dim3(256), 0, st, w, d->C, d->K, u);
    }
  } else if (backward) {
    const int n = d->K * (d->C / 4);
    lmh_launch(k_wino_weight_bwd, dim3((n + 255) / 256), dim3(256), 0, st, w, kscale, d->C, d->K, u);
  } else {
    const int n = d->C * (d->K / 4);
    lmh_launch(k_wino_weight_fwd, dim3((n + 255) / 256), dim3(256), 0, st, w, d->C, d->K, u);
  }
}

// The transformed weights depend on nothing but the weights (and, backward, the BN scale): a caller may compute them
// ahead of the call (u != NULL, produced under the SAME "wino_m" option); with u == NULL they are computed here.
extern "C" int lmh_conv2d_winograd_transform_weights(const lmh_conv_desc* d, const float* w, const float* kscale,
                                                     int backward, float* u, lmh_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  LMH_CHECK_ARG(w && u && wino_ok(d));
  wino_weights(d, wino_mo(), w, kscale, backward, u, (hipStream_t)stream);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

extern "C" size_t lmh_winograd_u_bytes(int C, int K) {
  const int mo = wino_mo();
  return (size_t)(mo + 2) * (mo + 2) * C * K * sizeof(float);
}

extern "C" int lmh_winograd_transform_weights_batch(const lmh_wino_weight_job* jobs, int n, int backward,
                                                    lmh_stream_t stream) {
  LMH_CHECK_ARG(jobs && n > 0);
  hipStream_t st = (hipStream_t)stream;
  if (wino_mo() != 4) {                      // F(2x2,3x3): the per-layer kernels
    for (int i = 0; i < n; ++i) {
      lmh_conv_desc d = {};
      d.C = jobs[i].C; d.K = jobs[i].K;
      wino_weights(&d, 2, jobs[i].w, jobs[i].kscale, backward, jobs[i].u, st);
    }
    LMH_CHECK_LAUNCH();
    return LMH_OK;
  }
  for (int i0 = 0; i0 < n; i0 += WINO_BATCH_MAX) {
    wino_weight_batch b;
    b.n = (n - i0) < WINO_BATCH_MAX ? (n - i0) : WINO_BATCH_MAX;
    b.backward = backward;
    int blocks = 0;
    for (int i = 0; i < b.n; ++i) {
      const lmh_wino_weight_job& j = jobs[i0 + i];
      LMH_CHECK_ARG(j.w && j.u && j.C > 0 && j.K > 0 && (j.C % BK) == 0 && (j.K % BK) == 0);
      b.w[i] = j.w; b.kscale[i] = j.kscale; b.u[i] = j.u; b.C[i] = j.C; b.K[i] = j.K;
      b.first_block[i] = blocks;
      blocks += (j.C * (j.K / 2) + 255) / 256;       // (c, k2) forward / (k, c2) backward: the same count
    }
    b.first_block[b.n] = blocks;
    lmh_launch(k_wino4_weight_batch, dim3(blocks), dim3(256), 0, st, b);
  }
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

extern "C" int lmh_conv2d_fwd_winograd(const lmh_conv_desc* d, const float* x, const float* w, const float* u,
                                       const float* scale, const float* shift, const float* residual, float* y,
                                       uint32_t* act_bits, float* v_keep, void* ws, size_t ws_bytes,
                                       lmh_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  LMH_CHECK_ARG(x && (w || u) && y && wino_ok(d));
  const int mo = wino_mo();
  float *U, *V, *Mo;
  rc = wino_carve(d, mo, ws, ws_bytes, d->C, d->K, &U, &V, &Mo);
  if (rc) return rc;
  if (v_keep) V = v_keep;          // the caller keeps B^T x B for this layer's weight gradient
  hipStream_t st = (hipStream_t)stream;
  if (!u) {
    wino_weights(d, mo, w, nullptr, 0, U, st);
    u = U;
  }
  const float lo = d->act ? 0.f : -INFINITY, hi = (d->act == 2) ? 6.f : INFINITY;
  return wino_run(d, mo, x, d->C, d->K, u, V, Mo, scale, shift, residual, lo, hi, y, act_bits, nullptr, st);
}

extern "C" int lmh_conv2d_bwd_data_winograd(const lmh_conv_desc* d, const float* dy, const float* w, const float* u,
                                            const float* kscale, const float* addend, const uint32_t* xbits,
                                            float* dx, void* ws, size_t ws_bytes, lmh_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  LMH_CHECK_ARG(dy && (w || u) && dx && wino_ok(d));
  const int mo = wino_mo();
  float *U, *V, *Mo;
  rc = wino_carve(d, mo, ws, ws_bytes, d->K, d->C, &U, &V, &Mo);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (!u) {
    wino_weights(d, mo, w, kscale, 1, U, st);
    u = U;
  }
  return wino_run(d, mo, dy, d->K, d->C, u, V, Mo, nullptr, nullptr, addend, -INFINITY, INFINITY, dx, nullptr, xbits, st);
}

// ---- weight gradient -------------------------------------------------------------------------------------------
static lmh_conv_desc wino_gemm_desc(const lmh_conv_desc* d, int T, int mo) {
  lmh_conv_desc g = *d;        // (mo+2)^2 stacked [T x C]^T [T x K] products as the taps of a fake (mo+2)x(mo+2) filter
  g.N = 1; g.H = T; g.W = 1; g.OH = T; g.OW = 1; g.R = mo + 2; g.S = mo + 2;
  g.stride = 1; g.dilation = 0; g.pad_top = 0; g.pad_left = 0; g.act = 0;
  return g;
}

static size_t wino_wgrad_bytes(const lmh_conv_desc* d, int mo, size_t* planes_out) {
  const int T = (int)wino_tiles(d, mo);
  const size_t P2 = (size_t)(mo + 2) * (mo + 2);
  const lmh_conv_desc g = wino_gemm_desc(d, T, mo);
  const size_t planes = lmh_align_up(P2 * sizeof(float) * ((size_t)T * d->C + (size_t)T * d->K + (size_t)d->C * d->K), 256);
  if (planes_out) *planes_out = planes;
  return planes + lmh_conv2d_bwd_weight_workspace_bytes(&g);
}

extern "C" size_t lmh_conv2d_bwd_weight_winograd_workspace_bytes(const lmh_conv_desc* d) {
  if (!d || !wino_ok(d)) return 0;
  const size_t a = wino_wgrad_bytes(d, 2, nullptr), b = wino_wgrad_bytes(d, 4, nullptr);
  return a > b ? a : b;
}

int lmh_colsum_rows_impl(const float* rows_, int nb, int K, float* out, hipStream_t st);   // elementwise.hip

// v_cached (may be NULL): the V planes lmh_conv2d_fwd_winograd kept for this x (same "wino_m"): the input transform is
// skipped.  colsum (may be NULL): K floats, WRITTEN with sum over pixels of dy per output channel (dbeta / dbias) — the
// (1,1) plane of A dY A^T is every tile's pixel sum (row 1 of A is all ones), so T rows of K are added instead of a pass
// over dy.
extern "C" int lmh_conv2d_bwd_weight_winograd(const lmh_conv_desc* d, const float* x, const float* dy, float* dw,
                                              const float* v_cached, float* colsum, void* ws, size_t ws_bytes,
                                              lmh_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  LMH_CHECK_ARG((x || v_cached) && dy && dw && wino_ok(d));
  if (!ws || ws_bytes < lmh_conv2d_bwd_weight_winograd_workspace_bytes(d)) {
    lmh_set_error("lmh_conv2d_bwd_weight_winograd: workspace too small");
    return LMH_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int mo = wino_mo();
  const int T = (int)wino_tiles(d, mo);
  const size_t P2 = (size_t)(mo + 2) * (mo + 2);
  size_t planes;
  (void)wino_wgrad_bytes(d, mo, &planes);
  float* V = reinterpret_cast<float*>(ws);
  float* dM = V + P2 * (size_t)T * d->C;
  float* dU = dM + P2 * (size_t)T * d->K;
  void* ws2 = reinterpret_cast<char*>(ws) + planes;
  const float* Vin = v_cached ? v_cached : V;
  if (mo == 4) {
    const int64_t n = (int64_t)T * (d->C / 2);
    if (!v_cached)
      lmh_launch(k_wino4_input, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, d->N, d->H, d->W, d->C, V);
    const int64_t m = (int64_t)T * (d->K / 2);
    lmh_launch(k_wino4_dy, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, dy, d->N, d->H, d->W, d->K, dM);
  } else {
    const int64_t n = (int64_t)T * (d->C / 4);
    if (!v_cached)
      lmh_launch(k_wino_input, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, d->N, d->H, d->W, d->C, V);
    const int64_t m = (int64_t)T * (d->K / 4);
    lmh_launch(k_wino_dy, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, dy, d->N, d->H, d->W, d->K, dM);
  }
  const float* cs_rows = dM + (size_t)(mo + 3) * T * d->K;      // plane (1,1) = index (mo+2)*1 + 1: the tiles' pixel sums
  if (colsum && mo != 4) {
    rc = lmh_colsum_rows_impl(cs_rows, T, d->K, colsum, st);
    if (rc) return rc;
  }
  const lmh_conv_desc g = wino_gemm_desc(d, T, mo);
  g_gb_slabs.want = (mo == 4);            // F(4x4): a split reduction stays in its slabs; k_wino4_dw adds them on load
  g_gb_slabs.splits = 0;
  rc = bwd_weight_launch(&g, Vin, dM, nullptr, dU, nullptr, ws2, ws_bytes - planes, st, true);   // gb: never deferred
  g_gb_slabs.want = false;
  if (rc) return rc;
  if (mo == 4) {
    const int n = d->C * (d->K / 2);
    const int nb_dw = (n + 255) / 256, nb_cs = colsum ? (d->K + 31) / 32 : 0;
    const bool slabs = g_gb_slabs.splits > 1;
    lmh_launch(k_wino4_dw, dim3(nb_dw + nb_cs), dim3(256), 0, st, slabs ? g_gb_slabs.slabs : (const float*)dU, d->C, d->K, dw,
               slabs ? g_gb_slabs.splits : 1, (size_t)P2 * d->C * d->K, cs_rows, T, colsum, nb_dw);
  } else {
    const int n = d->C * (d->K / 4);
    lmh_launch(k_wino_dw, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)dU, d->C, d->K, dw);
  }
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
