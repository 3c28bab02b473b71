// Streaming helpers around the convolutions (gfx950, HBM-bound): activation backward + column
// sums, BatchNorm parameter gradients, max pooling.
#include "lmh_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: loads stay loads

// ---------------------------------------------------------------------------
// Activations beyond the two the convolution epilogues fuse (1 relu, 2 relu6).  The reference takes ANY tf.nn.<name> as
// `activation_function` of the RPN convolution and the RCNN fully connected layers (luminoth/utils/vars.py:80-88,
// rpn.py:57-59, rcnn.py:73-74); these are the smooth / leaky ones of tf.nn in TF 1.x, applied in place behind the
// convolution (lmh_act_fwd) and differentiated like TF's own gradient ops: EluGrad / SeluGrad / SigmoidGrad / TanhGrad from
// the OUTPUT y, SoftplusGrad / SoftsignGrad from the INPUT z (in a saturated softsign 1 - |y| cancels: expressed in y its
// derivative loses three digits at |z| ~ 1e3, which an RPN convolution on an unnormalised feature map reaches) — so
// lmh_act_fwd writes those two next to their input instead of over it and lmh_act_bwd takes z as its second operand:
//   3 elu        z > 0 ? z : e^z - 1                         f' = y > 0 ? 1 : y + 1
//   4 selu       s * (z > 0 ? z : a (e^z - 1))               f' = y > 0 ? s : y + s a        (s 1.0507009873554805, a 1.6732632423543772)
//   5 softplus   log(1 + e^z)                                f' = 1 / (1 + e^-z)            (from z)
//   6 softsign   z / (1 + |z|)                               f' = 1 / (1 + |z|)^2           (from z)
//   7 sigmoid    1 / (1 + e^-z)                              f' = y (1 - y)
//   8 tanh                                                   f' = 1 - y^2
//   9 leaky_relu max(0.2 z, z)   (tf.nn.leaky_relu default)  f' = y > 0 ? 1 : 0.2
// ---------------------------------------------------------------------------
#define LMH_SELU_S 1.0507009873554805f
#define LMH_SELU_A 1.6732632423543772f
__device__ __forceinline__ float lmh_act_apply(float z, int act) {
  switch (act) {
    case 1: return fmaxf(z, 0.f);
    case 2: return fminf(fmaxf(z, 0.f), 6.f);
    case 3: return z > 0.f ? z : expm1f(z);
    case 4: return z > 0.f ? LMH_SELU_S * z : (LMH_SELU_S * LMH_SELU_A) * expm1f(z);
    case 5: return z > 15.f ? z : log1pf(expf(z));          // log1p(e^z) = z to fp32 beyond z = 15
    case 6: return z / (1.f + fabsf(z));
    case 7: return 1.f / (1.f + expf(-z));
    case 8: return tanhf(z);
    case 9: return fmaxf(0.2f * z, z);
    default: return z;
  }
}
// dy * f'(.): `y` is the layer OUTPUT, except for softplus / softsign (5, 6) where it is the pre-activation z.
// relu / relu6 stay a SELECT (the bits of dy pass unchanged; a product would turn an inf
// gradient under a dead unit into NaN)
__device__ __forceinline__ float lmh_act_grad(float dy, float y, int act, float hi) {
  switch (act) {
    case 0: return dy;
    case 1: case 2: return (y > 0.f && y < hi) ? dy : 0.f;
    case 3: return y > 0.f ? dy : dy * (y + 1.f);
    case 4: return dy * (y > 0.f ? LMH_SELU_S : y + LMH_SELU_S * LMH_SELU_A);
    case 5: return dy / (1.f + expf(-y));
    case 6: { const float t = 1.f + fabsf(y); return dy / (t * t); }
    case 7: return dy * (y * (1.f - y));
    case 8: return dy * (1.f - y * y);
    case 9: return y > 0.f ? dy : 0.2f * dy;
    default: return dy;
  }
}

__global__ void __launch_bounds__(256)
k_act_fwd(const float* __restrict__ z, float* __restrict__ y, int act, int64_t n) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 v = *reinterpret_cast<const f32x4*>(z + 4 * i);
    v[0] = lmh_act_apply(v[0], act); v[1] = lmh_act_apply(v[1], act);
    v[2] = lmh_act_apply(v[2], act); v[3] = lmh_act_apply(v[3], act);
    *reinterpret_cast<f32x4*>(y + 4 * i) = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    y[i] = lmh_act_apply(z[i], act);
  }
}

extern "C" int lmh_act_fwd(const float* z, float* y, int act, int64_t n, lmh_stream_t stream) {
  LMH_CHECK_ARG(z && y && n > 0 && act >= 0 && act <= LMH_ACT_MAX);
  LMH_CHECK_ARG(((((uintptr_t)z) | ((uintptr_t)y)) & 15) == 0);
  if (act == 0 && z == y) return LMH_OK;
  const int64_t n4 = (n + 3) >> 2;
  const int nb = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  lmh_launch(k_act_fwd, dim3(nb), dim3(256), 0, (hipStream_t)stream, z, y, act, n);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

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
  // column partials only when asked for: DYNAMIC shared memory, so the plain g = dy * act'(y) pass (most launches: the
  // weight-gradient kernels produce the channel sums themselves) holds no LDS and fits beside the MFMA blocks of the
  // weight-gradient stream, whose LDS rings would otherwise cap this kernel at two blocks per CU
  extern __shared__ float scol[];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  const float hi = (act == 2) ? 6.f : INFINITY;
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
            d4.x = lmh_act_grad(d4.x, y4.x, act, hi);
            d4.y = lmh_act_grad(d4.y, y4.y, act, hi);
            d4.z = lmh_act_grad(d4.z, y4.z, act, hi);
            d4.w = lmh_act_grad(d4.w, y4.w, act, hi);
          }
          if (g) *reinterpret_cast<float4*>(g + o) = d4;
          s.x += d4.x; s.y += d4.y; s.z += d4.z; s.w += d4.w;
        }
        if (partial) *reinterpret_cast<float4*>(&scol[rsub * K + 4 * cc]) = s;   // rstep * K <= 4096 floats
      }
    }
    if (partial && rstep > 1) {   // fold the rstep row-groups in a fixed order
      __syncthreads();
      for (int c = threadIdx.x; c < K; c += 256) {
        float t = scol[c];
        for (int g2 = 1; g2 < rstep; ++g2) t += scol[g2 * K + c];
        scol[c] = t;
      }
    }
  } else {
    for (int c = threadIdx.x; c < K; c += 256) {
      float sacc = 0.f;
      for (int64_t r = r0; r < r1; ++r) {
        const size_t o = (size_t)r * K + c;
        float d = dy[o];
        if (act) d = lmh_act_grad(d, y[o], act, hi);
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

#include "colsum_common.h"

__global__ void __launch_bounds__(256)
k_colsum_finish(const float* __restrict__ partial, int nb, int K, float* __restrict__ out) {
  colsum_finish_block(partial, nb, K, out, (int)blockIdx.x);
}

// per-channel sums of `nb` rows of K floats (deterministic, fixed order) — used by the Winograd weight gradient, whose
// transformed-gradient plane (1,1) holds every tile's pixel sum (conv_winograd.h)
int lmh_colsum_rows_impl(const float* rows_, int nb, int K, float* out, hipStream_t st) {
  lmh_launch(k_colsum_finish, dim3((K + 31) / 32), dim3(256), 0, st, rows_, nb, K, out);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

static int act_bwd_blocks(int64_t rows, int K, int* rpb_out) {
  int rpb = (int)((rows + 1023) / 1024);   // ~4 slabs per CU
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
  LMH_CHECK_ARG(act >= 0 && act <= LMH_ACT_MAX && (act == 0 || y != nullptr));
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
  const size_t lds = partial ? (size_t)ACT_MAX_K * sizeof(float) : 0;
  if ((K & 3) != 0)
    lmh_launch((k_act_bwd<false>), dim3(nb), dim3(256), lds, st, dy, y, act, rows, K, g, partial, rpb);
  else
    lmh_launch((k_act_bwd<true>), dim3(nb), dim3(256), lds, st, dy, y, act, rows, K, g, partial, rpb);
  if (colsum && g_lmh_defer_tail) {
    g_lmh_last_plan.colpart = partial;
    g_lmh_last_plan.colrows = nb;
  } else if (colsum) {
    lmh_launch(k_colsum_finish, dim3((K + 31) / 32), dim3(256), 0, st, partial, nb, K, colsum);
  }
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// ---- activation bit masks (standalone forms; the fast convolution kernels emit / apply them in their epilogues) ----
// bits[row][K/32]: bit c%32 of word c/32 = act'(y[row][c]) != 0  (relu: y > 0; relu6: 0 < y < 6).  One lane per float4,
// 8 adjacent lanes assemble a word.
__global__ void __launch_bounds__(256)
k_act_bits(const float* __restrict__ y, float hi, int64_t n4, uint32_t* __restrict__ bits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // float4 index; grid covers a multiple of 8
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) v = reinterpret_cast<const float4*>(y)[i];
  unsigned nib = ((v.x > 0.f && v.x < hi) ? 1u : 0u) | ((v.y > 0.f && v.y < hi) ? 2u : 0u) |
                 ((v.z > 0.f && v.z < hi) ? 4u : 0u) | ((v.w > 0.f && v.w < hi) ? 8u : 0u);
  nib <<= 4 * (threadIdx.x & 7);
  nib |= __shfl_xor(nib, 1);
  nib |= __shfl_xor(nib, 2);
  nib |= __shfl_xor(nib, 4);
  if ((threadIdx.x & 7) == 0 && i < n4) bits[i >> 3] = nib;
}

__global__ void __launch_bounds__(256)
k_apply_act_bits(float* __restrict__ dx, const uint32_t* __restrict__ bits, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const unsigned nib = bits[i >> 3] >> (4 * (i & 7));
  float4 v = reinterpret_cast<float4*>(dx)[i];
  v.x = (nib & 1u) ? v.x : 0.f;
  v.y = (nib & 2u) ? v.y : 0.f;
  v.z = (nib & 4u) ? v.z : 0.f;
  v.w = (nib & 8u) ? v.w : 0.f;
  reinterpret_cast<float4*>(dx)[i] = v;
}

int lmh_act_bits_impl(const float* y, int act, int64_t rows, int K, uint32_t* bits, hipStream_t st) {
  LMH_CHECK_ARG(y && bits && rows > 0 && K > 0 && (K & 31) == 0 && (act == 1 || act == 2));
  const int64_t n4 = rows * (K >> 2);
  lmh_launch(k_act_bits, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, y, act == 2 ? 6.f : INFINITY, n4,
                     bits);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
int lmh_apply_act_bits_impl(float* dx, const uint32_t* bits, int64_t rows, int C, hipStream_t st) {
  LMH_CHECK_ARG(dx && bits && rows > 0 && C > 0 && (C & 31) == 0);
  const int64_t n4 = rows * (C >> 2);
  lmh_launch(k_apply_act_bits, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, dx, bits, n4);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
extern "C" int lmh_act_bits(const float* y, int act, int64_t rows, int K, uint32_t* bits, lmh_stream_t stream) {
  return lmh_act_bits_impl(y, act, rows, K, bits, (hipStream_t)stream);
}
extern "C" int lmh_apply_act_bits(float* dx, const uint32_t* bits, int64_t rows, int C, lmh_stream_t stream) {
  return lmh_apply_act_bits_impl(dx, bits, rows, C, (hipStream_t)stream);
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
      *reinterpret_cast<float4*>(&scol[rsub * K + 4 * cc]) = s;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < K; c += 256) {
    float t = scol[c];
    for (int g2 = 1; g2 < rstep; ++g2) t += scol[g2 * K + c];
    partial[(size_t)blockIdx.x * K + c] = t;
  }
}

__global__ void __launch_bounds__(256)
k_bn_finish(const float* __restrict__ partial, int nb, int K, const float* __restrict__ dbeta,
            const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dgamma) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  red[g][cl] = (c < K) ? colsum_partial(partial, nb, K, c, g) : 0.f;
  __syncthreads();
  if (g == 0 && c < K) {
    float t = red[0][cl];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += red[i][cl];
    dgamma[c] = rstd[c] * (t - mean[c] * dbeta[c]);
  }
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
  lmh_launch(k_bn_wdot, dim3(nb), dim3(256), 0, st, w, dw_raw_inout, scale, rsc, K, partial, rpb);
  lmh_launch(k_bn_finish, dim3((K + 31) / 32), dim3(256), 0, st, partial, nb, K, dbeta, mean, rstd,
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

// 3 x 3 window (every pooling layer of ResNet / VGG16 here): the nine taps are requested together — clamped address,
// -inf where the tap lies outside the image — instead of load -> wait -> max one tap at a time (nine serial round trips
// of memory latency per output with the runtime-sized loops above; round 4).  max is exact: same values.
__global__ void __launch_bounds__(256)
k_maxpool3_fwd(const float* __restrict__ x, int N, int H, int W, int C, int stride, int pt, int pl, int OH, int OW,
               float* __restrict__ y) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)N * OH * OW * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    int64_t t = i / C4;
    const int ow = (int)(t % OW); t /= OW;
    const int oh = (int)(t % OH);
    const int n = (int)(t / OH);
    f32x4 v[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int ih = min(max(oh * stride - pt + r, 0), H - 1), iw = min(max(ow * stride - pl + q, 0), W - 1);
        v[3 * r + q] = *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + ih) * W + iw) * C + 4 * c4);
      }
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int ih = oh * stride - pt + r, iw = ow * stride - pl + q;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
          const f32x4 u = v[3 * r + q];
          m.x = fmaxf(m.x, u.x); m.y = fmaxf(m.y, u.y); m.z = fmaxf(m.z, u.z); m.w = fmaxf(m.w, u.w);
        }
      }
    *reinterpret_cast<f32x4*>(y + (size_t)i * 4) = m;
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
  if (ksize == 3)
    lmh_launch(k_maxpool3_fwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, stride, pad_top, pad_left,
               OH, OW, y);
  else
    lmh_launch(k_maxpool_fwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, ksize,
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
  lmh_launch(k_maxpool_bwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, dy, N, H, W, C,
                     ksize, stride, pad_top, pad_left, OH, OW, dx);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}


// ---------------------------------------------------------------------------------------------------------
// tf.image.resize_images(method=BILINEAR) as luminoth calls it (utils/image.py:92-95,126-129): TF 1.x legacy
// sampling — align_corners=False, no half-pixel centres: in = i * (in_size / out_size), lower = (int)in,
// upper = min(lower + 1, in_size - 1), lerp = in - lower; value = top + (bottom - top) * y_lerp with
// top/bottom lerped along x first.  Products and sums are rounded separately (no FMA contraction) so the
// result is bit-identical to the numpy restatement in oracle/image.py.  One thread per output pixel; HBM-bound.
template <typename T>
__global__ void __launch_bounds__(256)
k_resize_bilinear(const T* __restrict__ src, int H, int W, int C, float* __restrict__ dst, int OH, int OW,
                  float hscale, float wscale, int flip_lr, int flip_ud) {
  const int64_t total = (int64_t)OH * OW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int oy = (int)(i / OW), ox = (int)(i - (int64_t)oy * OW);
    const float in_y = __fmul_rn((float)oy, hscale), in_x = __fmul_rn((float)ox, wscale);
    int y0 = (int)in_y, x0 = (int)in_x;
    int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float yl = __fsub_rn(in_y, (float)y0), xl = __fsub_rn(in_x, (float)x0);
    // flip augmentation (utils/image.py:318-370) happens BEFORE the resize in the reference: sample the
    // mirrored source instead of materialising the flipped image
    if (flip_lr) { x0 = W - 1 - x0; x1 = W - 1 - x1; }
    if (flip_ud) { y0 = H - 1 - y0; y1 = H - 1 - y1; }
    const T* r0 = src + (size_t)y0 * W * C;
    const T* r1 = src + (size_t)y1 * W * C;
    float* o = dst + (size_t)i * C;
    for (int c = 0; c < C; ++c) {
      const float tl = (float)r0[(size_t)x0 * C + c], tr = (float)r0[(size_t)x1 * C + c];
      const float bl = (float)r1[(size_t)x0 * C + c], br = (float)r1[(size_t)x1 * C + c];
      const float top = __fadd_rn(tl, __fmul_rn(__fsub_rn(tr, tl), xl));
      const float bot = __fadd_rn(bl, __fmul_rn(__fsub_rn(br, bl), xl));
      o[c] = __fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), yl));
    }
  }
}

extern "C" int lmh_resize_bilinear(const void* src, int src_is_u8, int H, int W, int C, float* dst, int OH,
                                   int OW, int flip_lr, int flip_ud, lmh_stream_t stream) {
  LMH_CHECK_ARG(src && dst && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0);
  const float hscale = (float)H / (float)OH, wscale = (float)W / (float)OW;
  const int64_t total = (int64_t)OH * OW;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (src_is_u8)
    lmh_launch(k_resize_bilinear<uint8_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)src, H, W, C, dst, OH, OW, hscale, wscale, flip_lr, flip_ud);
  else
    lmh_launch(k_resize_bilinear<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float*)src, H, W, C, dst, OH, OW, hscale, wscale, flip_lr, flip_ud);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// ---- dropout (tf.nn.dropout in the RCNN head: models/fasterrcnn/rcnn.py:196,218) ------------------------------
// y = x * keep(i) / keep_prob with keep(i) = hash(seed, LMH_STREAM_DROPOUT, i) < keep_prob * 2^32.  The mask is a pure
// function of (seed, element index): the backward pass regenerates it (dx = dy * keep / keep_prob) instead of storing
// it.  TF's own Philox stream is not reproducible; the oracle twin is oracle/rng.py::dropout_mask.
__global__ void __launch_bounds__(256)
k_dropout(const float* __restrict__ x, int64_t n, uint32_t seed, uint32_t thr, float inv_keep, float* __restrict__ y) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const uint32_t h = lmh_hash_u32(seed, LMH_STREAM_DROPOUT, (uint32_t)i);
    y[i] = (h < thr) ? x[i] * inv_keep : 0.f;
  }
}

extern "C" int lmh_dropout(const float* x, int64_t n, float keep_prob, uint32_t seed, float* y, lmh_stream_t stream) {
  LMH_CHECK_ARG(x && y && n > 0 && n < (1ll << 32) && keep_prob > 0.f && keep_prob <= 1.f);
  const double t = (double)keep_prob * 4294967296.0;
  const uint32_t thr = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  lmh_launch(k_dropout, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, seed, thr, 1.f / keep_prob, y);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// ---- BatchNorm table refresh + loss scalars: the last host-framework arithmetic of the train step ------------------
// scale = gamma * rstd, shift = beta - mean * scale for every frozen-statistics BatchNorm layer of a network at once
// (slim batch_norm in inference mode folded into the convolution epilogues, base_network.py:84-89); same three
// roundings as the mul / mul / sub it replaces (the file is compiled with -ffp-contract=off).
__global__ void __launch_bounds__(256)
k_bn_refresh(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
             const float* __restrict__ rstd, int64_t n, float* __restrict__ scale, float* __restrict__ shift) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float s = gamma[i] * rstd[i];
  const float t = mean[i] * s;
  scale[i] = s;
  shift[i] = beta[i] - t;
}
extern "C" int lmh_bn_refresh(const float* gamma, const float* beta, const float* mean, const float* rstd, int64_t n,
                              float* scale, float* shift, lmh_stream_t stream) {
  LMH_CHECK_ARG(gamma && beta && mean && rstd && scale && shift && n > 0);
  lmh_launch(k_bn_refresh, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, rstd,
             n, scale, shift);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// total_loss of fasterrcnn.py:203-259: no_reg = ((t0 + t1) + t2) + ... over the weighted loss terms in the order the
// reference sums them, regularization = reg_a + reg_b (trainable + frozen regularised variables), total = no_reg + reg.
// out[0] = total, out[1] = no_reg, out[2] = regularization.  One thread: five additions.
struct lmh_loss_terms { const float* t[8]; };
__global__ void k_loss_sums(lmh_loss_terms terms, int n, const float* __restrict__ reg_a, const float* __restrict__ reg_b,
                            float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = terms.t[0][0];
  for (int i = 1; i < n; ++i) s = s + terms.t[i][0];
  const float r = (reg_a ? reg_a[0] : 0.f) + (reg_b ? reg_b[0] : 0.f);
  out[0] = s + r;
  out[1] = s;
  out[2] = r;
}
extern "C" int lmh_loss_sums(const float* const* terms, int n, const float* reg_a, const float* reg_b, float* out,
                             lmh_stream_t stream) {
  LMH_CHECK_ARG(terms && n >= 1 && n <= 8 && out);
  lmh_loss_terms t{};
  for (int i = 0; i < n; ++i) {
    LMH_CHECK_ARG(terms[i] != nullptr);
    t.t[i] = terms[i];
  }
  lmh_launch(k_loss_sums, dim3(1), dim3(64), 0, (hipStream_t)stream, t, n, reg_a, reg_b, out);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
