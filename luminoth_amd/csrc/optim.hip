// Fused multi-tensor momentum SGD (+ folded L2 regulariser) over one flat
// parameter buffer; L2 regularisation loss value (gfx950, HBM-bound).
//
// Reference: tf.train.MomentumOptimizer via luminoth/utils/training.py:64-81
// and train.py:79-91 (v = m*v + g; w -= lr*v, non-Nesterov); l2_regularizer
// (rpn.py:54-56, rcnn.py:59-60, slim weight_decay) enters total_loss, i.e. the
// gradient, as wd*w.
#include "lmh_common.h"

__device__ __forceinline__ int seg_find(const int64_t* __restrict__ off, int nseg, int64_t i) {
  int lo = 0, hi = nseg;  // off[lo] <= i < off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(256)
k_sgd_momentum(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ v, int64_t n,
               const int64_t* __restrict__ seg_offset, const float* __restrict__ seg_wd, int nseg,
               float lr, float momentum, float gscale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float wd = seg_wd[seg_find(seg_offset, nseg, i)];
    const float wi = w[i];
    const float gi = g[i] * gscale + wd * wi;
    const float vi = momentum * v[i] + gi;
    v[i] = vi;
    w[i] = wi - lr * vi;
  }
}

extern "C" int lmh_sgd_momentum(float* w, const float* g, float* v, int64_t n, const int64_t* seg_offset,
                                const float* seg_wd, int nseg, float lr, float momentum, float gscale,
                                lmh_stream_t stream) {
  LMH_CHECK_ARG(w && g && v && seg_offset && seg_wd && n > 0 && nseg > 0);
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_sgd_momentum, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, g, v, n,
                     seg_offset, seg_wd, nseg, lr, momentum, gscale);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

__global__ void __launch_bounds__(256)
k_l2_reg(const float* __restrict__ w, int64_t n, const int64_t* __restrict__ seg_offset,
         const float* __restrict__ seg_wd, int nseg, float* __restrict__ out) {
  __shared__ float sh[4];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float wd = seg_wd[seg_find(seg_offset, nseg, i)];
    const float wi = w[i];
    acc += wd * (wi * wi) * 0.5f;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, (sh[0] + sh[1]) + (sh[2] + sh[3]));
}

extern "C" int lmh_l2_reg_loss(const float* w, int64_t n, const int64_t* seg_offset, const float* seg_wd,
                               int nseg, float* out, lmh_stream_t stream) {
  LMH_CHECK_ARG(w && seg_offset && seg_wd && out && n > 0 && nseg > 0);
  const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(k_l2_reg, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, n, seg_offset,
                     seg_wd, nseg, out);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
