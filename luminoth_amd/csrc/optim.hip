// Fused multi-tensor momentum SGD (+ folded L2 regulariser) over one flat
// parameter buffer; L2 regularisation loss value (gfx950, HBM-bound).
//
// Reference: tf.train.MomentumOptimizer via luminoth/utils/training.py:64-81
// and train.py:79-91 (v = m*v + g; w -= lr*v, non-Nesterov); l2_regularizer
// (rpn.py:54-56, rcnn.py:59-60, slim weight_decay) enters total_loss, i.e. the
// gradient, as wd*w.
#include "lmh_common.h"

// The segment table is staged in LDS; every thread walks float4s with a fixed stride, so its segment
// index only moves forward: one binary search, then a short linear advance per element (segments may
// start anywhere; params.py happens to pad them to 16 bytes).
#define OPT_MAX_SEG 2048
__device__ __forceinline__ int seg_find(const int64_t* off, int nseg, int64_t i) {
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
  __shared__ int64_t s_off[OPT_MAX_SEG + 1];
  __shared__ float s_wd[OPT_MAX_SEG];
  for (int i = threadIdx.x; i <= nseg; i += 256) s_off[i] = seg_offset[i];
  for (int i = threadIdx.x; i < nseg; i += 256) s_wd[i] = seg_wd[i];
  __syncthreads();
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * 256;
  int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {   // scalar tail
    const int64_t i = 4 * n4 + threadIdx.x;
    const float wdt = s_wd[seg_find(s_off, nseg, i)];
    const float wi = w[i];
    const float vi = momentum * v[i] + (g[i] * gscale + wdt * wi);
    v[i] = vi;
    w[i] = wi - lr * vi;
  }
  if (i4 >= n4) return;
  int seg = seg_find(s_off, nseg, 4 * i4);
  for (; i4 < n4; i4 += stride) {
    float wd[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      while (s_off[seg + 1] <= 4 * i4 + e) ++seg;
      wd[e] = s_wd[seg];
    }
    float4 wi = reinterpret_cast<const float4*>(w)[i4];
    const float4 gi = reinterpret_cast<const float4*>(g)[i4];
    float4 vi = reinterpret_cast<const float4*>(v)[i4];
    vi.x = momentum * vi.x + (gi.x * gscale + wd[0] * wi.x);
    vi.y = momentum * vi.y + (gi.y * gscale + wd[1] * wi.y);
    vi.z = momentum * vi.z + (gi.z * gscale + wd[2] * wi.z);
    vi.w = momentum * vi.w + (gi.w * gscale + wd[3] * wi.w);
    wi.x -= lr * vi.x; wi.y -= lr * vi.y; wi.z -= lr * vi.z; wi.w -= lr * vi.w;
    reinterpret_cast<float4*>(v)[i4] = vi;
    reinterpret_cast<float4*>(w)[i4] = wi;
  }
}

extern "C" int lmh_sgd_momentum(float* w, const float* g, float* v, int64_t n, const int64_t* seg_offset,
                                const float* seg_wd, int nseg, float lr, float momentum, float gscale,
                                lmh_stream_t stream) {
  LMH_CHECK_ARG(w && g && v && seg_offset && seg_wd && n > 0 && nseg > 0 && nseg <= OPT_MAX_SEG);
  const int64_t n4 = n >> 2;
  const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 + 1 : 2048);
  lmh_launch(k_sgd_momentum, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, g, v, n,
                     seg_offset, seg_wd, nseg, lr, momentum, gscale);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// The same update over the float4-aligned range [lo, hi) of the flat buffer, with the learning rate read from DEVICE memory:
// a launch that can sit inside a recorded launch plan (a by-value rate would be frozen into it) and run as soon as the
// gradients of a range are final — behind the early tail batches, under the rest of the backward pass.
__device__ __forceinline__ void sgd_momentum_range_body(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ v, int64_t lo, int64_t hi,
                     const int64_t* __restrict__ seg_offset, const float* __restrict__ seg_wd, int nseg,
                     const float* __restrict__ lr_ptr, float momentum, float gscale) {
  __shared__ int64_t s_off[OPT_MAX_SEG + 1];
  __shared__ float s_wd[OPT_MAX_SEG];
  for (int i = threadIdx.x; i <= nseg; i += 256) s_off[i] = seg_offset[i];
  for (int i = threadIdx.x; i < nseg; i += 256) s_wd[i] = seg_wd[i];
  __syncthreads();
  const float lr = *lr_ptr;
  const int64_t lo4 = lo >> 2, hi4 = hi >> 2, stride = (int64_t)gridDim.x * 256;
  int64_t i4 = lo4 + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < (hi & 3)) {   // scalar tail (only when hi is the end of the buffer)
    const int64_t i = 4 * hi4 + threadIdx.x;
    const float wdt = s_wd[seg_find(s_off, nseg, i)];
    const float wi = w[i];
    const float vi = momentum * v[i] + (g[i] * gscale + wdt * wi);
    v[i] = vi;
    w[i] = wi - lr * vi;
  }
  if (i4 >= hi4) return;
  int seg = seg_find(s_off, nseg, 4 * i4);
  for (; i4 < hi4; i4 += stride) {
    float wd[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      while (s_off[seg + 1] <= 4 * i4 + e) ++seg;
      wd[e] = s_wd[seg];
    }
    float4 wi = reinterpret_cast<const float4*>(w)[i4];
    const float4 gi = reinterpret_cast<const float4*>(g)[i4];
    float4 vi = reinterpret_cast<const float4*>(v)[i4];
    vi.x = momentum * vi.x + (gi.x * gscale + wd[0] * wi.x);
    vi.y = momentum * vi.y + (gi.y * gscale + wd[1] * wi.y);
    vi.z = momentum * vi.z + (gi.z * gscale + wd[2] * wi.z);
    vi.w = momentum * vi.w + (gi.w * gscale + wd[3] * wi.w);
    wi.x -= lr * vi.x; wi.y -= lr * vi.y; wi.z -= lr * vi.z; wi.w -= lr * vi.w;
    reinterpret_cast<float4*>(v)[i4] = vi;
    reinterpret_cast<float4*>(w)[i4] = wi;
  }
}

// two names for the trace tools: a step ENDS with k_sgd_momentum / k_sgd_momentum_range (scripts/make_profile_summary.py,
// bench.py reduce_kernel_trace); the updates issued under the backward pass are k_sgd_early_range
__global__ void __launch_bounds__(256)
k_sgd_momentum_range(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ v, int64_t lo, int64_t hi,
                     const int64_t* __restrict__ seg_offset, const float* __restrict__ seg_wd, int nseg,
                     const float* __restrict__ lr_ptr, float momentum, float gscale) {
  sgd_momentum_range_body(w, g, v, lo, hi, seg_offset, seg_wd, nseg, lr_ptr, momentum, gscale);
}
__global__ void __launch_bounds__(256)
k_sgd_early_range(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ v, int64_t lo, int64_t hi,
                  const int64_t* __restrict__ seg_offset, const float* __restrict__ seg_wd, int nseg,
                  const float* __restrict__ lr_ptr, float momentum, float gscale) {
  sgd_momentum_range_body(w, g, v, lo, hi, seg_offset, seg_wd, nseg, lr_ptr, momentum, gscale);
}

extern "C" int lmh_sgd_momentum_range(float* w, const float* g, float* v, int64_t n, int64_t lo, int64_t hi,
                                      const int64_t* seg_offset, const float* seg_wd, int nseg, const float* lr_dev,
                                      float momentum, float gscale, int early, lmh_stream_t stream) {
  LMH_CHECK_ARG(w && g && v && seg_offset && seg_wd && lr_dev && n > 0 && nseg > 0 && nseg <= OPT_MAX_SEG);
  LMH_CHECK_ARG(lo >= 0 && lo < hi && hi <= n && (lo & 3) == 0 && ((hi & 3) == 0 || hi == n));
  const int64_t n4 = (hi - lo) >> 2;
  const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 + 1 : 2048);
  if (early)
    lmh_launch(k_sgd_early_range, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, g, v, lo, hi, seg_offset, seg_wd,
               nseg, lr_dev, momentum, gscale);
  else
    lmh_launch(k_sgd_momentum_range, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, g, v, lo, hi, seg_offset, seg_wd,
               nseg, lr_dev, momentum, gscale);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

__global__ void __launch_bounds__(256)
k_l2_reg(const float* __restrict__ w, int64_t n, const int64_t* __restrict__ seg_offset,
         const float* __restrict__ seg_wd, int nseg, float* __restrict__ out) {
  __shared__ int64_t s_off[OPT_MAX_SEG + 1];
  __shared__ float s_wd[OPT_MAX_SEG];
  __shared__ float sh[4];
  for (int i = threadIdx.x; i <= nseg; i += 256) s_off[i] = seg_offset[i];
  for (int i = threadIdx.x; i < nseg; i += 256) s_wd[i] = seg_wd[i];
  __syncthreads();
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * 256;
  int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  if (i4 < n4) {
    int seg = seg_find(s_off, nseg, 4 * i4);
    for (; i4 < n4; i4 += stride) {
      const float4 wi = reinterpret_cast<const float4*>(w)[i4];
      const float we[4] = {wi.x, wi.y, wi.z, wi.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        while (s_off[seg + 1] <= 4 * i4 + e) ++seg;
        acc += s_wd[seg] * (we[e] * we[e]) * 0.5f;
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {   // scalar tail
    const int64_t i = 4 * n4 + threadIdx.x;
    const float wi = w[i];
    acc += s_wd[seg_find(s_off, nseg, i)] * (wi * wi) * 0.5f;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  // one partial per block, added up by k_l2_reg_finish in block order: the reported scalar is the same bits every step
  // (round 6: the blocks used to add into `out` atomically, in arrival order — 1e-7 of run-to-run noise on a reported value)
  if (threadIdx.x == 0) out[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

#define L2_MAX_BLOCKS 1024
__global__ void __launch_bounds__(256)
k_l2_reg_finish(const float* __restrict__ partial, int nb, float* __restrict__ out) {
  __shared__ float sh[256];
  float a = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) a += partial[i];      // (fixed assignment, fixed order)
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

extern "C" size_t lmh_l2_reg_workspace_bytes(void) { return L2_MAX_BLOCKS * sizeof(float); }

extern "C" int lmh_l2_reg_loss(const float* w, int64_t n, const int64_t* seg_offset, const float* seg_wd,
                               int nseg, float* out, void* ws, size_t ws_bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(w && seg_offset && seg_wd && out && n > 0 && nseg > 0 && nseg <= OPT_MAX_SEG);
  if (!ws || ws_bytes < lmh_l2_reg_workspace_bytes()) { lmh_set_error("lmh_l2_reg_loss: workspace too small"); return LMH_ERR_WORKSPACE; }
  const int64_t n4 = n >> 2;
  const int blocks = (int)((n4 + 255) / 256 < L2_MAX_BLOCKS ? (n4 + 255) / 256 + 1 : L2_MAX_BLOCKS);
  float* partial = reinterpret_cast<float*>(ws);
  lmh_launch(k_l2_reg, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, n, seg_offset, seg_wd, nseg, partial);
  lmh_launch(k_l2_reg_finish, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partial, blocks, out);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// ============================================================================
// The rest of the optimizer surface of luminoth/utils/training.py (all non-default in base_config.yml):
//   * per-tensor clip_by_norm(g', 10) of `clip_gradients_by_norm` (training.py:84-120): g' = g*gscale + wd*w is
//     what TF differentiates (total_loss includes the L2 terms), so the norm is taken over g';
//   * tf.train.AdamOptimizer / RMSPropOptimizer / GradientDescentOptimizer of the OPTIMIZERS table (training.py:6-11).
// Two kernels: per-segment sum of squares -> clip factors, then one fused update over the flat buffer.
// ============================================================================
// Segment sums are accumulated with fp64 atomics: the order of the additions is not fixed, but with 53-bit partial
// sums the fp32 factor that comes out is the same in practice.
__global__ void __launch_bounds__(256)
k_seg_sqnorm(const float* __restrict__ w, const float* __restrict__ g, int64_t n,
             const int64_t* __restrict__ seg_offset, const float* __restrict__ seg_wd, int nseg, float gscale,
             double* __restrict__ ss) {
  __shared__ int64_t s_off[OPT_MAX_SEG + 1];
  __shared__ float s_wd[OPT_MAX_SEG];
  for (int i = threadIdx.x; i <= nseg; i += 256) s_off[i] = seg_offset[i];
  for (int i = threadIdx.x; i < nseg; i += 256) s_wd[i] = seg_wd[i];
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int seg = seg_find(s_off, nseg, i);
  double acc = 0.0;
  for (; i < n; i += stride) {
    int s2 = seg;
    while (s_off[s2 + 1] <= i) ++s2;
    if (s2 != seg) { atomicAdd(&ss[seg], acc); acc = 0.0; seg = s2; }
    const float gp = g[i] * gscale + s_wd[seg] * w[i];
    acc += (double)gp * (double)gp;
  }
  atomicAdd(&ss[seg], acc);
}

__global__ void k_clip_factor(const double* __restrict__ ss, int nseg, float clip, float* __restrict__ factor) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < nseg) {
    const float nrm = sqrtf((float)ss[s]);          // tf.clip_by_norm: g * clip / max(||g||, clip)
    factor[s] = clip / fmaxf(nrm, clip);
  }
}

extern "C" size_t lmh_grad_clip_workspace_bytes(int nseg) { return lmh_align_up((size_t)nseg * sizeof(double), 256); }

extern "C" int lmh_grad_clip_factors(const float* w, const float* g, int64_t n, const int64_t* seg_offset,
                                     const float* seg_wd, int nseg, float gscale, float clip_norm, float* factors,
                                     void* ws, size_t ws_bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(w && g && seg_offset && seg_wd && factors && ws && n > 0 && nseg > 0 && nseg <= OPT_MAX_SEG);
  LMH_CHECK_ARG(clip_norm > 0.f && ws_bytes >= lmh_grad_clip_workspace_bytes(nseg));
  hipStream_t st = (hipStream_t)stream;
  double* ss = reinterpret_cast<double*>(ws);
  LMH_CHECK_HIP(lmh_memset_async(ss, 0, sizeof(double) * nseg, st));
  const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  lmh_launch(k_seg_sqnorm, dim3(blocks), dim3(256), 0, st, w, g, n, seg_offset, seg_wd, nseg, gscale, ss);
  lmh_launch(k_clip_factor, dim3((nseg + 255) / 256), dim3(256), 0, st, (const double*)ss, nseg, clip_norm,
                     factors);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// kind 0: momentum / gradient descent  v = p1*v + g' ; w -= lr*v
// kind 1: Adam      m = p1*m + (1-p1)*g' ; v = p2*v + (1-p2)*g'^2 ; w -= lr * m / (sqrt(v) + eps)
//                   (lr = lr_t = lr0 * sqrt(1-p2^t) / (1-p1^t), computed by the host like TF's _prepare)
// kind 2: RMSProp   ms = p1*ms + (1-p1)*g'^2 ; mom = p2*mom + lr*g'/sqrt(ms + eps) ; w -= mom
// kind 3: Nesterov momentum (tf.train.MomentumOptimizer(use_nesterov=True), ApplyMomentum)
//                   v = p1*v + g' ; w -= g'*lr + v*p1*lr
// kind 4: centered RMSProp (tf.train.RMSPropOptimizer(centered=True), ApplyCenteredRMSProp)
//                   mg = p1*mg + (1-p1)*g' ; ms = p1*ms + (1-p1)*g'^2 ; mom = p2*mom + lr*g'/sqrt(ms - mg^2 + eps) ; w -= mom
template <int KIND>
__global__ void __launch_bounds__(256)
k_optimizer(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ s1, float* __restrict__ s2,
            float* __restrict__ s3, int64_t n, const int64_t* __restrict__ seg_offset, const float* __restrict__ seg_wd,
            const float* __restrict__ seg_factor, int nseg, float lr, float p1, float p2, float eps, float gscale) {
  __shared__ int64_t s_off[OPT_MAX_SEG + 1];
  __shared__ float s_wd[OPT_MAX_SEG], s_f[OPT_MAX_SEG];
  for (int i = threadIdx.x; i <= nseg; i += 256) s_off[i] = seg_offset[i];
  for (int i = threadIdx.x; i < nseg; i += 256) { s_wd[i] = seg_wd[i]; s_f[i] = seg_factor ? seg_factor[i] : 1.f; }
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int seg = seg_find(s_off, nseg, i);
  for (; i < n; i += stride) {
    while (s_off[seg + 1] <= i) ++seg;
    const float wi = w[i];
    const float gp = (g[i] * gscale + s_wd[seg] * wi) * s_f[seg];
    if (KIND == 0) {
      const float v = p1 * s1[i] + gp;
      s1[i] = v;
      w[i] = wi - lr * v;
    } else if (KIND == 1) {
      const float m = p1 * s1[i] + (1.f - p1) * gp;
      const float v = p2 * s2[i] + (1.f - p2) * (gp * gp);
      s1[i] = m;
      s2[i] = v;
      w[i] = wi - lr * m / (sqrtf(v) + eps);
    } else if (KIND == 2) {
      const float ms = p1 * s1[i] + (1.f - p1) * (gp * gp);
      const float mom = p2 * s2[i] + lr * gp / sqrtf(ms + eps);
      s1[i] = ms;
      s2[i] = mom;
      w[i] = wi - mom;
    } else if (KIND == 3) {
      const float v = p1 * s1[i] + gp;
      s1[i] = v;
      w[i] = wi - (gp * lr + v * p1 * lr);
    } else {
      const float mg = p1 * s3[i] + (1.f - p1) * gp;
      const float ms = p1 * s1[i] + (1.f - p1) * (gp * gp);
      const float mom = p2 * s2[i] + lr * gp / sqrtf(ms - mg * mg + eps);
      s3[i] = mg;
      s1[i] = ms;
      s2[i] = mom;
      w[i] = wi - mom;
    }
  }
}

extern "C" int lmh_optimizer_step(int kind, float* w, const float* g, float* slot1, float* slot2, float* slot3, int64_t n,
                                  const int64_t* seg_offset, const float* seg_wd, const float* seg_factor, int nseg,
                                  float lr, float p1, float p2, float eps, float gscale, lmh_stream_t stream) {
  LMH_CHECK_ARG(w && g && slot1 && seg_offset && seg_wd && n > 0 && nseg > 0 && nseg <= OPT_MAX_SEG);
  LMH_CHECK_ARG(kind >= 0 && kind <= 4 && (kind == 0 || kind == 3 || slot2 != nullptr) && (kind != 4 || slot3 != nullptr));
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_OPT(K_)                                                                                          \
  lmh_launch((k_optimizer<K_>), dim3(blocks), dim3(256), 0, st, w, g, slot1, slot2, slot3, n, seg_offset, seg_wd, \
                     seg_factor, nseg, lr, p1, p2, eps, gscale)
  if (kind == 0) LAUNCH_OPT(0);
  else if (kind == 1) LAUNCH_OPT(1);
  else if (kind == 2) LAUNCH_OPT(2);
  else if (kind == 3) LAUNCH_OPT(3);
  else LAUNCH_OPT(4);
#undef LAUNCH_OPT
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
