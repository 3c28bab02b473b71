// BatchNorm in TRAINING mode — `model.base_network.train_batch_norm: True` (luminoth/models/base/base_network.py:82-93,
// truncated_base_network.py:56-95: slim.batch_norm(is_training=True) inside resnet_arg_scope(batch_norm_epsilon=1e-5,
// batch_norm_decay=0.997, scale=True); the moving-average assignments are the UPDATE_OPS the train op depends on,
// train.py:87-88).  Non-default (base_config.yml:147), HBM-bound, written for clarity: the convolution kernels leave the
// RAW convolution z; this file normalises it with the statistics of the batch and differentiates through them.
//
//   forward :  mean[c] = sum_r z[r][c] / n ;  var[c] = sum_r (z[r][c] - mean[c])^2 / n      (two passes, like TF's CPU kernel)
//              y = act(z * s + t (+ residual)),  s = gamma * rsqrt(var + eps),  t = beta - mean * s
//              moving_mean -= (moving_mean - mean) * (1 - decay)
//              moving_var  -= (moving_var  - var * n / (n - 1)) * (1 - decay)               (tf.nn.fused_batch_norm hands the
//                                                                                            UNBIASED variance to the average)
//   backward:  g = dy * act'(y) arrives;  dbeta = sum_r g ;  dgamma = rstd * sum_r g (z - mean)
//              dz = gamma * rstd * (g - dbeta / n - (z - mean) * rstd * dgamma / n)
// Sums are two-stage and deterministic: a block reduces a slab of rows column by column (coalesced rows), one thread per
// column adds the slabs in order.
#include "lmh_common.h"

#define BNT_MAX_K 4096

template <int MODE>   // 0: sum z;  1: sum (z - mean)^2;  2: sum g and sum g (z - mean)
__global__ void __launch_bounds__(256)
k_bn_partial(const float* __restrict__ z, const float* __restrict__ g, const float* __restrict__ mean, int64_t rows, int K,
             int rpb, float* __restrict__ part) {
  const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
  for (int c = threadIdx.x; c < K; c += 256) {
    const float m = MODE ? mean[c] : 0.f;
    float a = 0.f, b = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      const float zv = z[(size_t)r * K + c];
      if (MODE == 0) a += zv;
      else if (MODE == 1) a += (zv - m) * (zv - m);
      else { const float gv = g[(size_t)r * K + c]; a += gv; b += gv * (zv - m); }
    }
    part[((size_t)blockIdx.x * 2 + 0) * K + c] = a;
    if (MODE == 2) part[((size_t)blockIdx.x * 2 + 1) * K + c] = b;
  }
}

__global__ void __launch_bounds__(256)
k_bn_finish_mean(const float* __restrict__ part, int nb, int64_t rows, int K, float* __restrict__ mean) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= K) return;
  float s = 0.f;
  for (int b = 0; b < nb; ++b) s += part[((size_t)b * 2) * K + c];
  mean[c] = s / (float)rows;
}

__global__ void __launch_bounds__(256)
k_bn_finish_var(const float* __restrict__ part, int nb, int64_t rows, int K, float eps, float decay,
                const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                float* __restrict__ moving_mean, float* __restrict__ moving_var, int update, float* __restrict__ rstd,
                float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= K) return;
  float s = 0.f;
  for (int b = 0; b < nb; ++b) s += part[((size_t)b * 2) * K + c];
  const float var = s / (float)rows;
  const float rs = 1.f / sqrtf(var + eps);
  const float sc = gamma[c] * rs;
  rstd[c] = rs;
  scale[c] = sc;
  shift[c] = beta[c] - mean[c] * sc;
  if (update) {      // assign_moving_average: variable -= (variable - value) * (1 - decay)
    const float unbiased = rows > 1 ? var * ((float)rows / (float)(rows - 1)) : var;
    moving_mean[c] -= (moving_mean[c] - mean[c]) * (1.f - decay);
    moving_var[c] -= (moving_var[c] - unbiased) * (1.f - decay);
  }
}

__global__ void __launch_bounds__(256)
k_bn_apply_fwd(const float* __restrict__ z, const float* __restrict__ scale, const float* __restrict__ shift,
               const float* __restrict__ residual, int act, int64_t n, int K, float* __restrict__ y) {
  const float hi = act == 2 ? 6.f : INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % K);
    float v = z[i] * scale[c] + shift[c];
    if (residual) v += residual[i];
    if (act) v = fminf(fmaxf(v, 0.f), hi);
    y[i] = v;
  }
}

__global__ void __launch_bounds__(256)
k_bn_finish_bwd(const float* __restrict__ part, int nb, int K, const float* __restrict__ rstd, float* __restrict__ dgamma,
                float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= K) return;
  float sg = 0.f, sgz = 0.f;
  for (int b = 0; b < nb; ++b) {
    sg += part[((size_t)b * 2 + 0) * K + c];
    sgz += part[((size_t)b * 2 + 1) * K + c];
  }
  dbeta[c] = sg;
  dgamma[c] = rstd[c] * sgz;
}

__global__ void __launch_bounds__(256)
k_bn_apply_bwd(const float* __restrict__ g, const float* __restrict__ z, const float* __restrict__ mean,
               const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ dgamma,
               const float* __restrict__ dbeta, const float* __restrict__ addend, int frozen, int64_t rows, int64_t n,
               int K, float* __restrict__ dz) {
  const float inv_n = 1.f / (float)rows;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % K);
    float v;
    if (frozen) {       // statistics are constants (inference-mode BatchNorm with trainable gamma / beta)
      v = gamma[c] * rstd[c] * g[i];
    } else {
      const float xhat = (z[i] - mean[c]) * rstd[c];
      v = gamma[c] * rstd[c] * (g[i] - dbeta[c] * inv_n - xhat * dgamma[c] * inv_n);
    }
    dz[i] = addend ? v + addend[i] : v;
  }
}

static int bnt_blocks(int64_t rows, int* rpb) {
  int r = (int)((rows + 1023) / 1024);
  if (r < 8) r = 8;
  *rpb = r;
  return (int)((rows + r - 1) / r);
}
extern "C" size_t lmh_bn_train_workspace_bytes(int64_t rows, int K) {
  int rpb;
  const int nb = bnt_blocks(rows, &rpb);
  return lmh_align_up(((size_t)nb * 2 * K + 2 * (size_t)K) * sizeof(float), 256);
}

extern "C" int lmh_bn_train_fwd(const float* z, int64_t rows, int K, const float* gamma, const float* beta, float eps,
                                float decay, float* moving_mean, float* moving_var, int update_moving,
                                const float* residual, int act, float* y, float* mean, float* rstd, void* ws,
                                size_t ws_bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(z && gamma && beta && y && mean && rstd && rows > 0 && K > 0 && K <= BNT_MAX_K && act >= 0 && act <= 2);
  LMH_CHECK_ARG(!update_moving || (moving_mean && moving_var));
  if (!ws || ws_bytes < lmh_bn_train_workspace_bytes(rows, K)) {
    lmh_set_error("lmh_bn_train_fwd: workspace too small");
    return LMH_ERR_WORKSPACE;
  }
  int rpb;
  const int nb = bnt_blocks(rows, &rpb);
  float* part = reinterpret_cast<float*>(ws);
  float* scale = part + (size_t)nb * 2 * K;
  float* shift = scale + K;
  hipStream_t st = (hipStream_t)stream;
  const int cb = (K + 255) / 256;
  const int64_t n = rows * K;
  const int eb = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  lmh_launch(k_bn_partial<0>, dim3(nb), dim3(256), 0, st, z, (const float*)nullptr, (const float*)nullptr, rows, K, rpb, part);
  lmh_launch(k_bn_finish_mean, dim3(cb), dim3(256), 0, st, (const float*)part, nb, rows, K, mean);
  lmh_launch(k_bn_partial<1>, dim3(nb), dim3(256), 0, st, z, (const float*)nullptr, (const float*)mean, rows, K, rpb, part);
  lmh_launch(k_bn_finish_var, dim3(cb), dim3(256), 0, st, (const float*)part, nb, rows, K, eps, decay, gamma, beta,
             (const float*)mean, moving_mean, moving_var, update_moving, rstd, scale, shift);
  lmh_launch(k_bn_apply_fwd, dim3(eb), dim3(256), 0, st, z, (const float*)scale, (const float*)shift, residual, act, n, K, y);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

extern "C" int lmh_bn_train_bwd(const float* g, const float* z, const float* mean, const float* rstd, const float* gamma,
                                int64_t rows, int K, const float* addend, int frozen_statistics, float* dgamma,
                                float* dbeta, float* dz, void* ws, size_t ws_bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(g && z && mean && rstd && gamma && dgamma && dbeta && rows > 0 && K > 0 && K <= BNT_MAX_K);
  if (!ws || ws_bytes < lmh_bn_train_workspace_bytes(rows, K)) {
    lmh_set_error("lmh_bn_train_bwd: workspace too small");
    return LMH_ERR_WORKSPACE;
  }
  int rpb;
  const int nb = bnt_blocks(rows, &rpb);
  float* part = reinterpret_cast<float*>(ws);
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = rows * K;
  const int eb = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  lmh_launch(k_bn_partial<2>, dim3(nb), dim3(256), 0, st, z, g, mean, rows, K, rpb, part);
  lmh_launch(k_bn_finish_bwd, dim3((K + 255) / 256), dim3(256), 0, st, (const float*)part, nb, K, rstd, dgamma, dbeta);
  if (dz)       // (NULL: only the parameter gradients are wanted — nothing below needs the data gradient)
    lmh_launch(k_bn_apply_bwd, dim3(eb), dim3(256), 0, st, g, z, mean, rstd, gamma, (const float*)dgamma,
               (const float*)dbeta, addend, frozen_statistics, rows, n, K, dz);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// y = act(z * scale[c] + shift[c] (+ residual)): a BatchNorm with frozen statistics that does NOT follow a convolution
// (the `preact` of slim's resnet_v2 units, base_network.py:94-101) — convolutions fold the same expression into their epilogue.
extern "C" int lmh_bn_apply(const float* z, int64_t rows, int K, const float* scale, const float* shift,
                            const float* residual, int act, float* y, lmh_stream_t stream) {
  LMH_CHECK_ARG(z && scale && shift && y && rows > 0 && K > 0 && act >= 0 && act <= 2);
  const int64_t n = rows * K;
  const int eb = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  lmh_launch(k_bn_apply_fwd, dim3(eb), dim3(256), 0, (hipStream_t)stream, z, scale, shift, residual, act, n, K, y);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
