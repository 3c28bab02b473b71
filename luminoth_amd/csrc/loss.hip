// Fused loss forward + gradient kernels (gfx950).
//
// RPN.loss   luminoth/models/fasterrcnn/rpn.py:219-309
// RCNN.loss  luminoth/models/fasterrcnn/rcnn.py:255-411
// smooth_l1  luminoth/utils/losses.py:4-32
// One block per image: deterministic in-block tree reductions (no float
// atomics), gradients written in the same launch.
#include "lmh_common.h"

#define LOSS_THREADS 1024

__device__ __forceinline__ float sl1_val(float d, float sigma2) {
  const float a = fabsf(d);
  return (a < 1.0f / sigma2) ? 0.5f * sigma2 * (a * a) : a - 0.5f / sigma2;
}
__device__ __forceinline__ float sl1_grad(float d, float sigma2) {
  const float a = fabsf(d);
  return (a < 1.0f / sigma2) ? sigma2 * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
}

// deterministic block sum of up to 4 values per thread
__device__ void block_sum4(float v[4], float* sh /* 4 * LOSS_THREADS/64 */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float x = v[q];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
    if (lane == 0) sh[q * 16 + wave] = x;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += sh[threadIdx.x * 16 + w];
    sh[64 + threadIdx.x] = t;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = sh[64 + q];
  __syncthreads();
}

__global__ void __launch_bounds__(LOSS_THREADS)
k_rpn_loss(const float* __restrict__ cls_score, const float* __restrict__ bbox_pred,
           const float* __restrict__ labels, const float* __restrict__ bbox_targets, int B, int N,
           float sigma2, float* __restrict__ per_image) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain beside MFMA kernels of other streams: win the issue arbitration
  __shared__ float sh[72];
  const int b = blockIdx.x;
  const float2* cs = reinterpret_cast<const float2*>(cls_score) + (size_t)b * N;
  const float4* bp = reinterpret_cast<const float4*>(bbox_pred) + (size_t)b * N;
  const float4* bt = reinterpret_cast<const float4*>(bbox_targets) + (size_t)b * N;
  const float* lab = labels + (size_t)b * N;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};  // ce_sum, n_cls, reg_sum, n_pos
  // four label loads in flight per thread (all but ~256 of the 49 152 labels are -1: the loop is load latency)
  for (int n0 = threadIdx.x; n0 < N; n0 += 4 * blockDim.x) {
    float l4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = n0 + u * blockDim.x;
      l4[u] = n < N ? lab[n] : -1.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = n0 + u * blockDim.x;
      const float l = l4[u];
      if (l != -1.f) {
        const float2 s = cs[n];
        const float m = fmaxf(s.x, s.y);
        const float lse = logf(expf(s.x - m) + expf(s.y - m));
        const float z = ((l == 1.f) ? s.y : s.x) - m;
        acc[0] += lse - z;
        acc[1] += 1.f;
      }
      if (l == 1.f) {
        const float4 p = bp[n], t = bt[n];
        acc[2] += ((sl1_val(p.x - t.x, sigma2) + sl1_val(p.y - t.y, sigma2)) + sl1_val(p.z - t.z, sigma2)) +
                  sl1_val(p.w - t.w, sigma2);
        acc[3] += 1.f;
      }
    }
  }
  block_sum4(acc, sh);
  if (threadIdx.x == 0) {
    per_image[b * 4 + 0] = acc[0] / acc[1];  // mean of empty -> NaN, as tf.reduce_mean
    per_image[b * 4 + 1] = acc[2] / acc[3];
    per_image[b * 4 + 2] = acc[1];
    per_image[b * 4 + 3] = acc[3];
  }
}

// Gradient pass of the RPN loss, one thread per anchor over the whole grid (the sums above are a one-block-per-image
// reduction; writing the 2+4 floats of all 49 152 anchors from that same block kept the step's critical path
// waiting ~50 us on a single CU).  Same arithmetic as before: gc / gr from the per-image counts.
__global__ void __launch_bounds__(256)
k_rpn_loss_grad(const float* __restrict__ cls_score, const float* __restrict__ bbox_pred,
                const float* __restrict__ labels, const float* __restrict__ bbox_targets, int B, int N,
                float sigma2, float w_cls, float w_reg, const float* __restrict__ per_image,
                float* __restrict__ d_cls, float* __restrict__ d_bbox) {
  __builtin_amdgcn_s_setprio(3);
  const int b = blockIdx.y;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float gc = w_cls / (per_image[b * 4 + 2] * (float)B);
  const float gr = w_reg / (per_image[b * 4 + 3] * (float)B);
  const size_t row = (size_t)b * N + n;
  const float l = labels[row];
  float2 g = make_float2(0.f, 0.f);
  float4 gb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (l != -1.f) {
    const float2 s = reinterpret_cast<const float2*>(cls_score)[row];
    const float m = fmaxf(s.x, s.y);
    const float e0 = expf(s.x - m), e1 = expf(s.y - m);
    const float den = e0 + e1;
    const float p0 = e0 / den, p1 = e1 / den;
    g.x = (p0 - ((l == 1.f) ? 0.f : 1.f)) * gc;
    g.y = (p1 - ((l == 1.f) ? 1.f : 0.f)) * gc;
  }
  if (l == 1.f) {
    const float4 p = reinterpret_cast<const float4*>(bbox_pred)[row];
    const float4 t = reinterpret_cast<const float4*>(bbox_targets)[row];
    gb.x = sl1_grad(p.x - t.x, sigma2) * gr;
    gb.y = sl1_grad(p.y - t.y, sigma2) * gr;
    gb.z = sl1_grad(p.z - t.z, sigma2) * gr;
    gb.w = sl1_grad(p.w - t.w, sigma2) * gr;
  }
  if (d_cls) reinterpret_cast<float2*>(d_cls)[row] = g;
  if (d_bbox) reinterpret_cast<float4*>(d_bbox)[row] = gb;
}

// losses[q] = mean_b per_image[b*4 + q], q in {0,1}
__global__ void k_loss_mean(const float* __restrict__ per_image, int B, float w0, float w1,
                            float* __restrict__ losses) {
  if (threadIdx.x < 2) {
    float t = 0.f;
    for (int b = 0; b < B; ++b) t += per_image[b * 4 + threadIdx.x];
    losses[threadIdx.x] = (t / (float)B) * (threadIdx.x == 0 ? w0 : w1);
  }
}

extern "C" int lmh_rpn_loss(const float* cls_score, const float* bbox_pred, const float* labels,
                            const float* bbox_targets, int B, int N, float sigma, float w_cls,
                            float w_reg, float* losses, float* per_image, float* d_cls_score,
                            float* d_bbox_pred, lmh_stream_t stream) {
  LMH_CHECK_ARG(cls_score && bbox_pred && labels && bbox_targets && losses && per_image);
  LMH_CHECK_ARG(B > 0 && N > 0);
  hipStream_t st = (hipStream_t)stream;
  lmh_launch(k_rpn_loss, dim3(B), dim3(LOSS_THREADS), 0, st, cls_score, bbox_pred, labels,
                     bbox_targets, B, N, sigma * sigma, per_image);
  if (d_cls_score || d_bbox_pred)
    lmh_launch(k_rpn_loss_grad, dim3((N + 255) / 256, B), dim3(256), 0, st, cls_score, bbox_pred, labels,
                       bbox_targets, B, N, sigma * sigma, w_cls, w_reg, per_image, d_cls_score, d_bbox_pred);
  lmh_launch(k_loss_mean, dim3(1), dim3(64), 0, st, per_image, B, w_cls, w_reg, losses);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// ---------------------------------------------------------------------------
// RCNN loss: rows (B,R); a wave per row, lanes over classes.
// ---------------------------------------------------------------------------
// 256 threads and four rows per trip (81 VGPRs, one wave per SIMD): such a block fits beside two resident MFMA blocks
// (188-208 VGPRs x 2 waves per SIMD) on any CU.  Rounds 3-4 history: 1024 threads waited up to 200 us for a CU with room
// beside the convolution blocks of the other streams; 512 threads x 129 VGPRs (eight rows per trip) still needed a CU with
// only ONE resident MFMA block: 31 us alone, 97-185 us inside the step (profiles/r04_bench_summary.md).
#define RCNN_LOSS_THREADS 256
#define RCNN_LOSS_ROWS 4      // rows of a wave per trip (all their loads in flight together)
__global__ void __launch_bounds__(RCNN_LOSS_THREADS)
k_rcnn_loss(const float* __restrict__ cls_score, const float* __restrict__ bbox_offsets,
            const float* __restrict__ labels, const float* __restrict__ targets, int B, int R, int C,
            float sigma2, float* __restrict__ per_image) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain beside MFMA kernels of other streams: win the issue arbitration
  __shared__ float sh[72];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int C1 = C + 1;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (C1 <= 128) {
    // The rows a wave owns, RCNN_LOSS_ROWS at a time with every load of the batch issued together: two rounds of memory latency per
    // chunk (labels, then scores + the few values only lane 0 consumes, fetched by lanes 0..8) instead of three dependent
    // rounds per row — the kernel is one block per image on the critical proposal -> RCNN chain (56-71 us inside the step
    // before).  Same arithmetic in the same order as the row-by-row loop below: bit-identical sums.
    for (int r0 = wave; r0 < R; r0 += RCNN_LOSS_ROWS * nw) {
      float l[RCNN_LOSS_ROWS], s0[RCNN_LOSS_ROWS], s1[RCNN_LOSS_ROWS], ex[RCNN_LOSS_ROWS];
#pragma unroll
      for (int q = 0; q < RCNN_LOSS_ROWS; ++q) {
        const int r = r0 + q * nw;
        l[q] = (r < R) ? labels[(size_t)b * R + r] : -1.f;
      }
#pragma unroll
      for (int q = 0; q < RCNN_LOSS_ROWS; ++q) {
        const size_t row = (size_t)b * R + r0 + q * nw;
        const bool on = l[q] >= 0.f;
        const float* s = cls_score + row * C1;
        s0[q] = (on && lane < C1) ? s[lane] : -INFINITY;
        s1[q] = (on && lane + 64 < C1) ? s[lane + 64] : -INFINITY;
        float e = 0.f;
        if (on && lane == 8) e = s[(int)l[q]];
        if (l[q] > 0.f && lane < 4) e = bbox_offsets[row * 4 * C + 4 * ((int)l[q] - 1) + lane];
        if (l[q] > 0.f && lane >= 4 && lane < 8) e = targets[row * 4 + (lane - 4)];
        ex[q] = e;
      }
#pragma unroll
      for (int q = 0; q < RCNN_LOSS_ROWS; ++q) {
        if (!(l[q] >= 0.f)) continue;              // wave-uniform
        float m = fmaxf(s0[q], s1[q]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float se = (lane < C1) ? expf(s0[q] - m) : 0.f;
        if (lane + 64 < C1) se += expf(s1[q] - m);
        for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
        const float sl = __shfl(ex[q], 8);
        const float d0 = __shfl(ex[q], 0) - __shfl(ex[q], 4), d1 = __shfl(ex[q], 1) - __shfl(ex[q], 5);
        const float d2 = __shfl(ex[q], 2) - __shfl(ex[q], 6), d3 = __shfl(ex[q], 3) - __shfl(ex[q], 7);
        if (lane == 0) {
          acc[0] += logf(se) - (sl - m);
          acc[1] += 1.f;
          if (l[q] > 0.f) {
            acc[2] += ((sl1_val(d0, sigma2) + sl1_val(d1, sigma2)) + sl1_val(d2, sigma2)) + sl1_val(d3, sigma2);
            acc[3] += 1.f;
          }
        }
      }
    }
  } else
  for (int r = wave; r < R; r += nw) {
    const size_t row = (size_t)b * R + r;
    const float l = labels[row];
    if (l >= 0.f) {
      const float* s = cls_score + row * C1;
      float m = -INFINITY;
      for (int c = lane; c < C1; c += 64) m = fmaxf(m, s[c]);
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      float se = 0.f;
      for (int c = lane; c < C1; c += 64) se += expf(s[c] - m);
      for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
      if (lane == 0) {
        acc[0] += logf(se) - (s[(int)l] - m);
        acc[1] += 1.f;
        if (l > 0.f) {
          const float* p = bbox_offsets + row * 4 * C + 4 * ((int)l - 1);
          const float* t = targets + row * 4;
          acc[2] += ((sl1_val(p[0] - t[0], sigma2) + sl1_val(p[1] - t[1], sigma2)) +
                     sl1_val(p[2] - t[2], sigma2)) + sl1_val(p[3] - t[3], sigma2);
          acc[3] += 1.f;
        }
      }
    }
  }
  block_sum4(acc, sh);
  if (threadIdx.x == 0) {
    per_image[b * 4 + 0] = acc[0] / acc[1];
    per_image[b * 4 + 1] = acc[2] / acc[3];
    per_image[b * 4 + 2] = acc[1];
    per_image[b * 4 + 3] = acc[3];
  }
}

// Gradient pass of the RCNN loss, one WAVE per row over the whole grid (round 3): written from the one block per image of
// the sum kernel, the 81 + 320 floats of each of the 512 rows kept the proposal / RCNN chain — the critical path of the
// middle of the step — on two CUs for 70 us.  Same arithmetic: gc / gr from the per-image counts.
__global__ void __launch_bounds__(256)
k_rcnn_loss_grad(const float* __restrict__ cls_score, const float* __restrict__ bbox_offsets,
                 const float* __restrict__ labels, const float* __restrict__ targets, int B, int R, int C,
                 float sigma2, float w_cls, float w_reg, float* __restrict__ d_cls, float* __restrict__ d_off) {
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= (int64_t)B * R) return;
  const int b = (int)(row / R);
  const int C1 = C + 1;
  // the two normalisers — #labelled and #positive rows of this image — counted by the wave itself from the image's R
  // labels (1 KB out of L2): the gradients then do not wait for the one-block-per-image sum kernel, which only feeds the
  // REPORTED loss values and can run anywhere behind (round 4).  Counts are integers: the same floats k_rcnn_loss sums.
  int n_lab = 0, n_pos = 0;
  for (int r = lane; r < R; r += 64) {
    const float lr = labels[(size_t)b * R + r];
    n_lab += (lr >= 0.f);
    n_pos += (lr > 0.f);
  }
  for (int o = 32; o > 0; o >>= 1) { n_lab += __shfl_xor(n_lab, o); n_pos += __shfl_xor(n_pos, o); }
  const float gc = w_cls / ((float)n_lab * (float)B);
  const float gr = w_reg / ((float)n_pos * (float)B);
  const float l = labels[row];
  if (d_cls) {
    const float* s = cls_score + row * C1;
    float* g = d_cls + row * C1;
    if (l >= 0.f) {
      float m = -INFINITY;
      for (int c = lane; c < C1; c += 64) m = fmaxf(m, s[c]);
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      float se = 0.f;
      for (int c = lane; c < C1; c += 64) se += expf(s[c] - m);
      for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
      for (int c = lane; c < C1; c += 64)
        g[c] = (expf(s[c] - m) / se - ((c == (int)l) ? 1.f : 0.f)) * gc;
    } else {
      for (int c = lane; c < C1; c += 64) g[c] = 0.f;
    }
  }
  if (d_off) {
    float* g = d_off + row * 4 * C;
    const int lo = (l > 0.f) ? 4 * ((int)l - 1) : -1;
    for (int c = lane; c < 4 * C; c += 64) {
      float v = 0.f;
      if (lo >= 0 && c >= lo && c < lo + 4)
        v = sl1_grad(bbox_offsets[row * 4 * C + c] - targets[row * 4 + (c - lo)], sigma2) * gr;
      g[c] = v;
    }
  }
}

static void rcnn_loss_grad_launch(const float* cls_score, const float* bbox_offsets, const float* labels,
                                  const float* targets, int B, int R, int C, float sigma, float w_cls, float w_reg,
                                  float* d_cls_score, float* d_bbox_offsets, hipStream_t st) {
  lmh_launch(k_rcnn_loss_grad, dim3((unsigned)(((int64_t)B * R + 3) / 4)), dim3(256), 0, st, cls_score, bbox_offsets,
             labels, targets, B, R, C, sigma * sigma, w_cls, w_reg, d_cls_score, d_bbox_offsets);
}

extern "C" int lmh_rcnn_loss(const float* cls_score, const float* bbox_offsets, const float* labels,
                             const float* targets, int B, int R, int C, float sigma, float w_cls,
                             float w_reg, float* losses, float* per_image, float* d_cls_score,
                             float* d_bbox_offsets, lmh_stream_t stream) {
  LMH_CHECK_ARG(cls_score && bbox_offsets && labels && targets && losses && per_image);
  LMH_CHECK_ARG(B > 0 && R > 0 && C > 0);
  hipStream_t st = (hipStream_t)stream;
  // gradients first: what follows them on the stream (the RCNN backward) does not need the sums
  if (d_cls_score || d_bbox_offsets)
    rcnn_loss_grad_launch(cls_score, bbox_offsets, labels, targets, B, R, C, sigma, w_cls, w_reg, d_cls_score,
                          d_bbox_offsets, st);
  lmh_launch(k_rcnn_loss, dim3(B), dim3(RCNN_LOSS_THREADS), 0, st, cls_score, bbox_offsets, labels,
                     targets, B, R, C, sigma * sigma, per_image);
  lmh_launch(k_loss_mean, dim3(1), dim3(64), 0, st, per_image, B, w_cls, w_reg, losses);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

extern "C" int lmh_rcnn_loss_grad(const float* cls_score, const float* bbox_offsets, const float* labels,
                                  const float* targets, int B, int R, int C, float sigma, float w_cls, float w_reg,
                                  float* d_cls_score, float* d_bbox_offsets, lmh_stream_t stream) {
  LMH_CHECK_ARG(cls_score && bbox_offsets && labels && targets && (d_cls_score || d_bbox_offsets));
  LMH_CHECK_ARG(B > 0 && R > 0 && C > 0);
  rcnn_loss_grad_launch(cls_score, bbox_offsets, labels, targets, B, R, C, sigma, w_cls, w_reg, d_cls_score,
                        d_bbox_offsets, (hipStream_t)stream);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// tf.nn.softmax over the last axis: one wave per row.
__global__ void __launch_bounds__(256)
k_softmax(const float* __restrict__ x, int64_t rows, int C, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* s = x + r * C;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, s[c]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float se = 0.f;
  for (int c = lane; c < C; c += 64) se += expf(s[c] - m);
  for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
  for (int c = lane; c < C; c += 64) y[r * C + c] = expf(s[c] - m) / se;
}
extern "C" int lmh_softmax(const float* x, int64_t rows, int C, float* y, lmh_stream_t stream) {
  LMH_CHECK_ARG(x && y && rows > 0 && C > 0);
  lmh_launch(k_softmax, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                     rows, C, y);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
