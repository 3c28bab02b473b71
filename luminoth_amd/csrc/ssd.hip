// SSD-specific stages (gfx950): conv4_3 L2 normalisation, multibox targets with hard-negative mining,
// the SSD loss.  Everything else SSD needs (convolutions incl. dilation, pools, softmax, per-class NMS)
// is shared with Faster R-CNN.
//
// Reference: luminoth/models/ssd/feature_extractor.py:75-89 (l2_normalize * gamma),
//            luminoth/models/ssd/target.py:35-200, luminoth/models/ssd/ssd.py:197-300,
//            luminoth/utils/losses.py:4-32, luminoth/utils/bbox_overlap.py:7-48,
//            luminoth/utils/bbox_transform_tf.py:18-38.
#include "lmh_common.h"

// ============================================================================
// tf.nn.l2_normalize(x, axis=3, epsilon) * gamma     x (P, C), one wave per pixel
// ============================================================================
#define L2N_MAX_C4 4   // float4 chunks per lane: C <= 1024

__global__ void __launch_bounds__(256)
k_l2norm_fwd(const float* __restrict__ x, const float* __restrict__ gamma, int64_t P, int C, float eps,
             float* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C4 = C >> 2;
  for (int64_t p = (int64_t)blockIdx.x * 4 + wave; p < P; p += (int64_t)gridDim.x * 4) {
    const float4* xp = reinterpret_cast<const float4*>(x + p * C);
    float4 v[L2N_MAX_C4];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < L2N_MAX_C4; ++i) {
      const int c4 = lane + 64 * i;
      v[i] = (c4 < C4) ? xp[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
      ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float r = rsqrtf(fmaxf(ss, eps));
#pragma unroll
    for (int i = 0; i < L2N_MAX_C4; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < C4) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[c4];
        reinterpret_cast<float4*>(y + p * C)[c4] =
            make_float4(v[i].x * r * g.x, v[i].y * r * g.y, v[i].z * r * g.z, v[i].w * r * g.w);
      }
    }
  }
}

// dx = r*(dn - n*<dn,n>) (dn = dy*gamma, n = x*r; the clamped branch ss < eps passes dn*r through);
// dgamma partial per block: part[block][C] = sum_pixels dy*n  (folded by k_l2norm_finish)
__global__ void __launch_bounds__(256)
k_l2norm_bwd(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma, int64_t P,
             int C, float eps, float* __restrict__ dx, float* __restrict__ part) {
  __shared__ float red[4][L2N_MAX_C4 * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C4 = C >> 2;
  float4 dg[L2N_MAX_C4];
#pragma unroll
  for (int i = 0; i < L2N_MAX_C4; ++i) dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t p = (int64_t)blockIdx.x * 4 + wave; p < P; p += (int64_t)gridDim.x * 4) {
    float4 v[L2N_MAX_C4], d[L2N_MAX_C4];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < L2N_MAX_C4; ++i) {
      const int c4 = lane + 64 * i;
      const bool ok = c4 < C4;
      v[i] = ok ? reinterpret_cast<const float4*>(x + p * C)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
      d[i] = ok ? reinterpret_cast<const float4*>(dy + p * C)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
      ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float r = rsqrtf(fmaxf(ss, eps));
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < L2N_MAX_C4; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < C4) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[c4];
        const float4 n = make_float4(v[i].x * r, v[i].y * r, v[i].z * r, v[i].w * r);
        dg[i].x += d[i].x * n.x; dg[i].y += d[i].y * n.y; dg[i].z += d[i].z * n.z; dg[i].w += d[i].w * n.w;
        d[i] = make_float4(d[i].x * g.x, d[i].y * g.y, d[i].z * g.z, d[i].w * g.w);   // dn
        dot += d[i].x * n.x + d[i].y * n.y + d[i].z * n.z + d[i].w * n.w;
        v[i] = n;
      }
    }
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    if (ss < eps) dot = 0.f;
#pragma unroll
    for (int i = 0; i < L2N_MAX_C4; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < C4)
        reinterpret_cast<float4*>(dx + p * C)[c4] =
            make_float4(r * (d[i].x - v[i].x * dot), r * (d[i].y - v[i].y * dot), r * (d[i].z - v[i].z * dot),
                        r * (d[i].w - v[i].w * dot));
    }
  }
#pragma unroll
  for (int i = 0; i < L2N_MAX_C4; ++i) {
    const int c = 4 * (lane + 64 * i);
    red[wave][c + 0] = dg[i].x; red[wave][c + 1] = dg[i].y; red[wave][c + 2] = dg[i].z; red[wave][c + 3] = dg[i].w;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256)
    part[(size_t)blockIdx.x * C + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

__global__ void __launch_bounds__(256)
k_l2norm_finish(const float* __restrict__ part, int nb, int C, float* __restrict__ dgamma) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < C)
    for (int b = g; b < nb; b += 8) s += part[(size_t)b * C + c];
  red[g][cl] = s;
  __syncthreads();
  if (g == 0 && c < C) {
    float t = red[0][cl];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += red[i][cl];
    dgamma[c] = t;
  }
}

static int l2n_blocks(int64_t P) { return (int)((P + 3) / 4 < 512 ? (P + 3) / 4 : 512); }

extern "C" int lmh_l2norm_scale_fwd(const float* x, const float* gamma, int64_t P, int C, float eps, float* y,
                                    lmh_stream_t stream) {
  LMH_CHECK_ARG(x && gamma && y && P > 0 && C > 0 && (C & 3) == 0 && C <= 256 * L2N_MAX_C4);
  lmh_launch(k_l2norm_fwd, dim3(l2n_blocks(P)), dim3(256), 0, (hipStream_t)stream, x, gamma, P, C, eps, y);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

extern "C" size_t lmh_l2norm_scale_bwd_workspace_bytes(int64_t P, int C) {
  return lmh_align_up((size_t)l2n_blocks(P) * C * sizeof(float), 256);
}

extern "C" int lmh_l2norm_scale_bwd(const float* x, const float* dy, const float* gamma, int64_t P, int C, float eps,
                                    float* dx, float* dgamma, void* ws, size_t ws_bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(x && dy && gamma && dx && dgamma && P > 0 && C > 0 && (C & 3) == 0 && C <= 256 * L2N_MAX_C4);
  if (!ws || ws_bytes < lmh_l2norm_scale_bwd_workspace_bytes(P, C)) {
    lmh_set_error("lmh_l2norm_scale_bwd: workspace too small");
    return LMH_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int nb = l2n_blocks(P);
  float* part = reinterpret_cast<float*>(ws);
  lmh_launch(k_l2norm_bwd, dim3(nb), dim3(256), 0, st, x, dy, gamma, P, C, eps, dx, part);
  lmh_launch(k_l2norm_finish, dim3((C + 31) / 32), dim3(256), 0, st, part, nb, C, dgamma);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// ============================================================================
// SSDTarget (models/ssd/target.py:35-200), batched.  anchors (N,4) fp32 are shared by all images.
// ============================================================================
struct ssd_ws {
  unsigned long long* gt_best;   // [B][Gmax]  (orderable(iou) << 32) | ~anchor  -> per-gt argmax, first occurrence
  uint64_t* keys;                // [B][Npad]  hard-negative ranking keys
  int32_t* best_gt;              // [B][N]
  int32_t* num_fg;               // [B]
};
static size_t ssd_ws_layout(int B, int N, int Npad, int Gmax, void* base, ssd_ws* w) {
  size_t off = 0;
  char* p = reinterpret_cast<char*>(base);
  w->gt_best = reinterpret_cast<unsigned long long*>(p + off); off += lmh_align_up((size_t)B * Gmax * 8, 256);
  w->keys = reinterpret_cast<uint64_t*>(p + off); off += lmh_align_up((size_t)B * Npad * 8, 256);
  w->best_gt = reinterpret_cast<int32_t*>(p + off); off += lmh_align_up((size_t)B * N * 4, 256);
  w->num_fg = reinterpret_cast<int32_t*>(p + off); off += lmh_align_up((size_t)B * 4, 256);
  return off;
}

// a: per-anchor max / argmax IoU and the fg rule; per-gt best anchor through an ordered u64 atomic max
__global__ void __launch_bounds__(256)
k_ssd_target_a(lmh_ssd_target_desc d, const float4* __restrict__ anchors, const float* __restrict__ gt,
               const int32_t* __restrict__ gt_count, float* __restrict__ labels, float* __restrict__ max_ov,
               ssd_ws w) {
  const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
  if (n >= d.N) return;
  const int G = min(gt_count[b], d.Gmax);
  const float4 a4 = anchors[n];
  const lmh_box a{a4.x, a4.y, a4.z, a4.w};
  const float aa = lmh_area_plus1(a);
  float best = -1.f;   // tf.reduce_max over an empty gt set is -inf; G == 0 leaves everything background-free
  int arg = 0;
  for (int g = 0; g < G; ++g) {
    const float* gp = gt + ((size_t)b * d.Gmax + g) * 5;
    const lmh_box gb{gp[0], gp[1], gp[2], gp[3]};
    const float iou = lmh_iou_plus1(a, aa, gb, lmh_area_plus1(gb));
    if (iou > best) { best = iou; arg = g; }
    const unsigned long long key = ((unsigned long long)lmh_float_orderable(iou) << 32) | (unsigned)(~(unsigned)n);
    atomicMax(&w.gt_best[(size_t)b * d.Gmax + g], key);
  }
  float lab = -1.f;
  if (G > 0 && best >= d.foreground_threshold) lab = gt[((size_t)b * d.Gmax + arg) * 5 + 4] + 1.f;
  labels[(size_t)b * d.N + n] = lab;
  max_ov[(size_t)b * d.N + n] = G > 0 ? best : 0.f;
  w.best_gt[(size_t)b * d.N + n] = arg;
}

// b: best-anchor override (last gt wins on duplicates), hard-negative score keys, fg count
__global__ void __launch_bounds__(256)
k_ssd_target_b(lmh_ssd_target_desc d, int Npad, const float* __restrict__ gt, const int32_t* __restrict__ gt_count,
               const float* __restrict__ probs, float* __restrict__ labels, const float* __restrict__ max_ov,
               ssd_ws w) {
  const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
  if (n >= Npad) return;
  uint64_t key = ~0ull;
  if (n < d.N) {
    const int G = min(gt_count[b], d.Gmax);
    float lab = labels[(size_t)b * d.N + n];
    for (int g = 0; g < G; ++g) {
      const unsigned bestn = ~(unsigned)(w.gt_best[(size_t)b * d.Gmax + g] & 0xFFFFFFFFull);
      if (bestn == (unsigned)n) lab = gt[((size_t)b * d.Gmax + g) * 5 + 4] + 1.f;
    }
    labels[(size_t)b * d.N + n] = lab;
    if (lab > 0.f) atomicAdd(&w.num_fg[b], 1);
    const float* pr = probs + ((size_t)b * d.N + n) * (d.C + 1);
    float mp = pr[1];
    for (int c = 2; c <= d.C; ++c) mp = fmaxf(mp, pr[c]);
    const bool cand = (max_ov[(size_t)b * d.N + n] <= d.background_threshold_high) && (lab <= 0.f);
    const float score = cand ? mp : -1.f;
    key = ((uint64_t)(~lmh_float_orderable(score)) << 32) | (uint32_t)n;   // ascending sort = top_k order
  }
  w.keys[(size_t)b * Npad + n] = key;
}

// c: the num_bg best-ranked rows become background (label 0) — whatever they were (target.py:149-160)
__global__ void __launch_bounds__(256)
k_ssd_target_c(lmh_ssd_target_desc d, int Npad, float* __restrict__ labels, ssd_ws w) {
  const int b = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
  int num_bg = (int)((float)w.num_fg[b] * d.hard_negative_ratio);
  if (num_bg > d.N) num_bg = d.N;
  if (r >= num_bg) return;
  const uint32_t idx = (uint32_t)(w.keys[(size_t)b * Npad + r] & 0xFFFFFFFFull);
  labels[(size_t)b * d.N + idx] = 0.f;
}

// d: bbox targets for the rows that are still foreground
__global__ void __launch_bounds__(256)
k_ssd_target_d(lmh_ssd_target_desc d, const float4* __restrict__ anchors, const float* __restrict__ gt,
               const float* __restrict__ labels, float4* __restrict__ targets, ssd_ws w) {
  const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
  if (n >= d.N) return;
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  if (labels[(size_t)b * d.N + n] > 0.f) {
    const float4 a4 = anchors[n];
    const float* gp = gt + ((size_t)b * d.Gmax + w.best_gt[(size_t)b * d.N + n]) * 5;
    float o[4];
    lmh_encode(lmh_box{a4.x, a4.y, a4.z, a4.w}, lmh_box{gp[0], gp[1], gp[2], gp[3]}, d.variance_xy, d.variance_wh, o);
    t = make_float4(o[0], o[1], o[2], o[3]);
  }
  targets[(size_t)b * d.N + n] = t;
}

extern "C" size_t lmh_ssd_target_workspace_bytes(const lmh_ssd_target_desc* d) {
  if (!d) return 0;
  ssd_ws w;
  return ssd_ws_layout(d->B, d->N, lmh_next_pow2(d->N), d->Gmax, nullptr, &w);
}

extern "C" int lmh_ssd_target(const lmh_ssd_target_desc* d, const float* anchors, const float* gt,
                              const int32_t* gt_count, const float* probs, float* labels, float* bbox_targets,
                              float* max_overlaps, void* ws, size_t ws_bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(d && anchors && gt && gt_count && probs && labels && bbox_targets && max_overlaps);
  LMH_CHECK_ARG(d->B > 0 && d->N > 0 && d->C > 0 && d->Gmax > 0);
  if (!ws || ws_bytes < lmh_ssd_target_workspace_bytes(d)) {
    lmh_set_error("lmh_ssd_target: workspace too small");
    return LMH_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int Npad = lmh_next_pow2(d->N);
  ssd_ws w;
  ssd_ws_layout(d->B, d->N, Npad, d->Gmax, ws, &w);
  LMH_CHECK_HIP(lmh_memset_async(w.gt_best, 0, (size_t)d->B * d->Gmax * 8, st));
  LMH_CHECK_HIP(lmh_memset_async(w.num_fg, 0, (size_t)d->B * 4, st));
  const dim3 gn((d->N + 255) / 256, d->B), gp((Npad + 255) / 256, d->B);
  lmh_launch(k_ssd_target_a, gn, dim3(256), 0, st, *d, reinterpret_cast<const float4*>(anchors), gt, gt_count,
                     labels, max_overlaps, w);
  lmh_launch(k_ssd_target_b, gp, dim3(256), 0, st, *d, Npad, gt, gt_count, probs, labels, max_overlaps, w);
  int rc = lmh_sort_u64_impl(w.keys, d->B, Npad, st);
  if (rc) return rc;
  lmh_launch(k_ssd_target_c, gp, dim3(256), 0, st, *d, Npad, labels, w);
  lmh_launch(k_ssd_target_d, gn, dim3(256), 0, st, *d, reinterpret_cast<const float4*>(anchors), gt, labels,
                     reinterpret_cast<float4*>(bbox_targets), w);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// ============================================================================
// SSD.loss (models/ssd/ssd.py:197-300): per image (sum CE over rows with label >= 0 + w_loc * sum
// smooth-L1(sigma 3) over rows with label > 0) / #positives, 0 without positives; batch = mean over
// images.  One block per image, deterministic reductions, gradients in the same launch.
// ============================================================================
#define SSD_LOSS_THREADS 1024
#define SSD_MAX_CLS 128

__device__ float ssd_block_sum(float x, float* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
  if (lane == 0) sh[wave] = x;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) {
    for (int w2 = 0; w2 < SSD_LOSS_THREADS / 64; ++w2) t += sh[w2];
    sh[32] = t;
  }
  __syncthreads();
  t = sh[32];
  __syncthreads();
  return t;
}

__global__ void __launch_bounds__(SSD_LOSS_THREADS)
k_ssd_loss(const float* __restrict__ cls_pred, const float* __restrict__ loc_pred, const float* __restrict__ labels,
           const float* __restrict__ targets, int B, int N, int C, float sigma2, float w_loc,
           float* __restrict__ per_image, float* __restrict__ d_cls, float* __restrict__ d_loc) {
  __shared__ float sh[40];
  const int b = blockIdx.x, K = C + 1;
  const float* lab = labels + (size_t)b * N;
  float cnt = 0.f;
  for (int n = threadIdx.x; n < N; n += SSD_LOSS_THREADS) cnt += (lab[n] > 0.f) ? 1.f : 0.f;
  const float npos = ssd_block_sum(cnt, sh);
  const float inv = (npos > 0.f) ? 1.f / (npos * (float)B) : 0.f;   // d(mean_b final_b)/d(sum terms of image b)
  float ce_sum = 0.f, reg_sum = 0.f;
  for (int n = threadIdx.x; n < N; n += SSD_LOSS_THREADS) {
    const float l = lab[n];
    const float* s = cls_pred + ((size_t)b * N + n) * K;
    float* dc = d_cls ? d_cls + ((size_t)b * N + n) * K : nullptr;
    float4 dl = make_float4(0.f, 0.f, 0.f, 0.f);
    if (l >= 0.f) {
      const int li = (int)l;
      float m = s[0];
      for (int c = 1; c < K; ++c) m = fmaxf(m, s[c]);
      float se = 0.f;
      for (int c = 0; c < K; ++c) se += expf(s[c] - m);
      const float lse = m + logf(se);
      ce_sum += lse - s[li];
      if (dc)
        for (int c = 0; c < K; ++c) dc[c] = (expf(s[c] - lse) - (c == li ? 1.f : 0.f)) * inv;
      if (l > 0.f) {
        const float4 p = reinterpret_cast<const float4*>(loc_pred)[(size_t)b * N + n];
        const float4 t = reinterpret_cast<const float4*>(targets)[(size_t)b * N + n];
        const float dd[4] = {p.x - t.x, p.y - t.y, p.z - t.z, p.w - t.w};
        float g4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = fabsf(dd[e]);
          const bool small = a < 1.0f / sigma2;
          reg_sum += small ? 0.5f * sigma2 * (a * a) : a - 0.5f / sigma2;
          g4[e] = (small ? sigma2 * dd[e] : (dd[e] > 0.f ? 1.f : (dd[e] < 0.f ? -1.f : 0.f))) * w_loc * inv;
        }
        dl = make_float4(g4[0], g4[1], g4[2], g4[3]);
      }
    } else if (dc) {
      for (int c = 0; c < K; ++c) dc[c] = 0.f;
    }
    if (d_loc) reinterpret_cast<float4*>(d_loc)[(size_t)b * N + n] = dl;
  }
  const float ce = ssd_block_sum(ce_sum, sh);
  const float reg = ssd_block_sum(reg_sum, sh);
  if (threadIdx.x == 0) {
    per_image[b * 4 + 0] = ce;
    per_image[b * 4 + 1] = reg;
    per_image[b * 4 + 2] = npos;
    per_image[b * 4 + 3] = (npos > 0.f) ? (ce + reg * w_loc) / npos : 0.f;
  }
}

// losses[0] = mean_b final_b, [1] = mean_b cls_sum_b, [2] = mean_b bbox_sum_b
__global__ void k_ssd_loss_mean(const float* __restrict__ per_image, int B, float* __restrict__ losses) {
  if (threadIdx.x < 3) {
    const int col = threadIdx.x == 0 ? 3 : threadIdx.x - 1;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += per_image[b * 4 + col];
    losses[threadIdx.x] = s / (float)B;
  }
}

extern "C" int lmh_ssd_loss(const float* cls_pred, const float* loc_pred, const float* labels, const float* targets,
                            int B, int N, int C, float sigma, float w_loc, float* losses, float* per_image,
                            float* d_cls_pred, float* d_loc_pred, lmh_stream_t stream) {
  LMH_CHECK_ARG(cls_pred && loc_pred && labels && targets && losses && per_image && B > 0 && N > 0 && C > 0);
  hipStream_t st = (hipStream_t)stream;
  lmh_launch(k_ssd_loss, dim3(B), dim3(SSD_LOSS_THREADS), 0, st, cls_pred, loc_pred, labels, targets, B, N, C,
                     sigma * sigma, w_loc, per_image, d_cls_pred, d_loc_pred);
  lmh_launch(k_ssd_loss_mean, dim3(1), dim3(64), 0, st, per_image, B, losses);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
