// Anchor targets (RPNTarget) and proposal targets (RCNNTarget) for a batch.
//
// Reference: luminoth/models/fasterrcnn/rpn_target.py:73-335,
// rcnn_target.py:48-299, rcnn.py:156-167.  The N x G IoU matrix is never
// materialised: row max/argmax live in registers, column maxima in LDS/L2
// atomics, and the random fg/bg subsampling is an exact block-wide radix
// select on the shared counter hash (oracle/rng.py twin).
#include "lmh_common.h"

// ---------------------------------------------------------------------------
// Block-wide: find the k-th smallest (hash, index) among candidates `pred(i)`,
// i in [0,n).  All threads must call.  `hist` = 256 u32 in LDS, `eq_list` =
// LMH_SELECT_MAX_EQ u32, `bc` = 4 u32 broadcast slots.
// ---------------------------------------------------------------------------
template <typename Pred>
__device__ void block_select_kth(int n, uint32_t k, uint32_t seed, uint32_t stream, Pred pred,
                                 uint32_t* hist, uint32_t* eq_list, uint32_t* bc,
                                 lmh_select_state* out) {
  // Radix select, 8 bits per pass.  Round 4: the 256-bin scan is done by wave 0 (4 bins per lane + a shuffle scan; it was a
  // serial loop of thread 0: ~7 us per pass on the one block per image these kernels run as), and as soon as the selected
  // bin holds <= 64 candidates — after the first pass for the 2000 proposals of an image, after the second for 49 152
  // anchors — they are ranked directly by one wave instead of walking the remaining passes.
  uint32_t prefix = 0, mask = 0, remaining = k;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      if (pred(i)) {
        const uint32_t h = lmh_hash_u32(seed, stream, (uint32_t)i);
        if ((h & mask) == prefix) atomicAdd(&hist[(h >> shift) & 255u], 1u);
      }
    }
    __syncthreads();
    if (wave == 0) {
      const uint32_t c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
      const uint32_t mine = c0 + c1 + c2 + c3;
      uint32_t incl = mine;
      for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
      uint32_t cum = incl - mine;
      if (cum < remaining && incl >= remaining) {          // exactly one lane: the first bin whose running count reaches k
        uint32_t dsel, cnt;
        if (cum + c0 >= remaining) { dsel = 0; cnt = c0; }
        else if (cum + c0 + c1 >= remaining) { dsel = 1; cnt = c1; cum += c0; }
        else if (cum + c0 + c1 + c2 >= remaining) { dsel = 2; cnt = c2; cum += c0 + c1; }
        else { dsel = 3; cnt = c3; cum += c0 + c1 + c2; }
        bc[0] = 4 * lane + dsel;
        bc[1] = remaining - cum;
        bc[2] = cnt;
      } else if (lane == 63 && incl < remaining) {         // k beyond the candidates (callers do not ask): the last bin
        bc[0] = 255; bc[1] = remaining - incl; bc[2] = c3;
      }
    }
    __syncthreads();
    prefix |= bc[0] << shift;
    mask |= 255u << shift;
    remaining = bc[1];
    const uint32_t cnt = bc[2];
    __syncthreads();
    if (pass < 3 && cnt <= 64u) {
      // ---- direct finish: the <= 64 candidates of the selected bin as (hash, index) pairs in hist[0..127]
      if (threadIdx.x == 0) bc[3] = 0;
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if (pred(i)) {
          const uint32_t h = lmh_hash_u32(seed, stream, (uint32_t)i);
          if ((h & mask) == prefix) {
            const uint32_t slot = atomicAdd(&bc[3], 1u);
            if (slot < 64u) { hist[2 * slot] = h; hist[2 * slot + 1] = (uint32_t)i; }
          }
        }
      }
      __syncthreads();
      if (wave == 0) {
        const uint32_t c = min(bc[3], 64u);
        const bool have = (uint32_t)lane < c;
        const uint32_t hme = have ? hist[2 * lane] : 0xFFFFFFFFu, ime = have ? hist[2 * lane + 1] : 0xFFFFFFFFu;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < c; ++j) {
          const uint32_t hj = hist[2 * j], ij = hist[2 * j + 1];
          rank += (hj < hme || (hj == hme && ij < ime)) ? 1u : 0u;
        }
        const unsigned long long kb = __ballot(have && rank + 1u == remaining);
        const int src = kb ? (int)__ffsll((long long)kb) - 1 : 0;
        const uint32_t thr = __shfl(hme, src);
        const unsigned long long eqb = __ballot(have && hme == thr);
        const uint32_t n_eq = (uint32_t)__popcll(eqb), below = (uint32_t)__popcll(__ballot(have && hme < thr));
        const uint32_t need = remaining - below;
        if (n_eq != need && have && hme == thr)
          eq_list[__popcll(eqb & ((1ull << lane) - 1ull))] = ime;
        if (lane == 0) { bc[0] = thr; bc[1] = need; bc[2] = (n_eq != need) ? n_eq : 0xFFFFFFFFu; }
      }
      __syncthreads();
      out->thr_hash = bc[0];
      out->need_eq = bc[1];
      out->eq_count = bc[2];
      __syncthreads();
      return;
    }
  }
  const uint32_t n_eq = bc[2];
  // collect the (rare) equal-hash candidates so ties break on the lower index
  if (threadIdx.x == 0) bc[3] = 0;
  __syncthreads();
  if (n_eq != remaining) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      if (pred(i) && lmh_hash_u32(seed, stream, (uint32_t)i) == prefix) {
        const uint32_t slot = atomicAdd(&bc[3], 1u);
        if (slot < LMH_SELECT_MAX_EQ) eq_list[slot] = (uint32_t)i;
      }
    }
  }
  __syncthreads();
  out->thr_hash = prefix;
  out->need_eq = remaining;
  out->eq_count = (n_eq != remaining) ? min(bc[3], (uint32_t)LMH_SELECT_MAX_EQ) : 0xFFFFFFFFu;
  __syncthreads();
}

__device__ __forceinline__ bool select_keep(const lmh_select_state& st, const uint32_t* eq_list,
                                            uint32_t seed, uint32_t stream, uint32_t i) {
  const uint32_t h = lmh_hash_u32(seed, stream, i);
  if (h < st.thr_hash) return true;
  if (h > st.thr_hash) return false;
  if (st.eq_count == 0xFFFFFFFFu) return true;  // every equal-hash candidate is kept
  uint32_t rank = 0;
  for (uint32_t q = 0; q < st.eq_count; ++q) rank += (eq_list[q] < i) ? 1u : 0u;
  return rank < st.need_eq;
}

// ---------------------------------------------------------------------------
// RPN target, kernel A: per inside anchor row max / argmax; per-gt column max.
// ---------------------------------------------------------------------------
#define RT_MAX_G 128

__global__ void __launch_bounds__(256)
k_rpn_target_rowmax(lmh_rpn_target_desc d, int N, const int32_t* __restrict__ anchor_ref,
                    const float* __restrict__ gt, const int32_t* __restrict__ gt_count,
                    float* __restrict__ max_overlaps, int32_t* __restrict__ argmax,
                    uint32_t* __restrict__ gt_max_bits) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain beside MFMA kernels of other streams: win the issue arbitration
  __shared__ lmh_box sgt[RT_MAX_G];
  __shared__ float sarea[RT_MAX_G];
  __shared__ uint32_t scolmax[RT_MAX_G];
  const int b = blockIdx.y;
  const int G = min(gt_count[b], d.Gmax);
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  int32_t a4[4] = {0, 0, 0, 0};
  bool inside = false;
  if (n < N) {
    lmh_anchor(anchor_ref, n, d.A, d.feat_w, d.anchor_stride, a4);
    const int bd = d.allowed_border;
    inside = a4[0] >= -bd && a4[1] >= -bd && a4[2] < d.im_w + bd && a4[3] < d.im_h + bd;
  }
  const lmh_box a = {(float)a4[0], (float)a4[1], (float)a4[2], (float)a4[3]};
  const float aa = lmh_area_plus1(a);
  float best = inside ? -1.f : 0.f;
  int besti = inside ? 0 : -1;  // -1 == outside the image
  // the reference has no bound on the number of gt boxes (rpn_target.py:137): they pass through LDS RT_MAX_G at a time
  for (int g0 = 0; g0 < G; g0 += RT_MAX_G) {
    const int gn = min(RT_MAX_G, G - g0);
    if (g0) __syncthreads();
    for (int g = threadIdx.x; g < gn; g += blockDim.x) {
      const float* p = gt + ((size_t)b * d.Gmax + g0 + g) * 5;
      lmh_box bx = {p[0], p[1], p[2], p[3]};
      sgt[g] = bx;
      sarea[g] = lmh_area_plus1(bx);
      scolmax[g] = 0u;
    }
    __syncthreads();
    if (inside) {
      for (int g = 0; g < gn; ++g) {
        const float iou = lmh_iou_plus1(a, aa, sgt[g], sarea[g]);
        if (iou > best) { best = iou; besti = g0 + g; }  // first occurrence of the max (tf.argmax)
        const uint32_t bits = __float_as_uint(iou);  // iou >= 0: uint order == float order
        if (bits > scolmax[g]) atomicMax(&scolmax[g], bits);
      }
    }
    __syncthreads();
    for (int g = threadIdx.x; g < gn; g += blockDim.x)
      if (scolmax[g]) atomicMax(&gt_max_bits[(size_t)b * d.Gmax + g0 + g], scolmax[g]);
  }
  if (n < N) {
    if (G == 0) best = 0.f;
    max_overlaps[(size_t)b * N + n] = inside ? best : 0.f;
    argmax[(size_t)b * N + n] = besti;
  }
}

// kernel B: labels before subsampling (rpn_target.py:142-202)
__global__ void __launch_bounds__(256)
k_rpn_target_labels(lmh_rpn_target_desc d, int N, const int32_t* __restrict__ anchor_ref,
                    const float* __restrict__ gt, const int32_t* __restrict__ gt_count,
                    const float* __restrict__ max_overlaps, const int32_t* __restrict__ argmax,
                    const uint32_t* __restrict__ gt_max_bits, float* __restrict__ labels,
                    float* __restrict__ labels_pre) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain beside MFMA kernels of other streams: win the issue arbitration
  __shared__ lmh_box sgt[RT_MAX_G];
  __shared__ float sarea[RT_MAX_G];
  __shared__ float scolmax[RT_MAX_G];
  const int b = blockIdx.y;
  const int G = min(gt_count[b], d.Gmax);
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t row = (size_t)b * N + n;
  const bool live = n < N && argmax[row] >= 0;      // inside the image
  int32_t a4[4] = {0, 0, 0, 0};
  if (live) lmh_anchor(anchor_ref, n, d.A, d.feat_w, d.anchor_stride, a4);
  const lmh_box a = {(float)a4[0], (float)a4[1], (float)a4[2], (float)a4[3]};
  const float aa = lmh_area_plus1(a);
  bool is_gt_argmax = false;
  for (int g0 = 0; g0 < G; g0 += RT_MAX_G) {        // gt boxes pass through LDS RT_MAX_G at a time (no bound on G)
    const int gn = min(RT_MAX_G, G - g0);
    if (g0) __syncthreads();
    for (int g = threadIdx.x; g < gn; g += blockDim.x) {
      const float* p = gt + ((size_t)b * d.Gmax + g0 + g) * 5;
      lmh_box bx = {p[0], p[1], p[2], p[3]};
      sgt[g] = bx;
      sarea[g] = lmh_area_plus1(bx);
      scolmax[g] = __uint_as_float(gt_max_bits[(size_t)b * d.Gmax + g0 + g]);
    }
    __syncthreads();
    if (live)
      for (int g = 0; g < gn; ++g)
        is_gt_argmax |= (lmh_iou_plus1(a, aa, sgt[g], sarea[g]) == scolmax[g]);
  }
  if (n >= N) return;
  float label = -1.f;
  if (live) {
    const float mo = max_overlaps[row];
    const bool neg = mo < d.background_threshold_high;
    if (!d.clobber_positives && neg) label = 0.f;
    if (is_gt_argmax) label = 1.f;
    if (mo >= d.foreground_threshold) label = 1.f;
    if (d.clobber_positives && neg) label = 0.f;
  }
  labels[row] = label;
  if (labels_pre) labels_pre[row] = label;
}

// kernel C (fallback for N > RT_LDS_MAX_N): exact random subsampling of fg then bg, one block per image
// (rpn_target.py:203-284), then bbox targets (rpn_target.py:289-304).
#define RT_SUB_THREADS 1024
__global__ void __launch_bounds__(RT_SUB_THREADS)
k_rpn_target_subsample_global(lmh_rpn_target_desc d, int N, const int32_t* __restrict__ anchor_ref,
                       const float* __restrict__ gt, const int32_t* __restrict__ gt_count,
                       const uint32_t* __restrict__ seeds, const int32_t* __restrict__ argmax,
                       float* __restrict__ labels, float* __restrict__ bbox_targets) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t eq_list[LMH_SELECT_MAX_EQ];
  __shared__ uint32_t bc[4];
  __shared__ uint32_t s_cnt[2];
  const int b = blockIdx.x;
  float* lab = labels + (size_t)b * N;
  const uint32_t seed = seeds[b];
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  {
    uint32_t cf = 0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) cf += (lab[i] == 1.f);
    for (int o = 32; o > 0; o >>= 1) cf += __shfl_down(cf, o);
    if ((threadIdx.x & 63) == 0 && cf) atomicAdd(&s_cnt[0], cf);
  }
  __syncthreads();
  const uint32_t num_fg = (uint32_t)(int)(d.foreground_fraction * (float)d.minibatch_size);
  uint32_t n_fg = s_cnt[0];
  if (n_fg > num_fg) {
    if (num_fg == 0) {
      for (int i = threadIdx.x; i < N; i += blockDim.x) if (lab[i] == 1.f) lab[i] = -1.f;
    } else {
      lmh_select_state st;
      block_select_kth(N, num_fg, seed, LMH_STREAM_RPN_FG, [&](int i) { return lab[i] == 1.f; }, hist,
                       eq_list, bc, &st);
      for (int i = threadIdx.x; i < N; i += blockDim.x)
        if (lab[i] == 1.f && !select_keep(st, eq_list, seed, LMH_STREAM_RPN_FG, (uint32_t)i)) lab[i] = -1.f;
    }
    n_fg = num_fg;
  }
  __syncthreads();
  {
    uint32_t cb = 0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) cb += (lab[i] == 0.f);
    for (int o = 32; o > 0; o >>= 1) cb += __shfl_down(cb, o);
    if ((threadIdx.x & 63) == 0 && cb) atomicAdd(&s_cnt[1], cb);
  }
  __syncthreads();
  const int num_bg_i = d.minibatch_size - (int)n_fg;
  const uint32_t num_bg = num_bg_i > 0 ? (uint32_t)num_bg_i : 0u;
  if (s_cnt[1] > num_bg) {
    if (num_bg == 0) {
      for (int i = threadIdx.x; i < N; i += blockDim.x) if (lab[i] == 0.f) lab[i] = -1.f;
    } else {
      lmh_select_state st;
      block_select_kth(N, num_bg, seed, LMH_STREAM_RPN_BG, [&](int i) { return lab[i] == 0.f; }, hist,
                       eq_list, bc, &st);
      for (int i = threadIdx.x; i < N; i += blockDim.x)
        if (lab[i] == 0.f && !select_keep(st, eq_list, seed, LMH_STREAM_RPN_BG, (uint32_t)i)) lab[i] = -1.f;
    }
  }
  __syncthreads();
  // bbox targets: encode(anchor, gt[argmax]) where label == 1 else 0
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    if (lab[i] == 1.f) {
      int32_t a4[4];
      lmh_anchor(anchor_ref, i, d.A, d.feat_w, d.anchor_stride, a4);
      const lmh_box a = {(float)a4[0], (float)a4[1], (float)a4[2], (float)a4[3]};
      const float* p = gt + ((size_t)b * d.Gmax + argmax[(size_t)b * N + i]) * 5;
      const lmh_box g = {p[0], p[1], p[2], p[3]};
      lmh_encode(a, g, 1.f, 1.f, t);
    }
    reinterpret_cast<float4*>(bbox_targets)[(size_t)b * N + i] = make_float4(t[0], t[1], t[2], t[3]);
  }
}

// kernel C: exact random subsampling of fg then bg (rpn_target.py:203-284), then bbox targets
// (rpn_target.py:289-304), one block per image.  The labels are staged ONCE into LDS as int8 (48 KiB for
// the 49 152 anchors of a 1024^2 image): the ~12 scans of the radix selects then never touch global memory
// (the global-memory version above spent 0.2 ms on dependent scan latency, on the step's critical path).
#define RT_LDS_MAX_N 61440
__global__ void __launch_bounds__(RT_SUB_THREADS)
k_rpn_target_subsample(lmh_rpn_target_desc d, int N, const int32_t* __restrict__ anchor_ref,
                       const float* __restrict__ gt, const int32_t* __restrict__ gt_count,
                       const uint32_t* __restrict__ seeds, const int32_t* __restrict__ argmax,
                       float* __restrict__ labels, float* __restrict__ bbox_targets) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain beside MFMA kernels of other streams: win the issue arbitration
  __shared__ int8_t sl[RT_LDS_MAX_N];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t eq_list[LMH_SELECT_MAX_EQ];
  __shared__ uint32_t bc[4];
  __shared__ uint32_t s_cnt[2];
  const int b = blockIdx.x;
  float* lab = labels + (size_t)b * N;
  const uint32_t seed = seeds[b];
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  uint32_t cf = 0, cb = 0;
  for (int i0 = threadIdx.x; i0 < N; i0 += 8 * RT_SUB_THREADS) {      // eight label reads in flight per trip (clamped index)
    float l[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) l[u] = lab[min(i0 + u * RT_SUB_THREADS, N - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * RT_SUB_THREADS;
      if (i < N) {
        sl[i] = (int8_t)(int)l[u];
        cf += (l[u] == 1.f);
      }
    }
  }
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) cf += __shfl_down(cf, o);
  if ((threadIdx.x & 63) == 0 && cf) atomicAdd(&s_cnt[0], cf);
  __syncthreads();
  const uint32_t num_fg = (uint32_t)(int)(d.foreground_fraction * (float)d.minibatch_size);
  uint32_t n_fg = s_cnt[0];
  if (n_fg > num_fg) {
    if (num_fg == 0) {
      for (int i = threadIdx.x; i < N; i += RT_SUB_THREADS) if (sl[i] == 1) sl[i] = -1;
    } else {
      lmh_select_state st;
      block_select_kth(N, num_fg, seed, LMH_STREAM_RPN_FG, [&](int i) { return sl[i] == 1; }, hist, eq_list, bc, &st);
      for (int i = threadIdx.x; i < N; i += RT_SUB_THREADS)
        if (sl[i] == 1 && !select_keep(st, eq_list, seed, LMH_STREAM_RPN_FG, (uint32_t)i)) sl[i] = -1;
    }
    n_fg = num_fg;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += RT_SUB_THREADS) cb += (sl[i] == 0);
  for (int o = 32; o > 0; o >>= 1) cb += __shfl_down(cb, o);
  if ((threadIdx.x & 63) == 0 && cb) atomicAdd(&s_cnt[1], cb);
  __syncthreads();
  const int num_bg_i = d.minibatch_size - (int)n_fg;
  const uint32_t num_bg = num_bg_i > 0 ? (uint32_t)num_bg_i : 0u;
  if (s_cnt[1] > num_bg) {
    if (num_bg == 0) {
      for (int i = threadIdx.x; i < N; i += RT_SUB_THREADS) if (sl[i] == 0) sl[i] = -1;
    } else {
      lmh_select_state st;
      block_select_kth(N, num_bg, seed, LMH_STREAM_RPN_BG, [&](int i) { return sl[i] == 0; }, hist, eq_list, bc, &st);
      for (int i = threadIdx.x; i < N; i += RT_SUB_THREADS)
        if (sl[i] == 0 && !select_keep(st, eq_list, seed, LMH_STREAM_RPN_BG, (uint32_t)i)) sl[i] = -1;
    }
  }
  __syncthreads();
  // final labels + bbox targets: encode(anchor, gt[argmax]) where label == 1 else 0
  for (int i = threadIdx.x; i < N; i += RT_SUB_THREADS) {
    const int l = sl[i];
    lab[i] = (float)l;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    if (l == 1) {
      int32_t a4[4];
      lmh_anchor(anchor_ref, i, d.A, d.feat_w, d.anchor_stride, a4);
      const lmh_box a = {(float)a4[0], (float)a4[1], (float)a4[2], (float)a4[3]};
      const float* p = gt + ((size_t)b * d.Gmax + argmax[(size_t)b * N + i]) * 5;
      const lmh_box g = {p[0], p[1], p[2], p[3]};
      lmh_encode(a, g, 1.f, 1.f, t);
    }
    reinterpret_cast<float4*>(bbox_targets)[(size_t)b * N + i] = make_float4(t[0], t[1], t[2], t[3]);
  }
}

extern "C" size_t lmh_rpn_target_workspace_bytes(const lmh_rpn_target_desc* d) {
  if (!d) return 0;
  const size_t N = (size_t)d->feat_h * d->feat_w * d->A;
  return lmh_align_up((size_t)d->B * N * 4, 256) + lmh_align_up((size_t)d->B * d->Gmax * 4, 256);
}

extern "C" int lmh_rpn_target(const lmh_rpn_target_desc* d, const int32_t* anchor_ref, const float* gt,
                              const int32_t* gt_count, const uint32_t* seeds, float* labels,
                              float* bbox_targets, float* max_overlaps, float* labels_pre, void* ws,
                              size_t ws_bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(d && anchor_ref && gt && gt_count && seeds && labels && bbox_targets && max_overlaps && ws);
  LMH_CHECK_ARG(d->B > 0 && d->A > 0 && d->feat_h > 0 && d->feat_w > 0);
  LMH_CHECK_ARG(d->Gmax > 0);
  if (ws_bytes < lmh_rpn_target_workspace_bytes(d)) {
    lmh_set_error("lmh_rpn_target: workspace too small");
    return LMH_ERR_WORKSPACE;
  }
  const int N = d->feat_h * d->feat_w * d->A;
  hipStream_t st = (hipStream_t)stream;
  int32_t* argmax = reinterpret_cast<int32_t*>(ws);
  uint32_t* gt_max = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ws) +
                                                 lmh_align_up((size_t)d->B * N * 4, 256));
  LMH_CHECK_HIP(lmh_memset_async(gt_max, 0, (size_t)d->B * d->Gmax * 4, st));
  dim3 g((N + 255) / 256, d->B);
  lmh_launch(k_rpn_target_rowmax, g, dim3(256), 0, st, *d, N, anchor_ref, gt, gt_count,
                     max_overlaps, argmax, gt_max);
  lmh_launch(k_rpn_target_labels, g, dim3(256), 0, st, *d, N, anchor_ref, gt, gt_count,
                     max_overlaps, argmax, gt_max, labels, labels_pre);
  if (N <= RT_LDS_MAX_N)
    lmh_launch(k_rpn_target_subsample, dim3(d->B), dim3(RT_SUB_THREADS), 0, st, *d, N,
                     anchor_ref, gt, gt_count, seeds, argmax, labels, bbox_targets);
  else
    lmh_launch(k_rpn_target_subsample_global, dim3(d->B), dim3(RT_SUB_THREADS), 0, st, *d, N,
                     anchor_ref, gt, gt_count, seeds, argmax, labels, bbox_targets);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// ---------------------------------------------------------------------------
// RCNN target: one 1024-thread block per image.  Per-proposal state (label, best gt, "is the best proposal of gt g",
// fg condition) lives in LDS for P <= CT_LDS_MAX_P — every default configuration — and in the caller's workspace
// beyond; gt boxes pass through LDS RT_MAX_G at a time.  The reference bounds neither (rcnn_target.py:48-66).
// ---------------------------------------------------------------------------
#define CT_THREADS 1024
#define CT_LDS_MAX_P 4096
#define CT_MAX_G 32767   // best-gt indices are int16

struct ct_state {
  float* lab; int16_t* bestgt; int16_t* bestg_of_p; uint8_t* fg;
};

#define CT_A16(x) ((((size_t)(x)) + 15) / 16 * 16)
static __host__ __device__ inline size_t ct_ws_per_image(int P) {
  // [carry IoU: P f32][lab: P f32][bestgt: P i16][bestg_of_p: P i16][fg: P u8]
  return CT_A16((size_t)P * 4) + CT_A16((size_t)P * 4) + 2 * CT_A16((size_t)P * 2) + CT_A16(P);
}

template <bool WS_STATE>
__global__ void __launch_bounds__(CT_THREADS)
k_rcnn_target(lmh_rcnn_target_desc d, const float* __restrict__ proposals,
              const int32_t* __restrict__ prop_count, const float* __restrict__ gt,
              const int32_t* __restrict__ gt_count, const uint32_t* __restrict__ seeds,
              float* __restrict__ labels, float* __restrict__ bbox_targets,
              float* __restrict__ labels_pre, float* __restrict__ rois,
              float* __restrict__ roi_labels, float* __restrict__ roi_targets,
              int32_t* __restrict__ roi_count, unsigned char* __restrict__ ws) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain beside MFMA kernels of other streams: win the issue arbitration
  __shared__ lmh_box sgt[RT_MAX_G];
  __shared__ float sarea[RT_MAX_G];
  __shared__ unsigned long long sbest[RT_MAX_G];  // (iou_bits << 32) | ~p : max == best, first p
  __shared__ float l_lab[WS_STATE ? 1 : CT_LDS_MAX_P];
  __shared__ int16_t l_bestgt[WS_STATE ? 1 : CT_LDS_MAX_P];
  __shared__ int16_t l_bestg_of_p[WS_STATE ? 1 : CT_LDS_MAX_P];  // max g whose best proposal is p, or -1
  __shared__ uint8_t l_fg[WS_STATE ? 1 : CT_LDS_MAX_P];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t eq_list[LMH_SELECT_MAX_EQ];
  __shared__ uint32_t bc[4];
  __shared__ uint32_t s_cnt[3];
  __shared__ uint32_t s_scan[CT_THREADS / 64];
  const int b = blockIdx.x;
  const int P = min(prop_count[b], d.P);
  const int G = min(gt_count[b], d.Gmax);
  const uint32_t seed = seeds[b];
  const float4* props = reinterpret_cast<const float4*>(proposals) + (size_t)b * d.P;
  const float* gtb = gt + (size_t)b * d.Gmax * 5;
  unsigned char* wsb = ws + (size_t)b * ct_ws_per_image(d.P);
  float* carry = reinterpret_cast<float*>(wsb);     // running row max between gt chunks (only touched when G > RT_MAX_G)
  float* slab;
  int16_t* sbestgt;
  int16_t* sbestg_of_p;
  uint8_t* sfgcond;
  if (WS_STATE) {
    unsigned char* q = wsb + CT_A16((size_t)d.P * 4);
    slab = reinterpret_cast<float*>(q); q += CT_A16((size_t)d.P * 4);
    sbestgt = reinterpret_cast<int16_t*>(q); q += CT_A16((size_t)d.P * 2);
    sbestg_of_p = reinterpret_cast<int16_t*>(q); q += CT_A16((size_t)d.P * 2);
    sfgcond = q;
  } else {
    slab = l_lab; sbestgt = l_bestgt; sbestg_of_p = l_bestg_of_p; sfgcond = l_fg;
  }
  if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
  for (int p = threadIdx.x; p < d.P; p += blockDim.x) sbestg_of_p[p] = -1;
  // pass 1: per proposal row max/argmax + label by thresholds (rcnn_target.py:66-136), gt chunk by gt chunk
  for (int g0 = 0; g0 < max(G, 1); g0 += RT_MAX_G) {
    const int gn = max(0, min(RT_MAX_G, G - g0));
    const bool first = g0 == 0, last = g0 + RT_MAX_G >= G;
    __syncthreads();
    for (int g = threadIdx.x; g < gn; g += blockDim.x) {
      const float* p = gtb + (size_t)(g0 + g) * 5;
      lmh_box bx = {p[0], p[1], p[2], p[3]};
      sgt[g] = bx;
      sarea[g] = lmh_area_plus1(bx);
      sbest[g] = 0ull;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
      const float4 v = props[p];
      const lmh_box a = {v.x, v.y, v.z, v.w};
      const float aa = lmh_area_plus1(a);
      float best = first ? -1.f : carry[p];
      int bi = first ? 0 : (int)sbestgt[p];
      for (int g = 0; g < gn; ++g) {
        const float iou = lmh_iou_plus1(a, aa, sgt[g], sarea[g]);
        if (iou > best) { best = iou; bi = g0 + g; }
        const unsigned long long key =
            ((unsigned long long)__float_as_uint(iou) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)p);
        if (key > sbest[g]) atomicMax(&sbest[g], key);
      }
      sbestgt[p] = (int16_t)bi;
      if (!last) {
        carry[p] = best;
      } else {
        float label = -1.f;
        if (best >= d.background_threshold_low && best < d.background_threshold_high) label = 0.f;
        const bool is_fg = best >= d.foreground_threshold;
        if (is_fg) label = gtb[(size_t)bi * 5 + 4] + 1.f;
        slab[p] = label;
        sfgcond[p] = is_fg ? 1 : 0;
      }
    }
    __syncthreads();
    // best proposal per gt; duplicates: last gt wins (sparse_to_dense, rcnn_target.py:140-153)
    if (threadIdx.x == 0) {
      for (int g = 0; g < gn && P > 0; ++g) {
        const uint32_t p = 0xFFFFFFFFu - (uint32_t)(sbest[g] & 0xFFFFFFFFull);
        sbestg_of_p[p] = (int16_t)(g0 + g);  // ascending g: last write wins
      }
    }
  }
  __syncthreads();
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    const int g = sbestg_of_p[p];
    if (g >= 0) { slab[p] = gtb[(size_t)g * 5 + 4] + 1.f; sfgcond[p] = 1; }
    if (labels_pre) labels_pre[(size_t)b * d.P + p] = slab[p];
  }
  __syncthreads();
  // fg subsample (rcnn_target.py:159-200): disabled -> -label
  {
    uint32_t c = 0;
    for (int p = threadIdx.x; p < P; p += blockDim.x) c += sfgcond[p];
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt[0], c);
  }
  __syncthreads();
  const uint32_t max_fg = (uint32_t)(int)(d.foreground_fraction * (float)d.minibatch_size);
  if (s_cnt[0] > max_fg) {
    if (max_fg == 0) {
      for (int p = threadIdx.x; p < P; p += blockDim.x) if (sfgcond[p]) slab[p] = -slab[p];
    } else {
      lmh_select_state st;
      block_select_kth(P, max_fg, seed, LMH_STREAM_RCNN_FG, [&](int i) { return sfgcond[i] != 0; }, hist,
                       eq_list, bc, &st);
      for (int p = threadIdx.x; p < P; p += blockDim.x)
        if (sfgcond[p] && !select_keep(st, eq_list, seed, LMH_STREAM_RCNN_FG, (uint32_t)p)) slab[p] = -slab[p];
    }
  }
  __syncthreads();
  {
    uint32_t cf = 0, cb = 0;
    for (int p = threadIdx.x; p < P; p += blockDim.x) { cf += (slab[p] > 0.f); cb += (slab[p] == 0.f); }
    for (int o = 32; o > 0; o >>= 1) { cf += __shfl_down(cf, o); cb += __shfl_down(cb, o); }
    if ((threadIdx.x & 63) == 0) { if (cf) atomicAdd(&s_cnt[1], cf); if (cb) atomicAdd(&s_cnt[2], cb); }
  }
  __syncthreads();
  const int max_bg_i = d.minibatch_size - (int)s_cnt[1];
  const uint32_t max_bg = max_bg_i > 0 ? (uint32_t)max_bg_i : 0u;
  if (s_cnt[2] >= max_bg && s_cnt[2] > 0) {  // rcnn_target.py:246-250 (>=)
    if (max_bg == 0) {
      for (int p = threadIdx.x; p < P; p += blockDim.x) if (slab[p] == 0.f) slab[p] = -1.f;
    } else if (s_cnt[2] > max_bg) {
      lmh_select_state st;
      block_select_kth(P, max_bg, seed, LMH_STREAM_RCNN_BG, [&](int i) { return slab[i] == 0.f; }, hist,
                       eq_list, bc, &st);
      for (int p = threadIdx.x; p < P; p += blockDim.x)
        if (slab[p] == 0.f && !select_keep(st, eq_list, seed, LMH_STREAM_RCNN_BG, (uint32_t)p)) slab[p] = -1.f;
    }
  }
  __syncthreads();
  // outputs: full labels/targets + order-preserving compaction of label >= 0
  const int R = d.minibatch_size;
  const int per = (d.P + CT_THREADS - 1) / CT_THREADS;
  const int p0 = threadIdx.x * per;
  uint32_t mine = 0;
  for (int q = 0; q < per; ++q) { const int p = p0 + q; if (p < P && slab[p] >= 0.f) ++mine; }
  // block exclusive scan of `mine`
  uint32_t incl = mine;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
  if (lane == 63) s_scan[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int w = 0; w < CT_THREADS / 64; ++w) { const uint32_t t = s_scan[w]; s_scan[w] = run; run += t; }
    roi_count[b] = (int32_t)min(run, (uint32_t)R);
    bc[0] = run;
  }
  __syncthreads();
  uint32_t pos = s_scan[wave] + incl - mine;
  for (int q = 0; q < per; ++q) {
    const int p = p0 + q;
    if (p >= d.P) break;
    float label = -1.f;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < P) {
      label = slab[p];
      v = props[p];
      if (label > 0.f) {
        const lmh_box a = {v.x, v.y, v.z, v.w};
        const float* gp = gtb + (size_t)sbestgt[p] * 5;
        lmh_encode(a, lmh_box{gp[0], gp[1], gp[2], gp[3]}, d.variance_xy, d.variance_wh, t);
      }
    }
    labels[(size_t)b * d.P + p] = label;
    reinterpret_cast<float4*>(bbox_targets)[(size_t)b * d.P + p] = make_float4(t[0], t[1], t[2], t[3]);
    if (p < P && label >= 0.f) {
      if (pos < (uint32_t)R) {
        reinterpret_cast<float4*>(rois)[(size_t)b * R + pos] = v;
        roi_labels[(size_t)b * R + pos] = label;
        reinterpret_cast<float4*>(roi_targets)[(size_t)b * R + pos] = make_float4(t[0], t[1], t[2], t[3]);
      }
      ++pos;
    }
  }
  __syncthreads();
  const uint32_t total = min(bc[0], (uint32_t)R);
  for (int r = total + threadIdx.x; r < R; r += blockDim.x) {
    reinterpret_cast<float4*>(rois)[(size_t)b * R + r] = make_float4(0.f, 0.f, 0.f, 0.f);
    roi_labels[(size_t)b * R + r] = -1.f;
    reinterpret_cast<float4*>(roi_targets)[(size_t)b * R + r] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

extern "C" size_t lmh_rcnn_target_workspace_bytes(const lmh_rcnn_target_desc* d) {
  if (!d || d->B <= 0 || d->P <= 0) return 0;
  return lmh_align_up((size_t)d->B * ct_ws_per_image(d->P), 256);
}

extern "C" int lmh_rcnn_target(const lmh_rcnn_target_desc* d, const float* proposals,
                               const int32_t* prop_count, const float* gt, const int32_t* gt_count,
                               const uint32_t* seeds, float* labels, float* bbox_targets,
                               float* labels_pre, float* rois, float* roi_labels, float* roi_targets,
                               int32_t* roi_count, void* ws, size_t ws_bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(d && proposals && prop_count && gt && gt_count && seeds && labels && bbox_targets &&
                rois && roi_labels && roi_targets && roi_count && ws);
  LMH_CHECK_ARG(d->B > 0 && d->P > 0 && d->Gmax > 0 && d->Gmax <= CT_MAX_G);
  LMH_CHECK_ARG(d->minibatch_size > 0);
  if (ws_bytes < lmh_rcnn_target_workspace_bytes(d)) {
    lmh_set_error("lmh_rcnn_target: workspace %zu < %zu", ws_bytes, lmh_rcnn_target_workspace_bytes(d));
    return LMH_ERR_WORKSPACE;
  }
  if (d->P <= CT_LDS_MAX_P)
    lmh_launch(k_rcnn_target<false>, dim3(d->B), dim3(CT_THREADS), 0, (hipStream_t)stream, *d,
                       proposals, prop_count, gt, gt_count, seeds, labels, bbox_targets, labels_pre, rois,
                       roi_labels, roi_targets, roi_count, reinterpret_cast<unsigned char*>(ws));
  else
    lmh_launch(k_rcnn_target<true>, dim3(d->B), dim3(CT_THREADS), 0, (hipStream_t)stream, *d,
                       proposals, prop_count, gt, gt_count, seeds, labels, bbox_targets, labels_pre, rois,
                       roi_labels, roi_targets, roi_count, reinterpret_cast<unsigned char*>(ws));
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
