// Shared host/device helpers for libluminoth_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/luminoth_hip.h"

#define LMH_WAVE 64

void lmh_set_error(const char* fmt, ...);
// shared building blocks (proposals.hip)
int lmh_sort_u64_impl(uint64_t* keys, int B, int n_pad, hipStream_t st);
int lmh_nms_impl(const float* boxes, const int32_t* counts, int B, int K, float thr, int max_out,
                 int32_t* keep_idx, int32_t* keep_count, void* ws, hipStream_t st);

// Deferred weight-gradient tails (tail.hip): while lmh_tail_defer(1) is in effect on the calling thread,
// lmh_conv2d_bwd_weight / lmh_act_bwd launch their main kernel only, leave the split-K slabs / column-sum partial rows
// where they are and record them here; lmh_wgrad_tail_batch later finishes many layers in two launches.
struct lmh_tail_plan {
  const float* slabs;    // [splits][n] partial gradients, or NULL when dw already holds the raw gradient
  int splits;
  const float* colpart;  // [colrows][K] partial column sums, or NULL
  int colrows;
};
extern thread_local int g_lmh_defer_tail;
extern thread_local lmh_tail_plan g_lmh_last_plan;

#define LMH_CHECK_ARG(cond)                                                 \
  do {                                                                      \
    if (!(cond)) {                                                          \
      lmh_set_error("%s:%d invalid argument: %s", __FILE__, __LINE__, #cond); \
      return LMH_ERR_INVALID;                                               \
    }                                                                       \
  } while (0)

#define LMH_CHECK_HIP(expr)                                                          \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess) {                                                          \
      lmh_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      return LMH_ERR_LAUNCH;                                                         \
    }                                                                                \
  } while (0)

#define LMH_CHECK_LAUNCH() LMH_CHECK_HIP(hipGetLastError())

// ---- launch plans (plan.hip): every kernel launch, memset and stream-to-stream wait of this library goes through the
// three functions below.  Normally they just issue the HIP call.  While the calling thread RECORDS a plan
// (lmh_plan_begin ... lmh_plan_end) they also append the launch — function, geometry, stream and a private copy of the
// argument values — to the plan, so that lmh_plan_run can re-issue the identical sequence later without going through
// the host code above it (one C call per train step instead of ~250 Python -> ctypes round trips).
void lmh_launch_raw(const void* fn, dim3 grid, dim3 block, unsigned shmem, hipStream_t st, void** args,
                    const size_t* sizes, const size_t* aligns, int nargs);
hipError_t lmh_memset_async(void* ptr, int value, size_t bytes, hipStream_t st);
hipError_t lmh_memcpy_d2d_async(void* dst, const void* src, size_t bytes, hipStream_t st);

#ifdef __HIPCC__
#include <tuple>
#include <utility>
template <typename Tuple, size_t... I>
inline void lmh_launch_tuple_(const void* fn, dim3 grid, dim3 block, unsigned shmem, hipStream_t st, Tuple& vals,
                              std::index_sequence<I...>) {
  void* ptrs[sizeof...(I) + 1] = {(void*)&std::get<I>(vals)..., nullptr};
  const size_t sizes[sizeof...(I) + 1] = {sizeof(std::tuple_element_t<I, Tuple>)..., 0};
  const size_t aligns[sizeof...(I) + 1] = {alignof(std::tuple_element_t<I, Tuple>)..., 0};
  lmh_launch_raw(fn, grid, block, shmem, st, ptrs, sizes, aligns, (int)sizeof...(I));
}
// Drop-in for lmh_launch(kernel, grid, block, shmem, stream, args...): arguments are converted to the kernel's
// parameter types exactly like a call would.
template <typename... KArgs, typename... Args>
inline void lmh_launch(void (*kern)(KArgs...), dim3 grid, dim3 block, unsigned shmem, hipStream_t st, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
  std::tuple<KArgs...> vals{static_cast<KArgs>(std::forward<Args>(args))...};
  lmh_launch_tuple_(reinterpret_cast<const void*>(kern), grid, block, shmem, st, vals,
                    std::index_sequence_for<KArgs...>{});
}
#endif

static inline size_t lmh_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int lmh_next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// ---- shared counter-based hash (bit-identical twin: oracle/rng.py) ----------
__host__ __device__ inline uint32_t lmh_fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__host__ __device__ inline uint32_t lmh_hash_u32(uint32_t seed, uint32_t stream, uint32_t idx) {
  uint32_t h = seed ^ (idx * 0x9E3779B1u);
  h = lmh_fmix32(h);
  h ^= stream * 0x85EBCA77u;
  return lmh_fmix32(h);
}
enum { LMH_STREAM_RPN_FG = 0, LMH_STREAM_RPN_BG = 1, LMH_STREAM_RCNN_FG = 2, LMH_STREAM_RCNN_BG = 3,
       LMH_STREAM_SSD = 4, LMH_STREAM_DROPOUT = 5 };

#ifdef __HIPCC__
// ---- box arithmetic, op-for-op the reference's fp32 graph (compiled with
// -ffp-contract=off so nothing fuses into FMAs) ------------------------------
struct lmh_box { float x1, y1, x2, y2; };

// luminoth/utils/bbox_overlap.py:7-48 (+1 convention)
__device__ __forceinline__ float lmh_iou_plus1(const lmh_box& a, float a_area, const lmh_box& b,
                                               float b_area) {
  float xI1 = fmaxf(a.x1, b.x1), yI1 = fmaxf(a.y1, b.y1);
  float xI2 = fminf(a.x2, b.x2), yI2 = fminf(a.y2, b.y2);
  float inter = fmaxf(xI2 - xI1 + 1.f, 0.f) * fmaxf(yI2 - yI1 + 1.f, 0.f);
  float uni = (a_area + b_area) - inter;
  return fmaxf(inter / uni, 0.f);
}
__device__ __forceinline__ float lmh_area_plus1(const lmh_box& a) {
  return (a.x2 - a.x1 + 1.f) * (a.y2 - a.y1 + 1.f);
}

// luminoth/utils/bbox_transform_tf.py:41-66
__device__ __forceinline__ lmh_box lmh_decode(const lmh_box& roi, float dx, float dy, float dw,
                                              float dh, float v0, float v1) {
  float w = roi.x2 - roi.x1 + 1.f, h = roi.y2 - roi.y1 + 1.f;
  float cx = roi.x1 + .5f * w, cy = roi.y1 + .5f * h;
  float px = dx * w * v0 + cx, py = dy * h * v0 + cy;
  float pw = expf(dw * v1) * w, ph = expf(dh * v1) * h;
  lmh_box o;
  o.x1 = px - .5f * pw;
  o.y1 = py - .5f * ph;
  o.x2 = px + .5f * pw - 1.f;
  o.y2 = py + .5f * ph - 1.f;
  return o;
}
// luminoth/utils/bbox_transform_tf.py:18-38
__device__ __forceinline__ void lmh_encode(const lmh_box& b, const lmh_box& g, float v0, float v1,
                                           float* out4) {
  float w = b.x2 - b.x1 + 1.f, h = b.y2 - b.y1 + 1.f;
  float cx = b.x1 + .5f * w, cy = b.y1 + .5f * h;
  float gw = g.x2 - g.x1 + 1.f, gh = g.y2 - g.y1 + 1.f;
  float gcx = g.x1 + .5f * gw, gcy = g.y1 + .5f * gh;
  out4[0] = (gcx - cx) / (w * v0);
  out4[1] = (gcy - cy) / (h * v0);
  out4[2] = logf(gw / w) / v1;
  out4[3] = logf(gh / h) / v1;
}
// luminoth/utils/bbox_transform_tf.py:69-99
__device__ __forceinline__ lmh_box lmh_clip(const lmh_box& b, float im_h, float im_w) {
  lmh_box o;
  o.x1 = fmaxf(fminf(b.x1, im_w - 1.f), 0.f);
  o.x2 = fmaxf(fminf(b.x2, im_w - 1.f), 0.f);
  o.y1 = fmaxf(fminf(b.y1, im_h - 1.f), 0.f);
  o.y2 = fmaxf(fminf(b.y2, im_h - 1.f), 0.f);
  return o;
}
// anchor n of the grid: fasterrcnn.py:261-308 (int32: truncated reference + shift)
__device__ __forceinline__ void lmh_anchor(const int32_t* ref, int n, int A, int feat_w, int stride,
                                           int32_t* out4) {
  int a = n % A, cell = n / A;
  int x = cell % feat_w, y = cell / feat_w;
  int sx = x * stride, sy = y * stride;
  out4[0] = ref[a * 4 + 0] + sx;
  out4[1] = ref[a * 4 + 1] + sy;
  out4[2] = ref[a * 4 + 2] + sx;
  out4[3] = ref[a * 4 + 3] + sy;
}

__device__ __forceinline__ uint32_t lmh_float_orderable(float f) {
  uint32_t b = __float_as_uint(f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);  // ascending uint == ascending float
}

// Block-wide exact selection of the k smallest (hash, index) keys among the
// candidates of one stream (twin of oracle/rng.py: keep_k_smallest).
// Returns through *thr_hash / *n_less / eq list the data needed to decide
// membership:  keep(i) <=> hash_i < thr || (hash_i == thr && rank_among_equal(i) < need_eq)
struct lmh_select_state {
  uint32_t thr_hash;   // k-th smallest hash value
  uint32_t need_eq;    // how many of the hash==thr candidates (lowest index first) are kept
  uint32_t eq_count;   // number of candidates with hash == thr (<= LMH_SELECT_MAX_EQ)
};
#define LMH_SELECT_MAX_EQ 64
#endif  // __HIPCC__
