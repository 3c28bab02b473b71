// Streaming kernels of the half-STORAGE path (gfx950, HBM-bound; BASELINE configs[4]): casts at the fp32 / half boundaries
// of the trunk, the working copies of the weights, max pooling on half tensors.  `dtype`: 1 = f16, 2 = bf16 (the values of
// lmh_conv_desc.compute).  The convolutions themselves: conv_hs.h.
#include "lmh_common.h"

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DT> struct HS;
template <> struct HS<1> { typedef _Float16 T; typedef h16x8 V8; };
template <> struct HS<2> { typedef __bf16 T; typedef b16x8 V8; };

// fp32 -> 16-bit with round-to-nearest-even; f16 SATURATES at its largest finite value (5 exponent bits: an overflow to inf in
// a stored activation or scaled gradient would reach the fp32 master weights through the weight gradient — ADVICE r3);
// bf16 has fp32's exponent range and converts as is.
template <int DT> __device__ __forceinline__ typename HS<DT>::T hs_sat(float v) {
  return (typename HS<DT>::T)(DT == 1 ? fminf(fmaxf(v, -65504.f), 65504.f) : v);
}

// q(v * ks) for the backward weight copy: ONE rounding of the exact product for f16 (v_fma_mixlo_f16: fp32 sources, the
// fused result rounded straight to f16 — what the compiler picks for (T)(v * ks) in some contexts and not in others; tests
// pin the single rounding); bf16: the fp32 product, then round-to-nearest-even.
template <int DT> __device__ __forceinline__ typename HS<DT>::T hs_mul_round(float v, float ks);
template <> __device__ __forceinline__ _Float16 hs_mul_round<1>(float v, float ks) {
  uint32_t r = 0;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(r) : "v"(v), "v"(ks));
  return __builtin_bit_cast(_Float16, (uint16_t)r);
}
template <> __device__ __forceinline__ __bf16 hs_mul_round<2>(float v, float ks) { return (__bf16)(v * ks); }

#define HS_DISPATCH(dtype, CALL)                                  \
  do {                                                            \
    if ((dtype) == 1) { CALL(1); } else { CALL(2); }              \
  } while (0)

// ---- fp32 -> half:  y = q( x * mul * mask ), mask = activation bits [rows][C / 32] (NULL: none) --------------------
template <int DT>
__global__ void __launch_bounds__(256)
k_cast_to_half(const float* __restrict__ x, int64_t rows, int C, float mul, const uint32_t* __restrict__ bits,
               typename HS<DT>::T* __restrict__ y) {
  typedef typename HS<DT>::T T;
  typedef typename HS<DT>::V8 V8;
  const int C8 = C >> 3;
  const int64_t total = rows * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / C8;
    const int c8 = (int)(i - row * C8);
    const float* px = x + i * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(px), b = *reinterpret_cast<const f32x4*>(px + 4);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t m = 0xFFu;
    if (bits) m = bits[row * (C >> 5) + (c8 >> 2)] >> (8 * (c8 & 3));
    V8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = hs_sat<DT>(((m >> e) & 1u) ? v[e] * mul : 0.f);
    *reinterpret_cast<V8*>(y + i * 8) = h;
  }
}

template <int DT>
__global__ void __launch_bounds__(256)
k_cast_to_f32(const typename HS<DT>::T* __restrict__ x, int64_t n8, float mul, float* __restrict__ y) {
  typedef typename HS<DT>::V8 V8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const V8 h = *reinterpret_cast<const V8*>(x + i * 8);
    *reinterpret_cast<f32x4*>(y + i * 8) = f32x4{(float)h[0] * mul, (float)h[1] * mul, (float)h[2] * mul, (float)h[3] * mul};
    *reinterpret_cast<f32x4*>(y + i * 8 + 4) = f32x4{(float)h[4] * mul, (float)h[5] * mul, (float)h[6] * mul, (float)h[7] * mul};
  }
}

static inline int stream_blocks(int64_t total) { return (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384); }

extern "C" int lmh_cast_to_half(const float* x, int64_t rows, int C, float mul, const uint32_t* bits, void* y, int dtype,
                                lmh_stream_t stream) {
  LMH_CHECK_ARG(x && y && rows > 0 && C > 0 && (C & 7) == 0 && (dtype == 1 || dtype == 2));
  LMH_CHECK_ARG(bits == nullptr || (C & 31) == 0);
  const int64_t total = rows * (C >> 3);
#define CALL(DT_) lmh_launch(k_cast_to_half<DT_>, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, \
                                     rows, C, mul, bits, reinterpret_cast<HS<DT_>::T*>(y))
  HS_DISPATCH(dtype, CALL);
#undef CALL
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

extern "C" int lmh_cast_to_f32(const void* x, int64_t n, float mul, float* y, int dtype, lmh_stream_t stream) {
  LMH_CHECK_ARG(x && y && n > 0 && (n & 7) == 0 && (dtype == 1 || dtype == 2));
#define CALL(DT_) lmh_launch(k_cast_to_f32<DT_>, dim3(stream_blocks(n >> 3)), dim3(256), 0, (hipStream_t)stream, \
                                     reinterpret_cast<const HS<DT_>::T*>(x), n >> 3, mul, y)
  HS_DISPATCH(dtype, CALL);
#undef CALL
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// ---- working copies of the weights (conv_hs.h): w (RS, C, K) fp32 ->  w_fwd = q(w),  w_bwd = q(w * kscale[k])
// One launch for many layers: 64 x 64 tiles of the (RS*C) x K matrix through LDS.
//
// Layout (round 6, third session).  Both copies are the B operand of a gather-GEMM  out[p][n] = sum_q A[p][q] * B[n][q]:
//     forward    n = output channel k,  q = tap * C + c      (B[n][q] = q(w[tap][c][k]))
//     backward   n = input channel c,   q = tap * K + k      (B[n][q] = q(w[tap][c][k] * kscale[k]))
// When C % 64 == 0 and K % 64 == 0 (every layer conv_hs.h accepts) a copy is stored in the FRAGMENT ORDER of
// v_mfma_f32_32x32x16_{f16,bf16}:  [n / 32][q / 64 (stage)][(q % 64) / 16 (k-step)][lane = 32 * ((q % 16) / 8) + n % 32][q % 8]
// — the 16 bytes lane `lane` holds as the B operand of k-step (q % 64) / 16, so that one wave instruction moves one
// contiguous 1 KB whether it goes to LDS (global_load_lds, lane-linear) or straight into the operand registers.
// (Rounds 3-5 kept row-major copies, [K][RS*C] and [RS*C][K]: a B row of a stage was 128 contiguous bytes and always went
// through LDS.)  Other shapes keep the row-major copies (no kernel reads them).
#define HW_BATCH_MAX 48
struct half_weight_batch {
  lmh_half_weight_job job[HW_BATCH_MAX];
  int first_block[HW_BATCH_MAX + 1];
  int n;
};

template <int DT>
__global__ void __launch_bounds__(256)
k_half_weights(half_weight_batch b) {
  typedef typename HS<DT>::T T;
  typedef typename HS<DT>::V8 V8;
  __shared__ float tile[64][65];
  __shared__ float sks[64];
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.first_block[j + 1]) ++j;
  const lmh_half_weight_job jb = b.job[j];
  const int rows = jb.RS * jb.C, K = jb.K;
  const int tiles_k = (K + 63) / 64;
  const int t = blockIdx.x - b.first_block[j];
  const int r0 = (t / tiles_k) * 64, k0 = (t % tiles_k) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  T* wf = reinterpret_cast<T*>(jb.w_fwd);
  T* wb = reinterpret_cast<T*>(jb.w_bwd);
  const bool frag = (jb.C % 64) == 0 && (K % 64) == 0;
  const float ks = (jb.kscale && k0 + tx < K) ? jb.kscale[k0 + tx] : 1.f;
  if (ty == 0) sks[tx] = ks;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, k = k0 + tx;
    float v = 0.f;
    if (r < rows && k < K) {
      v = jb.w[(size_t)r * K + k];
      if (wb && !frag) wb[(size_t)r * K + k] = hs_mul_round<DT>(v, ks);     // weights: no saturation
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  if (!frag) {
    if (wf) {
      for (int i = ty; i < 64; i += 4) {
        const int k = k0 + i, r = r0 + tx;
        if (k < K && r < rows) wf[(size_t)k * rows + r] = (T)tile[tx][i];
      }
    }
    return;
  }
  // fragment order: the tile is one stage of the forward copy (q = r0 .. r0 + 63) for two 32-column groups (k0 / 32 + {0, 1}),
  // and one stage of the backward copy (tap r0 / C, q = tap * K + k0 ..) for the two groups c0 / 32 + {0, 1}
  const int tap = r0 / jb.C, c0 = r0 - tap * jb.C;
  const int KTf = rows >> 6, KTb = (jb.RS * K) >> 6;
  const int tf = r0 >> 6, tb = tap * (K >> 6) + (k0 >> 6);
#pragma unroll
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int ntl = e >> 8, s = (e >> 6) & 3, lane = e & 63;
    const int hi = lane >> 5, l31 = lane & 31, q0 = 16 * s + 8 * hi;
    if (wf) {
      V8 h;
#pragma unroll
      for (int i = 0; i < 8; ++i) h[i] = (T)tile[q0 + i][32 * ntl + l31];
      *reinterpret_cast<V8*>(wf + ((((size_t)((k0 >> 5) + ntl) * KTf + tf) * 4 + s) * 64 + lane) * 8) = h;
    }
    if (wb) {
      V8 h;
#pragma unroll
      for (int i = 0; i < 8; ++i) h[i] = hs_mul_round<DT>(tile[32 * ntl + l31][q0 + i], sks[q0 + i]);
      *reinterpret_cast<V8*>(wb + ((((size_t)((c0 >> 5) + ntl) * KTb + tb) * 4 + s) * 64 + lane) * 8) = h;
    }
  }
}

extern "C" int lmh_half_weights_batch(const lmh_half_weight_job* jobs, int n, int dtype, lmh_stream_t stream) {
  LMH_CHECK_ARG(jobs && n > 0 && (dtype == 1 || dtype == 2));
  for (int s = 0; s < n; s += HW_BATCH_MAX) {
    half_weight_batch b;
    b.n = n - s < HW_BATCH_MAX ? n - s : HW_BATCH_MAX;
    int blocks = 0;
    for (int i = 0; i < b.n; ++i) {
      const lmh_half_weight_job& jb = jobs[s + i];
      LMH_CHECK_ARG(jb.w && (jb.w_fwd || jb.w_bwd) && jb.RS > 0 && jb.C > 0 && jb.K > 0);
      b.job[i] = jb;
      b.first_block[i] = blocks;
      blocks += ((jb.RS * jb.C + 63) / 64) * ((jb.K + 63) / 64);
    }
    b.first_block[b.n] = blocks;
#define CALL(DT_) lmh_launch(k_half_weights<DT_>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b)
    HS_DISPATCH(dtype, CALL);
#undef CALL
  }
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// ---- max pool, NHWC, half output (input fp32 or half): max then round == round then max (rounding is monotone) -----------
template <int DT, bool IN_F32>
__global__ void __launch_bounds__(256)
k_maxpool_fwd_hs(const void* __restrict__ xv, int N, int H, int W, int C, int ks, int stride, int pt, int pl, int OH, int OW,
                 typename HS<DT>::T* __restrict__ y) {
  typedef typename HS<DT>::T T;
  typedef typename HS<DT>::V8 V8;
  const int C8 = C >> 3;
  const int64_t total = (int64_t)N * OH * OW * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    int64_t t = i / C8;
    const int ow = (int)(t % OW); t /= OW;
    const int oh = (int)(t % OH);
    const int n = (int)(t / OH);
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    if (ks == 3) {
      // the nine taps requested together (clamped address; a tap outside the image is skipped below) instead of load ->
      // wait -> max one tap at a time: nine serial round trips of memory latency per output (round 4).  max is exact.
      f32x4 fa[IN_F32 ? 9 : 1], fb[IN_F32 ? 9 : 1];
      V8 hv[IN_F32 ? 1 : 9];
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int ih = min(max(oh * stride - pt + q / 3, 0), H - 1), iw = min(max(ow * stride - pl + q % 3, 0), W - 1);
        const size_t o = ((size_t)(n * H + ih) * W + iw) * C + 8 * c8;
        if (IN_F32) {
          const float* px = reinterpret_cast<const float*>(xv) + o;
          fa[q] = *reinterpret_cast<const f32x4*>(px);
          fb[q] = *reinterpret_cast<const f32x4*>(px + 4);
        } else {
          hv[q] = *reinterpret_cast<const V8*>(reinterpret_cast<const T*>(xv) + o);
        }
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int ih = oh * stride - pt + q / 3, iw = ow * stride - pl + q % 3;
        if ((unsigned)ih >= (unsigned)H || (unsigned)iw >= (unsigned)W) continue;
        if (IN_F32) {
          m[0] = fmaxf(m[0], fa[q].x); m[1] = fmaxf(m[1], fa[q].y); m[2] = fmaxf(m[2], fa[q].z); m[3] = fmaxf(m[3], fa[q].w);
          m[4] = fmaxf(m[4], fb[q].x); m[5] = fmaxf(m[5], fb[q].y); m[6] = fmaxf(m[6], fb[q].z); m[7] = fmaxf(m[7], fb[q].w);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (float)hv[q][e]);
        }
      }
    } else
    for (int r = 0; r < ks; ++r) {
      const int ih = oh * stride - pt + r;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int s = 0; s < ks; ++s) {
        const int iw = ow * stride - pl + s;
        if ((unsigned)iw >= (unsigned)W) continue;
        const size_t o = ((size_t)(n * H + ih) * W + iw) * C + 8 * c8;
        if (IN_F32) {
          const float* px = reinterpret_cast<const float*>(xv) + o;
          const f32x4 a = *reinterpret_cast<const f32x4*>(px), b = *reinterpret_cast<const f32x4*>(px + 4);
          m[0] = fmaxf(m[0], a.x); m[1] = fmaxf(m[1], a.y); m[2] = fmaxf(m[2], a.z); m[3] = fmaxf(m[3], a.w);
          m[4] = fmaxf(m[4], b.x); m[5] = fmaxf(m[5], b.y); m[6] = fmaxf(m[6], b.z); m[7] = fmaxf(m[7], b.w);
        } else {
          const V8 h = *reinterpret_cast<const V8*>(reinterpret_cast<const T*>(xv) + o);
#pragma unroll
          for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (float)h[e]);
        }
      }
    }
    V8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = hs_sat<DT>(m[e]);
    *reinterpret_cast<V8*>(y + i * 8) = h;
  }
}

extern "C" int lmh_maxpool_fwd_hs(const void* x, int x_is_f32, int N, int H, int W, int C, int ksize, int stride, int pad_top,
                                  int pad_left, int OH, int OW, void* y, int dtype, lmh_stream_t stream) {
  LMH_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0 && ksize > 0 && stride > 0);
  LMH_CHECK_ARG(dtype == 1 || dtype == 2);
  const int64_t total = (int64_t)N * OH * OW * (C / 8);
#define CALL(DT_)                                                                                                        \
  do {                                                                                                                   \
    if (x_is_f32) lmh_launch((k_maxpool_fwd_hs<DT_, true>), dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream, \
                                     x, N, H, W, C, ksize, stride, pad_top, pad_left, OH, OW, reinterpret_cast<HS<DT_>::T*>(y)); \
    else lmh_launch((k_maxpool_fwd_hs<DT_, false>), dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, N, \
                            H, W, C, ksize, stride, pad_top, pad_left, OH, OW, reinterpret_cast<HS<DT_>::T*>(y));          \
  } while (0)
  HS_DISPATCH(dtype, CALL);
#undef CALL
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// resnet_utils.subsample (1x1 max pool, stride s) backward: dx[n, oh*s, ow*s, :] = dy[n, oh, ow, :], zero elsewhere.
// Written as a gather over dx (every element exactly once: no zero fill, no atomics); 16-bit elements moved as raw bits.
__global__ void __launch_bounds__(256)
k_subsample_bwd_hs(const uint4* __restrict__ dy, int N, int H, int W, int C8, int stride, int OH, int OW, uint4* __restrict__ dx) {
  const int64_t total = (int64_t)N * H * W * C8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    int64_t t = i / C8;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    const int oh = h / stride, ow = w / stride;
    if (oh * stride == h && ow * stride == w && oh < OH && ow < OW) v = dy[((size_t)(n * OH + oh) * OW + ow) * C8 + c8];
    dx[i] = v;
  }
}

extern "C" int lmh_subsample_bwd_hs(const void* dy, int N, int H, int W, int C, int stride, int OH, int OW, void* dx,
                                    lmh_stream_t stream) {
  LMH_CHECK_ARG(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0 && stride > 0 && OH > 0 && OW > 0);
  const int64_t total = (int64_t)N * H * W * (C / 8);
  lmh_launch(k_subsample_bwd_hs, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint4*>(dy), N, H, W, C / 8, stride, OH, OW, reinterpret_cast<uint4*>(dx));
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
