// Mixed-precision implicit-GEMM convolutions (gfx950): fp32 tensors in HBM, f16 / bf16 operands in LDS,
// v_mfma_f32_32x32x16_{f16,bf16} with fp32 accumulation, fp32 epilogue.  BASELINE configs[4] ("fp16 MFMA path").
//
// What changes against conv_fast.h (v_mfma_f32_32x32x2_f32, 1/16 of this MFMA rate):
//   * operands are rounded to half precision (RNE, v_cvt_pk_*) on their way from the staging registers into LDS, so
//     nothing outside these three kernels changes: activations, gradients and master weights stay fp32 in HBM;
//   * BOTH operands sit in LDS K-contiguous as [row][BK + 8] halfs (80-byte rows: 16-byte aligned and conflict-free
//     for the 4 x 16-lane groups of ds_read_b128), because a 32x32x16 fragment is 8 consecutive k per lane = one
//     ds_read_b128.  Operands whose reduction axis is the SLOW axis in memory (HWIO weights in the forward pass,
//     both pixel-major operands of the weight gradient) are transposed in registers: a thread loads a 4(k) x 4(col)
//     block as four float4 rows and writes four 8-byte k-quads, lanes ordered k-quad fastest so that a 16-lane
//     group of ds_write_b64 covers all 32 banks;
//   * backward operands are multiplied by a power-of-two `gscale` before rounding and the accumulators by 1/gscale
//     in the epilogue (static loss scaling inside the kernel: f16 has 5 exponent bits; exact in fp32);
//   * with the matrix pipe 16x faster the kernels are bound by the staging path (L2 -> VGPR -> cvt -> LDS), so the
//     pipeline is the plain one: global loads of tile t+1 before the MFMAs of tile t, converted and written to the
//     other LDS buffer after them, one barrier per stage.
//
// PF = 3 ("warp-specialised", 512 threads): see HALF_PIPELINE.
// DT = 3, "bf16x3": fp32 ARITHMETIC on the bf16 matrix pipe.  Every fp32 operand is split EXACTLY into three bf16 pieces by
// truncation (a = a0 + a1 + a2, 8 significant bits each: 24 = the fp32 significand), the pieces sit in three LDS planes,
// and a product is six MFMAs into the same fp32 accumulator: a0b2, a2b0, a1b1, a0b1, a1b0, a0b0 (the three dropped
// cross terms are <= 2^-23 of the product — below the rounding of the fp32 accumulation itself; every kept product of
// two 8-bit pieces is exact).  v_mfma_f32_32x32x16_bf16 is 16x the rate of v_mfma_f32_32x32x2_f32, so six of them are
// 2.67x the fp32-MFMA roofline at fp32 accuracy (the "6-pass bf16" scheme XLA uses for fp32 matmuls on TPUs).  The
// split costs ~5.5 VALU ops per staged element; one LDS buffer (61 KB at 128x128) so two blocks share a CU and one
// block's split / staging runs under the other's MFMAs.
#pragma once
#include <type_traits>
#include "conv_fast.h"

#define LDH (BK + 8)   // halfs per LDS row

template <int DT> struct HT;
template <> struct HT<1> {
  typedef _Float16 T;
  typedef _Float16 V4 __attribute__((ext_vector_type(4)));
  typedef _Float16 V8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct HT<2> {
  typedef __bf16 T;
  typedef __bf16 V4 __attribute__((ext_vector_type(4)));
  typedef __bf16 V8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

template <> struct HT<3> : HT<2> {};      // bf16 pieces of the exact 3-way split
template <int DT> struct half_cfg { static constexpr int NS = (DT == 3) ? 3 : 1; };

// exact split of an fp32 value into three bf16 pieces by truncation; returned as fp32 bit patterns whose HIGH halves are
// the pieces (the low half of the last one is zero by construction)
__device__ __forceinline__ void split3(float a, uint32_t& h0, uint32_t& h1, uint32_t& h2) {
  h0 = __float_as_uint(a) & 0xFFFF0000u;
  const float r1 = a - __uint_as_float(h0);
  h1 = __float_as_uint(r1) & 0xFFFF0000u;
  h2 = __float_as_uint(r1 - __uint_as_float(h1));
}
// (hi16(e1) << 16) | hi16(e0)
__device__ __forceinline__ uint32_t pack_hi(uint32_t e0, uint32_t e1) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }

// four fp32 values -> three planes of four bf16 at S (plane stride `pl` elements)
__device__ __forceinline__ void st_split4(__bf16* __restrict__ S, int pl, float a, float b, float c, float d) {
  uint32_t x0[4], x1[4], x2[4];
  split3(a, x0[0], x1[0], x2[0]);
  split3(b, x0[1], x1[1], x2[1]);
  split3(c, x0[2], x1[2], x2[2]);
  split3(d, x0[3], x1[3], x2[3]);
  *reinterpret_cast<uint2*>(S) = make_uint2(pack_hi(x0[0], x0[1]), pack_hi(x0[2], x0[3]));
  *reinterpret_cast<uint2*>(S + pl) = make_uint2(pack_hi(x1[0], x1[1]), pack_hi(x1[2], x1[3]));
  *reinterpret_cast<uint2*>(S + 2 * pl) = make_uint2(pack_hi(x2[0], x2[1]), pack_hi(x2[2], x2[3]));
}

template <int DT>
__device__ __forceinline__ typename HT<DT>::V4 cvt4(float a, float b, float c, float d) {
  typedef typename HT<DT>::T T;
  typename HT<DT>::V4 r;
  r[0] = (T)a; r[1] = (T)b; r[2] = (T)c; r[3] = (T)d;
  return r;
}

// K-contiguous source row -> LDS row segment (4 halfs at k = 4*kq)
template <int DT>
__device__ __forceinline__ void st_kc(typename HT<DT>::T* __restrict__ S, int pl, int row, int kq, f32x4 v) {
  if constexpr (DT == 3) st_split4(&S[row * LDH + 4 * kq], pl, v.x, v.y, v.z, v.w);
  else *reinterpret_cast<typename HT<DT>::V4*>(&S[row * LDH + 4 * kq]) = cvt4<DT>(v.x, v.y, v.z, v.w);
}
// 4(k) x 4(col) register block of a K-major source -> four LDS rows (cols), k-quad kq4
template <int DT>
__device__ __forceinline__ void st_km(typename HT<DT>::T* __restrict__ S, int pl, int col0, int kq4, const f32x4 (&r)[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if constexpr (DT == 3) st_split4(&S[(col0 + e) * LDH + 4 * kq4], pl, r[0][e], r[1][e], r[2][e], r[3][e]);
    else *reinterpret_cast<typename HT<DT>::V4*>(&S[(col0 + e) * LDH + 4 * kq4]) = cvt4<DT>(r[0][e], r[1][e], r[2][e], r[3][e]);
  }
}

// one BK = 32 stage: 2 k-steps of 16, fragments by ds_read_b128
template <int DT, int TM, int TN>
__device__ __forceinline__ void mfma_stage_h(const typename HT<DT>::T* __restrict__ As,
                                             const typename HT<DT>::T* __restrict__ Bs, f32x16 (&acc)[TM][TN],
                                             int a_off, int b_off, int lane, int a_pl = 0, int b_pl = 0) {
  typedef typename HT<DT>::V8 V8;
  const int l31 = lane & 31, kh = 8 * (lane >> 5);
  if constexpr (DT == 3) {
    // six products per fragment pair, smallest terms first; consecutive MFMAs go to different accumulators
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      V8 a[3][TM], b[3][TN];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int t = 0; t < TM; ++t)
          a[p][t] = *reinterpret_cast<const V8*>(&As[p * a_pl + (a_off + t * 32 + l31) * LDH + s * 16 + kh]);
#pragma unroll
        for (int t = 0; t < TN; ++t)
          b[p][t] = *reinterpret_cast<const V8*>(&Bs[p * b_pl + (b_off + t * 32 + l31) * LDH + s * 16 + kh]);
      }
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = HT<DT>::mfma(a[PA[q]][tm], b[PB[q]][tn], acc[tm][tn]);
    }
    return;
  }
  V8 a[2][TM], b[2][TN];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int t = 0; t < TM; ++t) a[s][t] = *reinterpret_cast<const V8*>(&As[(a_off + t * 32 + l31) * LDH + s * 16 + kh]);
#pragma unroll
    for (int t = 0; t < TN; ++t) b[s][t] = *reinterpret_cast<const V8*>(&Bs[(b_off + t * 32 + l31) * LDH + s * 16 + kh]);
  }
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = HT<DT>::mfma(a[s][tm], b[s][tn], acc[tm][tn]);
}

// Software pipeline shared by the three kernels.  The caller defines
//   load(S)        global -> staging register set S (std::integral_constant<int, 0|1>) at the current pointers
//   store(buf, S)  register set S -> LDS buffer `buf` (convert / transpose)
//   step()         advance the pointers to the next tile (no-op past the last tile: loads then re-read a valid tile)
//   mma(buf)       MFMAs of one stage out of LDS buffer `buf`
// PF = 1: one register set, tile t+1 is loaded while tile t is multiplied (one MFMA phase = 256 matrix cycles of cover);
// PF = 2: two register sets, tile t+2 / t+3 are in flight while tile t is multiplied: every load has a whole further
//         stage (two barriers) to land — the half-precision MFMA phase is 16x shorter than the fp32 one the single-set
//         schedule of conv_fast.h was built for.
#define HALF_PIPELINE(PF_, KT_)                                                                     \
  do {                                                                                              \
    typedef std::integral_constant<int, 0> S0;                                                      \
    typedef std::integral_constant<int, 1> S1;                                                      \
    if constexpr ((PF_) == 4) { /* warp-specialised, TWO register sets: tiles t+2 / t+3 in flight while t+1 is staged */ \
      const bool producer_ = threadIdx.x >= 256;                                                    \
      if (producer_) { load(S0()); store(0, S0()); step(); load(S0()); step(); load(S1()); }        \
      __syncthreads();                                                                              \
      for (int kt_ = 0; kt_ < (KT_); kt_ += 2) {                                                    \
        if (producer_) { store(1, S0()); step(); load(S0()); }                                      \
        else mma(0);                                                                                \
        __syncthreads();                                                                            \
        if (kt_ + 1 >= (KT_)) break;                                                                \
        if (producer_) { store(0, S1()); step(); load(S1()); }                                      \
        else mma(1);                                                                                \
        __syncthreads();                                                                            \
      }                                                                                             \
    } else if constexpr ((PF_) == 3) { /* warp-specialised: waves 4-7 stage tile t+1 (split + LDS writes) while waves 0-3 multiply tile t */ \
      const bool producer_ = threadIdx.x >= 256;                                                    \
      if (producer_) { load(S0()); store(0, S0()); step(); load(S0()); }                            \
      __syncthreads();                                                                              \
      for (int kt_ = 0; kt_ < (KT_); ++kt_) {                                                       \
        if (producer_) { store((kt_ & 1) ^ 1, S0()); step(); load(S0()); }                          \
        else mma(kt_ & 1);                                                                          \
        __syncthreads();                                                                            \
      }                                                                                             \
    } else if constexpr ((PF_) == 0) { /* ONE LDS buffer, two barriers per stage; loads of tile t+1 fly under the MFMAs of t */ \
      load(S0());                                                                                   \
      for (int kt_ = 0; kt_ < (KT_); ++kt_) {                                                       \
        store(0, S0()); step(); load(S0());                                                         \
        __syncthreads();                                                                            \
        mma(0);                                                                                     \
        __syncthreads();                                                                            \
      }                                                                                             \
    } else if constexpr ((PF_) == 1) {                                                              \
      load(S0()); store(0, S0()); step(); load(S0());                                               \
      __syncthreads();                                                                              \
      for (int kt_ = 0; kt_ < (KT_); ++kt_) {                                                       \
        mma(kt_ & 1); store((kt_ & 1) ^ 1, S0()); step(); load(S0());                               \
        __syncthreads();                                                                            \
      }                                                                                             \
    } else {                                                                                        \
      load(S0()); store(0, S0()); step(); load(S0()); step(); load(S1());                           \
      __syncthreads();                                                                              \
      for (int kt_ = 0; kt_ < (KT_); kt_ += 2) {                                                    \
        mma(0); store(1, S0()); step(); load(S0());                                                 \
        __syncthreads();                                                                            \
        if (kt_ + 1 >= (KT_)) break;                                                                \
        mma(1); store(0, S1()); step(); load(S1());                                                 \
        __syncthreads();                                                                            \
      }                                                                                             \
    }                                                                                               \
  } while (0)

// LDS floats: NB buffers x NS planes x (BM + BN) rows of LDH halfs, or the fp32 epilogue tile, whichever is larger
template <int DT, int BM, int BN, int PF>
struct half_smem {
  static constexpr int NS = half_cfg<DT>::NS, NB = PF == 0 ? 1 : 2;     // (PF 3: double buffer)
  static constexpr int stage = (NB * NS * (BM + BN) * LDH + 1) / 2, epi = BM * (BN + 4);
  static constexpr int floats = stage > epi ? stage : epi;
};

// ============================================================================
// forward:  y[p,k] = act( sum_{r,s,c} x[pix(p,r,s),c] * w[r,s,c,k] * scale[k] + shift[k] + res[p,k] )
//   needs C % 32 == 0, K % 4 == 0.  A: gather (K-contiguous).  B: HWIO rows (K-major) -> transposed in registers.
// ============================================================================
// GB: `gbatch` independent problems of the same shape stacked in x / w / y (the 16 transformed-domain GEMMs of a Winograd
// convolution: conv_winograd.h), one grid, plane index slowest.
template <int DT, int BM, int BN, int PF, bool GB = false>
__global__ void __launch_bounds__(PF >= 3 ? 512 : 256, (PF >= 3 || (DT == 3 && PF != 0)) ? 1 : 2)
k_conv_fwd_h(lmh_conv_desc d, const float* __restrict__ x, const float* __restrict__ w,
             const float* __restrict__ scale, const float* __restrict__ shift,
             const float* __restrict__ residual, float* __restrict__ y, int gbatch = 1) {
  typedef typename HT<DT>::T HTT;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AJ = BM / 32;
  constexpr int A_SZ = BM * LDH, B_SZ = BN * LDH;
  constexpr int LDC = BN + 4;
  constexpr int NS = half_cfg<DT>::NS, NB = PF == 0 ? 1 : 2, NR = (PF == 0 || PF == 3) ? 1 : 2 - (PF == 1), NT = PF >= 3 ? 512 : 256;
  constexpr int A_BUF = NS * A_SZ, B_BUF = NS * B_SZ;
  __shared__ __attribute__((aligned(16))) float smem[half_smem<DT, BM, BN, PF>::floats];
  HTT* const As = reinterpret_cast<HTT*>(smem);       // [NB][NS][BM][LDH]
  HTT* const Bs = As + NB * A_BUF;                    // [NB][NS][BN][LDH]
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;      // (PF == 3: producers 256..511 share the staging map)
  const int wm = wave >> 1, wn = wave & 1;
  const int M = d.N * d.OH * d.OW, K = d.K, C = d.C;
  const int tiles_n = (K + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n * (GB ? gbatch : 1));
  if (GB) {
    const int gi = tile / (tiles_m * tiles_n);
    tile -= gi * (tiles_m * tiles_n);
    x += (size_t)gi * M * C;
    w += (size_t)gi * d.R * d.S * C * K;
    y += (size_t)gi * M * K;
  }
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int CC = C / BK, KT = d.R * d.S * CC;
  // ---- A gather state (as k_conv_fwd)
  const int kq = tid & 7, arow = tid >> 3;
  int a_n[AJ], a_ih0[AJ], a_iw0[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int p = m0 + arow + 32 * j;
    if (p < M) {
      const int ow = p % d.OW, t = p / d.OW;
      a_n[j] = t / d.OH;
      a_ih0[j] = (t % d.OH) * d.stride - d.pad_top;
      a_iw0[j] = ow * d.stride - d.pad_left;
    } else { a_n[j] = -1; a_ih0[j] = 0; a_iw0[j] = 0; }
  }
  const float* pa[AJ];
  int inca[AJ];
  auto setup_rs = [&](int rs_) {
    const int r_ = rs_ / d.S, s_ = rs_ - r_ * d.S;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int ih = a_ih0[j] + r_ * d.dilation, iw = a_iw0[j] + s_ * d.dilation;
      const bool ok = a_n[j] >= 0 && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
      pa[j] = ok ? x + ((size_t)(a_n[j] * d.H + ih) * d.W + iw) * C + 4 * kq : lmh_zero_page;
      inca[j] = ok ? BK : 0;
    }
  };
  // ---- B: rows (rs*C + c) of w are consecutive GEMM-k rows; 4x4 block per thread: k-quad kq, column quad cq
  const int cq = tid >> 3;                             // 0..31
  const bool b_act = cq < BN / 4;
  const bool b_ok = b_act && (n0 + 4 * cq) < K;
  const float* pb = b_ok ? w + (size_t)(4 * kq) * K + n0 + 4 * cq : lmh_zero_page;
  const size_t rowb = b_ok ? (size_t)K : 0, incb = b_ok ? (size_t)BK * K : 0;

  f32x4 ra[NR][AJ], rb[NR][4];
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  int rs = 0, cc = 0, ptile = 0;
  setup_rs(0);
  auto step = [&]() {
    if (ptile + 1 >= KT) return;
    ++ptile;
    if (++cc == CC) { cc = 0; ++rs; setup_rs(rs); }
    else {
#pragma unroll
      for (int j = 0; j < AJ; ++j) pa[j] += inca[j];
    }
    pb += incb;
  };
  auto load = [&](auto S) {
    constexpr int s_ = decltype(S)::value;
#pragma unroll
    for (int j = 0; j < AJ; ++j) ra[s_][j] = *reinterpret_cast<const f32x4*>(pa[j]);
    if (b_act) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rb[s_][i] = *reinterpret_cast<const f32x4*>(pb + i * rowb);
    }
  };
  auto store = [&](int buf, auto S) {
    constexpr int s_ = decltype(S)::value;
    HTT* Ad = As + buf * A_BUF;
    HTT* Bd = Bs + buf * B_BUF;
#pragma unroll
    for (int j = 0; j < AJ; ++j) st_kc<DT>(Ad, A_SZ, arow + 32 * j, kq, ra[s_][j]);
    if (b_act) st_km<DT>(Bd, B_SZ, 4 * cq, kq, rb[s_]);
  };
  auto mma = [&](int buf) {
    mfma_stage_h<DT, TM, TN>(As + buf * A_BUF, Bs + buf * B_BUF, acc, wm * (BM / 2), wn * (BN / 2), lane, A_SZ, B_SZ);
  };
  HALF_PIPELINE(PF, KT);
  // ---- epilogue through LDS (fp32), as k_conv_fwd
  constexpr int CT = BN / 4, RSTEP = NT / CT;
  const int c4 = (int)threadIdx.x % CT, r0 = (int)threadIdx.x / CT;
  const int col = n0 + 4 * c4;
  if (threadIdx.x < 256) acc_to_lds<BM, BN, TM, TN>(smem, acc, wm, wn, lane);
  __syncthreads();
  if (col < K) {
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (scale) sc = *reinterpret_cast<const f32x4*>(scale + col);
    if (shift) sh = *reinterpret_cast<const f32x4*>(shift + col);
    const float act_lo = d.act ? 0.f : -INFINITY, act_hi = (d.act == 2) ? 6.f : INFINITY;
    for (int r = r0; r < BM; r += RSTEP) {
      const int row = m0 + r;
      if (row >= M) break;
      f32x4 v = *reinterpret_cast<const f32x4*>(&smem[r * LDC + 4 * c4]);
      v = v * sc + sh;
      if (residual) v += *reinterpret_cast<const f32x4*>(residual + (size_t)row * K + col);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], act_lo), act_hi);
      *reinterpret_cast<f32x4*>(y + (size_t)row * K + col) = v;
    }
  }
}

// ============================================================================
// backward data:  dx[p,c] = sum_{r,s,k} dy[opix(p,r,s),k] * kscale[k] * w[r,s,c,k]  (+ addend)
//   needs K % 32 == 0, C % 4 == 0.  A: dy gather (K-contiguous).  B: w[rs][c][k] rows (K-contiguous).
// ============================================================================
template <int DT, int BM, int BN, int PF>
__global__ void __launch_bounds__(PF >= 3 ? 512 : 256, (PF >= 3 || (DT == 3 && PF != 0)) ? 1 : 2)
k_conv_bwd_data_h(lmh_conv_desc d, const float* __restrict__ dy, const float* __restrict__ w,
                  const float* __restrict__ kscale, const float* __restrict__ addend, float gscale,
                  float* __restrict__ dx) {
  typedef typename HT<DT>::T HTT;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AJ = BM / 32, BJ = BN / 32;
  constexpr int A_SZ = BM * LDH, B_SZ = BN * LDH;
  constexpr int LDC = BN + 4;
  constexpr int NS = half_cfg<DT>::NS, NB = PF == 0 ? 1 : 2, NR = (PF == 0 || PF == 3) ? 1 : 2 - (PF == 1), NT = PF >= 3 ? 512 : 256;
  constexpr int A_BUF = NS * A_SZ, B_BUF = NS * B_SZ;
  __shared__ __attribute__((aligned(16))) float smem[half_smem<DT, BM, BN, PF>::floats];
  HTT* const As = reinterpret_cast<HTT*>(smem);
  HTT* const Bs = As + NB * A_BUF;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;      // (PF == 3: producers 256..511 share the staging map)
  const int wm = wave >> 1, wn = wave & 1;
  const int M = d.N * d.H * d.W, K = d.K, C = d.C;
  const int KC = K / BK;
  const int tiles_n = (C + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int kq = tid & 7, arow = tid >> 3;
  const int KT = d.R * d.S * KC;
  int a_n[AJ], a_h[AJ], a_w[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int p = m0 + arow + 32 * j;
    if (p < M) {
      const int t = p / d.W;
      a_w[j] = p - t * d.W + d.pad_left;
      a_n[j] = t / d.H;
      a_h[j] = t - a_n[j] * d.H + d.pad_top;
    } else { a_n[j] = -1; a_h[j] = 0; a_w[j] = 0; }
  }
  const float* pa[AJ];
  int inca[AJ];
  auto setup_rs = [&](int rs_) {
    const int r_ = rs_ / d.S, s_ = rs_ - r_ * d.S;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int th = a_h[j] - r_ * d.dilation, tw = a_w[j] - s_ * d.dilation;
      int oh = th, ow = tw;
      bool ok = a_n[j] >= 0 && th >= 0 && tw >= 0;
      if (d.stride > 1) {
        oh = th / d.stride; ow = tw / d.stride;
        ok = ok && (oh * d.stride == th) && (ow * d.stride == tw);
      }
      ok = ok && oh < d.OH && ow < d.OW;
      pa[j] = ok ? dy + ((size_t)(a_n[j] * d.OH + oh) * d.OW + ow) * K + 4 * kq : lmh_zero_page;
      inca[j] = ok ? BK : 0;
    }
  };
  const float* pb[BJ];
  int incb[BJ];
  size_t tapb[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int c = n0 + arow + 32 * j;
    const bool ok = c < C;
    pb[j] = ok ? w + (size_t)c * K + 4 * kq : lmh_zero_page;
    incb[j] = ok ? BK : 0;
    tapb[j] = ok ? (size_t)C * K - K + BK : 0;     // end of this tap's k range -> start of the next tap
  }
  const float* pks = kscale ? kscale + 4 * kq : lmh_zero_page;
  const int incks = kscale ? BK : 0;
  f32x4 ra[NR][AJ], rb[NR][BJ], ks[NR];
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  int rs = 0, kc = 0, ptile = 0;
  setup_rs(0);
  auto step = [&]() {
    if (ptile + 1 >= KT) return;
    ++ptile;
    if (++kc == KC) {
      kc = 0; ++rs;
      setup_rs(rs);
#pragma unroll
      for (int j = 0; j < BJ; ++j) pb[j] += tapb[j];
      pks -= (KC - 1) * incks;
    } else {
#pragma unroll
      for (int j = 0; j < AJ; ++j) pa[j] += inca[j];
#pragma unroll
      for (int j = 0; j < BJ; ++j) pb[j] += incb[j];
      pks += incks;
    }
  };
  auto load = [&](auto S) {
    constexpr int s_ = decltype(S)::value;
    ks[s_] = *reinterpret_cast<const f32x4*>(pks);
#pragma unroll
    for (int j = 0; j < AJ; ++j) ra[s_][j] = *reinterpret_cast<const f32x4*>(pa[j]);
#pragma unroll
    for (int j = 0; j < BJ; ++j) rb[s_][j] = *reinterpret_cast<const f32x4*>(pb[j]);
  };
  auto store = [&](int buf, auto S) {
    constexpr int s_ = decltype(S)::value;
    HTT* Ad = As + buf * A_BUF;
    HTT* Bd = Bs + buf * B_BUF;
    const f32x4 m = kscale ? ks[s_] * gscale : f32x4{gscale, gscale, gscale, gscale};
#pragma unroll
    for (int j = 0; j < AJ; ++j) st_kc<DT>(Ad, A_SZ, arow + 32 * j, kq, ra[s_][j] * m);
#pragma unroll
    for (int j = 0; j < BJ; ++j) st_kc<DT>(Bd, B_SZ, arow + 32 * j, kq, rb[s_][j]);
  };
  auto mma = [&](int buf) {
    mfma_stage_h<DT, TM, TN>(As + buf * A_BUF, Bs + buf * B_BUF, acc, wm * (BM / 2), wn * (BN / 2), lane, A_SZ, B_SZ);
  };
  HALF_PIPELINE(PF, KT);
  constexpr int CT = BN / 4, RSTEP = NT / CT;
  const int c4 = (int)threadIdx.x % CT, r0 = (int)threadIdx.x / CT;
  const int col = n0 + 4 * c4;
  if (threadIdx.x < 256) acc_to_lds<BM, BN, TM, TN>(smem, acc, wm, wn, lane);
  __syncthreads();
  if (col < C) {
    const float inv = 1.f / gscale;
    for (int r = r0; r < BM; r += RSTEP) {
      const int row = m0 + r;
      if (row >= M) break;
      f32x4 v = *reinterpret_cast<const f32x4*>(&smem[r * LDC + 4 * c4]) * inv;
      if (addend) v += *reinterpret_cast<const f32x4*>(addend + (size_t)row * C + col);
      *reinterpret_cast<f32x4*>(dx + (size_t)row * C + col) = v;
    }
  }
}

// ============================================================================
// backward weight:  dw[rs,c,k] = sum_p x[pix(p,r,s),c] * g[p,k]; reduction split over the pixels (slabs in `out`,
//   reduced by k_splitk_reduce).  needs C % 4 == 0, K % 4 == 0.  Both operands pixel-major -> register transposes.
// ============================================================================
// GB: the R*S "taps" are independent GEMMs stacked in x / g (Winograd weight gradient: tap rs reads x + rs*P*C, g + rs*P*K)
template <int DT, int BM, int BN, int PF, bool GB = false>
__global__ void __launch_bounds__(PF >= 3 ? 512 : 256, (PF >= 3 || (DT == 3 && PF != 0)) ? 1 : 2)
k_conv_bwd_weight_h(lmh_conv_desc d, const float* __restrict__ x, const float* __restrict__ g,
                    float* __restrict__ out, int kt_per_split, lmh_fastdiv div_ow, lmh_fastdiv div_oh,
                    float gscale, int tiles_x, int tiles_y, int splits, float* __restrict__ colpart = nullptr) {
  typedef typename HT<DT>::T HTT;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_SZ = BM * LDH, B_SZ = BN * LDH;
  constexpr int LDC = BN + 4;
  constexpr int NS = half_cfg<DT>::NS, NB = PF == 0 ? 1 : 2, NR = (PF == 0 || PF == 3) ? 1 : 2 - (PF == 1), NT = PF >= 3 ? 512 : 256;
  constexpr int A_BUF = NS * A_SZ, B_BUF = NS * B_SZ;
  __shared__ __attribute__((aligned(16))) float smem[half_smem<DT, BM, BN, PF>::floats];
  HTT* const As = reinterpret_cast<HTT*>(smem);
  HTT* const Bs = As + NB * A_BUF;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;      // (PF == 3: producers 256..511 share the staging map)
  const int wm = wave >> 1, wn = wave & 1;
  const int P = d.N * d.OH * d.OW, K = d.K, C = d.C;
  const int lin = xcd_remap(blockIdx.x, tiles_x * tiles_y * splits);
  const int bz = lin / (tiles_x * tiles_y), rem = lin - bz * (tiles_x * tiles_y);
  const int tiles_c = (C + BM - 1) / BM;
  int by, bx;
  if (GB) {                                     // plane-major: one XCD streams a plane's operands once (conv_fast.h)
    const int per_plane = tiles_c * tiles_y;
    const int plane = rem / per_plane, r2 = rem - plane * per_plane;
    by = r2 / tiles_c;
    bx = plane * tiles_c + (r2 - by * tiles_c);
  } else {
    by = rem / tiles_x;
    bx = rem - by * tiles_x;
  }
  const int rs = bx / tiles_c, m0 = (bx % tiles_c) * BM;
  const int n0 = by * BN;
  const int r = rs / d.S, s = rs - r * d.S;
  const int KT_all = (P + BK - 1) / BK;
  const int kt_begin = bz * kt_per_split;
  const int kt_end = min(KT_all, kt_begin + kt_per_split);
  const int dh0 = r * d.dilation - d.pad_top, dw0 = s * d.dilation - d.pad_left;
  // 4(pixel) x 4(channel) block per thread and operand: pixel quad kq (fastest over lanes), channel quad cq
  const int kq = tid & 7, cq = tid >> 3;
  const bool a_act = cq < BM / 4, b_act = cq < BN / 4;
  const bool a_ok = a_act && (m0 + 4 * cq) < C, b_ok = b_act && (n0 + 4 * cq) < K;
  const float* xb = x + m0 + 4 * cq + (GB ? (size_t)rs * P * C : 0);
  const float* gb = g + n0 + 4 * cq + (GB ? (size_t)rs * P * K : 0);
  int p0 = kt_begin * BK + 4 * kq;       // first of this thread's 4 pixels in the tile the pointers stand on
  f32x4 ra[NR][4], rb[NR][4];
  auto step = [&]() { p0 += BK; };       // past the split's end: pixels of the next split or (>= P) the zero page
  auto load = [&](auto S) {
    constexpr int s_ = decltype(S)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned p = (unsigned)(p0 + i);
      const unsigned t = lmh_div(p, div_ow), ow = p - t * (unsigned)d.OW;
      const unsigned n = lmh_div(t, div_oh), oh = t - n * (unsigned)d.OH;
      const int ih = (int)oh * d.stride + dh0, iw = (int)ow * d.stride + dw0;
      const bool oka = a_ok && n < (unsigned)d.N && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
      const float* pa_ = oka ? xb + ((size_t)((int)n * d.H + ih) * d.W + iw) * C : lmh_zero_page;
      if (a_act) ra[s_][i] = *reinterpret_cast<const f32x4*>(pa_);
      const bool okb = b_ok && (int)p < P;
      const float* pb_ = okb ? gb + (size_t)p * K : lmh_zero_page;
      if (b_act) rb[s_][i] = *reinterpret_cast<const f32x4*>(pb_);
    }
  };
  auto store = [&](int buf, auto S) {
    constexpr int s_ = decltype(S)::value;
    HTT* Ad = As + buf * A_BUF;
    HTT* Bd = Bs + buf * B_BUF;
    if (a_act) st_km<DT>(Ad, A_SZ, 4 * cq, kq, ra[s_]);
    if (b_act) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rb[s_][i] *= gscale;
      st_km<DT>(Bd, B_SZ, 4 * cq, kq, rb[s_]);
    }
  };
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  // bf16x3 only: the per-channel sums of g (dbeta / dbias) from the B tile while it sits in LDS — the three pieces of an
  // element add up to it exactly — by the blocks of ONE tile column (tap 0, first channel tile); [splits][K] partial rows
  // in `colpart`, folded by the reduce / tail launch.  Fixed order: deterministic.
  const bool do_col = DT == 3 && !GB && colpart != nullptr && bx == 0 && threadIdx.x < BN;
  float csum = 0.f;
  auto mma = [&](int buf) {
    if (do_col) {
      const HTT* row = Bs + buf * B_BUF + (int)threadIdx.x * LDH;
#pragma unroll
      for (int pz = 0; pz < NS; ++pz)
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
          const typename HT<DT>::V8 v = *reinterpret_cast<const typename HT<DT>::V8*>(row + pz * B_SZ + 8 * q);
#pragma unroll
          for (int e = 0; e < 8; ++e) csum += (float)v[e];
        }
    }
    mfma_stage_h<DT, TM, TN>(As + buf * A_BUF, Bs + buf * B_BUF, acc, wm * (BM / 2), wn * (BN / 2), lane, A_SZ, B_SZ);
  };
  const int n_st = kt_end - kt_begin;
  HALF_PIPELINE(PF, n_st);
  if (do_col && n0 + (int)threadIdx.x < K) colpart[(size_t)bz * K + n0 + threadIdx.x] = csum / gscale;
  if (threadIdx.x < 256) acc_to_lds<BM, BN, TM, TN>(smem, acc, wm, wn, lane);
  __syncthreads();
  float* o = out + (size_t)bz * ((size_t)d.R * d.S * C * K) + (size_t)rs * C * K;
  constexpr int CT = BN / 4, RSTEP = NT / CT;
  const int c4 = (int)threadIdx.x % CT, r0 = (int)threadIdx.x / CT;
  const int col = n0 + 4 * c4;
  if (col < K) {
    const float inv = 1.f / gscale;
    for (int rr = r0; rr < BM; rr += RSTEP) {
      const int row = m0 + rr;
      if (row >= C) break;
      *reinterpret_cast<f32x4*>(o + (size_t)row * K + col) = *reinterpret_cast<const f32x4*>(&smem[rr * LDC + 4 * c4]) * inv;
    }
  }
}
