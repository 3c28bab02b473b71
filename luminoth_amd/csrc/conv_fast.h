// Fast-path implicit-GEMM convolution kernels (gfx950, v_mfma_f32_32x32x2_f32).
//
// Structure shared by forward / backward-data / backward-weight:
//   * 256 threads = 4 waves (2x2), block tile BM x BN in {128x128, 128x64, 64x128, 64x64}, BK = 32;
//   * global -> VGPR prefetch of stage t+1 is issued (unconditionally: the last
//     iteration re-reads a valid address and drops the data) before the MFMAs
//     of stage t and written to the OTHER LDS buffer after them: one barrier
//     per stage, HBM/L2 latency hidden under 64 MFMAs (4096 matrix cycles);
//   * LDS -> VGPR operand fragments are software pipelined in 4 groups of 4
//     k-pairs: group g+1 is read while the MFMAs of group g issue, so one wave
//     per SIMD is enough to keep the matrix pipe busy;
//   * K-contiguous operands sit in LDS as [row][BK+4] and are read with
//     conflict-free ds_read_b128; K-major operands sit as [BK][cols] and are
//     read with conflict-free ds_read_b32;
//   * the epilogue goes through LDS (the operand buffers are dead by then):
//     accumulators are transposed into a row-major tile and leave the CU as
//     float4 rows (512 contiguous bytes per 32 lanes) with scale / shift /
//     residual / activation fused in.
// No lambdas here on purpose: the prefetch registers must stay in VGPRs (an
// earlier lambda-based version was demoted to scratch by the compiler).
#pragma once
#include "conv_common.h"

template <int T, bool KC, int LD>
__device__ __forceinline__ void load_frag(const float* __restrict__ S, int off, int g, int h, int l31,
                                          float (&f)[T][4]) {
  const int kb = 16 * h + 4 * g;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (KC) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&S[(off + t * 32 + l31) * LD + kb]);
      f[t][0] = v.x; f[t][1] = v.y; f[t][2] = v.z; f[t][3] = v.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) f[t][i] = S[(kb + i) * LD + off + t * 32 + l31];
    }
  }
}

template <int TM, int TN>
__device__ __forceinline__ void mfma_group(const float (&a)[TM][4], const float (&b)[TN][4],
                                           f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][i], b[tn][i], acc[tm][tn], 0, 0, 0);
}

// One BK=32 stage out of LDS, fragment reads pipelined one group ahead.
template <int TM, int TN, bool A_KC, bool B_KC, int LDA, int LDB>
__device__ __forceinline__ void mfma_stage_pipelined(const float* __restrict__ As, const float* __restrict__ Bs,
                                                     f32x16 (&acc)[TM][TN], int a_off, int b_off, int lane) {
  const int h = lane >> 5, l31 = lane & 31;
  float a0[TM][4], b0[TN][4], a1[TM][4], b1[TN][4];
  // sched_barrier(0): nothing moves across — keeps the reads of group g+1 AHEAD of the MFMAs of group g
  // (left alone, the scheduler sinks every read to just before its first use and the wave stalls on LDS).
  load_frag<TM, A_KC, LDA>(As, a_off, 0, h, l31, a0);
  load_frag<TN, B_KC, LDB>(Bs, b_off, 0, h, l31, b0);
  load_frag<TM, A_KC, LDA>(As, a_off, 1, h, l31, a1);
  load_frag<TN, B_KC, LDB>(Bs, b_off, 1, h, l31, b1);
  __builtin_amdgcn_sched_barrier(0);
  mfma_group<TM, TN>(a0, b0, acc);
  __builtin_amdgcn_sched_barrier(0);
  load_frag<TM, A_KC, LDA>(As, a_off, 2, h, l31, a0);
  load_frag<TN, B_KC, LDB>(Bs, b_off, 2, h, l31, b0);
  __builtin_amdgcn_sched_barrier(0);
  mfma_group<TM, TN>(a1, b1, acc);
  __builtin_amdgcn_sched_barrier(0);
  load_frag<TM, A_KC, LDA>(As, a_off, 3, h, l31, a1);
  load_frag<TN, B_KC, LDB>(Bs, b_off, 3, h, l31, b1);
  __builtin_amdgcn_sched_barrier(0);
  mfma_group<TM, TN>(a0, b0, acc);
  mfma_group<TM, TN>(a1, b1, acc);
  __builtin_amdgcn_sched_barrier(0);
}

// Full pipeline stage (the "split write / re-issue" schedule): on entry the staging registers hold tile
// t+1 (loaded during the previous stage).  Its ds_writes into the OTHER LDS buffer are interleaved with the
// MFMAs of fragment group 0, the global loads of tile t+2 (same registers) with the MFMAs of group 1, so
// neither the LDS-write pass nor the VMEM issue ever runs with the matrix pipe idle; fragment reads stay
// one group ahead.  sched_group_barrier pins the MFMA : DS_WRITE / VMEM interleave inside each region,
// sched_barrier(0) pins the regions.
template <int TM, int TN, bool A_KC, bool B_KC, int LDA, int LDB, int NWR, int NLD, class WriteF, class LoadF>
__device__ __forceinline__ void mfma_stage_split(const float* __restrict__ As, const float* __restrict__ Bs,
                                                 f32x16 (&acc)[TM][TN], int a_off, int b_off, int lane,
                                                 WriteF&& do_writes, LoadF&& do_loads) {
  constexpr int NM = 4 * TM * TN;   // MFMAs per fragment group
  const int h = lane >> 5, l31 = lane & 31;
  float a0[TM][4], b0[TN][4], a1[TM][4], b1[TN][4];
  load_frag<TM, A_KC, LDA>(As, a_off, 0, h, l31, a0);
  load_frag<TN, B_KC, LDB>(Bs, b_off, 0, h, l31, b0);
  load_frag<TM, A_KC, LDA>(As, a_off, 1, h, l31, a1);
  load_frag<TN, B_KC, LDB>(Bs, b_off, 1, h, l31, b1);
  __builtin_amdgcn_sched_barrier(0);
  // ---- group 0 MFMAs || ds_write of tile t+1
  mfma_group<TM, TN>(a0, b0, acc);
  do_writes();
  {
    constexpr int PER = (NM / NWR) > 0 ? (NM / NWR) : 1;
#pragma unroll
    for (int i = 0; i < NWR; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // DS write
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- group 1 MFMAs || fragment reads of group 2 || global loads of tile t+2
  load_frag<TM, A_KC, LDA>(As, a_off, 2, h, l31, a0);
  load_frag<TN, B_KC, LDB>(Bs, b_off, 2, h, l31, b0);
  __builtin_amdgcn_sched_barrier(0);
  mfma_group<TM, TN>(a1, b1, acc);
  do_loads();
  {
    constexpr int PER = (NM / NLD) > 0 ? (NM / NLD) : 1;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  load_frag<TM, A_KC, LDA>(As, a_off, 3, h, l31, a1);
  load_frag<TN, B_KC, LDB>(Bs, b_off, 3, h, l31, b1);
  __builtin_amdgcn_sched_barrier(0);
  mfma_group<TM, TN>(a0, b0, acc);
  mfma_group<TM, TN>(a1, b1, acc);
  __builtin_amdgcn_sched_barrier(0);
}

// g = dy * act'(y): act 1 = relu (y > 0), 2 = relu6 (0 < y < 6); y is the layer OUTPUT (post-activation)
__device__ __forceinline__ f32x4 act_mask(f32x4 dy, f32x4 y, float hi) {
  f32x4 g;
  g.x = (y.x > 0.f && y.x < hi) ? dy.x : 0.f;
  g.y = (y.y > 0.f && y.y < hi) ? dy.y : 0.f;
  g.z = (y.z > 0.f && y.z < hi) ? dy.z : 0.f;
  g.w = (y.w > 0.f && y.w < hi) ? dy.w : 0.f;
  return g;
}

template <int TM, int TN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[tm][tn][i] = 0.f;
}

// accumulators -> row-major LDS tile Cs[BM][BN + 4]
template <int BM, int BN, int TM, int TN>
__device__ __forceinline__ void acc_to_lds(float* __restrict__ Cs, const f32x16 (&acc)[TM][TN], int wm, int wn,
                                           int lane) {
  constexpr int LDC = BN + 4;
  const int l31 = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = wm * (BM / 2) + tm * 32 + (i & 3) + 8 * (i >> 2) + rbase;
        Cs[row * LDC + wn * (BN / 2) + tn * 32 + l31] = acc[tm][tn][i];
      }
}

// ============================================================================
// forward (fast path: C % 32 == 0, K % 4 == 0)
//   y[p, k] = act( sum_{r,s,c} x[pix(p,r,s), c] * w[r,s,c,k] * scale[k] + shift[k] + res[p,k] )
//   GEMM M = N*OH*OW, N = K, Kg = R*S*C.  A: gather, K-contiguous.  B: HWIO, K-major.
// ============================================================================
// GB: `gbatch` independent problems of the same shape stacked behind each other in x / w / y (the 16 transformed-
// domain GEMMs of a Winograd convolution are launched as ONE grid: 64 tiles each would leave the chip empty).
template <int BM, int BN, bool GB = false>
__global__ void __launch_bounds__(256)
k_conv_fwd(lmh_conv_desc d, const float* __restrict__ x, const float* __restrict__ w,
           const float* __restrict__ scale, const float* __restrict__ shift,
           const float* __restrict__ residual, float* __restrict__ y, int gbatch = 1,
           uint32_t* __restrict__ act_bits = nullptr) {
  // act_bits (optional, K % 32 == 0): one bit per output element, [pixel][K / 32] words, bit = act'(y) != 0 (relu:
  // y > 0; relu6: 0 < y < 6).  The backward-data kernel of the layer ABOVE applies it in its epilogue, so the
  // gradient that reaches this layer already is g = dy * act'(y) and no separate lmh_act_bwd pass exists.
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AJ = BM / 32, BJ = BN / 32;
  constexpr int A_SZ = BM * LDK, B_SZ = BK * BN;
  constexpr int LDC = BN + 4;
  constexpr int SMEM = (2 * (A_SZ + B_SZ) > BM * LDC) ? 2 * (A_SZ + B_SZ) : BM * LDC;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  float* const As = smem;               // [2][BM][LDK]
  float* const Bs = smem + 2 * A_SZ;    // [2][BK][BN]
  conv_stagger((160 * 1024) / (SMEM * 4));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = d.N * d.OH * d.OW, K = d.K, C = d.C;
  const int tiles_n = (K + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n * (GB ? gbatch : 1));
  if (GB) {
    const int g = tile / (tiles_m * tiles_n);
    tile -= g * (tiles_m * tiles_n);
    x += (size_t)g * M * C;
    w += (size_t)g * d.R * d.S * C * K;
    y += (size_t)g * M * K;
  }
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int dbg = conv_probe_bits();
  if (dbg & 256) residual = nullptr;
  const int CC = C / BK, KT = (dbg & 1024) ? 0 : d.R * d.S * CC;

  // ---- A gather state: AJ rows (output pixels) per thread, 4 channels each
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
#define FWD_SETUP_RS(rs_)                                                                         \
  do {                                                                                            \
    const int r_ = (rs_) / d.S, s_ = (rs_) - r_ * d.S;                                            \
    _Pragma("unroll") for (int j = 0; j < AJ; ++j) {                                              \
      const int ih = a_ih0[j] + r_ * d.dilation, iw = a_iw0[j] + s_ * d.dilation;                 \
      const bool ok = a_n[j] >= 0 && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W; \
      pa[j] = ok ? x + ((size_t)(a_n[j] * d.H + ih) * d.W + iw) * C + 4 * kq : lmh_zero_page;     \
      inca[j] = ok ? BK : 0;                                                                      \
    }                                                                                             \
  } while (0)
  // ---- B (weights, [Kg][K]) state
  constexpr int BROW_T = BN / 4, BROW_STEP = 256 / BROW_T;
  const int bx4 = tid % BROW_T, bk = tid / BROW_T;
  const bool b_ok = (n0 + 4 * bx4) < K;
  const float* pb = b_ok ? w + (size_t)bk * K + n0 + 4 * bx4 : lmh_zero_page;
  const size_t incb = b_ok ? (size_t)BK * K : 0;          // one stage
  const size_t rowb = b_ok ? (size_t)BROW_STEP * K : 0;   // next row slot of this thread

  f32x4 ra[AJ], rb[BJ];
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);

  int rs = 0, cc = 0;
  FWD_SETUP_RS(0);
  // block-uniform pointer advance to the next BK slice (tap change re-derives the gather pointers)
#define FWD_ADVANCE()                                                          \
  do {                                                                         \
    if (++cc == CC) { cc = 0; ++rs; FWD_SETUP_RS(rs); }                        \
    else { _Pragma("unroll") for (int j = 0; j < AJ; ++j) pa[j] += inca[j]; }  \
    pb += incb;                                                                \
  } while (0)
#define FWD_LOAD()                                                                                     \
  do {                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < AJ; ++j) ra[j] = *reinterpret_cast<const f32x4*>(pa[j]);     \
    _Pragma("unroll") for (int j = 0; j < BJ; ++j) rb[j] = *reinterpret_cast<const f32x4*>(pb + j * rowb); \
  } while (0)
#define FWD_STORE(buf_)                                                                                \
  do {                                                                                                 \
    float* Ad = As + (buf_) * A_SZ;                                                                    \
    float* Bd = Bs + (buf_) * B_SZ;                                                                    \
    _Pragma("unroll") for (int j = 0; j < AJ; ++j)                                                     \
        *reinterpret_cast<f32x4*>(&Ad[(arow + 32 * j) * LDK + 4 * kq]) = ra[j];                        \
    _Pragma("unroll") for (int j = 0; j < BJ; ++j)                                                     \
        *reinterpret_cast<f32x4*>(&Bd[(bk + BROW_STEP * j) * BN + 4 * bx4]) = rb[j];                   \
  } while (0)
  FWD_LOAD();                       // tile 0
  FWD_STORE(0);
  if (KT > 1) FWD_ADVANCE();
  FWD_LOAD();                       // tile 1 (or tile 0 again: never stored)
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 2 < KT) FWD_ADVANCE();   // pointers -> tile kt+2 (else they stay on a valid tile; data unused)
    mfma_stage_split<TM, TN, true, false, LDK, BN, AJ + BJ, AJ + BJ>(
        As + cur * A_SZ, Bs + cur * B_SZ, acc, wm * (BM / 2), wn * (BN / 2), lane,
        [&]() { FWD_STORE(cur ^ 1); },   // tile kt+1 (harmless duplicate after the last tile)
        [&]() { FWD_LOAD(); });          // tile kt+2
    __syncthreads();
  }
#undef FWD_ADVANCE
#undef FWD_LOAD
#undef FWD_STORE
#undef FWD_SETUP_RS

  // ---- epilogue through LDS: float4 rows.  EVERY load of the epilogue — the residual rows of all chunks, scale and
  // shift — is requested BEFORE the accumulator transpose, so that one round of (L2 / HBM) latency runs under it and
  // nothing but stores follows the first store.  Round 4 (ISA check): loads and stores count on the same vmcnt and may
  // retire out of order with respect to each other, so with both kinds in flight the compiler can only wait for
  // vmcnt(0): the second residual chunk requested after the first chunk's stores, and the scale / shift vectors
  // requested after the barrier, each cost a full store + load round trip per tile (three exposed trips in all).
  constexpr int CT = BN / 4, RSTEP = 256 / CT;
  constexpr int NR = BM / RSTEP, NRC = NR < 8 ? NR : 8, NCH = NR / NRC;
  const int c4 = tid % CT, r0 = tid / CT;
  const int col = n0 + 4 * c4;
  const bool col_ok = col < K;
  f32x4 ex[NCH][NRC];
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (col_ok) {
    if (scale) sc = *reinterpret_cast<const f32x4*>(scale + col);
    if (shift) sh = *reinterpret_cast<const f32x4*>(shift + col);
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int i = 0; i < NRC; ++i) {
      const int row = m0 + r0 + (ch * NRC + i) * RSTEP;
      ex[ch][i] = (residual && col_ok && row < M) ? *reinterpret_cast<const f32x4*>(residual + (size_t)row * K + col)
                                                  : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  acc_to_lds<BM, BN, TM, TN>(smem, acc, wm, wn, lane);
  __syncthreads();
  // every requested row is awaited HERE, in front of the first store: a wait the compiler places later, at the join
  // behind the first (conditional) row, would be a vmcnt(0) that also waits for that row's stores
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int i = 0; i < NRC; ++i) asm volatile("" ::"v"(ex[ch][i]));
  if (col_ok) {
    const float act_lo = d.act ? 0.f : -INFINITY, act_hi = (d.act == 2) ? 6.f : INFINITY;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
      for (int i = 0; i < NRC; ++i) {
        const int r = r0 + (ch * NRC + i) * RSTEP;
        const int row = m0 + r;
        if (row < M) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&smem[r * LDC + 4 * c4]);
          if (scale) v *= sc;
          v += sh;
          v += ex[ch][i];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], act_lo), act_hi);
          if (!(dbg & 512) || v.x == 12345.678f) *reinterpret_cast<f32x4*>(y + (size_t)row * K + col) = v;
          if (act_bits) {      // 8 adjacent lanes hold the 32 channels of one mask word (same row: all active together)
            unsigned nib = ((v.x > 0.f && v.x < act_hi) ? 1u : 0u) | ((v.y > 0.f && v.y < act_hi) ? 2u : 0u) |
                           ((v.z > 0.f && v.z < act_hi) ? 4u : 0u) | ((v.w > 0.f && v.w < act_hi) ? 8u : 0u);
            nib <<= 4 * (c4 & 7);
            nib |= __shfl_xor(nib, 1);
            nib |= __shfl_xor(nib, 2);
            nib |= __shfl_xor(nib, 4);
            if ((c4 & 7) == 0) act_bits[(size_t)row * (K >> 5) + (col >> 5)] = nib;
          }
        }
      }
    }
  }
}

// ============================================================================
// backward data (fast path: K % 32 == 0, C % 4 == 0)
//   dx[p, c] = sum_{r,s,k} dy[opix(p,r,s), k] * kscale[k] * w[r,s,c,k]   (+ addend)
//   GEMM M = N*H*W, N = C, Kg = R*S*K.  A: dy gather (K-contiguous).  B: w[rs][c][k] (K-contiguous).
// ============================================================================
template <int BM, int BN, bool YACT>   // YACT: A operand is dy * act'(yact) (fused activation backward)
__global__ void __launch_bounds__(256)
k_conv_bwd_data(lmh_conv_desc d, const float* __restrict__ dy, const float* __restrict__ w,
                const float* __restrict__ kscale, const float* __restrict__ addend,
                const float* __restrict__ yact, const uint32_t* __restrict__ xbits,
                float* __restrict__ dx) {
  // xbits (optional, C % 32 == 0): the activation bit mask of the layer INPUT x ([pixel][C / 32] words, written by the
  // forward kernel that produced x): the epilogue emits dx * act'(x), the pre-activation gradient of the layer below.
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AJ = BM / 32, BJ = BN / 32;
  constexpr int A_SZ = BM * LDK, B_SZ = BN * LDK;
  constexpr int LDC = BN + 4;
  constexpr int SMEM = (2 * (A_SZ + B_SZ) > BM * LDC) ? 2 * (A_SZ + B_SZ) : BM * LDC;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  float* const As = smem;               // [2][BM][LDK]
  float* const Bs = smem + 2 * A_SZ;    // [2][BN][LDK]
  conv_stagger((160 * 1024) / (SMEM * 4));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = d.N * d.H * d.W, K = d.K, C = d.C;
  const int KC = K / BK;
  const int tiles_n = (C + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int kq = tid & 7, arow = tid >> 3;
  // Stride-2 3x3: an input pixel (h, w) only receives the taps with r = (h + pad_top) and s = (w + pad_left)
  // mod 2 — 1 to 4 of the 9.  The rows of the GEMM are therefore enumerated parity class by parity class
  // (class = 2*(h&1) + (w&1), M/4 pixels each), so that a whole tile shares its tap set and the other taps are
  // skipped instead of multiplied against the zero page (2.25 of 9 taps on average: 4x fewer MFMA stages).
  const int Mq = M >> 2;
  const bool par = d.stride == 2 && d.dilation == 1 && d.R * d.S > 1 && d.R <= 3 && d.S <= 3 &&
                   !(d.H & 1) && !(d.W & 1) && (Mq % BM) == 0;
  const int H2 = d.H >> 1, W2 = d.W >> 1;
  auto pixel = [&](int p, int& n, int& h, int& ww) {
    if (par) {
      const int cls = p / Mq, q = p - cls * Mq;
      const int t = q / W2;
      ww = 2 * (q - t * W2) + (cls & 1);
      n = t / H2;
      h = 2 * (t - n * H2) + (cls >> 1);
    } else {
      const int t = p / d.W;
      ww = p - t * d.W;
      n = t / d.H;
      h = t - n * d.H;
    }
  };
  unsigned taps = 0;      // 4-bit tap ids (r*S + s), in increasing order
  int ntap = 0;
  if (par) {
    const int cls = m0 / Mq;
    for (int r_ = 0; r_ < d.R; ++r_)
      for (int s_ = 0; s_ < d.S; ++s_)
        if ((((cls >> 1) + d.pad_top - r_) & 1) == 0 && (((cls & 1) + d.pad_left - s_) & 1) == 0)
          taps |= (unsigned)(r_ * d.S + s_) << (4 * ntap++);
    if (ntap == 0) { taps = 0; ntap = 1; }   // cannot happen for 3x3 (every parity has a tap); keep the loop well formed
  } else {
    ntap = d.R * d.S;
  }
  const int KT = ntap * KC;
#define BD_TAP(i_) (par ? (int)((taps >> (4 * (i_))) & 15u) : (i_))
  int a_n[AJ], a_h[AJ], a_w[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int p = m0 + arow + 32 * j;
    if (p < M) {
      int n_, h_, w_;
      pixel(p, n_, h_, w_);
      a_w[j] = w_ + d.pad_left;
      a_h[j] = h_ + d.pad_top;
      a_n[j] = n_;
    } else { a_n[j] = -1; a_h[j] = 0; a_w[j] = 0; }
  }
  const float* pa[AJ];
  const float* py[AJ];   // activation output at the same positions (fused g = dy*act'(y)); zero page if unused
  int inca[AJ];
  const float act_hi = (d.act == 2) ? 6.f : INFINITY;
#define BD_SETUP_RS(rs_)                                                                                \
  do {                                                                                                  \
    const int r_ = (rs_) / d.S, s_ = (rs_) - r_ * d.S;                                                  \
    _Pragma("unroll") for (int j = 0; j < AJ; ++j) {                                                    \
      const int th = a_h[j] - r_ * d.dilation, tw = a_w[j] - s_ * d.dilation;                           \
      int oh = th, ow = tw;                                                                             \
      bool ok = a_n[j] >= 0 && th >= 0 && tw >= 0;                                                      \
      if (d.stride > 1) {                                                                               \
        oh = th / d.stride; ow = tw / d.stride;                                                         \
        ok = ok && (oh * d.stride == th) && (ow * d.stride == tw);                                      \
      }                                                                                                 \
      ok = ok && oh < d.OH && ow < d.OW;                                                                \
      const size_t off_ = ((size_t)(a_n[j] * d.OH + oh) * d.OW + ow) * K + 4 * kq;                      \
      pa[j] = ok ? dy + off_ : lmh_zero_page;                                                           \
      py[j] = (ok && YACT) ? yact + off_ : lmh_zero_page;                                               \
      inca[j] = ok ? BK : 0;                                                                            \
    }                                                                                                   \
  } while (0)
  // B rows = input channels c; pointer walks k within a tap, then jumps to the next tap
  const float* pb[BJ];
  int incb[BJ];
  size_t tapb[BJ], tapx[BJ];
  const int rs0 = BD_TAP(0);
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int c = n0 + arow + 32 * j;
    const bool ok = c < C;
    pb[j] = ok ? w + ((size_t)rs0 * C + c) * K + 4 * kq : lmh_zero_page;
    incb[j] = ok ? BK : 0;
    tapb[j] = ok ? (size_t)C * K - K + BK : 0;   // end of this tap's k range -> start of the next tap
    tapx[j] = ok ? (size_t)C * K : 0;            // one whole tap (skipped taps of the stride-2 tap list)
  }
  const float* pks = kscale ? kscale + 4 * kq : lmh_zero_page;
  const int incks = kscale ? BK : 0;

  f32x4 ra[AJ], ry[AJ], rb[BJ], ks;
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  int ti = 0, rs = rs0, kc = 0;
  BD_SETUP_RS(rs0);
#define BD_ADVANCE()                                                              \
  do {                                                                            \
    if (++kc == KC) {                                                             \
      kc = 0; ++ti;                                                               \
      const int nrs_ = BD_TAP(ti);                                                \
      const int skip_ = nrs_ - rs - 1;                                            \
      rs = nrs_;                                                                  \
      BD_SETUP_RS(rs);                                                            \
      _Pragma("unroll") for (int j = 0; j < BJ; ++j) pb[j] += tapb[j] + (size_t)skip_ * tapx[j]; \
      pks -= (KC - 1) * incks;                                                    \
    } else {                                                                      \
      _Pragma("unroll") for (int j = 0; j < AJ; ++j) { pa[j] += inca[j]; if (YACT) py[j] += inca[j]; } \
      _Pragma("unroll") for (int j = 0; j < BJ; ++j) pb[j] += incb[j];            \
      pks += incks;                                                               \
    }                                                                             \
  } while (0)
#define BD_LOAD()                                                                                  \
  do {                                                                                             \
    ks = *reinterpret_cast<const f32x4*>(pks);                                                     \
    _Pragma("unroll") for (int j = 0; j < AJ; ++j) ra[j] = *reinterpret_cast<const f32x4*>(pa[j]); \
    if (YACT) { _Pragma("unroll") for (int j = 0; j < AJ; ++j) ry[j] = *reinterpret_cast<const f32x4*>(py[j]); } \
    _Pragma("unroll") for (int j = 0; j < BJ; ++j) rb[j] = *reinterpret_cast<const f32x4*>(pb[j]); \
  } while (0)
#define BD_STORE(buf_)                                                                             \
  do {                                                                                             \
    float* Ad = As + (buf_) * A_SZ;                                                                \
    float* Bd = Bs + (buf_) * B_SZ;                                                                \
    if (YACT) { _Pragma("unroll") for (int j = 0; j < AJ; ++j) ra[j] = act_mask(ra[j], ry[j], act_hi); } \
    if (kscale) { _Pragma("unroll") for (int j = 0; j < AJ; ++j) ra[j] *= ks; }                    \
    _Pragma("unroll") for (int j = 0; j < AJ; ++j)                                                 \
        *reinterpret_cast<f32x4*>(&Ad[(arow + 32 * j) * LDK + 4 * kq]) = ra[j];                    \
    _Pragma("unroll") for (int j = 0; j < BJ; ++j)                                                 \
        *reinterpret_cast<f32x4*>(&Bd[(arow + 32 * j) * LDK + 4 * kq]) = rb[j];                    \
  } while (0)
  BD_LOAD();
  BD_STORE(0);
  if (KT > 1) BD_ADVANCE();
  BD_LOAD();
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 2 < KT) BD_ADVANCE();
    mfma_stage_split<TM, TN, true, true, LDK, LDK, AJ + BJ, (YACT ? 2 : 1) * AJ + BJ + 1>(
        As + cur * A_SZ, Bs + cur * B_SZ, acc, wm * (BM / 2), wn * (BN / 2), lane,
        [&]() { BD_STORE(cur ^ 1); }, [&]() { BD_LOAD(); });
    __syncthreads();
  }
#undef BD_ADVANCE
#undef BD_LOAD
#undef BD_STORE
#undef BD_SETUP_RS
#undef BD_TAP

  // epilogue: every addend row and mask word of the tile is requested before the accumulator transpose and nothing but
  // stores follows the first store (see the forward kernel: loads and stores in flight together force vmcnt(0) waits)
  constexpr int CT = BN / 4, RSTEP = 256 / CT;
  constexpr int NR = BM / RSTEP, NRC = NR < 8 ? NR : 8, NCH = NR / NRC;
  const int c4 = tid % CT, r0 = tid / CT;
  const int col = n0 + 4 * c4;
  const bool col_ok = col < C;
  f32x4 ex[NCH][NRC];
  uint32_t xw[NCH][NRC];
  auto out_row = [&](int p) {      // GEMM row -> pixel index of dx / addend / mask words
    if (!par) return p;
    int n_, h_, w_;
    pixel(p, n_, h_, w_);
    return (n_ * d.H + h_) * d.W + w_;
  };
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int i = 0; i < NRC; ++i) {
      const int prow_ = m0 + r0 + (ch * NRC + i) * RSTEP;
      const bool ok = col_ok && prow_ < M;
      const int row = ok ? out_row(prow_) : 0;
      ex[ch][i] = (addend && ok) ? *reinterpret_cast<const f32x4*>(addend + (size_t)row * C + col) : f32x4{0.f, 0.f, 0.f, 0.f};
      xw[ch][i] = (xbits && ok) ? xbits[(size_t)row * (C >> 5) + (col >> 5)] : 0u;
    }
  acc_to_lds<BM, BN, TM, TN>(smem, acc, wm, wn, lane);
  __syncthreads();
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)          // all loads awaited in front of the first store (forward kernel)
#pragma unroll
    for (int i = 0; i < NRC; ++i) asm volatile("" ::"v"(ex[ch][i]), "v"(xw[ch][i]));
  if (col_ok) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
      for (int i = 0; i < NRC; ++i) {
        const int r = r0 + (ch * NRC + i) * RSTEP;
        if (m0 + r < M) {
          const int row = out_row(m0 + r);
          f32x4 v = *reinterpret_cast<const f32x4*>(&smem[r * LDC + 4 * c4]);
          v += ex[ch][i];
          // x is the (post-activation) output of the layer below: emitting dx * act'(x) hands that layer its
          // pre-activation gradient directly — no separate lmh_act_bwd pass over dx (2 KB of mask words per tile,
          // requested before the accumulator transpose: their latency runs under it)
          if (xbits) {
            const unsigned nib = xw[ch][i] >> (4 * (c4 & 7));
            v.x = (nib & 1u) ? v.x : 0.f;
            v.y = (nib & 2u) ? v.y : 0.f;
            v.z = (nib & 4u) ? v.z : 0.f;
            v.w = (nib & 8u) ? v.w : 0.f;
          }
          *reinterpret_cast<f32x4*>(dx + (size_t)row * C + col) = v;
        }
      }
    }
  }
}

// ============================================================================
// backward weight (fast path: C % 4 == 0, K % 4 == 0)
//   dw[rs, c, k] = sum_p x[pix(p,r,s), c] * dy[p, k];  reduction split over gridDim.z
//   GEMM M = C (per tap), N = K, Kg = P = N*OH*OW.  Both operands K-major ([pixel][channel]).
// ============================================================================
// GB: the R*S "taps" are independent GEMMs stacked in x / dy (the 16 transformed planes of a Winograd weight
// gradient, launched with a fake 4x4 filter, dilation 0, no padding): tap rs reads x + rs*P*C and dy + rs*P*K.
template <int BM, int BN, bool YACT, bool GB = false>   // YACT: B operand is dy * act'(yact) (fused activation backward)
__global__ void __launch_bounds__(256)
k_conv_bwd_weight(lmh_conv_desc d, const float* __restrict__ x, const float* __restrict__ dy,
                  float* __restrict__ out, int kt_per_split, lmh_fastdiv div_ow, lmh_fastdiv div_oh,
                  const float* __restrict__ yact, float* __restrict__ colsum_part, int tiles_x, int tiles_y,
                  int splits) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AJ = BM / 32, BJ = BN / 32;
  constexpr int A_SZ = BK * BM, B_SZ = BK * BN;
  constexpr int LDC = BN + 4;
  constexpr int SMEM = (2 * (A_SZ + B_SZ) > BM * LDC) ? 2 * (A_SZ + B_SZ) : BM * LDC;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  float* const As = smem;               // [2][BK][BM]
  float* const Bs = smem + 2 * A_SZ;    // [2][BK][BN]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int P = d.N * d.OH * d.OW, K = d.K, C = d.C;
  // 1-D grid, XCD-aware: the blocks of one XCD (bid % 8) cover a contiguous range of (split, k-tile,
  // tap x c-tile) ids with the SPLIT slowest, so the pixel range of a split — the x rows and dy rows every
  // one of its tiles re-reads — is fetched into that XCD's private L2 once (PMC before: 144 MB fetched
  // per launch for ~42 MB of operands, every XCD streaming all of x).
  const int lin = xcd_remap(blockIdx.x, tiles_x * tiles_y * splits);
  const int bz = lin / (tiles_x * tiles_y), rem = lin - bz * (tiles_x * tiles_y);
  const int tiles_c = (C + BM - 1) / BM;
  int by, bx;
  if (GB) {
    // stacked planes are independent GEMMs: a plane's tiles stay together, so ONE XCD streams that plane's x and dy
    // (PMC on the RPN Winograd weight gradient with the k-tile slowest: 604 MB fetched for 201 MB of operands —
    // every k-tile row, on its own XCD, re-read all 16 x planes)
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
  constexpr int AROW_T = BM / 4, AROW_STEP = 256 / AROW_T;
  constexpr int BROW_T = BN / 4, BROW_STEP = 256 / BROW_T;
  const int ax4 = tid % AROW_T, ak = tid / AROW_T;
  const int bx4 = tid % BROW_T, bk = tid / BROW_T;
  const bool a_col_ok = (m0 + 4 * ax4) < C, b_col_ok = (n0 + 4 * bx4) < K;
  const int dh0 = r * d.dilation - d.pad_top, dw0 = s * d.dilation - d.pad_left;
  // A rows of this thread = output pixels pa0 + AROW_STEP*j (+BK per stage), decoded with magic-number
  // division (branch-free: the whole stage stays one basic block for the MFMA/VMEM interleave)
  int pa0 = kt_begin * BK + ak;
  int bp = kt_begin * BK + bk;
  const float* xb = x + m0 + 4 * ax4 + (GB ? (size_t)rs * P * C : 0);
  const size_t dy_off0 = (size_t)bp * K + n0 + 4 * bx4 + (GB ? (size_t)rs * P * K : 0);
  const float* pdy = dy + dy_off0;
  const float* pyy = YACT ? yact + dy_off0 : lmh_zero_page;
  const size_t dy_row = (size_t)BROW_STEP * K, dy_stage = (size_t)BK * K;
  const float act_hi = (d.act == 2) ? 6.f : INFINITY;
  // per-channel sums of g (= dbeta / dbias) ride along in the blocks of the first (tap, c-tile) column
  const bool do_colsum = colsum_part != nullptr && bx == 0;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  f32x4 ra[AJ], rb[BJ], ryb[BJ];
#define BW_LOAD()                                                                                          \
  do {                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < AJ; ++j) {                                                       \
      const unsigned p = (unsigned)(pa0 + AROW_STEP * j);                                                  \
      const unsigned t = lmh_div(p, div_ow), ow = p - t * (unsigned)d.OW;                                  \
      const unsigned n = lmh_div(t, div_oh), oh = t - n * (unsigned)d.OH;                                  \
      const int ih = (int)oh * d.stride + dh0, iw = (int)ow * d.stride + dw0;                              \
      const bool ok = a_col_ok && n < (unsigned)d.N && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W; \
      const float* p_ = ok ? xb + ((size_t)((int)n * d.H + ih) * d.W + iw) * C : lmh_zero_page;            \
      ra[j] = *reinterpret_cast<const f32x4*>(p_);                                                         \
    }                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < BJ; ++j) {                                                       \
      const bool ok = b_col_ok && (bp + BROW_STEP * j) < P;                                                \
      const float* p_ = ok ? pdy + j * dy_row : lmh_zero_page;                                             \
      rb[j] = *reinterpret_cast<const f32x4*>(p_);                                                         \
      if (YACT) ryb[j] = *reinterpret_cast<const f32x4*>((ok ? pyy + j * dy_row : lmh_zero_page));         \
    }                                                                                                      \
  } while (0)
#define BW_ADVANCE() do { pa0 += BK; bp += BK; pdy += dy_stage; if (YACT) pyy += dy_stage; } while (0)
#define BW_STORE(buf_)                                                                                     \
  do {                                                                                                     \
    float* Ad = As + (buf_) * A_SZ;                                                                        \
    float* Bd = Bs + (buf_) * B_SZ;                                                                        \
    if (YACT) { _Pragma("unroll") for (int j = 0; j < BJ; ++j) rb[j] = act_mask(rb[j], ryb[j], act_hi); }  \
    if (do_colsum) { _Pragma("unroll") for (int j = 0; j < BJ; ++j) csum += rb[j]; }                       \
    _Pragma("unroll") for (int j = 0; j < AJ; ++j)                                                         \
        *reinterpret_cast<f32x4*>(&Ad[(ak + AROW_STEP * j) * BM + 4 * ax4]) = ra[j];                       \
    _Pragma("unroll") for (int j = 0; j < BJ; ++j)                                                         \
        *reinterpret_cast<f32x4*>(&Bd[(bk + BROW_STEP * j) * BN + 4 * bx4]) = rb[j];                       \
  } while (0)
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  BW_LOAD();          // tile 0 of this split
  BW_STORE(0);
  BW_ADVANCE();
  BW_LOAD();          // tile 1 (past the split's end the rows are still valid pixels, or the zero page)
  __syncthreads();
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (kt - kt_begin) & 1;
    BW_ADVANCE();
    const f32x4 csum_before = csum;
    mfma_stage_split<TM, TN, false, false, BM, BN, AJ + BJ, AJ + (YACT ? 2 : 1) * BJ>(
        As + cur * A_SZ, Bs + cur * B_SZ, acc, wm * (BM / 2), wn * (BN / 2), lane,
        [&]() { BW_STORE(cur ^ 1); }, [&]() { BW_LOAD(); });
    if (kt + 1 >= kt_end) csum = csum_before;   // the tile stored in the last stage belongs to the next split
    __syncthreads();
  }
#undef BW_ADVANCE
#undef BW_LOAD
#undef BW_STORE

  acc_to_lds<BM, BN, TM, TN>(smem, acc, wm, wn, lane);
  __syncthreads();
  float* o = out + (size_t)bz * ((size_t)d.R * d.S * C * K) + (size_t)rs * C * K;
  constexpr int CT = BN / 4, RSTEP = 256 / CT;
  const int c4 = tid % CT, r0 = tid / CT;
  const int col = n0 + 4 * c4;
  if (col < K) {
#pragma unroll 4
    for (int rr = r0; rr < BM; rr += RSTEP) {
      const int row = m0 + rr;
      if (row >= C) break;
      *reinterpret_cast<f32x4*>(o + (size_t)row * K + col) = *reinterpret_cast<const f32x4*>(&smem[rr * LDC + 4 * c4]);
    }
  }
  if (do_colsum) {   // fold the BROW_STEP row-groups of every column group in a fixed order (deterministic)
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem);          // [BROW_STEP][BROW_T]
    red[bk * BROW_T + bx4] = csum;
    __syncthreads();
    if (tid < BROW_T && (n0 + 4 * tid) < K) {
      f32x4 t = red[tid];
#pragma unroll
      for (int g2 = 1; g2 < BROW_STEP; ++g2) t += red[g2 * BROW_T + tid];
      *reinterpret_cast<f32x4*>(colsum_part + (size_t)bz * K + n0 + 4 * tid) = t;
    }
  }
}

// ============================================================================
// ResNet stem: 7x7 / stride 2, C = 3 -> K = 64 (slim resnet_v1 conv1, conv2d_same: pad 3, VALID), fused
// channel-mean subtraction (base_network.py:153-177), frozen BN and ReLU.
//
// Kg = 147 is far too shallow for the staged pipeline above (5 BK stages, each gathering 3-channel pixels
// element by element: the generic kernel ran at 42 TF/s).  Here a persistent block keeps the WHOLE weight
// matrix (147 -> 160 rows x 64) in LDS and, per tile of 8 x 16 output pixels, the 21 x 37 x 3 input patch
// (mean-subtracted, zero outside the image).  The im2col never exists: lane (pixel i, k) reads
// patch[pixbase(i) + koff(k)] with koff(k) = k + 90*(k / 21) — for a fully unrolled k loop that is a
// compile-time immediate off one of two per-lane base registers.  No global traffic and no barrier inside
// the 160-MFMA tile loop; the next tile's patch is prefetched into registers under it.  Measured: 232 -> 134 us
// (74 TF/s); the MFMA phase runs at 87 % of peak, the rest is the 134 MB output write (2.4 TB/s with every
// block storing in lockstep) which does not yet overlap the next tile's MFMAs.
// ============================================================================
#define STEM_TH 8
#define STEM_TW 16
#define STEM_PH ((STEM_TH - 1) * 2 + 7)   // 21
#define STEM_PW ((STEM_TW - 1) * 2 + 7)   // 37
#define STEM_ROWF (STEM_PW * 3)           // 111 floats per patch row
#define STEM_PATCH (STEM_PH * STEM_ROWF)  // 2331
#define STEM_KP 160                       // 147 padded to a multiple of 32 (zero weight rows)
#define STEM_NLD ((STEM_PATCH + 255) / 256)

template <int dbg>   // dbg: compile-time ablation switches (1 no stores, 2 10% of the MFMAs, 4 no patch fetch); product = 0
__global__ void __launch_bounds__(256)
k_conv_stem7x7s2(lmh_conv_desc d, const float* __restrict__ x, const float* __restrict__ w,
                 const float* __restrict__ scale, const float* __restrict__ shift,
                 const float* __restrict__ in_sub, float* __restrict__ y, int tiles_h, int tiles_w) {
  __shared__ __attribute__((aligned(16))) float Ws[STEM_KP * 64];
  __shared__ __attribute__((aligned(16))) float Ps[2][STEM_PATCH + 64 * 3 + 32];   // slack: k in [147,160) reads
  __shared__ __attribute__((aligned(16))) float Sc[64], Sh[64];                     // BN scale / shift (epilogue)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  {
    // the weight matrix goes to LDS with ALL of a thread's loads in flight together (ten float4 + scale / shift).  The
    // loop it replaces (`Ws[i] = i < 147 * 64 ? w[i] : 0`, one float per trip) compiled to load -> s_waitcnt vmcnt(0) ->
    // ds_write: forty serial L2 round trips per thread before the first tile, a fifth of the kernel (round 4; ISA check).
    constexpr int NV = STEM_KP * 64 / 4, NW = 147 * 64 / 4, NQ = (NV + 255) / 256;      // float4 slots: 2560 / 2352 real / 10 per thread
    f32x4 wv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = tid + 256 * q;
      wv[q] = *reinterpret_cast<const f32x4*>(w + 4 * min(i, NW - 1));
    }
    const float sc_ = (scale && tid < 64) ? scale[tid] : 1.f, sh_ = (shift && tid < 64) ? shift[tid] : 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = tid + 256 * q;
      if (i < NV) *reinterpret_cast<f32x4*>(&Ws[4 * i]) = (i < NW) ? wv[q] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (tid < 64) { Sc[tid] = sc_; Sh[tid] = sh_; }
  }
  const float act_lo = d.act ? 0.f : -INFINITY, act_hi = (d.act == 2) ? 6.f : INFINITY;   // branch-free activation
  for (int i = tid; i < 2 * (STEM_PATCH + 64 * 3 + 32); i += 256) (&Ps[0][0])[i] = 0.f;
  const float sub[3] = {in_sub ? in_sub[0] : 0.f, in_sub ? in_sub[1] : 0.f, in_sub ? in_sub[2] : 0.f};
  const int ntiles = d.N * tiles_h * tiles_w;
  // per-thread patch element slots (fixed across tiles): e = tid + 256*q -> (row, col3)
  int prow[STEM_NLD], pcol[STEM_NLD];
#pragma unroll
  for (int q = 0; q < STEM_NLD; ++q) {
    const int e = tid + 256 * q;
    prow[q] = e / STEM_ROWF;
    pcol[q] = e - prow[q] * STEM_ROWF;
  }
  float pre[STEM_NLD], psub[STEM_NLD];
#define STEM_FETCH(t_)                                                                       \
  do {                                                                                       \
    const int tw_ = (t_) % tiles_w, tq_ = (t_) / tiles_w;                                    \
    const int th_ = tq_ % tiles_h, n_ = tq_ / tiles_h;                                       \
    const int ih0 = th_ * STEM_TH * 2 - d.pad_top, iw0 = tw_ * STEM_TW * 2 - d.pad_left;     \
    _Pragma("unroll") for (int q = 0; q < STEM_NLD; ++q) {                                   \
      const int e = tid + 256 * q;                                                           \
      const int ih = ih0 + prow[q], iwc = iw0 * 3 + pcol[q];                                 \
      const bool ok = e < STEM_PATCH && (unsigned)ih < (unsigned)d.H && iwc >= 0 && iwc < d.W * 3; \
      const int ch = pcol[q] % 3;                                                            \
      /* unconditional load (the zero page when outside the image), used unconditionally: as `ok ? x[..] : 0` every   \
         one of the STEM_NLD loads sat behind a branch and a full s_waitcnt, i.e. the "prefetch" of the next patch was   \
         ten serial round trips BEFORE the MFMAs of this tile instead of one under them (round 4; ISA check) */         \
      const float* xp_ = ok ? x + ((size_t)n_ * d.H + ih) * d.W * 3 + iwc : lmh_zero_page;   \
      pre[q] = *xp_;                                  /* first use: STEM_COMMIT, after the MFMAs */ \
      psub[q] = ok ? sub[ch] : 0.f;                                                          \
    }                                                                                        \
  } while (0)
#define STEM_COMMIT(buf_)                                                                    \
  do {                                                                                       \
    _Pragma("unroll") for (int q = 0; q < STEM_NLD; ++q) {                                   \
      const int e = tid + 256 * q;                                                           \
      if (e < STEM_PATCH) Ps[buf_][e] = pre[q] - psub[q];                                    \
    }                                                                                        \
  } while (0)
  // this wave's 32 pixels: rows 2*wave, 2*wave+1 of the 8 x 16 tile
  const int py = 2 * wave + (l31 >> 4), px = l31 & 15;
  const int pixbase = py * 2 * STEM_ROWF + px * 6;
  int t = blockIdx.x;
  if (t < ntiles) STEM_FETCH(t);
  __syncthreads();                      // Ws / zeroed patches visible
  if (t < ntiles) STEM_COMMIT(0);
  __syncthreads();
  int buf = 0;
  for (; t < ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < ntiles && !(dbg & 4)) STEM_FETCH(tn);    // next patch: global -> registers, under the MFMAs below
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const float* P = &Ps[buf][pixbase + h];          // k even/odd split: lane half h takes k = 2s + h
    const float* P91 = &Ps[buf][pixbase + 91 * h];   // steps where 2s+1 starts a new filter row
    const float* Wl = &Ws[h * 64 + l31];
    // operands of k-pair group g+1 are read while the MFMAs of group g issue (4 k-pairs per group,
    // sched_barrier-pinned: left alone the scheduler puts every ds_read right before its MFMA)
    float fa[2][4], fb0[2][4], fb1[2][4];
#define STEM_LOAD(g_, set_)                                                              \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                        \
    const int k0 = 2 * (4 * (g_) + u);                                                   \
    const int koff0 = k0 + 90 * (k0 / 21);                                               \
    fa[set_][u] = (((k0 + 1) % 21) == 0) ? P91[koff0] : P[koff0];                        \
    fb0[set_][u] = Wl[k0 * 64];                                                          \
    fb1[set_][u] = Wl[k0 * 64 + 32];                                                     \
  }
    STEM_LOAD(0, 0)
#pragma unroll
    for (int g = 0; g < ((dbg & 2) ? 2 : STEM_KP / 8); ++g) {
      const int cur = g & 1;
      if (g + 1 < STEM_KP / 8) { STEM_LOAD(g + 1, cur ^ 1) }
      __builtin_amdgcn_sched_barrier(0);
      // transposed product D[channel][pixel]: a lane then owns 4 CONSECUTIVE channels of one pixel per
      // accumulator quad -> float4 stores (4x fewer store instructions than D[pixel][channel])
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb0[cur][u], fa[cur][u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb1[cur][u], fa[cur][u], acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef STEM_LOAD
    // next patch -> LDS BEFORE this tile's stores: the vmcnt wait for the prefetch loads then covers no store of
    // this tile (memory ops retire in order; the previous tile's stores finished under the MFMAs above)
    if (tn < ntiles) STEM_COMMIT(buf ^ 1);
    // epilogue: lane = pixel l31 of this wave; accumulator quad g = channels 8g + 4h .. +3 (+32 for acc1)
    const int tw_ = t % tiles_w, tq_ = t / tiles_w;
    const int th_ = tq_ % tiles_h, n_ = tq_ / tiles_h;
    const int oh = th_ * STEM_TH + py, ow = tw_ * STEM_TW + px;
    if (oh < d.OH && ow < d.OW) {
      float* o = y + (((size_t)n_ * d.OH + oh) * d.OW + ow) * 64 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int c0 = 8 * g + 4 * h + 32 * half;
          const f32x4 sc = *reinterpret_cast<const f32x4*>(&Sc[c0]);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(&Sh[c0]);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a_ = half ? acc1[4 * g + e] : acc0[4 * g + e];
            v[e] = fminf(fmaxf(a_ * sc[e] + sh[e], act_lo), act_hi);
          }
          if (!(dbg & 1)) *reinterpret_cast<f32x4*>(o + 8 * g + 32 * half) = v;
        }
      }
    }
    // raw barrier: __syncthreads() would also drain vmcnt(0), i.e. wait for this tile's output stores to be
    // acknowledged (PMC: 47 % of the wave cycles sat in that wait); only the LDS patch writes must be complete
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    buf ^= 1;
  }
#undef STEM_FETCH
#undef STEM_COMMIT
}
