// bf16x3 convolutions, round 6: fp32 ARITHMETIC on the bf16 matrix pipe (exact three-way split of every fp32 operand,
// six v_mfma_f32_32x32x16_bf16 per fragment pair, fp32 accumulate — conv_half.h explains the arithmetic and
// tests/test_gpu_x3.py pins it), as ONE software-pipelined instruction stream per wave.
//
// Why a second set of kernels.  The round-2 bf16x3 kernels (k_conv_*_h<3, ...>, conv_half.h) run "load -> split -> LDS ->
// barrier -> MFMA -> barrier" phase by phase and rely on a second resident block of the same CU to fill the matrix pipe
// while one block splits.  Measured at the start of round 6 (scripts/bench_conv.py, MI355X): the 2-4 GFLOP layers of the
// ResNet-50 step take 20-55 us against 5-12 us of bf16 matrix time — a layer with one tile per CU (block3 512->256: 256
// tiles, 16 stages) spends ~2300 cycles per stage for 768 cycles of MFMAs, because nothing overlaps inside a block.  They
// also predate the round-3/4 epilogues (activation bit masks, every load in front of the first store), so the bf16x3 step
// carried 41 extra k_act_bits / k_apply_act_bits launches.
//
// Here (the structure of conv_fast.h's mfma_stage_split, re-balanced for a 16x faster matrix pipe):
//   * LDS is double buffered (3 planes x (BM + BN) rows x 80 B per buffer: 120 KB at 128 x 128 — one block per CU, one
//     wave per SIMD) and the staging registers are TWO sets: while the MFMAs of tile t issue, the SAME wave splits tile
//     t+1 (set A) into the other LDS buffer and has the global loads of tile t+2 (set B) in flight, so a load has a whole
//     stage to land and the split's ~180 VALU + 24 ds_write_b64 per thread sit in the shadow of the stage's 48 MFMAs
//     (sched_group_barrier pins ~4 VALU + 0.5 DS write per MFMA; left alone the scheduler emits them phase by phase);
//   * ONE barrier per stage; all fragment reads of a stage (2 k-steps x 3 planes) are requested before its first MFMA;
//   * epilogues of the round-4 native kernels: residual / addend rows and mask words requested before the accumulator
//     transpose and awaited once in front of the first store; the forward writes the activation bit mask, the backward
//     data applies the mask of its input (no separate mask passes);
//   * the per-channel sums of g (dbeta / dbias) of the weight gradient come off the matrix pipe: three extra MFMAs per
//     k-step against an all-ones operand in the blocks of one tile column (exact: the pieces of an element add up to it).
// The sums are formed in the SAME order as in k_conv_*_h<3, ...> (k-step, product, tile), so the two sets of kernels are
// bit-identical (tests/test_gpu_x3.py::test_pipelined_bf16x3_kernels_equal_the_round2_kernels).
#pragma once
#include <type_traits>
#include "conv_half.h"

typedef HT<3>::T x3_t;
typedef HT<3>::V8 x3_v8;

// MFMA : other-instruction interleave of one half stage.  NM MFMAs; the groups below are requests to the scheduler, an
// instruction class that runs out is simply skipped.
template <int NM, int NV, int NDS>
__device__ __forceinline__ void x3_interleave() {
  // NV VALU per MFMA and a DS write every (NM / NDS) MFMAs
  constexpr int DSTEP = NDS > 0 ? ((NM / NDS) > 0 ? (NM / NDS) : 1) : NM + 1;
#pragma unroll
  for (int i = 0; i < NM; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     // MFMA
    __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);                    // VALU
    if ((i % DSTEP) == DSTEP - 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
  }
}

// One BK = 32 stage.  do_loads: global loads of tile t+2; do_w1 / do_w2: split + LDS writes of the two operand tiles of
// tile t+1.  COL: also accumulate ones x B into ccol (per-column sums of the B tile).
template <int TM, int TN, bool COL, int NV1, int ND1, int NV2, int ND2, class LoadF, class W1F, class W2F>
__device__ __forceinline__ void x3_stage(const x3_t* __restrict__ As, const x3_t* __restrict__ Bs, int a_pl, int b_pl,
                                         f32x16 (&acc)[TM][TN], f32x16 (&ccol)[TN], int a_off, int b_off, int lane,
                                         LoadF&& do_loads, W1F&& do_w1, W2F&& do_w2) {
  constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first (conv_half.h)
  constexpr int NM = 6 * TM * TN + (COL ? 3 * TN : 0);
  const int l31 = lane & 31, kh = 8 * (lane >> 5);
  x3_v8 a[2][3][TM], b[2][3][TN];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int t = 0; t < TM; ++t)
        a[s][p][t] = *reinterpret_cast<const x3_v8*>(&As[p * a_pl + (a_off + t * 32 + l31) * LDH + s * 16 + kh]);
#pragma unroll
      for (int t = 0; t < TN; ++t)
        b[s][p][t] = *reinterpret_cast<const x3_v8*>(&Bs[p * b_pl + (b_off + t * 32 + l31) * LDH + s * 16 + kh]);
    }
  do_loads();          // tile t+2: in flight for this whole stage (first use: the split of the NEXT stage)
  x3_v8 ones;
  if (COL) {
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (x3_t)1.0f;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = HT<3>::mfma(a[s][PA[q]][tm], b[s][PB[q]][tn], acc[tm][tn]);
    if (COL) {
#pragma unroll
      for (int p = 2; p >= 0; --p)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) ccol[tn] = HT<3>::mfma(ones, b[s][p][tn], ccol[tn]);
    }
    if (s == 0) {
      do_w1();
      x3_interleave<NM, NV1, ND1>();
    } else {
      do_w2();
      x3_interleave<NM, NV2, ND2>();
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Two blocks of the phase-by-phase kernels share a CU and start together: left alone they split together and multiply
// together.  The block in the CU's SECOND LDS slot (LDS_BASE != 0: scripts/probes/lds_base_probe.hip) sleeps `units` x 64
// cycles before its first stage so that one block's split runs under the other's MFMAs.
__device__ __forceinline__ void x3_stagger(int units) {
  if (units > 0 && (__builtin_amdgcn_s_getreg((7 << 11) | 6) & 0xff) != 0)
    for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(1);
}

// LDS floats: two buffers x three planes x (BM + BN) rows of LDH halfs, or the fp32 epilogue tile
template <int BM, int BN, int PIPE>
struct x3_smem {
  static constexpr int NB = PIPE ? 2 : 1;
  static constexpr int stage = (NB * 3 * (BM + BN) * LDH + 1) / 2, epi = BM * (BN + 4);
  static constexpr int floats = stage > epi ? stage : epi;
};

// The pipeline around x3_stage.  The caller defines (S = std::integral_constant<int, 0 | 1>: staging register set)
//   load(S)            global -> set S at the current pointers (branch-free: the stage must stay ONE basic block for the
//                      MFMA / VALU / DS interleave)
//   advance()          pointers -> next tile (past the last tile they stay: the loads re-read a valid tile and the data is
//                      never multiplied — the tile count is even); called between the stages
//   store_a(buf, S) / store_b(buf, S)   set S -> LDS buffer buf (split, transposes)
//   stage(buf, SL, SW) x3_stage on LDS buffer buf with loads into set SL and the writes of set SW into buffer buf ^ 1
// PIPE == 0: the round-2 schedule (ONE LDS buffer, two barriers per stage, 61 KB at 128 x 128: two blocks share a CU and
// one block's split runs under the other's MFMAs); PIPE == 1: the software pipeline below (one block per CU).
#define X3_RUN(KT_)                                                                                 \
  do {                                                                                              \
    if constexpr (PIPE == 0) {                                                                      \
      typedef std::integral_constant<int, 0> S0;                                                    \
      load(S0());                                                                                   \
      advance();                                                                                    \
      x3_stagger(stagger & 255);                                                                    \
      for (int kt_ = 0; kt_ < (KT_); ++kt_) {                                                       \
        store_a(0, S0());                                                                           \
        store_b(0, S0());                                                                           \
        load(S0());                                                                                 \
        advance();                                                                                  \
        __syncthreads();                                                                            \
        mma(0);                                                                                     \
        __syncthreads();                                                                            \
      }                                                                                             \
    } else {                                                                                        \
      X3_PIPELINE(KT_);                                                                             \
    }                                                                                               \
  } while (0)

#define X3_PIPELINE(KT_)                                                                            \
  do {                                                                                              \
    typedef std::integral_constant<int, 0> S0;                                                      \
    typedef std::integral_constant<int, 1> S1;                                                      \
    load(S0());                      /* tile 0 */                                                   \
    advance();                                                                                      \
    load(S1());                      /* tile 1 */                                                   \
    advance();                                                                                      \
    store_a(0, S0());                                                                               \
    store_b(0, S0());                                                                               \
    __syncthreads();                                                                                \
    /* stages in PAIRS (compile-time register-set / buffer indices) with a single loop exit: a mid-loop break made the \
       compiler shuffle all 64 accumulator registers (v_accvgpr_mov) at the merge points of every stage.  The tile \
       count must be EVEN (x3_ok_* on the host; odd counts stay with the round-2 kernels). */             \
    for (int kt_ = 0; kt_ < (KT_); kt_ += 2) {                                                      \
      stage(0, S0(), S1());          /* MFMAs of tile kt_, tile kt_+1 -> buffer 1, tile kt_+2 -> set 0 */ \
      advance();                                                                                    \
      __syncthreads();                                                                              \
      stage(1, S1(), S0());                                                                         \
      advance();                                                                                    \
      __syncthreads();                                                                              \
    }                                                                                               \
  } while (0)

// ============================================================================
// forward:  y[p,k] = act( sum_{r,s,c} x[pix(p,r,s),c] * w[r,s,c,k] * scale[k] + shift[k] + res[p,k] )   (+ bit mask)
//   needs C % 32 == 0, K % 4 == 0.  A: gather (K-contiguous).  B: HWIO rows (K-major) -> transposed in registers.
//   GB: `gbatch` independent problems of one shape stacked in x / w / y (the transformed-domain GEMMs of a Winograd
//   convolution), one grid, plane index slowest.
// ============================================================================
template <int BM, int BN, bool GB, int PIPE>
__global__ void __launch_bounds__(256, (PIPE && (BM + BN) > 128) ? 1 : 2)
k_x3_fwd(lmh_conv_desc d, const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
         const float* __restrict__ shift, const float* __restrict__ residual, float* __restrict__ y, int gbatch,
         uint32_t* __restrict__ act_bits, int stagger, lmh_fastdiv dvw, lmh_fastdiv dvh) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AJ = BM / 32;
  constexpr int A_SZ = BM * LDH, B_SZ = BN * LDH, A_BUF = 3 * A_SZ, B_BUF = 3 * B_SZ;
  constexpr int LDC = BN + 4;
  __shared__ __attribute__((aligned(16))) float smem[x3_smem<BM, BN, PIPE>::floats];
  x3_t* const As = reinterpret_cast<x3_t*>(smem);       // [2][3][BM][LDH]
  x3_t* const Bs = As + (PIPE ? 2 : 1) * A_BUF;                      // [2][3][BN][LDH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
      const int t = (int)lmh_div((unsigned)p, dvw), ow = p - t * d.OW;      // (magic-number division: a plain / and % pair
      a_n[j] = (int)lmh_div((unsigned)t, dvh);                                //  is ~60 instructions per row in front of the first load)
      a_ih0[j] = (t - a_n[j] * d.OH) * d.stride - d.pad_top;
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
  constexpr bool B_ALL = BN / 4 >= 32;                 // every thread stages a block of B (no exec-masked region in the stage)
  const bool b_act = cq < BN / 4;
  const bool b_ok = b_act && (n0 + 4 * cq) < K;
  const float* pb = b_ok ? w + (size_t)(4 * kq) * K + n0 + 4 * cq : lmh_zero_page;
  const size_t rowb = b_ok ? (size_t)K : 0, incb = b_ok ? (size_t)BK * K : 0;

  // timing decomposition (probe builds only: LMH_PROBES=1 bash build.sh; scripts/r6_x3_decomp.py): bits 8.. of `stagger`
  // switch parts of the kernel off — 1 split + LDS writes of B, 2 loads of B, 4 split + writes of A, 8 loads of A, 16 the
  // MFMA phase, 32 the epilogue's residual loads and stores.  Wrong results; what each part costs inside the launch.
#ifdef LMH_PROBES
  const int dbg = stagger >> 8;
#else
  constexpr int dbg = 0;
#endif
  f32x4 ra[2][AJ], rb[2][4];
  f32x16 acc[TM][TN], cdummy[TN];
  zero_acc<TM, TN>(acc);
  int rs = 0, cc = 0, ptile = 0;
  setup_rs(0);
  auto load = [&](auto S) {
    constexpr int s_ = decltype(S)::value;
#pragma unroll
    for (int j = 0; j < AJ; ++j) ra[s_][j] = *reinterpret_cast<const f32x4*>((dbg & 8) ? lmh_zero_page : pa[j]);
    if ((B_ALL || b_act) && !(dbg & 2)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rb[s_][i] = *reinterpret_cast<const f32x4*>(pb + i * rowb);
    }
  };
  auto advance = [&]() {       // (the tile count is even — host check — so a tile past the last one is never multiplied)
    if (ptile + 1 < KT) {
      ++ptile;
      if (++cc == CC) { cc = 0; ++rs; setup_rs(rs); }
      else {
#pragma unroll
        for (int j = 0; j < AJ; ++j) pa[j] += inca[j];
      }
      pb += incb;
    }
  };
  auto store_a = [&](int buf, auto S) {
    constexpr int s_ = decltype(S)::value;
    x3_t* Ad = As + buf * A_BUF;
    if (!(dbg & 4))
#pragma unroll
      for (int j = 0; j < AJ; ++j) st_kc<3>(Ad, A_SZ, arow + 32 * j, kq, ra[s_][j]);
  };
  auto store_b = [&](int buf, auto S) {
    constexpr int s_ = decltype(S)::value;
    if ((B_ALL || b_act) && !(dbg & 1)) st_km<3>(Bs + buf * B_BUF, B_SZ, 4 * cq, kq, rb[s_]);
  };
  auto stage = [&](int buf, auto SL, auto SW) {
    x3_stage<TM, TN, false, 5, AJ * 3, 5, 12>(
        As + buf * A_BUF, Bs + buf * B_BUF, A_SZ, B_SZ, acc, cdummy, wm * (BM / 2), wn * (BN / 2), lane,
        [&]() { load(SL); }, [&]() { store_a(buf ^ 1, SW); }, [&]() { store_b(buf ^ 1, SW); });
  };
  auto mma = [&](int buf) {
    if (dbg & 16) return;
    x3_stage<TM, TN, false, 0, 0, 0, 0>(As + buf * A_BUF, Bs + buf * B_BUF, A_SZ, B_SZ, acc, cdummy, wm * (BM / 2),
                                        wn * (BN / 2), lane, []() {}, []() {}, []() {});
  };
  X3_RUN(KT);

  // ---- epilogue through LDS (fp32): k_conv_fwd's — every load in front of the first store
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
      ex[ch][i] = (residual && col_ok && row < M && !(dbg & 32)) ? *reinterpret_cast<const f32x4*>(residual + (size_t)row * K + col)
                                                  : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  acc_to_lds<BM, BN, TM, TN>(smem, acc, wm, wn, lane);
  __syncthreads();
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
        if (row < M && !(dbg & 32)) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&smem[r * LDC + 4 * c4]);
          v = v * sc + sh;                 // (k_conv_fwd_h's expression: bit-identical results)
          if (residual) v += ex[ch][i];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], act_lo), act_hi);
          *reinterpret_cast<f32x4*>(y + (size_t)row * K + col) = v;
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
// Round 6, second session: PRE-SPLIT WEIGHTS.  The timing decomposition of k_x3_fwd (scripts/r6_x3_decomp.py, probe build)
// says the phases of a launch add up instead of overlapping — block3 256->1024: 41 us = 7 (empty launch) + 12 (loads + splits)
// + 16 (MFMA phases) + 11-15 (epilogue) — and that the staging cost is the SPLIT (VALU + LDS writes), not the loads: without the
// split of B 35.5 us, without B altogether 31.8.  B is the weight matrix: every one of the 64 row tiles of a launch splits the
// same 32 x 128 slab again, every stage.  Here the weights are split ONCE per step (k_x3_split_w, all layers in one launch) into
// the MFMA fragment order, and a wave loads its B fragments straight from global memory (L2-resident: a layer's planes are
// 1.5 MB) into registers one stage ahead: no LDS traffic, no VALU work and no barrier dependency for B at all.
//   W3 layout: [stage of 32 GEMM-k][plane 0..2][column group of 32][k-step 0..1][lane 0..63] x 16 bytes (8 bf16):
//   lane = 32 * (k-half) + (column in the group) — exactly what lane holds as the B operand of v_mfma_f32_32x32x16_bf16 — so one
//   wave instruction is one contiguous 1 KB.  The pieces are those of split3 (same bits as the in-kernel split: the results
//   of k_x3_fwd_ws are bit-identical to k_x3_fwd's).
// ============================================================================
__host__ __device__ __forceinline__ size_t x3_w3_stage(int ncols) { return (size_t)3 * (ncols >> 5) * 2 * 64; }   // uint4 per stage

// FWD: w is [RS * C][K] (HWIO): GEMM-k = (tap, c), columns = output channels.  BWD (backward data): columns = input
// channels, GEMM-k = (tap, k): element (kk = tap * K + k, n = c) = w[(tap * C + c) * K + k].
template <bool FWD>
__device__ __forceinline__ void x3_split_w_body(const float* __restrict__ w, int RS, int C, int K, uint4* __restrict__ out,
                                                int64_t idx) {
  const int N = FWD ? K : C, NG = N >> 5;
  const int T = FWD ? (RS * C) >> 5 : RS * (K >> 5);
  if (idx >= (int64_t)T * NG * 128) return;
  const int lane = (int)(idx & 63), s = (int)((idx >> 6) & 1);
  const int64_t r = idx >> 7;
  const int ng = (int)(r % NG), t = (int)(r / NG);
  const int n = 32 * ng + (lane & 31), k0 = 16 * s + 8 * (lane >> 5);
  uint32_t h[3][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v;
    if (FWD) v = w[(size_t)(32 * t + k0 + e) * K + n];
    else {
      const int KC = K >> 5, tap = t / KC, kc = t - tap * KC;
      v = w[((size_t)tap * C + n) * K + 32 * kc + k0 + e];
    }
    split3(v, h[0][e], h[1][e], h[2][e]);
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    uint4 o;
    o.x = pack_hi(h[p][0], h[p][1]); o.y = pack_hi(h[p][2], h[p][3]);
    o.z = pack_hi(h[p][4], h[p][5]); o.w = pack_hi(h[p][6], h[p][7]);
    out[(((size_t)(t * 3 + p) * NG + ng) * 2 + s) * 64 + lane] = o;
  }
}

#define X3_SPLIT_MAX 96
struct x3_split_batch {
  const float* w[X3_SPLIT_MAX];
  uint4* out[X3_SPLIT_MAX];
  int32_t RS[X3_SPLIT_MAX], C[X3_SPLIT_MAX], K[X3_SPLIT_MAX];
  int32_t first_block[X3_SPLIT_MAX + 1];
  int32_t n, fwd;
};
__global__ void __launch_bounds__(256)
k_x3_split_w(x3_split_batch b) {
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.first_block[j + 1]) ++j;
  const int64_t idx = (int64_t)((int)blockIdx.x - b.first_block[j]) * 256 + threadIdx.x;
  if (b.fwd) x3_split_w_body<true>(b.w[j], b.RS[j], b.C[j], b.K[j], b.out[j], idx);
  else x3_split_w_body<false>(b.w[j], b.RS[j], b.C[j], b.K[j], b.out[j], idx);
}

// the MFMA phase of one BK = 32 stage with the B fragments in registers (same order of products as x3_stage).  ONE register
// set for B: the fragments of k-step s are re-loaded for the NEXT stage right behind the MFMAs that read them (a second
// set cost 48 more VGPRs at 128 x 128 and spilled); a load then has the rest of this stage and the split phase of the next to land.
template <int TM, int TN, class ReloadF>
__device__ __forceinline__ void x3_mma_rb(const x3_t* __restrict__ As, int a_pl, f32x16 (&acc)[TM][TN], int a_off, int lane,
                                          x3_v8 (&b)[2][3][TN], ReloadF&& reload) {
  constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first (conv_half.h)
  const int l31 = lane & 31, kh = 8 * (lane >> 5);
  x3_v8 a[2][3][TM];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < TM; ++t)
        a[s][p][t] = *reinterpret_cast<const x3_v8*>(&As[p * a_pl + (a_off + t * 32 + l31) * LDH + s * 16 + kh]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = HT<3>::mfma(a[s][PA[q]][tm], b[s][PB[q]][tn], acc[tm][tn]);
    __builtin_amdgcn_sched_barrier(0);
    reload(s);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// forward with pre-split weights: k_x3_fwd's gather, split of A, epilogue; B as above.  Needs K % 32 == 0 besides C % 32 == 0.
template <int BM, int BN, bool GB>
__global__ void __launch_bounds__(256, 2)
k_x3_fwd_ws(lmh_conv_desc d, const float* __restrict__ x, const uint4* __restrict__ w3, const float* __restrict__ scale,
         const float* __restrict__ shift, const float* __restrict__ residual, float* __restrict__ y, int gbatch,
         uint32_t* __restrict__ act_bits, int stagger, lmh_fastdiv dvw, lmh_fastdiv dvh) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AJ = BM / 32;
  constexpr int A_SZ = BM * LDH, A_BUF = 3 * A_SZ;
  constexpr int LDC = BN + 4;
  __shared__ __attribute__((aligned(16))) float smem[(3 * BM * LDH / 2) > BM * (BN + 4) ? (3 * BM * LDH / 2) : BM * (BN + 4)];   // A planes, then the fp32 epilogue tile
  x3_t* const As = reinterpret_cast<x3_t*>(smem);       // [3][BM][LDH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = d.N * d.OH * d.OW, K = d.K, C = d.C;
  const int tiles_n = (K + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n * (GB ? gbatch : 1));
  if (GB) {
    const int gi = tile / (tiles_m * tiles_n);
    tile -= gi * (tiles_m * tiles_n);
    x += (size_t)gi * M * C;
    w3 += (size_t)gi * d.R * d.S * (C / BK) * x3_w3_stage(K);
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
      const int t = (int)lmh_div((unsigned)p, dvw), ow = p - t * d.OW;      // (magic-number division: a plain / and % pair
      a_n[j] = (int)lmh_div((unsigned)t, dvh);                                //  is ~60 instructions per row in front of the first load)
      a_ih0[j] = (t - a_n[j] * d.OH) * d.stride - d.pad_top;
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
  // ---- B: pre-split fragments straight from global memory (x3_w3_*): this wave's TN column groups of 32
  const int NG = K >> 5;
  int bgrp[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int g_ = (n0 + wn * (BN / 2)) / 32 + t;
    bgrp[t] = g_ < NG ? g_ : 0;                        // (columns past K are never stored)
  }
  const size_t w3_stage = x3_w3_stage(K);              // uint4 per stage
  f32x4 ra[1][AJ];
  x3_v8 bq[2][3][TN];
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  int rs = 0, cc = 0, ptile = 0;
  setup_rs(0);
  auto load = [&](auto S) {
    constexpr int s_ = decltype(S)::value;
#pragma unroll
    for (int j = 0; j < AJ; ++j) ra[s_][j] = *reinterpret_cast<const f32x4*>(pa[j]);
  };
  auto load_b = [&](int s, int t_) {          // k-step s of stage t_ (past the end: the last stage again, never multiplied)
    const uint4* base = w3 + (size_t)(t_ < KT ? t_ : KT - 1) * w3_stage;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const uint4 v = base[((size_t)(p * NG + bgrp[t]) * 2 + s) * 64 + lane];
        bq[s][p][t] = __builtin_bit_cast(x3_v8, v);
      }
  };
  auto advance = [&]() {       // (the tile count is even — host check — so a tile past the last one is never multiplied)
    if (ptile + 1 < KT) {
      ++ptile;
      if (++cc == CC) { cc = 0; ++rs; setup_rs(rs); }
      else {
#pragma unroll
        for (int j = 0; j < AJ; ++j) pa[j] += inca[j];
      }
    }
  };
  auto store_a = [&](int buf, auto S) {
    constexpr int s_ = decltype(S)::value;
    x3_t* Ad = As + buf * A_BUF;
#pragma unroll
    for (int j = 0; j < AJ; ++j) st_kc<3>(Ad, A_SZ, arow + 32 * j, kq, ra[s_][j]);
  };
  typedef std::integral_constant<int, 0> S0;
  load(S0());                        // A tile 0 -> registers
  advance();
  load_b(0, 0);
  load_b(1, 0);
  for (int kt = 0; kt < KT; ++kt) {
    store_a(0, S0());
    load(S0());                      // A tile kt + 1 (past the end: a valid tile that is never multiplied)
    advance();
    __syncthreads();
    x3_mma_rb<TM, TN>(As, A_SZ, acc, wm * (BM / 2), lane, bq, [&](int s) { load_b(s, kt + 1); });
    __syncthreads();
  }

  // ---- epilogue through LDS (fp32): k_conv_fwd's — every load in front of the first store
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
          v = v * sc + sh;                 // (k_conv_fwd_h's expression: bit-identical results)
          if (residual) v += ex[ch][i];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], act_lo), act_hi);
          *reinterpret_cast<f32x4*>(y + (size_t)row * K + col) = v;
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
// backward data:  dx[p,c] = sum_{r,s,k} dy[opix(p,r,s),k] * kscale[k] * w[r,s,c,k]  (+ addend) (x mask of x)
//   needs K % 32 == 0, C % 4 == 0.  A: dy gather (K-contiguous).  B: w[rs][c][k] rows (K-contiguous).
// ============================================================================
template <int BM, int BN, int PIPE>
__global__ void __launch_bounds__(256, (PIPE && (BM + BN) > 128) ? 1 : 2)
k_x3_bwd_data(lmh_conv_desc d, const float* __restrict__ dy, const float* __restrict__ w,
              const float* __restrict__ kscale, const float* __restrict__ addend, const uint32_t* __restrict__ xbits,
              float* __restrict__ dx, int stagger, lmh_fastdiv dvw, lmh_fastdiv dvh) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AJ = BM / 32, BJ = BN / 32;
  constexpr int A_SZ = BM * LDH, B_SZ = BN * LDH, A_BUF = 3 * A_SZ, B_BUF = 3 * B_SZ;
  constexpr int LDC = BN + 4;
  __shared__ __attribute__((aligned(16))) float smem[x3_smem<BM, BN, PIPE>::floats];
  x3_t* const As = reinterpret_cast<x3_t*>(smem);
  x3_t* const Bs = As + (PIPE ? 2 : 1) * A_BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
      const int t = (int)lmh_div((unsigned)p, dvw);
      a_w[j] = p - t * d.W + d.pad_left;
      a_n[j] = (int)lmh_div((unsigned)t, dvh);
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
  f32x4 ra[2][AJ], rb[2][BJ], ks[2];
  f32x16 acc[TM][TN], cdummy[TN];
  zero_acc<TM, TN>(acc);
  int rs = 0, kc = 0, ptile = 0;
  setup_rs(0);
  auto load = [&](auto S) {
    constexpr int s_ = decltype(S)::value;
    ks[s_] = *reinterpret_cast<const f32x4*>(pks);
#pragma unroll
    for (int j = 0; j < AJ; ++j) ra[s_][j] = *reinterpret_cast<const f32x4*>(pa[j]);
#pragma unroll
    for (int j = 0; j < BJ; ++j) rb[s_][j] = *reinterpret_cast<const f32x4*>(pb[j]);
  };
  auto advance = [&]() {
    if (ptile + 1 < KT) {       // (even tile count — host check: a tile past the last one is never multiplied)
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
    }
  };
  auto store_a = [&](int buf, auto S) {
    constexpr int s_ = decltype(S)::value;
    x3_t* Ad = As + buf * A_BUF;
    // (k_conv_bwd_data_h multiplies by kscale * gscale with gscale = 1 for bf16x3: the same product)
    const f32x4 m = kscale ? ks[s_] * 1.f : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int j = 0; j < AJ; ++j) st_kc<3>(Ad, A_SZ, arow + 32 * j, kq, ra[s_][j] * m);
  };
  auto store_b = [&](int buf, auto S) {
    constexpr int s_ = decltype(S)::value;
    x3_t* Bd = Bs + buf * B_BUF;
#pragma unroll
    for (int j = 0; j < BJ; ++j) st_kc<3>(Bd, B_SZ, arow + 32 * j, kq, rb[s_][j]);
  };
  auto stage = [&](int buf, auto SL, auto SW) {
    x3_stage<TM, TN, false, 5, AJ * 3, 5, BJ * 3>(
        As + buf * A_BUF, Bs + buf * B_BUF, A_SZ, B_SZ, acc, cdummy, wm * (BM / 2), wn * (BN / 2), lane,
        [&]() { load(SL); }, [&]() { store_a(buf ^ 1, SW); }, [&]() { store_b(buf ^ 1, SW); });
  };
  auto mma = [&](int buf) {
    x3_stage<TM, TN, false, 0, 0, 0, 0>(As + buf * A_BUF, Bs + buf * B_BUF, A_SZ, B_SZ, acc, cdummy, wm * (BM / 2),
                                        wn * (BN / 2), lane, []() {}, []() {}, []() {});
  };
  X3_RUN(KT);

  // epilogue (k_conv_bwd_data's): addend rows and mask words requested before the accumulator transpose
  constexpr int CT = BN / 4, RSTEP = 256 / CT;
  constexpr int NR = BM / RSTEP, NRC = NR < 8 ? NR : 8, NCH = NR / NRC;
  const int c4 = tid % CT, r0 = tid / CT;
  const int col = n0 + 4 * c4;
  const bool col_ok = col < C;
  f32x4 ex[NCH][NRC];
  uint32_t xw[NCH][NRC];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int i = 0; i < NRC; ++i) {
      const int row = m0 + r0 + (ch * NRC + i) * RSTEP;
      const bool ok = col_ok && row < M;
      ex[ch][i] = (addend && ok) ? *reinterpret_cast<const f32x4*>(addend + (size_t)row * C + col) : f32x4{0.f, 0.f, 0.f, 0.f};
      xw[ch][i] = (xbits && ok) ? xbits[(size_t)row * (C >> 5) + (col >> 5)] : 0u;
    }
  acc_to_lds<BM, BN, TM, TN>(smem, acc, wm, wn, lane);
  __syncthreads();
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int i = 0; i < NRC; ++i) asm volatile("" ::"v"(ex[ch][i]), "v"(xw[ch][i]));
  if (col_ok) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
      for (int i = 0; i < NRC; ++i) {
        const int r = r0 + (ch * NRC + i) * RSTEP;
        const int row = m0 + r;
        if (row < M) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&smem[r * LDC + 4 * c4]) * 1.f;
          if (addend) v += ex[ch][i];
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

// backward data with pre-split weights (backward arrangement: columns = input channels): k_x3_bwd_data's gather, kscale
// product and split of dy, epilogue; B fragments as in k_x3_fwd_ws.  Needs C % 32 == 0 besides K % 32 == 0.  Bit-identical.
template <int BM, int BN>
__global__ void __launch_bounds__(256, 2)
k_x3_bwd_data_ws(lmh_conv_desc d, const float* __restrict__ dy, const uint4* __restrict__ w3,
              const float* __restrict__ kscale, const float* __restrict__ addend, const uint32_t* __restrict__ xbits,
              float* __restrict__ dx, int stagger, lmh_fastdiv dvw, lmh_fastdiv dvh) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AJ = BM / 32;
  constexpr int A_SZ = BM * LDH, A_BUF = 3 * A_SZ;
  constexpr int LDC = BN + 4;
  __shared__ __attribute__((aligned(16))) float smem[(3 * BM * LDH / 2) > BM * (BN + 4) ? (3 * BM * LDH / 2) : BM * (BN + 4)];
  x3_t* const As = reinterpret_cast<x3_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
      const int t = (int)lmh_div((unsigned)p, dvw);
      a_w[j] = p - t * d.W + d.pad_left;
      a_n[j] = (int)lmh_div((unsigned)t, dvh);
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
  // ---- B: pre-split fragments (columns = input channels) straight from global memory
  const int NG = C >> 5;
  int bgrp[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int g_ = (n0 + wn * (BN / 2)) / 32 + t;
    bgrp[t] = g_ < NG ? g_ : 0;                        // (columns past C are never stored)
  }
  const size_t w3_stage = x3_w3_stage(C);
  const float* pks = kscale ? kscale + 4 * kq : lmh_zero_page;
  const int incks = kscale ? BK : 0;
  f32x4 ra[1][AJ], ks[1];
  x3_v8 bq[2][3][TN];
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  int rs = 0, kc = 0, ptile = 0;
  setup_rs(0);
  auto load = [&](auto S) {
    constexpr int s_ = decltype(S)::value;
    ks[s_] = *reinterpret_cast<const f32x4*>(pks);
#pragma unroll
    for (int j = 0; j < AJ; ++j) ra[s_][j] = *reinterpret_cast<const f32x4*>(pa[j]);
  };
  auto load_b = [&](int s, int t_) {
    const uint4* base = w3 + (size_t)(t_ < KT ? t_ : KT - 1) * w3_stage;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const uint4 v = base[((size_t)(p * NG + bgrp[t]) * 2 + s) * 64 + lane];
        bq[s][p][t] = __builtin_bit_cast(x3_v8, v);
      }
  };
  auto advance = [&]() {
    if (ptile + 1 < KT) {       // (even tile count — host check: a tile past the last one is never multiplied)
      ++ptile;
      if (++kc == KC) {
        kc = 0; ++rs;
        setup_rs(rs);
        pks -= (KC - 1) * incks;
      } else {
#pragma unroll
        for (int j = 0; j < AJ; ++j) pa[j] += inca[j];
        pks += incks;
      }
    }
  };
  auto store_a = [&](int buf, auto S) {
    constexpr int s_ = decltype(S)::value;
    x3_t* Ad = As + buf * A_BUF;
    // (k_conv_bwd_data_h multiplies by kscale * gscale with gscale = 1 for bf16x3: the same product)
    const f32x4 m = kscale ? ks[s_] * 1.f : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int j = 0; j < AJ; ++j) st_kc<3>(Ad, A_SZ, arow + 32 * j, kq, ra[s_][j] * m);
  };
  typedef std::integral_constant<int, 0> S0;
  load(S0());
  advance();
  load_b(0, 0);
  load_b(1, 0);
  for (int kt = 0; kt < KT; ++kt) {
    store_a(0, S0());
    load(S0());
    advance();
    __syncthreads();
    x3_mma_rb<TM, TN>(As, A_SZ, acc, wm * (BM / 2), lane, bq, [&](int s) { load_b(s, kt + 1); });
    __syncthreads();
  }


  // epilogue (k_conv_bwd_data's): addend rows and mask words requested before the accumulator transpose
  constexpr int CT = BN / 4, RSTEP = 256 / CT;
  constexpr int NR = BM / RSTEP, NRC = NR < 8 ? NR : 8, NCH = NR / NRC;
  const int c4 = tid % CT, r0 = tid / CT;
  const int col = n0 + 4 * c4;
  const bool col_ok = col < C;
  f32x4 ex[NCH][NRC];
  uint32_t xw[NCH][NRC];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int i = 0; i < NRC; ++i) {
      const int row = m0 + r0 + (ch * NRC + i) * RSTEP;
      const bool ok = col_ok && row < M;
      ex[ch][i] = (addend && ok) ? *reinterpret_cast<const f32x4*>(addend + (size_t)row * C + col) : f32x4{0.f, 0.f, 0.f, 0.f};
      xw[ch][i] = (xbits && ok) ? xbits[(size_t)row * (C >> 5) + (col >> 5)] : 0u;
    }
  acc_to_lds<BM, BN, TM, TN>(smem, acc, wm, wn, lane);
  __syncthreads();
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int i = 0; i < NRC; ++i) asm volatile("" ::"v"(ex[ch][i]), "v"(xw[ch][i]));
  if (col_ok) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
      for (int i = 0; i < NRC; ++i) {
        const int r = r0 + (ch * NRC + i) * RSTEP;
        const int row = m0 + r;
        if (row < M) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&smem[r * LDC + 4 * c4]) * 1.f;
          if (addend) v += ex[ch][i];
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
// backward weight:  dw[rs,c,k] = sum_p x[pix(p,r,s),c] * g[p,k]; reduction split over the pixels (slabs in `out`).
//   needs C % 4 == 0, K % 4 == 0.  Both operands pixel-major -> 4 x 4 register transposes.
//   GB: the R*S "taps" are independent GEMMs stacked in x / g (Winograd weight gradient).
//   colpart (not GB): [splits][K] per-channel sums of g by the blocks of tile column bx == 0, off the matrix pipe.
// ============================================================================
//   PLAIN (host-checked: stride 1, no padding, OH x OW == H x W, one tap or GB): see `load` below.
template <int BM, int BN, bool GB, int PIPE, bool PLAIN = false>
__global__ void __launch_bounds__(256, (PIPE && (BM + BN) > 128) ? 1 : 2)
k_x3_bwd_weight(lmh_conv_desc d, const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ out,
                int kt_per_split, lmh_fastdiv div_ow, lmh_fastdiv div_oh, int tiles_x, int tiles_y, int splits,
                float* __restrict__ colpart, int stagger) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_SZ = BM * LDH, B_SZ = BN * LDH, A_BUF = 3 * A_SZ, B_BUF = 3 * B_SZ;
  constexpr int LDC = BN + 4;
  __shared__ __attribute__((aligned(16))) float smem[x3_smem<BM, BN, PIPE>::floats];
  x3_t* const As = reinterpret_cast<x3_t*>(smem);
  x3_t* const Bs = As + (PIPE ? 2 : 1) * A_BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  constexpr bool A_ALL = BM / 4 >= 32, B_ALL = BN / 4 >= 32;    // every thread stages a block (no exec-masked region in the stage)
  const bool a_act = cq < BM / 4, b_act = cq < BN / 4;
  const bool a_ok = a_act && (m0 + 4 * cq) < C, b_ok = b_act && (n0 + 4 * cq) < K;
  const float* xb = x + m0 + 4 * cq + (GB ? (size_t)rs * P * C : 0);
  const float* gb = g + n0 + 4 * cq + (GB ? (size_t)rs * P * K : 0);
  int p0 = kt_begin * BK + 4 * kq;       // first of this thread's 4 pixels in the tile the pointers stand on
  const unsigned p_end = (unsigned)min(P, kt_end * BK);      // pixels of this split: [kt_begin * BK, p_end)
#ifdef LMH_PROBES      // timing decomposition (scripts/r6_x3_decomp.py): 1 / 2 split + writes of g / x, 4 loads, 16 MFMA phase, 32 slab store
  const int dbg = stagger >> 8;
#else
  constexpr int dbg = 0;
#endif
  f32x4 ra[2][4], rb[2][4];
  // PLAIN: the source pixel of x IS the output pixel (1x1 / stride-1 layers, the stacked GEMMs of a Winograd weight gradient) —
  // no decode, and both operands are walked with pointers that advance by one stage.  The decode below is two magic-number
  // divisions, the bounds and a 64-bit address product per pixel: 46 quarter-rate integer instructions per thread and stage
  // (v_mul_lo / v_mul_hi / v_mad_u64: ISA check, round 6), as many issue cycles as the split itself — and for 23 + 10 of the
  // 33 weight-gradient launches of the ResNet-50 step it is the identity.  Same addresses, same bits.
  const float* xpp = xb + (size_t)p0 * C;
  const float* gpp = gb + (size_t)p0 * K;
  auto load = [&](auto S) {
    constexpr int s_ = decltype(S)::value;
    if constexpr (PLAIN) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool inp = (unsigned)(p0 + i) < p_end;
        const float* pa_ = (a_ok && inp && !(dbg & 4)) ? xpp + (size_t)i * C : lmh_zero_page;
        if (A_ALL || a_act) ra[s_][i] = *reinterpret_cast<const f32x4*>(pa_);
        const float* pb_ = (b_ok && inp && !(dbg & 4)) ? gpp + (size_t)i * K : lmh_zero_page;
        if (B_ALL || b_act) rb[s_][i] = *reinterpret_cast<const f32x4*>(pb_);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned p = (unsigned)(p0 + i);
      const unsigned t = lmh_div(p, div_ow), ow = p - t * (unsigned)d.OW;
      const unsigned n = lmh_div(t, div_oh), oh = t - n * (unsigned)d.OH;
      const int ih = (int)oh * d.stride + dh0, iw = (int)ow * d.stride + dw0;
      const bool oka = a_ok && p < p_end && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
      const float* pa_ = (oka && !(dbg & 4)) ? xb + ((size_t)((int)n * d.H + ih) * d.W + iw) * C : lmh_zero_page;
      if (A_ALL || a_act) ra[s_][i] = *reinterpret_cast<const f32x4*>(pa_);
      const bool okb = b_ok && p < p_end;
      const float* pb_ = (okb && !(dbg & 4)) ? gpp + (size_t)i * K : lmh_zero_page;
      if (B_ALL || b_act) rb[s_][i] = *reinterpret_cast<const f32x4*>(pb_);
    }
  };
  auto advance = [&]() { p0 += BK; xpp += (size_t)BK * C; gpp += (size_t)BK * K; };   // past the split's end (p_end): the zero page
  auto store_a = [&](int buf, auto S) {
    constexpr int s_ = decltype(S)::value;
    if ((A_ALL || a_act) && !(dbg & 2)) st_km<3>(As + buf * A_BUF, A_SZ, 4 * cq, kq, ra[s_]);
  };
  auto store_b = [&](int buf, auto S) {
    constexpr int s_ = decltype(S)::value;
    if ((B_ALL || b_act) && !(dbg & 1)) st_km<3>(Bs + buf * B_BUF, B_SZ, 4 * cq, kq, rb[s_]);
  };
  f32x16 acc[TM][TN], ccol[TN];
  zero_acc<TM, TN>(acc);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int i = 0; i < 16; ++i) ccol[tn][i] = 0.f;
  const bool do_col = !GB && colpart != nullptr && bx == 0;      // block-uniform
  const int n_st = kt_end - kt_begin;
  if (do_col) {
    auto stage = [&](int buf, auto SL, auto SW) {
      x3_stage<TM, TN, true, 5, 12, 5, 12>(
          As + buf * A_BUF, Bs + buf * B_BUF, A_SZ, B_SZ, acc, ccol, wm * (BM / 2), wn * (BN / 2), lane,
          [&]() { load(SL); }, [&]() { store_a(buf ^ 1, SW); }, [&]() { store_b(buf ^ 1, SW); });
    };
    auto mma = [&](int buf) {
      x3_stage<TM, TN, true, 0, 0, 0, 0>(As + buf * A_BUF, Bs + buf * B_BUF, A_SZ, B_SZ, acc, ccol, wm * (BM / 2),
                                         wn * (BN / 2), lane, []() {}, []() {}, []() {});
    };
    X3_RUN(n_st);
    // every row of ones x B is the column sum: row 0 lives in accumulator element 0 of lanes 0..31
    if (wm == 0 && lane < 32) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int colk = n0 + wn * (BN / 2) + tn * 32 + lane;
        if (colk < K) colpart[(size_t)bz * K + colk] = ccol[tn][0];
      }
    }
  } else {
    auto stage = [&](int buf, auto SL, auto SW) {
      x3_stage<TM, TN, false, 5, 12, 5, 12>(
          As + buf * A_BUF, Bs + buf * B_BUF, A_SZ, B_SZ, acc, ccol, wm * (BM / 2), wn * (BN / 2), lane,
          [&]() { load(SL); }, [&]() { store_a(buf ^ 1, SW); }, [&]() { store_b(buf ^ 1, SW); });
    };
    auto mma = [&](int buf) {
      if (dbg & 16) return;
      x3_stage<TM, TN, false, 0, 0, 0, 0>(As + buf * A_BUF, Bs + buf * B_BUF, A_SZ, B_SZ, acc, ccol, wm * (BM / 2),
                                          wn * (BN / 2), lane, []() {}, []() {}, []() {});
    };
    X3_RUN(n_st);
  }
  acc_to_lds<BM, BN, TM, TN>(smem, acc, wm, wn, lane);
  __syncthreads();
  float* o = out + (size_t)bz * ((size_t)d.R * d.S * C * K) + (size_t)rs * C * K;
  constexpr int CT = BN / 4, RSTEP = 256 / CT;
  const int c4 = tid % CT, r0 = tid / CT;
  const int col = n0 + 4 * c4;
  if (col < K && !(dbg & 32)) {
#pragma unroll 4
    for (int rr = r0; rr < BM; rr += RSTEP) {
      const int row = m0 + rr;
      if (row >= C) break;
      *reinterpret_cast<f32x4*>(o + (size_t)row * K + col) = *reinterpret_cast<const f32x4*>(&smem[rr * LDC + 4 * c4]) * 1.f;
    }
  }
}
