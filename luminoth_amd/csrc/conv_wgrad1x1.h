// Weight gradient of 1x1 / stride-1 convolutions and Sonnet Linear layers (gfx950, v_mfma_f32_32x32x2_f32):
//
//     dW[c][k] = sum_p x[p][c] * g[p][k]            ("TN" GEMM: M = C, N = K, reduction over the P pixels)
//
// This is 22 of the 26 direct weight-gradient launches of the ResNet-50 step, and the class that sat at 0.38 of
// the fp32-MFMA peak in round 1.  These GEMMs are skinny (C x K = 256x128 ... 512x1024 outputs against 8k-32k
// pixels) and sit at the roofline ridge (43-86 FLOP/B with every operand byte read once), so the reduction must
// be split over the pixels to fill 256 CUs, and the price of a split is its fp32 partial slab: total slab traffic
// is 2 * 4 B * (resident blocks) * BM * BN whatever the layer, i.e. it grows with the TILE AREA.  Small tiles are
// therefore right for these shapes — but the register-staged pipeline of conv_fast.h spends, per 64x64 tile stage,
// 8 global loads + 8 ds_write_b128 + two magic-number pixel decodes on 16 MFMAs: the matrix pipe starves on issue
// slots, not on bandwidth.
//
// Here both operands are pixel-major rows, so a BK x BM tile of x (and a BK x BN tile of g) is a set of contiguous
// row segments and its LDS image [BK][BM] is exactly lane-linear: the tiles are brought in with direct-to-LDS loads
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass, no address arithmetic beyond one pointer bump per
// instruction), through an NBUF-deep LDS ring with prefetch distance NBUF-1, ONE raw s_barrier per stage and
// counted vmcnt waits (a plain __syncthreads() would drain the ring: cdna_hip_programming.md §5).  Fragments are
// read k-major with conflict-free ds_read_b32 one group ahead of the MFMAs (same reader as conv_fast.h); the
// accumulators leave the CU straight from registers (a 32-lane row of the C/D layout is 128 contiguous bytes).
// Out-of-range rows / columns read a zero page (branch-free), so any C % 4 == 0, K % 4 == 0 and any P work.
#pragma once
#include "conv_fast.h"

template <int BM, int BN, int NBUF>
__global__ void __launch_bounds__(256)
k_wgrad_1x1(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ out, int P, int C, int K,
            int kt_per_split, int tiles_c, int tiles_k, int splits, float* __restrict__ colpart) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_SZ = BK * BM, B_SZ = BK * BN, STAGE = A_SZ + B_SZ;
  constexpr int A_LPR = BM / 4, B_LPR = BN / 4;            // lanes per tile row (16 B each)
  constexpr int A_RPI = 64 / A_LPR, B_RPI = 64 / B_LPR;    // tile rows per wave instruction (1 KB)
  constexpr int A_NI = BK / A_RPI / 4, B_NI = BK / B_RPI / 4;   // instructions per wave per stage
  constexpr int NLD = A_NI + B_NI;
  constexpr int D = NBUF - 1;                               // prefetch distance (stages in flight)
  __shared__ __attribute__((aligned(16))) float smem[NBUF * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware 1-D grid, split slowest: the tiles of one pixel range share one XCD's L2
  const int lin = xcd_remap(blockIdx.x, tiles_c * tiles_k * splits);
  const int bz = lin / (tiles_c * tiles_k), rem = lin - bz * (tiles_c * tiles_k);
  const int m0 = (rem % tiles_c) * BM, n0 = (rem / tiles_c) * BN;
  const int KT_all = (P + BK - 1) / BK;
  const int kt_begin = bz * kt_per_split;
  const int n_st = min(KT_all, kt_begin + kt_per_split) - kt_begin;

  // per-lane source rows of this wave's instructions: instruction j of the A set covers tile rows
  // (wave * A_NI + j) * A_RPI .. + A_RPI - 1
  const int a_r = lane / A_LPR, a_c = (lane % A_LPR) * 4;
  const int b_r = lane / B_LPR, b_c = (lane % B_LPR) * 4;
  const bool a_ok = (m0 + a_c) < C, b_ok = (n0 + b_c) < K;
  const int a_row0 = wave * A_NI * A_RPI + a_r, b_row0 = wave * B_NI * B_RPI + b_r;
  const float* pa = x + (size_t)(kt_begin * BK + a_row0) * C + m0 + a_c;
  const float* pb = g + (size_t)(kt_begin * BK + b_row0) * K + n0 + b_c;
  int prow = kt_begin * BK;     // first pixel of the next stage to issue
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
#define WG_ISSUE(buf_)                                                                                    \
  do {                                                                                                    \
    float* As_ = smem + (buf_) * STAGE;                                                                   \
    float* Bs_ = As_ + A_SZ;                                                                              \
    _Pragma("unroll") for (int j = 0; j < A_NI; ++j) {                                                    \
      const bool ok = a_ok && (prow + a_row0 + j * A_RPI) < P;                                            \
      const float* src = ok ? pa + (size_t)j * A_RPI * C : lmh_zero_page;                                 \
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(As_ + (wave * A_NI + j) * A_RPI * BM), 16, 0, 0); \
    }                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < B_NI; ++j) {                                                    \
      const bool ok = b_ok && (prow + b_row0 + j * B_RPI) < P;                                            \
      const float* src = ok ? pb + (size_t)j * B_RPI * K : lmh_zero_page;                                 \
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(Bs_ + (wave * B_NI + j) * B_RPI * BN), 16, 0, 0); \
    }                                                                                                     \
    pa += (size_t)BK * C; pb += (size_t)BK * K; prow += BK;                                               \
  } while (0)

  // Per-channel sums of g (dbeta / dbias) ride along for free: the g tile is in LDS anyway.  Every c-tile block of a
  // (split, k-tile) sees the same g tile, so the 32 pixel rows of a stage are dealt out over the c-tiles: block `ci`
  // adds rows ci, ci + CS, ... of its BN columns (thread = column) and writes ONE partial row; the rows are summed by
  // the reduce / tail kernel in a fixed order.  CS = min(tiles_c, 32) partial rows per split.
  const int CS = min(tiles_c, BK);
  const int ci = rem % tiles_c;
  const bool do_col = colpart != nullptr && ci < CS && tid < BN;
  float csum = 0.f;
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < n_st) WG_ISSUE(s);
  int cur = 0;                   // ring slot of tile t
  for (int t = 0; t < n_st; ++t) {
    // tile t has landed once at most min(D - 1, n_st - 1 - t) younger stages are still in flight
    if (n_st - 1 - t >= D - 1) {
      if (D - 1 == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (NLD * (D - 1) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (NLD * (D - 1) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (NLD * (D - 1) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (NLD * (D - 1) == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (NLD * (D - 1) == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (NLD * (D - 1) == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
      else if (NLD * (D - 1) == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // raw barrier (no vmcnt drain): every wave's pieces of tile t are in LDS, and every wave has finished reading
    // the buffer of tile t - 1, which the issue below overwrites
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + D < n_st) WG_ISSUE(cur == 0 ? NBUF - 1 : cur - 1);     // slot of tile t + D == slot of tile t - 1
    const float* As = smem + cur * STAGE;
    if (do_col) {
      const float* Bt = As + A_SZ + tid;
      for (int k = ci; k < BK; k += CS) csum += Bt[k * BN];
    }
    mfma_stage_pipelined<TM, TN, false, false, BM, BN>(As, As + A_SZ, acc, wm * (BM / 2), wn * (BN / 2), lane);
    cur = (cur + 1 == NBUF) ? 0 : cur + 1;
  }
#undef WG_ISSUE

  if (do_col && (n0 + tid) < K) colpart[(size_t)(bz * CS + ci) * K + n0 + tid] = csum;
  // epilogue: registers -> global; lanes 0..31 of one accumulator register hold 32 consecutive k of one c row
  float* o = out + (size_t)bz * ((size_t)C * K);
  const int l31 = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = n0 + wn * (BN / 2) + tn * 32 + l31;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = m0 + wm * (BM / 2) + tm * 32 + (i & 3) + 8 * (i >> 2) + rbase;
        if (row < C && col < K) o[(size_t)row * K + col] = acc[tm][tn][i];
      }
    }
}
