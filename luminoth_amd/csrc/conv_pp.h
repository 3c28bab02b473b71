// Persistent, software-pipelined 1x1 forward convolution (gfx950, v_mfma_f32_32x32x2_f32) — round 5.
//
// What it is for.  The dominant kernel class of the fp32 train step is the 1x1 convolution with a SHORT reduction (ResNet
// bottleneck "expand" layers: C = 128 / 256 -> K = 512 / 1024, 4 or 8 stages of BK = 32 per 128 x 128 tile).  With
// k_conv_fwd every tile is a block of its own and a launch is one or two lock-step waves of blocks that all do
// "first loads -> LDS -> ... -> accumulator transpose through LDS -> residual + 64 KB of stores" at the same time, the matrix
// pipe idle in both ends.  Timing decomposition on MI355X (scripts/r5_epilogue_decomp.py, profiles/r05_epilogue_decomp.log):
// block3 256 -> 1024 + residual 58.3 us, of which the 8 stages of every tile ~38 us; an "empty" launch of the same grid (no
// main loop, no output traffic) 11.1 us, the output stores 7-9 us; block2 128 -> 512 + residual 71.0 us: main loop ~34 us,
// empty launch 20.3 us (two rounds of blocks), residual + stores 16.5 us.
//
// Here ONE block per compute unit walks a contiguous range of tiles and the whole launch is a single flattened stage loop:
//   * the operand pipeline (global -> VGPR prefetch two stages ahead, ds_write one stage ahead: mfma_stage_split of
//     conv_fast.h) runs straight across tile boundaries — the first loads of tile t + 1 are in flight during the last
//     stages of tile t, so only the first tile of a block pays a load latency;
//   * a finished tile's accumulators are copied to 64 holding registers per lane and drained DURING the first four stages of
//     the next tile, one 32 x 32 quadrant per stage, in the MFMA register layout itself: lane (l31, half) of a quadrant
//     holds 16 values of column l31 — 32 lanes write 128 contiguous bytes of a row — so no transpose through LDS is needed,
//     the activation mask word of a row is one ballot, and the residual row segments are requested one stage ahead;
//   * all global stores / residual loads are issued in the second MFMA group of a stage and the barrier at the end of a
//     stage is a raw s_barrier (no vmcnt drain): the next stage's ds_write waits for the operand loads, by which time the
//     stores are three quarters of a stage old.
// Arithmetic is k_conv_fwd's, element for element (same MFMA chain over k, then * scale + shift + residual, clamp):
// bit-identical results (tests/test_gpu_kernels.py::test_conv1x1_pp_equals_tiled_kernel).
#pragma once
#include "conv_fast.h"

#define PP_VMEM_ALL 0x010
#define PP_VALU 0x002
#define PP_SALU 0x004
#define PP_MFMA 0x008

// One BK = 32 stage like mfma_stage_split, plus two pieces of epilogue work of the PREVIOUS tile (straight-line code) in MFMA
// groups 0 and 1.  EVERY global memory operation of a stage is issued in its first half: gfx9 counts loads and stores on one
// vmcnt and stores may retire out of order, so the wait in front of the next stage's ds_write (for the operand loads) is in
// effect a wait for everything issued before it — with the stores issued early it finds them 1.2+ stages of matrix work old.
template <int TM, int TN, class WriteF, class LoadF, class H1, class H2>
__device__ __forceinline__ void pp_stage(const float* __restrict__ As, const float* __restrict__ Bs, f32x16 (&acc)[TM][TN],
                                         int a_off, int b_off, int lane, WriteF&& do_writes, LoadF&& do_loads,
                                         H1&& h1, H2&& h2) {
  constexpr int NM = 4 * TM * TN;
  const int h = lane >> 5, l31 = lane & 31;
  float a0[TM][4], b0[TN][4], a1[TM][4], b1[TN][4];
  load_frag<TM, true, LDK>(As, a_off, 0, h, l31, a0);
  load_frag<TN, false, 128>(Bs, b_off, 0, h, l31, b0);
  load_frag<TM, true, LDK>(As, a_off, 1, h, l31, a1);
  load_frag<TN, false, 128>(Bs, b_off, 1, h, l31, b1);
  __builtin_amdgcn_sched_barrier(0);
  // ---- group 0 MFMAs || ds_write of stage g + 1 || operand loads of stage g + 2 || epilogue piece 1
  mfma_group<TM, TN>(a0, b0, acc);
  do_writes();
  do_loads();
  h1();
#pragma unroll
  for (int i = 0; i < NM; ++i) {
    __builtin_amdgcn_sched_group_barrier(PP_MFMA, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // DS write
    __builtin_amdgcn_sched_group_barrier(PP_VALU, 6, 0);
    __builtin_amdgcn_sched_group_barrier(PP_SALU, 1, 0);
    __builtin_amdgcn_sched_group_barrier(PP_VMEM_ALL, 2, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- group 1 MFMAs || fragment reads of group 2 || epilogue piece 2
  load_frag<TM, true, LDK>(As, a_off, 2, h, l31, a0);
  load_frag<TN, false, 128>(Bs, b_off, 2, h, l31, b0);
  __builtin_amdgcn_sched_barrier(0);
  mfma_group<TM, TN>(a1, b1, acc);
  h2();
#pragma unroll
  for (int i = 0; i < NM; ++i) {
    __builtin_amdgcn_sched_group_barrier(PP_MFMA, 1, 0);
    __builtin_amdgcn_sched_group_barrier(PP_VALU, 6, 0);
    __builtin_amdgcn_sched_group_barrier(PP_SALU, 1, 0);
    __builtin_amdgcn_sched_group_barrier(PP_VMEM_ALL, 3, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- groups 2 and 3: matrix work only
  load_frag<TM, true, LDK>(As, a_off, 3, h, l31, a1);
  load_frag<TN, false, 128>(Bs, b_off, 3, h, l31, b1);
  __builtin_amdgcn_sched_barrier(0);
  mfma_group<TM, TN>(a0, b0, acc);
  mfma_group<TM, TN>(a1, b1, acc);
  __builtin_amdgcn_sched_barrier(0);
}

// Epilogue addressing through buffer resources: an out-of-range offset (a row past M, the lanes that do not write a mask
// word, every offset of an absent residual) makes the hardware drop the store / return 0 for the load — the drain code is
// straight-line, which is what lets it be interleaved with the MFMAs of the stage it rides in.
#define PP_OOB 0x7FFFFFF0u
struct pp_epi {
  __amdgpu_buffer_rsrc_t res, out, bits;
  int M, K;
  float act_lo, act_hi;
};

// A wave owns a 64 x 32 strip of the tile (two 32 x 32 accumulator blocks, one above the other): rows m0 + 64 wm + 32 QM + ..,
// columns n0 + 32 wn + l31.  Lane (l31, half) holds elements i = 0..15 of a block: row (i & 3) + 8 (i >> 2) + 4 half.
// residual values of elements [I0, I1) of block QM -> rr[I0..I1) (0 past the last row / without a residual)
template <int QM, int I0, int I1>
__device__ __forceinline__ void pp_res_load(const pp_epi& e, int m0, int n0, int wm, int wn, int lane, float (&rr)[16]) {
  const unsigned col = (unsigned)(n0 + wn * 32 + (lane & 31));
  const unsigned rbase = (unsigned)(m0 + wm * 64 + QM * 32 + 4 * (lane >> 5));
  const unsigned base = (rbase * (unsigned)e.K + col) * 4u, rstep = (unsigned)e.K * 4u;
#pragma unroll
  for (int i = I0; i < I1; ++i)
    rr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(e.res, base + (unsigned)((i & 3) + 8 * (i >> 2)) * rstep, 0, 0));
}

// y = clamp(acc * scale + shift + residual) for elements [I0, I1) of block QM, held in `hv`; the activation mask word of a
// row is a ballot over the 32 lanes that hold its 32 channels (lane 0 of each half writes it)
template <int QM, bool BITS, int I0, int I1>
__device__ __forceinline__ void pp_drain(const pp_epi& e, const f32x16& hv, int m0, int n0, int wm, int wn, int lane,
                                         const float (&rr)[16], float sc, float sh) {
  const int l31 = lane & 31, half = lane >> 5;
  const unsigned col0 = (unsigned)(n0 + wn * 32), col = col0 + (unsigned)l31;
  const unsigned rbase = (unsigned)(m0 + wm * 64 + QM * 32 + 4 * half);
  const unsigned base = (rbase * (unsigned)e.K + col) * 4u, rstep = (unsigned)e.K * 4u;
  const unsigned wstep = (unsigned)(e.K >> 5) * 4u;
  const unsigned wbase = (l31 == 0) ? (rbase * (unsigned)(e.K >> 5) + (col0 >> 5)) * 4u : PP_OOB;
#pragma unroll
  for (int i = I0; i < I1; ++i) {
    const unsigned r = (unsigned)((i & 3) + 8 * (i >> 2));
    float v = hv[i];
    v *= sc;
    v += sh;
    v += rr[i];
    v = fminf(fmaxf(v, e.act_lo), e.act_hi);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), e.out, base + r * rstep, 0, 0);
    if (BITS) {
      const unsigned long long bal = __ballot(v > 0.f && v < e.act_hi);
      __builtin_amdgcn_raw_buffer_store_b32(half ? (unsigned)(bal >> 32) : (unsigned)bal, e.bits, wbase + r * wstep, 0, 0);
    }
  }
}

// grid: nblk blocks (<= one per compute unit) of 512 threads = 8 waves (2 x 4), each wave a 64 x 32 strip of the 128 x 128
// tile — two waves per SIMD, 64 accumulator + holding registers per lane instead of 128.  Block b walks tiles
// [b * nsub, min((b + 1) * nsub, ntiles)), tile t = (row block t / tiles_n, column block t % tiles_n).
// Needs C % 32 == 0, K % 128 == 0, (M + 128) * K * 4 < 2^31; CC4: C == 128 (four stages per tile: the last drain step and
// the next tile's first residual request share stage 3).
template <bool CC4, bool BITS>
__global__ void __launch_bounds__(512)
k_conv1x1_pp(int M, int C, int K, int act, const float* __restrict__ x, const float* __restrict__ w,
             const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ residual,
             float* __restrict__ y, uint32_t* __restrict__ act_bits, int ntiles, int tiles_n, int nsub) {
  constexpr int BM = 128, BN = 128, AJ = 2, BJ = 2, NT = 512;
  constexpr int A_SZ = BM * LDK, B_SZ = BK * BN;
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_SZ + B_SZ)];
  float* const As = smem;               // [2][BM][LDK]
  float* const Bs = smem + 2 * A_SZ;    // [2][BK][BN]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int blk = xcd_remap(blockIdx.x, gridDim.x);
  const int t0 = blk * nsub, t1 = min(t0 + nsub, ntiles);
  if (t0 >= t1) return;
  const int CC = CC4 ? 4 : C / BK;
  const int G = (t1 - t0) * CC;
  pp_epi e;
  const unsigned ybytes = (unsigned)M * (unsigned)K * 4u;
  e.res = __builtin_amdgcn_make_buffer_rsrc((void*)residual, 0, residual ? ybytes : 0u, 0x00020000);
  e.out = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, ybytes, 0x00020000);
  e.bits = __builtin_amdgcn_make_buffer_rsrc((void*)act_bits, 0, (BITS && act_bits) ? ybytes >> 5 : 0u, 0x00020000);
  e.M = M; e.K = K;
  e.act_lo = act ? 0.f : -INFINITY;
  e.act_hi = (act == 2) ? 6.f : INFINITY;

  // ---- operand load state (the tile / k-chunk the NEXT global load reads)
  const int kq = tid & 7, arow = tid >> 3;                 // 64 rows per pass, AJ = 2 passes
  constexpr int BROW_T = BN / 4, BROW_STEP = NT / BROW_T;    // 32 float4 per row, 16 rows per pass, BJ = 2 passes
  const int bx4 = tid % BROW_T, bk = tid / BROW_T;
  const float* pa[AJ];
  int inca[AJ];
  const float* pb;
  const size_t incb = (size_t)BK * K, rowb = (size_t)BROW_STEP * K;
  int ld_t = t0, ld_kc = 0;
#define PP_SETUP(t_)                                                                               \
  do {                                                                                             \
    const int m0_ = ((t_) / tiles_n) * BM, n0_ = ((t_) % tiles_n) * BN;                            \
    _Pragma("unroll") for (int j = 0; j < AJ; ++j) {                                               \
      const int p_ = m0_ + arow + 64 * j;                                                          \
      const bool ok_ = p_ < M;                                                                     \
      pa[j] = ok_ ? x + (size_t)p_ * C + 4 * kq : lmh_zero_page;                                   \
      inca[j] = ok_ ? BK : 0;                                                                      \
    }                                                                                              \
    pb = w + (size_t)bk * K + n0_ + 4 * bx4;                                                       \
  } while (0)
#define PP_ADVANCE()                                                                               \
  do {                                                                                             \
    if (++ld_kc == CC) { ld_kc = 0; ++ld_t; PP_SETUP(ld_t); }                                      \
    else { _Pragma("unroll") for (int j = 0; j < AJ; ++j) pa[j] += inca[j]; pb += incb; }          \
  } while (0)
  f32x4 ra[AJ], rb[BJ];
#define PP_LOAD()                                                                                      \
  do {                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < AJ; ++j) ra[j] = *reinterpret_cast<const f32x4*>(pa[j]);     \
    _Pragma("unroll") for (int j = 0; j < BJ; ++j) rb[j] = *reinterpret_cast<const f32x4*>(pb + j * rowb); \
  } while (0)
#define PP_STORE(buf_)                                                                                 \
  do {                                                                                                 \
    float* Ad = As + (buf_) * A_SZ;                                                                    \
    float* Bd = Bs + (buf_) * B_SZ;                                                                    \
    _Pragma("unroll") for (int j = 0; j < AJ; ++j)                                                     \
        *reinterpret_cast<f32x4*>(&Ad[(arow + 64 * j) * LDK + 4 * kq]) = ra[j];                        \
    _Pragma("unroll") for (int j = 0; j < BJ; ++j)                                                     \
        *reinterpret_cast<f32x4*>(&Bd[(bk + BROW_STEP * j) * BN + 4 * bx4]) = rb[j];                   \
  } while (0)

  f32x16 acc[2][1], hold[2][1];
  zero_acc<2, 1>(acc);
  zero_acc<2, 1>(hold);
  float rr[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) rr[i] = 0.f;
  float sc = 1.f, sh = 0.f;      // of the tile being drained, this lane's column
  // origin of the tile in `hold`.  Before the first tile is complete the drain steps run all the same — on rows past M, which
  // the buffer addressing drops — so that every stage of the loop is the same straight-line code
  int dm0 = M, dn0 = 0;
  // scale / shift of tile t_ for this lane's column + the residual of its upper block: requested in the LAST stage of the tile,
  // one stage before its first drain step
#define PP_TILE_EPI_BEGIN(t_)                                                                      \
  do {                                                                                             \
    const int m0_ = ((t_) / tiles_n) * BM, n0_ = ((t_) % tiles_n) * BN;                            \
    const int col_ = n0_ + wn * 32 + (lane & 31);                                                  \
    sc = scale ? scale[col_] : 1.f;                                                                \
    sh = shift ? shift[col_] : 0.f;                                                                \
    pp_res_load<0, 0, 16>(e, m0_, n0_, wm, wn, lane, rr);                                          \
  } while (0)
  // drain step of the tile at (dm0, dn0) held in `hold`: elements [I0_, I1_) of block QM_ leave; NEXT_: the same elements of
  // the OTHER block's residual are requested into the registers just freed
#define PP_PIECE(QM_, NEXT_, I0_, I1_)                                                                              \
  do {                                                                                                              \
    pp_drain<QM_, BITS, I0_, I1_>(e, hold[QM_][0], dm0, dn0, wm, wn, lane, rr, sc, sh);                              \
    if (NEXT_) pp_res_load<1, I0_, I1_>(e, dm0, dn0, wm, wn, lane, rr);                                              \
  } while (0)

  // ---- prologue: stage 0 into LDS, stage 1 in registers
  PP_SETUP(t0);
  PP_LOAD();
  PP_STORE(0);
  if (G > 1) PP_ADVANCE();
  PP_LOAD();
  __syncthreads();
  int g = 0;
  // one stage of tile t: MFMAs into `acc`; H1_ / H2_ = the epilogue pieces that ride in MFMA groups 0 / 1
#define PP_STAGE(H1_, H2_)                                                                         \
  do {                                                                                             \
    const int cur = g & 1;                                                                         \
    if (g + 2 < G) PP_ADVANCE();                                                                   \
    pp_stage<2, 1>(As + cur * A_SZ, Bs + cur * B_SZ, acc, wm * 64, wn * 32, lane,                  \
                   [&]() { PP_STORE(cur ^ 1); }, [&]() { PP_LOAD(); },                             \
                   [&]() { H1_; }, [&]() { H2_; });                                                \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                \
    ++g;                                                                                           \
  } while (0)
#define PP_PLAIN_STAGE()                                                                           \
  do {                                                                                             \
    const int cur = g & 1;                                                                         \
    if (g + 2 < G) PP_ADVANCE();                                                                   \
    mfma_stage_split<2, 1, true, false, LDK, 128, AJ + BJ, AJ + BJ>(                               \
        As + cur * A_SZ, Bs + cur * B_SZ, acc, wm * 64, wn * 32, lane,                             \
        [&]() { PP_STORE(cur ^ 1); }, [&]() { PP_LOAD(); });                                       \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                \
    ++g;                                                                                           \
  } while (0)
  for (int t = t0; t < t1; ++t) {
    // four drain steps of 8 elements: upper block (its residual came with the tile's last stage), then the lower block
    PP_STAGE(PP_PIECE(0, true, 0, 4), PP_PIECE(0, true, 4, 8));
    PP_STAGE(PP_PIECE(0, true, 8, 12), PP_PIECE(0, true, 12, 16));
    PP_STAGE(PP_PIECE(1, false, 0, 4), PP_PIECE(1, false, 4, 8));
    if (CC4) {
      PP_STAGE(PP_PIECE(1, false, 8, 12), PP_PIECE(1, false, 12, 16); PP_TILE_EPI_BEGIN(t));
    } else {
      PP_STAGE(PP_PIECE(1, false, 8, 12), PP_PIECE(1, false, 12, 16));
      for (int kc = 4; kc < CC - 1; ++kc) PP_PLAIN_STAGE();
      PP_STAGE((void)0, PP_TILE_EPI_BEGIN(t));
    }
    // the tile is complete: its accumulators move to the holding registers (drained under the next tile's first stages)
    dm0 = (t / tiles_n) * BM;
    dn0 = (t % tiles_n) * BN;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      hold[a][0] = acc[a][0];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][0][i] = 0.f;
    }
  }
  // ---- the last tile: nothing left to hide it under
  PP_PIECE(0, true, 0, 16);
  PP_PIECE(1, false, 0, 16);
#undef PP_PLAIN_STAGE
#undef PP_STAGE
#undef PP_PIECE
#undef PP_TILE_EPI_BEGIN
#undef PP_STORE
#undef PP_LOAD
#undef PP_ADVANCE
#undef PP_SETUP
}
