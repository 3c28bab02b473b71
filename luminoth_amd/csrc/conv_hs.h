// Half-STORAGE convolutions (gfx950): activations, activation gradients and the working copy of the weights are f16 / bf16
// tensors in HBM; v_mfma_f32_32x32x16_{f16,bf16} accumulate in fp32; master weights and weight gradients stay fp32.
// BASELINE configs[4] ("fp16 MFMA path", SURVEY.md §8(d): "fp16 activations/weights with fp32 accumulate + fp32 master
// weights").  conv_half.h keeps fp32 tensors and rounds operands on their way into LDS — every byte it moves is still an
// fp32 byte and the kernels are bound by that staging path; here an operand row of 64 reduction elements is 128 bytes
// and goes from HBM to LDS as eight 16-byte pieces with no arithmetic in between.
//
// Forward and backward-data are ONE kernel: both are gather-GEMMs whose two operands are contiguous along the reduction
// axis once the weight copy has the right layout — the cast pass of the optimizer step (k_half_weights, halfstore.hip) writes
//     w_fwd  =  q(w[r][s][c][k])                  as B[n = k][q = (tap, c)]
//     w_bwd  =  q(w[r][s][c][k] * bn_scale[k])    as B[n = c][q = (tap, k)]   (the frozen-BatchNorm scale folded in)
// in the FRAGMENT ORDER of the MFMA's B operand ([n / 32][stage q / 64][k-step][lane][8 halfs], halfstore.hip; round 6 —
// rounds 3-5: row-major [n][q]), so neither pass transposes anything and a wave instruction moves one contiguous 1 KB of B:
//     forward    y[p][k]  = q( act( sum_{tap,c} x[src(p,tap)][c] * w_fwd[k][tap][c] * scale[k] + shift[k] + res[p][k] ) )
//     backward   dx[p][c] = q( ( sum_{tap,k} g[dst(p,tap)][k] * w_bwd[tap][c][k] + addend[p][c] ) * act'(x[p][c]) )
// with act'(x) read from the activation bit mask the forward epilogue of the producing layer wrote (conv_fast.h).
// Gradients carry the step's loss scale (a power of two chosen by the host, 1 for bf16); the weight gradient divides it
// out of its fp32 accumulators.
//
// Tile: BM x BN outputs, BK = 64 reduction elements per stage, a ring of 3-4 LDS stages of [BM + BN][64] halfs filled by
// direct-to-LDS loads (k_conv_hs).
//
// The weight gradient (k_wgrad_hs_tr, below) reduces over pixels: both operands are pixel-major, its tiles land in LDS
// lane-linearly and are transposed by the LDS read itself.  (A first version transposed 4 x 4 blocks in registers with
// v_perm, the staging map of k_conv_bwd_weight_h: 880 us over the 17 distinct layers against 650 us now.)
#pragma once
#include "conv_half.h"

#define HS_BK 64

struct hs_epilogue {
  const float* scale;        // forward: per output channel (NULL = 1)
  const float* shift;        // forward: per output channel (NULL = 0)
  const void* extra;         // forward: residual; backward: addend — [M][Ncols] halfs (NULL = none)
  const uint32_t* bits_in;   // backward: activation mask of the layer input, [M][Ncols / 32] (NULL = none)
  uint32_t* bits_out;        // forward: activation mask of y to write (NULL = none)
  void* out;                 // [M][Ncols] halfs, or floats when out_f32
  int out_f32;
  float mul;                 // accumulator multiplier (backward: 1)
};

// One stage out of LDS.  A: the swizzled image — row R holds its eight 16-byte k-chunks at chunk position c ^ ((R >> 1) & 7), so
// the 16 rows a lane group of ds_read_b128 touches ({0-3,12-15,20-27} ...) land on 16 different 4-bank groups.  B: the 1 KB
// chunks of the fragment-order weight copy as they are (lane-linear: conflict-free).
template <int DT, int TM, int TN>
__device__ __forceinline__ void hs_mma_stage(const typename HT<DT>::T* __restrict__ As,
                                             const typename HT<DT>::T* __restrict__ Bs, f32x16 (&acc)[TM][TN],
                                             int a_off, int b_off, int lane) {
  typedef typename HT<DT>::V8 V8;
  const int l31 = lane & 31, hi = lane >> 5, swz = (l31 >> 1) & 7;      // (tile offsets are multiples of 32 rows)
  // The fragments of k-step s + 1 are read while the MFMAs of k-step s issue (two register sets, regions pinned with
  // sched_barrier).  Round 3 read each k-step's fragments right in front of its MFMAs: the compiler put an
  // s_waitcnt lgkmcnt(0) before every MFMA pair, i.e. a wave paid the LDS latency (~100+ cycles) for every 64 cycles of
  // matrix work and only the other block of the CU could fill it (ISA check, round 4).
  V8 a[2][TM], b[2][TN];
#define HS_FRAG(set_, s_)                                                                                       \
  do {                                                                                                          \
    const int ch_ = ((2 * (s_) + hi) ^ swz) * 8;                                                                \
    _Pragma("unroll") for (int t = 0; t < TM; ++t)                                                              \
      a[set_][t] = *reinterpret_cast<const V8*>(&As[(a_off + t * 32 + l31) * HS_BK + ch_]);                     \
    _Pragma("unroll") for (int t = 0; t < TN; ++t)     /* B: 1 KB chunks in fragment order, (column group, k-step) */ \
      b[set_][t] = *reinterpret_cast<const V8*>(&Bs[(((b_off >> 5) + t) * (HS_BK / 16) + (s_)) * 512 + lane * 8]); \
  } while (0)
  HS_FRAG(0, 0);
#pragma unroll
  for (int s = 0; s < HS_BK / 16; ++s) {
    const int cur = s & 1;
    if (s + 1 < HS_BK / 16) HS_FRAG(cur ^ 1, s + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = HT<DT>::mfma(a[cur][tm], b[cur][tn], acc[tm][tn]);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef HS_FRAG
}

// The same stage with the B fragments already in registers (BG kernels: B straight from global memory, below): only the A
// fragments come out of LDS.
template <int DT, int TM, int TN>
__device__ __forceinline__ void hs_mma_stage_rb(const typename HT<DT>::T* __restrict__ As, f32x16 (&acc)[TM][TN], int a_off,
                                                int lane, const typename HT<DT>::V8 (&b)[HS_BK / 16][TN]) {
  typedef typename HT<DT>::V8 V8;
  const int l31 = lane & 31, hi = lane >> 5, swz = (l31 >> 1) & 7;
  V8 a[2][TM];
#define HS_FRAG_A(set_, s_)                                                                                     \
  do {                                                                                                          \
    const int ch_ = ((2 * (s_) + hi) ^ swz) * 8;                                                                \
    _Pragma("unroll") for (int t = 0; t < TM; ++t)                                                              \
      a[set_][t] = *reinterpret_cast<const V8*>(&As[(a_off + t * 32 + l31) * HS_BK + ch_]);                     \
  } while (0)
  HS_FRAG_A(0, 0);
#pragma unroll
  for (int s = 0; s < HS_BK / 16; ++s) {
    const int cur = s & 1;
    if (s + 1 < HS_BK / 16) HS_FRAG_A(cur ^ 1, s + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = HT<DT>::mfma(a[cur][tm], b[s][tn], acc[tm][tn]);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef HS_FRAG_A
}

template <int BM, int BN, int NBUF, bool BG>
struct hs_smem {
  static constexpr int ring = NBUF * (BM + (BG ? 0 : BN)) * HS_BK / 2, epi = BM * (BN + 4);      // floats
  static constexpr int floats = ring > epi ? ring : epi;
};
// ring depth by tile: 64 KB (two blocks per CU) for the small tiles, 96 KB (one block) at 128 x 128; BG kernels (A only in
// LDS): three stages — their B loads are awaited one stage after they are issued, and with loads returning in order nothing
// issued before them can stay in flight longer than that
template <int BM, int BN, bool BG> struct hs_nbuf { static constexpr int value = BG ? 3 : ((BM + BN == 128) ? 4 : 3); };

template <int N> __device__ __forceinline__ void hs_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// BWD = false: forward (A = x, B = w_fwd);  BWD = true: backward data (A = g, B = w_bwd).
// Needs (reduction channels) % 64 == 0 and (output channels) % 64 == 0.
// A rows go from global memory straight into LDS (global_load_lds_dwordx4: one wave instruction = 8 rows of 128
// bytes, lane-linear in LDS; the k-chunk a lane FETCHES is permuted so that the image is the swizzled one above), through
// an NBUF-deep ring with NBUF - 1 stages in flight, one raw s_barrier per stage and counted vmcnt waits
// (conv_wgrad1x1.h has the same pipeline): at 64 x 64 a stage is only 4 MFMAs per wave, so the kernel lives on how many
// loads it keeps in the air, not on bandwidth.
// BG = false: the B chunks of a stage ride in the same ring (BN / 8 more LDS-DMA instructions per block and stage).
// BG = true (round 6, third session): a wave loads the B fragments of its own column groups straight into registers, one
// stage ahead (two register sets).  What bounded the LDS variant was the rate of the LDS-DMA instructions themselves
// (scripts/r6_hs_decomp.py: the main loop without MFMAs and with every load served from one cached line still took 94 of the
// RPN convolution's 180 us; a wave gets one 1 KB global_load_lds through every ~100 cycles, a plain global_load_dwordx4 of a
// contiguous 1 KB takes the same path in ~16): B is a third (128 x 64) to a half (128 x 128) of a stage's LDS-DMA
// instructions, it is the same for every row tile of a launch (L2-resident), and in fragment order it needs no LDS at all.
template <int DT, int BM, int BN, bool BWD, bool BG>
__global__ void __launch_bounds__(256, (BM * BN > 128 * 128 || (!BG && BM * BN == 128 * 128)) ? 1 : 2)
k_conv_hs(lmh_conv_desc d, const typename HT<DT>::T* __restrict__ A, const typename HT<DT>::T* __restrict__ B,
          hs_epilogue e) {
  typedef typename HT<DT>::T HTT;
  typedef typename HT<DT>::V8 V8;
  constexpr int NBUF = hs_nbuf<BM, BN, BG>::value, D = NBUF - 1;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AJ = BM / 32, BJ = BG ? 0 : BN / 32;  // LDS-DMA instructions per wave and stage (1 KB each, 4 waves)
  constexpr int NLD = AJ + BJ;
  constexpr int A_SZ = BM * HS_BK, STAGE = (BM + (BG ? 0 : BN)) * HS_BK;
  constexpr int LDC = BN + 4;
  constexpr int KS = HS_BK / 16;                      // k-steps per stage
  __shared__ __attribute__((aligned(16))) float smem[hs_smem<BM, BN, NBUF, BG>::floats];
  HTT* const ring = reinterpret_cast<HTT*>(smem);     // [NBUF][BM (+ BN)][HS_BK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = BWD ? d.N * d.H * d.W : d.N * d.OH * d.OW;
  const int KR = BWD ? d.K : d.C;                     // reduction channels per tap
  const int NC = BWD ? d.C : d.K;                     // output channels
  const int RS = d.R * d.S;
  const int tiles_n = (NC + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int KC = KR / HS_BK, KT = RS * KC;
  // lane -> (row within the instruction's 8 rows, LDS chunk slot); instruction j of this wave covers tile rows
  // (wave * AJ + j) * 8 .. + 7
  const int lrow = lane >> 3, slot = lane & 7;
  // ---- A rows: output pixels (forward) / input pixels (backward)
  const int PW = BWD ? d.W : d.OW, PH = BWD ? d.H : d.OH;
  int a_n[AJ], a_h0[AJ], a_w0[AJ], a_ch[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int r = (wave * AJ + j) * 8 + lrow;
    a_ch[j] = 8 * (slot ^ ((r >> 1) & 7));              // the global k-chunk this lane fetches (halfs)
    const int p = m0 + r;
    if (p < M) {
      const int t = p / PW, pw = p - t * PW;
      a_n[j] = t / PH;
      const int ph = t - a_n[j] * PH;
      a_h0[j] = BWD ? ph + d.pad_top : ph * d.stride - d.pad_top;
      a_w0[j] = BWD ? pw + d.pad_left : pw * d.stride - d.pad_left;
    } else { a_n[j] = -1; a_h0[j] = 0; a_w0[j] = 0; }
  }
  const HTT* const zero = reinterpret_cast<const HTT*>(lmh_zero_page);
  const HTT* pa[AJ];
  int inca[AJ];
  auto setup_tap = [&](int rs_) {
    const int r_ = rs_ / d.S, s_ = rs_ - r_ * d.S;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      bool ok = a_n[j] >= 0;
      size_t off;
      if (BWD) {
        const int th = a_h0[j] - r_ * d.dilation, tw = a_w0[j] - s_ * d.dilation;
        int oh = th, ow = tw;
        ok = ok && th >= 0 && tw >= 0;
        if (d.stride > 1) {
          oh = th / d.stride; ow = tw / d.stride;
          ok = ok && (oh * d.stride == th) && (ow * d.stride == tw);
        }
        ok = ok && oh < d.OH && ow < d.OW;
        off = ((size_t)(a_n[j] * d.OH + oh) * d.OW + ow) * KR;
      } else {
        const int ih = a_h0[j] + r_ * d.dilation, iw = a_w0[j] + s_ * d.dilation;
        ok = ok && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
        off = ((size_t)(a_n[j] * d.H + ih) * d.W + iw) * KR;
      }
      pa[j] = ok ? A + off + a_ch[j] : zero;
      inca[j] = ok ? HS_BK : 0;
    }
  };
  // ---- B: fragment-order chunks of 1 KB, [column group][stage][k-step] (halfstore.hip); stages follow each other across the
  // taps.  Column groups past NC (a forced tile wider than the layer) read the last group: their columns are never stored.
  const int NG = NC >> 5;
  // BG = false: instruction j of this wave moves chunk c = wave * BJ + j of the block's BN / 32 * 4 chunks into LDS
  const HTT* pb[BJ > 0 ? BJ : 1];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int c = wave * BJ + j, g_ = min((n0 >> 5) + c / KS, NG - 1);
    pb[j] = B + ((size_t)g_ * KT * KS + (c % KS)) * 512 + lane * 8;
  }
  // BG = true: this wave's TN column groups, straight into registers
  const HTT* bp[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t)
    bp[t] = B + (size_t)min((n0 >> 5) + wn * TN + t, NG - 1) * KT * KS * 512 + lane * 8;
  // The loads are inline asm: the compiler's own wait for a tracked load of the PREVIOUS loop iteration is a vmcnt(0) in front
  // of the stage's first MFMA (ISA check), which also drains the A stages just issued — the ring would never be more than one
  // stage deep.  Here the waits are the counted ones at the top of a stage, and tie_b makes the registers a stage multiplies
  // depend on that wait (the compiler neither knows that they are still in flight before it nor may move a use above it).
  static_assert(KS == 4, "load_b issues the four k-steps of a stage");
  V8 bq[2][KS][TN];
  // (macros, not lambdas: an asm operand cannot name a captured array element)
#define HS_LOAD_B(set_, t_)                                                                                \
  do {                                                                                                     \
    _Pragma("unroll") for (int tb_ = 0; tb_ < TN; ++tb_) {                                                 \
      const HTT* p_ = bp[tb_] + (size_t)(t_) * (KS * 512);                                                 \
      asm volatile("global_load_dwordx4 %0, %4, off\n\t"                                                  \
                   "global_load_dwordx4 %1, %4, off offset:1024\n\t"                                      \
                   "global_load_dwordx4 %2, %4, off offset:2048\n\t"                                      \
                   "global_load_dwordx4 %3, %4, off offset:3072"                                           \
                   : "=&v"(bq[set_][0][tb_]), "=&v"(bq[set_][1][tb_]), "=&v"(bq[set_][2][tb_]),            \
                     "=&v"(bq[set_][3][tb_])                                                               \
                   : "v"(p_)                                                                               \
                   : "memory");                                                                            \
    }                                                                                                      \
  } while (0)
#define HS_TIE_B(set_)                                                                                     \
  do {                                                                                                     \
    _Pragma("unroll") for (int sb_ = 0; sb_ < KS; ++sb_)                                                   \
      _Pragma("unroll") for (int tb_ = 0; tb_ < TN; ++tb_) asm volatile("" : "+v"(bq[set_][sb_][tb_]));    \
  } while (0)
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  int rs = 0, kc = 0;
  setup_tap(0);
  // timing decomposition (probe builds only; scripts/r6_hs_decomp.py): bits 8.. of the probe word — 1 no residual / addend /
  // mask reads, 2 no output stores, 4 no main loop, 8 A rows from the zero page, 16 no MFMA phase, 32 no B loads past stage 0.
  // Wrong results.
  const int dbg = conv_probe_bits() >> 8;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  // issue the LDS-DMA loads of the stage the pointers stand on into ring slot `buf`, then advance the pointers by one stage
#define HS_ISSUE(buf_)                                                                                     \
  do {                                                                                                     \
    HTT* As_ = ring + (buf_) * STAGE;                                                                      \
    HTT* Bs_ = As_ + A_SZ;                                                                                 \
    _Pragma("unroll") for (int j = 0; j < AJ; ++j)                                                         \
      __builtin_amdgcn_global_load_lds((glb_ptr)((dbg & 8) ? zero : pa[j]), (lds_ptr)(As_ + (wave * AJ + j) * 8 * HS_BK), 16, 0, 0); \
    _Pragma("unroll") for (int j = 0; j < BJ; ++j) {                                                       \
      __builtin_amdgcn_global_load_lds((glb_ptr)((dbg & 8) ? zero : pb[j]), (lds_ptr)(Bs_ + (wave * BJ + j) * 512), 16, 0, 0); \
      pb[j] += KS * 512;                                                                                   \
    }                                                                                                      \
    if (++kc == KC) {                                                                                      \
      kc = 0; ++rs;                                                                                        \
      if (rs < RS) setup_tap(rs);                                                                          \
    } else {                                                                                               \
      _Pragma("unroll") for (int j = 0; j < AJ; ++j) pa[j] += inca[j];                                     \
    }                                                                                                      \
  } while (0)
  if (BG) {
    HS_LOAD_B(0, 0);                       // (the oldest loads in flight: awaited together with stage 0)
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < KT) HS_ISSUE(s);
  int cur = 0;
  // one stage (set_: the B register set it multiplies, literal): stage t has landed once at most `allowed` younger LDS-DMA
  // instructions are still in flight
#define HS_STAGE(set_, t_)                                                                                 \
  do {                                                                                                     \
    const int t = (t_);                                                                                    \
    if (!BG) {                                                                                             \
      if (KT - 1 - t >= D - 1) hs_wait_vm<NLD * (D - 1)>(); else hs_wait_vm<0>();                          \
    } else if (t == 0) {                                                                                   \
      /* issued so far: B(0), A(0) .. A(min(D, KT) - 1) */                                                 \
      if (KT >= D) hs_wait_vm<AJ * (D - 1)>(); else hs_wait_vm<0>();                                       \
    } else {                                                                                               \
      /* iteration t - 1 issued B(t), then A(t - 1 + D): everything but that last stage has to be in — B(t) with it */ \
      /* (loads return in order) */                                                                        \
      if (t - 1 + D < KT) hs_wait_vm<AJ>(); else hs_wait_vm<0>();                                          \
    }                                                                                                      \
    if (BG) HS_TIE_B(set_);                                                                                \
    /* raw barrier (no vmcnt drain): every wave's rows of stage t are in LDS and every wave is done reading the slot of */ \
    /* stage t - 1, which the issue below overwrites */                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                        \
    if (BG) {                                                                                              \
      if (t + 1 < KT && !(dbg & 32)) HS_LOAD_B((set_) ^ 1, t + 1);                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
    }                                                                                                      \
    if (t + D < KT) HS_ISSUE(cur == 0 ? NBUF - 1 : cur - 1);                                               \
    const HTT* As = ring + cur * STAGE;                                                                    \
    if (!(dbg & 16)) {                                                                                     \
      if (BG) hs_mma_stage_rb<DT, TM, TN>(As, acc, wm * (BM / 2), lane, bq[set_]);                         \
      else hs_mma_stage<DT, TM, TN>(As, As + A_SZ, acc, wm * (BM / 2), wn * (BN / 2), lane);               \
    }                                                                                                      \
    cur = (cur + 1 == NBUF) ? 0 : cur + 1;                                                                 \
  } while (0)
  {
    const int kt_run = (dbg & 4) ? 0 : KT;
    int t2 = 0;
    for (; t2 + 1 < kt_run; t2 += 2) { HS_STAGE(0, t2); HS_STAGE(1, t2 + 1); }
    if (t2 < kt_run) HS_STAGE(0, t2);
  }
#undef HS_STAGE
#undef HS_TIE_B
#undef HS_LOAD_B
#undef HS_ISSUE
  __syncthreads();      // the epilogue tile overlays the ring
  // ---- epilogue through LDS: a thread owns 8 consecutive output channels of a row (one 16-byte half store).
  // Every load of the epilogue (scale / shift, the residual or addend rows, the input-mask words) is requested BEFORE the
  // accumulator transpose and awaited once, in front of the first store.  Round 4 (ISA check): with the loads inside the
  // row loop each row was load -> wait -> store, and since loads and stores count on the same vmcnt and retire out of
  // order with respect to each other the wait was a vmcnt(0) that also waited for the previous row's store — four to
  // eight serial memory round trips per tile in every backward launch (mask words) and every residual forward launch.
  constexpr int CT = BN / 8, RSTEP = 256 / CT, NRW = BM / RSTEP;
  const int c8 = tid % CT, r0 = tid / CT;
  const int col = n0 + 8 * c8;
  const bool col_ok = col < NC;
  const int words = NC >> 5, wcol = col >> 5, bsh = 8 * (c8 & 3);
  f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
  if (!BWD && col_ok) {
    if (e.scale) { sc0 = *reinterpret_cast<const f32x4*>(e.scale + col); sc1 = *reinterpret_cast<const f32x4*>(e.scale + col + 4); }
    if (e.shift) { sh0 = *reinterpret_cast<const f32x4*>(e.shift + col); sh1 = *reinterpret_cast<const f32x4*>(e.shift + col + 4); }
  }
  V8 x8[NRW];
  uint32_t mw[NRW];
#pragma unroll
  for (int i = 0; i < NRW; ++i) {
    const int row = m0 + r0 + i * RSTEP;
    const bool ok = col_ok && row < M;
    const size_t o = ok ? (size_t)row * NC + col : 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) x8[i][q] = (HTT)0.f;
    if (e.extra && ok && !(dbg & 1)) x8[i] = *reinterpret_cast<const V8*>(reinterpret_cast<const HTT*>(e.extra) + o);
    mw[i] = (BWD && e.bits_in && ok && !(dbg & 1)) ? e.bits_in[(size_t)row * words + wcol] : 0xFFFFFFFFu;
  }
  acc_to_lds<BM, BN, TM, TN>(smem, acc, wm, wn, lane);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NRW; ++i) asm volatile("" ::"v"(x8[i]), "v"(mw[i]));      // loads awaited in front of the first store
  if (col_ok) {
    const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
    const float sh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
    const float act_lo = (!BWD && d.act) ? 0.f : -INFINITY, act_hi = (!BWD && d.act == 2) ? 6.f : INFINITY;
#pragma unroll
    for (int i = 0; i < NRW; ++i) {
      const int r = r0 + i * RSTEP;
      const int row = m0 + r;
      if (row >= M || (dbg & 2)) break;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(&smem[r * LDC + 8 * c8]);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(&smem[r * LDC + 8 * c8 + 4]);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      const size_t o = (size_t)row * NC + col;
      if (BWD) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] *= e.mul;
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = fmaf(v[q], sc[q], sh[q]);
      }
      if (e.extra) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += (float)x8[i][q];
      }
      if (BWD) {
        if (e.bits_in) {
          const uint32_t m = mw[i] >> bsh;
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = ((m >> q) & 1u) ? v[q] : 0.f;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = fminf(fmaxf(v[q], act_lo), act_hi);
      }
      if (e.out_f32) {
        float* yo = reinterpret_cast<float*>(e.out) + o;
        *reinterpret_cast<f32x4*>(yo) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(yo + 4) = f32x4{v[4], v[5], v[6], v[7]};
      } else {
        V8 h;
        // f16 has 5 exponent bits: a value beyond +-65504 would round to inf and poison the fp32 master weights through
        // the weight gradient; it is stored as the largest finite f16 instead (bf16 has fp32's range)
#pragma unroll
        for (int q = 0; q < 8; ++q) h[q] = (HTT)(DT == 1 ? fminf(fmaxf(v[q], -65504.f), 65504.f) : v[q]);
        *reinterpret_cast<V8*>(reinterpret_cast<HTT*>(e.out) + o) = h;
        // the mask follows the STORED value: a positive sum that rounds to zero (f16 underflow) is a dead unit for
        // the next layer and for the backward pass alike
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (float)h[q];
      }
      if (!BWD && e.bits_out) {   // 4 adjacent lanes hold the 32 channels of one mask word (same row: active together)
        uint32_t m = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) m |= ((v[q] > 0.f && v[q] < act_hi) ? 1u : 0u) << q;
        m <<= bsh;
        m |= __shfl_xor(m, 1);
        m |= __shfl_xor(m, 2);
        if ((c8 & 3) == 0) e.bits_out[(size_t)row * words + wcol] = m;
      }
    }
  }
}

// ============================================================================
// Weight gradient: operands straight into LDS, fragments by the transposing LDS read.
//
// Both operands are pixel-major, so a [64 pixel][BM channel] tile of x (and of g) is a set of contiguous row segments — the
// lane-linear image global_load_lds writes — but an MFMA fragment wants 8 consecutive PIXELS of one channel per lane.
// gfx950's ds_read_b64_tr_b16 does that transpose on the way out of LDS: the 16 lanes of a group hand in the addresses of
// the four 8-byte quarters of four rows (lane i: row i / 4, elements 4 (i % 4) .. + 3) and lane i receives column i of that
// 4 x 16 block (measured: scripts/probes/tr16_probe.hip).  Two such reads make one 32x32x16 operand (k = 8 hi + 0..7).
// No staging registers, no v_perm, no ds_write: a stage costs each wave 4-8 load instructions and (TM + TN) * 8 LDS reads.
// Rows are swizzled by 64-byte quarters (the chunk a lane FETCHES is permuted) so that the 4 rows x 64 bytes a 32-lane
// phase reads fall on 64 different banks: chunk' = chunk ^ 4 ((row >> 1) & 1) for 128-byte rows, ^ 4 (row & 3) for 256.
// The per-channel sums of g come from the matrix pipe as well: one extra MFMA per k-step against an all-ones operand.
// ============================================================================
#define HSW_BK 64      // pixels per stage

template <int BM> __device__ __forceinline__ int hsw_swz(int row) { return BM == 64 ? 4 * ((row >> 1) & 1) : 4 * (row & 3); }

// MFMA with the accumulator in VECTOR registers (gfx950 takes either file).  The column-sum accumulators of k_wgrad_hs_tr are
// only touched in the blocks of one tile column; through the builtin the compiler gave them VGPR homes and copied all 32 of
// them to AGPRs and back around the two MFMAs of EVERY k-step (64 v_accvgpr moves + the MFMA result latency): those
// blocks ran ~3x longer per stage than the others, and a launch ends with its slowest blocks (ISA check, round 4).
// Inline asm is outside the compiler's hazard tracking: s_nop 1 covers a vector write of an operand right in front, the
// caller waits out the write-back before it reads the accumulator.
template <int DT>
__device__ __forceinline__ void hs_mfma_vgpr(f32x16& c, typename HT<DT>::V8 a, typename HT<DT>::V8 b) {
  if (DT == 1) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

typedef short hs_s16x4 __attribute__((ext_vector_type(4)));
typedef short hs_s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ hs_s16x4 hs_tr_read(const void* p) {      // 16-bit elements as raw bits: one builtin for f16 and bf16
  typedef __attribute__((address_space(3))) hs_s16x4* lp;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)p);
}

// RS = 0: the tiles go from global memory into the LDS ring with LDS-DMA instructions (rounds 3-5).  RS = 4 (round 6, third
// session): through registers — global_load_dwordx4 into one of four register sets, three stages ahead, ds_write_b128 into
// the same lane-linear image (two LDS slots) one stage ahead.  An unsplit 256-tile launch of the LDS-DMA kernel takes 1.16 us per
// 32 KB stage and CU = 28 GB/s per CU (scripts/probes/r6_wgrad_group_potential.py), the rate MI355X_MICROARCH.md measures for
// the LDS-DMA path itself (~25 GB/s per CU, 6.4 TB/s over the chip) — a fifth of what the vector caches deliver to registers.
template <int DT, int BM, int BN, bool GATHER, int RS = 0>
__global__ void __launch_bounds__(256, (BM + BN == 256) ? 1 : 2)
k_wgrad_hs_tr(lmh_conv_desc d, const typename HT<DT>::T* __restrict__ x, const typename HT<DT>::T* __restrict__ g,
              float* __restrict__ out, int kt_per_split, lmh_fastdiv div_ow, lmh_fastdiv div_oh, float inv_scale,
              int tiles_x, int tiles_y, int splits, float* __restrict__ colpart) {
  typedef typename HT<DT>::T HTT;
  typedef typename HT<DT>::V8 V8;
  constexpr int NBUF = RS ? 2 : ((BM + BN == 128) ? 4 : 3), D = NBUF - 1;
  static_assert(RS == 0 || RS == 4, "register sets: four (unrolled by hand)");
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_LPR = BM / 8, B_LPR = BN / 8;              // lanes (16-byte chunks) per tile row
  constexpr int A_RPI = 64 / A_LPR, B_RPI = 64 / B_LPR;      // tile rows per wave instruction
  constexpr int A_NI = HSW_BK / A_RPI / 4, B_NI = HSW_BK / B_RPI / 4;     // instructions per wave and stage
  constexpr int NLD = A_NI + B_NI;
  constexpr int A_SZ = HSW_BK * BM, STAGE = HSW_BK * (BM + BN);           // halfs
  __shared__ __attribute__((aligned(16))) typename HT<DT>::T ring[NBUF * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int P = d.N * d.OH * d.OW, K = d.K, C = d.C;
  const int lin = xcd_remap(blockIdx.x, tiles_x * tiles_y * splits);
  const int bz = lin / (tiles_x * tiles_y), rem = lin - bz * (tiles_x * tiles_y);
  const int tiles_c = (C + BM - 1) / BM;
  const int by = rem / tiles_x, bx = rem - by * tiles_x;
  const int rs = bx / tiles_c, m0 = (bx % tiles_c) * BM;
  const int n0 = by * BN;
  const int r_ = rs / d.S, s_ = rs - r_ * d.S;
  const int KT_all = (P + HSW_BK - 1) / HSW_BK;
  const int kt_begin = bz * kt_per_split;
  const int n_st = min(KT_all, kt_begin + kt_per_split) - kt_begin;
  const int dh0 = r_ * d.dilation - d.pad_top, dw0 = s_ * d.dilation - d.pad_left;
  const HTT* const zero = reinterpret_cast<const HTT*>(lmh_zero_page);
  // ---- loader state: instruction j of this wave covers tile rows (wave * NI + j) * RPI .. + RPI - 1
  int a_row[A_NI], a_col[A_NI], b_row[B_NI], b_col[B_NI];
  bool a_cok[A_NI], b_cok[B_NI];
#pragma unroll
  for (int j = 0; j < A_NI; ++j) {
    a_row[j] = (wave * A_NI + j) * A_RPI + lane / A_LPR;
    a_col[j] = m0 + 8 * ((lane % A_LPR) ^ hsw_swz<BM>(a_row[j]));
    a_cok[j] = a_col[j] < C;
  }
#pragma unroll
  for (int j = 0; j < B_NI; ++j) {
    b_row[j] = (wave * B_NI + j) * B_RPI + lane / B_LPR;
    b_col[j] = n0 + 8 * ((lane % B_LPR) ^ hsw_swz<BN>(b_row[j]));
    b_cok[j] = b_col[j] < K;
  }
  int pst = kt_begin * HSW_BK;           // first pixel of the next stage to issue
  const HTT* xrow[A_NI];
  const HTT* grow[B_NI];
#pragma unroll
  for (int j = 0; j < A_NI; ++j) xrow[j] = x + (size_t)(pst + a_row[j]) * C + a_col[j];
#pragma unroll
  for (int j = 0; j < B_NI; ++j) grow[j] = g + (size_t)(pst + b_row[j]) * K + b_col[j];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  // source of instruction j of this wave for the stage at `pst` (the zero page for rows / channels outside)
#define HSW_SRC_A(j, src)                                                                                   \
      const unsigned p = (unsigned)(pst + a_row[j]);                                                        \
      const HTT* src = zero;                                                                                \
      if (GATHER) {                                                                                         \
        const unsigned t = lmh_div(p, div_ow), ow = p - t * (unsigned)d.OW;                                 \
        const unsigned n = lmh_div(t, div_oh), oh = t - n * (unsigned)d.OH;                                 \
        const int ih = (int)oh * d.stride + dh0, iw = (int)ow * d.stride + dw0;                             \
        if (a_cok[j] && n < (unsigned)d.N && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W)  \
          src = x + ((size_t)((int)n * d.H + ih) * d.W + iw) * C + a_col[j];                                \
      } else if (a_cok[j] && (int)p < P) {                                                                  \
        src = xrow[j];                                                                                      \
      }
#define HSW_SRC_B(j, src)                                                                                   \
      const int p = pst + b_row[j];                                                                         \
      const HTT* src = (b_cok[j] && p < P) ? grow[j] : zero;
  // running row pointers of the un-gathered operands (advanced with pst: a 64-bit add per instruction and stage instead of a
  // quarter-rate 64-bit multiply-add)
#define HSW_ADVANCE()                                                                                       \
  do {                                                                                                      \
    pst += HSW_BK;                                                                                          \
    if (!GATHER) { _Pragma("unroll") for (int j = 0; j < A_NI; ++j) xrow[j] += (size_t)HSW_BK * C; }        \
    _Pragma("unroll") for (int j = 0; j < B_NI; ++j) grow[j] += (size_t)HSW_BK * K;                         \
  } while (0)
#define HSW_ISSUE(buf_)                                                                                     \
  do {                                                                                                      \
    HTT* As_ = ring + (buf_) * STAGE;                                                                       \
    HTT* Bs_ = As_ + A_SZ;                                                                                  \
    _Pragma("unroll") for (int j = 0; j < A_NI; ++j) {                                                      \
      HSW_SRC_A(j, src)                                                                                     \
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(As_ + (wave * A_NI + j) * A_RPI * BM), 16, 0, 0); \
    }                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < B_NI; ++j) {                                                      \
      HSW_SRC_B(j, src)                                                                                     \
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(Bs_ + (wave * B_NI + j) * B_RPI * BN), 16, 0, 0); \
    }                                                                                                       \
    HSW_ADVANCE();                                                                                          \
  } while (0)
  // RS: the same 16 bytes per lane into register set set_ (literal) / from that set into LDS slot buf_ (the lane-linear image the
  // LDS-DMA instruction writes).  Inline asm loads: the waits are counted by hand (vector-memory loads return in order), and the
  // empty asm in front of a store makes the registers depend on the wait that precedes it.
  V8 rg[RS ? RS : 1][NLD];
#define HSW_LOAD(set_)                                                                                      \
  do {                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < A_NI; ++j) {                                                      \
      HSW_SRC_A(j, src)                                                                                     \
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(rg[set_][j]) : "v"(src) : "memory");           \
    }                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < B_NI; ++j) {                                                      \
      HSW_SRC_B(j, src)                                                                                     \
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(rg[set_][A_NI + j]) : "v"(src) : "memory");    \
    }                                                                                                       \
    HSW_ADVANCE();                                                                                          \
  } while (0)
#define HSW_STORE(set_, buf_)                                                                               \
  do {                                                                                                      \
    HTT* As_ = ring + (buf_) * STAGE;                                                                       \
    HTT* Bs_ = As_ + A_SZ;                                                                                  \
    _Pragma("unroll") for (int j = 0; j < A_NI; ++j) {                                                      \
      asm volatile("" : "+v"(rg[set_][j]));                                                                 \
      *reinterpret_cast<V8*>(As_ + (wave * A_NI + j) * A_RPI * BM + lane * 8) = rg[set_][j];                \
    }                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < B_NI; ++j) {                                                      \
      asm volatile("" : "+v"(rg[set_][A_NI + j]));                                                          \
      *reinterpret_cast<V8*>(Bs_ + (wave * B_NI + j) * B_RPI * BN + lane * 8) = rg[set_][A_NI + j];         \
    }                                                                                                       \
  } while (0)

  // ---- fragment addressing (bytes inside a tile): group gq = lane >> 4 reads rows 8 (gq >> 1) + 4 half + (i >> 2) of the
  // k-step, channels 16 (gq & 1) + 4 (i & 3) .. + 3 of the 32-channel MFMA tile
  const int gq = lane >> 4, li = lane & 15;
  const int frow = 8 * (gq >> 1) + (li >> 2);                  // + 16 s + 4 half
  const int fch = 16 * (gq & 1) + 4 * (li & 3);                // channel inside the 32-wide tile
  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);
  // per-channel sums of g: blocks of ONE tile column (tap 0, first channel tile), waves wm == 0, on the matrix pipe
  // (wave-uniform, and told so: as a per-lane condition the branch around the column-sum MFMAs was an exec-mask region whose
  // accumulators lived in VGPRs and were copied to AGPRs and back — 64 v_accvgpr moves plus the MFMA result latency —
  // in EVERY k-step of the blocks that carry the sums, i.e. of the blocks every launch ends with; ISA check, round 4)
  const bool do_col = __builtin_amdgcn_readfirstlane((int)(colpart != nullptr && bx == 0 && wm == 0)) != 0;
  f32x16 cacc[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) cacc[t][i] = 0.f;
  V8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (HTT)1.0f;

  // the MFMA phase of one stage out of LDS slot slot_: fragments of k-step s + 1 are read (transposing LDS reads) while the MFMAs
  // of k-step s issue: two register sets, regions pinned with sched_barrier (round 4; before, every MFMA pair sat behind an
  // s_waitcnt lgkmcnt(0))
#define HSW_FRAG(set_, s_)                                                                                      \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                             \
      const int row = 16 * (s_) + 4 * h + frow;                                                                 \
      _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) {                                                       \
        const int c = wm * (BM / 2) + tm * 32 + fch;                                                            \
        const hs_s16x4 v = hs_tr_read(As + row * (BM * 2) + (((c >> 3) ^ hsw_swz<BM>(row)) << 4) + ((c & 7) << 1)); \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) ar[set_][tm][4 * h + e] = v[e];                            \
      }                                                                                                         \
      _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) {                                                       \
        const int c = wn * (BN / 2) + tn * 32 + fch;                                                            \
        const hs_s16x4 v = hs_tr_read(Bs + row * (BN * 2) + (((c >> 3) ^ hsw_swz<BN>(row)) << 4) + ((c & 7) << 1)); \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) br[set_][tn][4 * h + e] = v[e];                            \
      }                                                                                                         \
    }
#define HSW_MMA(slot_)                                                                                          \
  do {                                                                                                          \
    const char* As = reinterpret_cast<const char*>(ring + (slot_) * STAGE);                                     \
    const char* Bs = As + A_SZ * 2;                                                                             \
    hs_s16x8 ar[2][TM], br[2][TN];                                                                              \
    HSW_FRAG(0, 0)                                                                                              \
    _Pragma("unroll") for (int s = 0; s < HSW_BK / 16; ++s) {                                                   \
      const int fc = s & 1;                                                                                     \
      if (s + 1 < HSW_BK / 16) { HSW_FRAG(fc ^ 1, s + 1) }                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
      V8 a[TM], b[TN];                                                                                          \
      _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) a[tm] = __builtin_bit_cast(V8, ar[fc][tm]);             \
      _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) b[tn] = __builtin_bit_cast(V8, br[fc][tn]);             \
      _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                                                         \
        _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = HT<DT>::mfma(a[tm], b[tn], acc[tm][tn]); \
      if (do_col) {                                                                                             \
        _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) hs_mfma_vgpr<DT>(cacc[tn], ones, b[tn]);              \
      }                                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
    }                                                                                                           \
  } while (0)
  if constexpr (RS == 4) {
    // stage s lives in register set s % 4 and LDS slot s % 2.  Prologue: four stages requested, stage 0 stored.
    if (n_st > 0) HSW_LOAD(0);
    if (n_st > 1) HSW_LOAD(1);
    if (n_st > 2) HSW_LOAD(2);
    if (n_st > 3) HSW_LOAD(3);
    if (n_st > 0) {
      if (n_st > 3) hs_wait_vm<3 * NLD>(); else if (n_st == 3) hs_wait_vm<2 * NLD>(); else if (n_st == 2) hs_wait_vm<NLD>(); else hs_wait_vm<0>();
      HSW_STORE(0, 0);
    }
    // iteration t (set_ = t % 4, literal): [barrier] request stage t + 4 into the set stage t left, multiply stage t, then await
    // stage t + 1 (requested three iterations ago; up to three younger stages stay in flight) and store it into the other slot
#define HSW_STAGE_R(set_, t_)                                                                               \
  do {                                                                                                      \
    const int t = (t_);                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                         \
    if (t + 4 < n_st) HSW_LOAD(set_);                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    HSW_MMA((set_) & 1);                                                                                    \
    if (t + 1 < n_st) {                                                                                     \
      const int younger = n_st - t - 2;                                                                     \
      if (younger >= 3) hs_wait_vm<3 * NLD>(); else if (younger == 2) hs_wait_vm<2 * NLD>();                \
      else if (younger == 1) hs_wait_vm<NLD>(); else hs_wait_vm<0>();                                       \
      HSW_STORE(((set_) + 1) & 3, ((set_) + 1) & 1);                                                        \
    }                                                                                                       \
  } while (0)
    int t4 = 0;
    for (; t4 + 4 <= n_st; t4 += 4) { HSW_STAGE_R(0, t4); HSW_STAGE_R(1, t4 + 1); HSW_STAGE_R(2, t4 + 2); HSW_STAGE_R(3, t4 + 3); }
    if (t4 < n_st) HSW_STAGE_R(0, t4);
    if (t4 + 1 < n_st) HSW_STAGE_R(1, t4 + 1);
    if (t4 + 2 < n_st) HSW_STAGE_R(2, t4 + 2);
#undef HSW_STAGE_R
  } else {
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < n_st) HSW_ISSUE(s);
  int cur = 0;
  for (int t = 0; t < n_st; ++t) {
    if (n_st - 1 - t >= D - 1) hs_wait_vm<NLD * (D - 1)>(); else hs_wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + D < n_st) HSW_ISSUE(cur == 0 ? NBUF - 1 : cur - 1);
    HSW_MMA(cur);
    cur = (cur + 1 == NBUF) ? 0 : cur + 1;
  }
  }
#undef HSW_MMA
#undef HSW_FRAG
#undef HSW_STORE
#undef HSW_LOAD
#undef HSW_ADVANCE
#undef HSW_SRC_B
#undef HSW_SRC_A
#undef HSW_ISSUE
  const int l31 = lane & 31, rbase = 4 * (lane >> 5);
  if (do_col) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last column-sum MFMA (inline asm: no hazard tracking) has written back
  if (do_col && lane < 32) {          // every row of cacc holds the column sums; row 0 = register 0 of lanes 0..31
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = n0 + wn * (BN / 2) + tn * 32 + l31;
      if (col < K) colpart[(size_t)bz * K + col] = cacc[tn][0] * inv_scale;
    }
  }
  // epilogue: registers -> global; lanes 0..31 of one accumulator register hold 32 consecutive k of one c row
  float* o = out + (size_t)bz * ((size_t)d.R * d.S * C * K) + (size_t)rs * C * K;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = n0 + wn * (BN / 2) + tn * 32 + l31;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = m0 + wm * (BM / 2) + tm * 32 + (i & 3) + 8 * (i >> 2) + rbase;
        if (row < C && col < K) o[(size_t)row * K + col] = acc[tm][tn][i] * inv_scale;
      }
    }
}
