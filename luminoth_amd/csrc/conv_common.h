// Shared pieces of the implicit-GEMM convolution kernels (gfx950).
#pragma once
#include "lmh_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: loads stay loads (float4 struct copies
                                                            // through address-space-ambiguous pointers become memcpy -> scratch)
extern __device__ float lmh_zero_page[16];

// Unsigned division by a launch-time constant: q = (mulhi(n, m) + n) >> s, exact for n < 2^31.
struct lmh_fastdiv { uint32_t m, s; };
static inline lmh_fastdiv lmh_make_fastdiv(uint32_t dv) {
  lmh_fastdiv f;
  uint32_t s = 0;
  while ((1ull << s) < dv) ++s;
  f.s = s;
  f.m = (uint32_t)(((1ull << 32) * ((1ull << s) - dv)) / dv + 1);
  return f;
}
__device__ __forceinline__ unsigned lmh_div(unsigned n, lmh_fastdiv f) { return (__umulhi(n, f.m) + n) >> f.s; }

#define BK 32
#define LDK (BK + 4)  // row stride (floats) of K-contiguous LDS tiles: 9*m mod 16 slots, conflict-free b128

// PROBES (round 5; compiled only with -DLMH_PROBES: `LMH_PROBES=1 bash build.sh`, not part of the product build).
// Start-time stagger of co-resident blocks: on an idle chip blocks b, b + 256, b + 512, ... of a grid share a CU
// (scripts/probes/lds_base_probe.hip); block slot s = (b >> 8) % NRES sleeps s * units x ~1 us (2048 cycles) before its
// first load.  Bits 8..10 of the same word are a timing decomposition (results are WRONG with them): 256 = no residual /
// addend read, 512 = no output stores, 1024 = no main loop.  Set with lmh_conv_set_stagger; 0 = off.
// (profiles/r05_tile_stagger_sweep.log, profiles/r05_epilogue_decomp.log.)
#ifdef LMH_PROBES
__device__ int g_conv_stagger = 0;
__device__ __forceinline__ int conv_probe_bits() { return g_conv_stagger; }
__device__ __forceinline__ void conv_stagger(int nres) {
  const int units = g_conv_stagger & 255;
  if (units > 0) {
    const int slot = (blockIdx.x >> 8) % nres;
    for (int i = 0; i < slot * units; ++i) __builtin_amdgcn_s_sleep(32);
  }
}
#else
__device__ __forceinline__ int conv_probe_bits() { return 0; }
__device__ __forceinline__ void conv_stagger(int) {}
#endif

// bijective XCD remap: hardware places block b on XCD b % 8
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

// ---- MFMA stage: A tile (BM x 32), B tile (32 x BN) from LDS -----------------
template <int TM, int TN, bool A_KC, bool B_KC, int LDA, int LDB>
__device__ __forceinline__ void mfma_stage(const float* __restrict__ As, const float* __restrict__ Bs,
                                           f32x16 (&acc)[TM][TN], int a_off, int b_off, int lane) {
  const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int kb = 16 * h + 4 * g;
    float a[TM][4], b[TN][4];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      if (A_KC) {
        const float4 v = *reinterpret_cast<const float4*>(&As[(a_off + tm * 32 + l31) * LDA + kb]);
        a[tm][0] = v.x; a[tm][1] = v.y; a[tm][2] = v.z; a[tm][3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[tm][i] = As[(kb + i) * LDA + a_off + tm * 32 + l31];
      }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      if (B_KC) {
        const float4 v = *reinterpret_cast<const float4*>(&Bs[(b_off + tn * 32 + l31) * LDB + kb]);
        b[tn][0] = v.x; b[tn][1] = v.y; b[tn][2] = v.z; b[tn][3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) b[tn][i] = Bs[(kb + i) * LDB + b_off + tn * 32 + l31];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][i], b[tn][i], acc[tm][tn], 0, 0, 0);
  }
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return fminf(fmaxf(v, 0.f), 6.f);
  return v;
}

