// Per-channel sums of `nb` rows of K floats in a fixed order (deterministic): the last stage of every dbeta / dbias.
// Shared by elementwise.hip (k_colsum_finish) and conv_winograd.h (the same sums as extra blocks of k_wino4_dw).
#pragma once
#include "lmh_common.h"

// out[c] = sum_b partial[b][c]: 32 columns x 8 row-groups per block, fixed summation tree (deterministic)
__device__ __forceinline__ float colsum_partial(const float* __restrict__ partial, int nb, int K, int c, int g) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = g;
  // sixteen rows requested together, added in the order of the loop below (same four chains, same sums): with four loads
  // per trip a thread walked its 64 rows of a 512-row plane as 16 serial round trips of memory latency (round 4)
  for (; b + 120 < nb; b += 128) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = partial[(size_t)(b + 8 * q) * K + c];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) { s0 += v[4 * q]; s1 += v[4 * q + 1]; s2 += v[4 * q + 2]; s3 += v[4 * q + 3]; }
  }
  for (; b + 24 < nb; b += 32) {
    s0 += partial[(size_t)b * K + c];
    s1 += partial[(size_t)(b + 8) * K + c];
    s2 += partial[(size_t)(b + 16) * K + c];
    s3 += partial[(size_t)(b + 24) * K + c];
  }
  for (; b < nb; b += 8) s0 += partial[(size_t)b * K + c];
  return (s0 + s1) + (s2 + s3);
}

// one 256-thread block: columns [32 * blk, 32 * blk + 32)
__device__ __forceinline__ void colsum_finish_block(const float* __restrict__ partial, int nb, int K, float* __restrict__ out,
                                                    int blk) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blk * 32 + cl;
  red[g][cl] = (c < K) ? colsum_partial(partial, nb, K, c, g) : 0.f;
  __syncthreads();
  if (g == 0 && c < K) {
    float t = red[0][cl];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += red[i][cl];
    out[c] = t;
  }
}
