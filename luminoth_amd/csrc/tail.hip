// Deferred "tails" of the weight gradients, many layers per launch (gfx950, HBM / latency bound).
//
// After its MFMA kernel every trainable convolution still needs: the deterministic split-K reduction of its partial
// slabs, for frozen-BatchNorm layers the gamma gradient  dgamma[k] = rstd[k] * (sum_i w[i,k] * dW_raw[i,k] - mean[k] *
// dbeta[k])  together with the BN scaling of the weight gradient  dW = dW_raw * (gamma * rstd)[k], and the last stage
// of the per-channel sums of g (dbeta / dbias).  Round 1 ran that as 3-4 tiny launches per layer: ~110 launches per
// ResNet-50 step, each a 5-10 us latency chain of its own (0.6 ms of the 10.6 ms serial kernel time) and — worse — the
// chain the main stream waits for at the very end of the step.  None of it is needed before the optimizer (or the
// gradient all-reduce) reads the gradients, so the layers now only QUEUE a descriptor and the whole backlog is
// finished by TWO launches:
//   k_tail_reduce : slabs -> raw gradient (fixed summation order), x w -> per-block column partials, x scale -> dW
//   k_tail_finish : column partials -> dgamma; column-sum partials -> dbeta / dbias
// Descriptors travel in the kernel arguments (<= LMH_TAIL_MAX per launch), blocks find their layer by a short scan.
#include "lmh_common.h"

#define LMH_TAIL_MAX 20
#define TAIL_MAX_K 4096

struct tail_args {
  lmh_wgrad_tail t[LMH_TAIL_MAX];
  int32_t blk0[LMH_TAIL_MAX + 1];     // first block of layer i (k_tail_reduce) / first column block (k_tail_finish)
  int32_t rpb[LMH_TAIL_MAX];          // rows of [RSC][K] per block
  int64_t part0[LMH_TAIL_MAX + 1];    // offset (floats) of layer i's [nblocks][K] partial rows in the workspace
  int32_t count;
};

__device__ __forceinline__ int tail_find(const int32_t* blk0, int count, int b) {
  int i = 0;
  while (i + 1 < count && blk0[i + 1] <= b) ++i;
  return i;
}

__global__ void __launch_bounds__(256)
k_tail_reduce(tail_args a, float* __restrict__ partial) {
  __shared__ float scol[TAIL_MAX_K];
  const int li = tail_find(a.blk0, a.count, blockIdx.x);
  const lmh_wgrad_tail& t = a.t[li];
  const int K = t.K, K4 = K >> 2;
  const int64_t rsc = t.n / K;
  const int lb = blockIdx.x - a.blk0[li];
  const int64_t r0 = (int64_t)lb * a.rpb[li];
  const int64_t r1 = min(rsc, r0 + a.rpb[li]);
  const int tpr = min(K4, 256), rstep = 256 / tpr;
  const int c4 = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
  const bool bn = t.dgamma != nullptr;
  if (rsub < rstep) {
    // FOUR rows per trip, and four slabs per round of a trip: sixteen float4 loads in flight, every load of a trip in
    // front of its first store.  The loop this replaces took one row per trip — slab loads four at a time, then w, then
    // the store, and the next trip's loads behind that store (loads and stores share vmcnt: a vmcnt(0) round trip per
    // row) — and ran at 2.5 TB/s.  Same additions in the same order: slab 0, 1, 2, ... per element, rows in order.
    constexpr int RB = 4, SPB = 4;
    for (int cc = c4; cc < K4; cc += tpr) {
      float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
      if (t.scale) sc = *reinterpret_cast<const float4*>(t.scale + 4 * cc);
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int64_t rb = r0 + rsub; rb < r1; rb += (int64_t)RB * rstep) {
        size_t o[RB];
        float4 d[RB], wv[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) o[u] = (size_t)min(rb + (int64_t)u * rstep, r1 - 1) * K + 4 * cc;   // clamped: result unused
        if (t.splits > 0) {                        // deterministic: slab 0, 1, 2, ... in order
#pragma unroll
          for (int u = 0; u < RB; ++u) d[u] = *reinterpret_cast<const float4*>(t.slabs + o[u]);
          for (int sp = 1; sp < t.splits; sp += SPB) {
            float4 e[SPB][RB];
#pragma unroll
            for (int q = 0; q < SPB; ++q)
#pragma unroll
              for (int u = 0; u < RB; ++u)
                e[q][u] = *reinterpret_cast<const float4*>(t.slabs + (size_t)min(sp + q, t.splits - 1) * t.n + o[u]);
            __builtin_amdgcn_sched_barrier(0);     // all sixteen requested before the first is added (the scheduler sinks loads to their use)
#pragma unroll
            for (int q = 0; q < SPB; ++q) {
              if (sp + q >= t.splits) break;
#pragma unroll
              for (int u = 0; u < RB; ++u) { d[u].x += e[q][u].x; d[u].y += e[q][u].y; d[u].z += e[q][u].z; d[u].w += e[q][u].w; }
            }
          }
        } else {
#pragma unroll
          for (int u = 0; u < RB; ++u) d[u] = *reinterpret_cast<const float4*>(t.dw + o[u]);
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) wv[u] = bn ? *reinterpret_cast<const float4*>(t.w + o[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < RB; ++u) asm volatile("" ::"v"(d[u].x), "v"(d[u].y), "v"(d[u].z), "v"(d[u].w), "v"(wv[u].x), "v"(wv[u].y), "v"(wv[u].z), "v"(wv[u].w));
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          if (rb + (int64_t)u * rstep >= r1) break;
          float4 dd = d[u];
          if (bn) { s.x += wv[u].x * dd.x; s.y += wv[u].y * dd.y; s.z += wv[u].z * dd.z; s.w += wv[u].w * dd.w; }
          if (t.scale) { dd.x *= sc.x; dd.y *= sc.y; dd.z *= sc.z; dd.w *= sc.w; }
          if (t.splits > 0 || t.scale) *reinterpret_cast<float4*>(t.dw + o[u]) = dd;
        }
      }
      if (bn) *reinterpret_cast<float4*>(&scol[rsub * K + 4 * cc]) = s;
    }
  }
  if (!bn) return;
  __syncthreads();
  float* prow = partial + a.part0[li] + (size_t)lb * K;
  for (int c = threadIdx.x; c < K; c += 256) {
    float v = scol[c];
    for (int g2 = 1; g2 < rstep; ++g2) v += scol[g2 * K + c];
    prow[c] = v;
  }
}

// fixed-tree column sum of `nb` partial rows (same tree as elementwise.hip::colsum_partial)
__device__ __forceinline__ float tail_colsum(const float* __restrict__ p, int nb, int K, int c, int g) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;       // four independent chains: the loads of a round are in flight together
  int b = g;
  // sixteen rows requested together, added in the order of the loop below (same four chains, same sums): with four loads
  // per trip a thread walked its 64 rows of a 512-row plane as 16 serial round trips of memory latency (round 4)
  for (; b + 120 < nb; b += 128) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = p[(size_t)(b + 8 * q) * K + c];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) { s0 += v[4 * q]; s1 += v[4 * q + 1]; s2 += v[4 * q + 2]; s3 += v[4 * q + 3]; }
  }
  for (; b + 24 < nb; b += 32) {
    s0 += p[(size_t)b * K + c];
    s1 += p[(size_t)(b + 8) * K + c];
    s2 += p[(size_t)(b + 16) * K + c];
    s3 += p[(size_t)(b + 24) * K + c];
  }
  for (; b < nb; b += 8) s0 += p[(size_t)b * K + c];
  return (s0 + s1) + (s2 + s3);
}

__global__ void __launch_bounds__(256)
k_tail_finish(tail_args a, const float* __restrict__ partial) {
  __shared__ float red[2][8][33];
  const int li = tail_find(a.blk0, a.count, blockIdx.x);
  const lmh_wgrad_tail& t = a.t[li];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = (blockIdx.x - a.blk0[li]) * 32 + cl;
  const bool ok = c < t.K;
  const int nb = (int)((a.part0[li + 1] - a.part0[li]) / (t.K > 0 ? t.K : 1));
  red[0][g][cl] = (ok && t.colpart) ? tail_colsum(t.colpart, t.colrows, t.K, c, g) : 0.f;
  red[1][g][cl] = (ok && t.dgamma) ? tail_colsum(partial + a.part0[li], nb, t.K, c, g) : 0.f;
  __syncthreads();
  if (g == 0 && ok) {
    float dbeta;
    if (t.colpart) {
      dbeta = red[0][0][cl];
#pragma unroll
      for (int i = 1; i < 8; ++i) dbeta += red[0][i][cl];
      t.colsum[c] = dbeta;
    } else {
      dbeta = t.colsum ? t.colsum[c] : 0.f;
    }
    if (t.dgamma) {
      float dot = red[1][0][cl];
#pragma unroll
      for (int i = 1; i < 8; ++i) dot += red[1][i][cl];
      t.dgamma[c] = t.rstd[c] * (dot - t.mean[c] * dbeta);
    }
  }
}

static int tail_blocks(const lmh_wgrad_tail* t, int* rpb_out) {
  const int64_t rsc = t->n / t->K;
  int rpb = (int)((rsc + 127) / 128);              // <= 128 row slabs per layer
  const int k4 = t->K >> 2;
  const int rstep = k4 >= 256 ? 1 : 256 / k4;
  if (rpb < 2 * rstep) rpb = 2 * rstep;
  *rpb_out = rpb;
  return (int)((rsc + rpb - 1) / rpb);
}

extern "C" size_t lmh_wgrad_tail_batch_workspace_bytes(const lmh_wgrad_tail* tails, int count) {
  size_t fl = 0;
  for (int i = 0; i < count; ++i) {
    int rpb;
    if (tails[i].dgamma) fl += (size_t)tail_blocks(&tails[i], &rpb) * tails[i].K;
  }
  return lmh_align_up(fl * sizeof(float) + 256, 256);
}

extern "C" int lmh_wgrad_tail_batch(const lmh_wgrad_tail* tails, int count, void* ws, size_t ws_bytes,
                                    lmh_stream_t stream) {
  LMH_CHECK_ARG(tails && count > 0);
  if (!ws || ws_bytes < lmh_wgrad_tail_batch_workspace_bytes(tails, count)) {
    lmh_set_error("lmh_wgrad_tail_batch: workspace too small");
    return LMH_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* partial = reinterpret_cast<float*>(ws);
  int64_t part_off = 0;
  for (int base = 0; base < count; base += LMH_TAIL_MAX) {
    const int n = count - base < LMH_TAIL_MAX ? count - base : LMH_TAIL_MAX;
    tail_args ra, fa;
    memset(&ra, 0, sizeof(ra));
    int nb_r = 0, nb_f = 0;
    for (int i = 0; i < n; ++i) {
      const lmh_wgrad_tail& t = tails[base + i];
      LMH_CHECK_ARG(t.dw && t.K > 0 && (t.K & 3) == 0 && t.K <= TAIL_MAX_K && t.n > 0 && t.n % t.K == 0);
      LMH_CHECK_ARG(t.splits == 0 || t.slabs != nullptr);
      LMH_CHECK_ARG(!t.dgamma || (t.w && t.mean && t.rstd && t.scale));
      LMH_CHECK_ARG(!t.colpart || (t.colsum && t.colrows > 0));
      ra.t[i] = t;
      ra.blk0[i] = nb_r;
      int rpb;
      const int nb = tail_blocks(&t, &rpb);
      ra.rpb[i] = rpb;
      ra.part0[i] = part_off;
      const bool work = t.splits > 0 || t.scale || t.dgamma;
      nb_r += work ? nb : 0;
      if (t.dgamma) part_off += (int64_t)nb * t.K;
    }
    ra.blk0[n] = nb_r;
    ra.count = n;
    fa = ra;
    for (int i = 0; i < n; ++i) {
      const lmh_wgrad_tail& t = tails[base + i];
      fa.blk0[i] = nb_f;
      nb_f += (t.colpart || t.dgamma) ? (t.K + 31) / 32 : 0;
    }
    fa.blk0[n] = nb_f;
    // part0[n] closes the last layer's partial range (k_tail_finish derives the row count from the difference)
    ra.part0[n] = part_off;
    fa.part0[n] = part_off;
    if (nb_r > 0) lmh_launch(k_tail_reduce, dim3(nb_r), dim3(256), 0, st, ra, partial);
    if (nb_f > 0) lmh_launch(k_tail_finish, dim3(nb_f), dim3(256), 0, st, fa, (const float*)partial);
  }
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
