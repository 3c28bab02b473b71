// RPN proposal pipeline + batched sort + NMS (gfx950).
//
// Reference semantics: luminoth/models/fasterrcnn/rpn_proposal.py:41-197,
// tf.nn.top_k, tf.image.non_max_suppression (TF 1.x CPU kernels, restated in
// oracle/tfops.py).  All kernels are HBM/latency-bound integer + fp32 scalar
// work: coalesced SoA-free float4 box loads, LDS bitonic sort, 64x64 IoU
// bit-mask tiles (one u64 word per lane) and a wave-serial greedy reduce.
#include <stdlib.h>
#include "lmh_common.h"
int lmh_opt(const char* name);   // api.hip: the lmh_set_option registry

// ----------------------------------------------------------------------------
// 1. decode + clip + validity + sort key   (one thread per anchor)
// ----------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_rpn_decode(lmh_rpn_proposal_desc d, int N, int Npad, const float* __restrict__ cls_score,
             const float* __restrict__ bbox_pred, const int32_t* __restrict__ anchor_ref,
             float* __restrict__ cls_prob, float4* __restrict__ boxes, uint64_t* __restrict__ keys,
             int32_t* __restrict__ n_valid) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain beside MFMA kernels of other streams: win the issue arbitration
  const int b = blockIdx.y;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= Npad) return;
  uint64_t key = ~0ull;
  if (n < N) {
    const size_t row = (size_t)b * N + n;
    // tf.nn.softmax over 2 logits (rpn.py:163): max-subtracted exp / sum.
    const float2 s = reinterpret_cast<const float2*>(cls_score)[row];
    const float m = fmaxf(s.x, s.y);
    const float e0 = expf(s.x - m), e1 = expf(s.y - m);
    const float den = e0 + e1;
    const float p0 = e0 / den, p1 = e1 / den;
    reinterpret_cast<float2*>(cls_prob)[row] = make_float2(p0, p1);

    int32_t a4[4];
    lmh_anchor(anchor_ref, n, d.A, d.feat_w, d.anchor_stride, a4);
    bool ok = true;
    if (d.filter_outside_anchors) {  // rpn_proposal.py:69-90 (int32 compares)
      ok = a4[0] >= 0 && a4[1] >= 0 && a4[2] < (int)d.im_w && a4[3] < (int)d.im_h;
    }
    lmh_box roi = {(float)a4[0], (float)a4[1], (float)a4[2], (float)a4[3]};
    const float4 dl = reinterpret_cast<const float4*>(bbox_pred)[row];
    lmh_box p = lmh_decode(roi, dl.x, dl.y, dl.z, dl.w, 1.f, 1.f);
    // zero/negative area filter has NO +1 (rpn_proposal.py:101-105)
    const bool area_ok = fmaxf(p.x2 - p.x1, 0.f) * fmaxf(p.y2 - p.y1, 0.f) > 0.f;
    ok = ok && area_ok && (p1 >= d.min_prob_threshold);
    if (!d.clip_after_nms) p = lmh_clip(p, d.im_h, d.im_w);
    boxes[row] = make_float4(p.x1, p.y1, p.x2, p.y2);
    if (ok) {
      // ascending u64 sort == (score desc, index asc): tf.nn.top_k order.
      key = ((uint64_t)(~lmh_float_orderable(p1)) << 32) | (uint32_t)n;
    }
  }
  keys[(size_t)b * Npad + n] = key;
  // per-image valid count: one atomic per wave
  const unsigned long long bal = __ballot(key != ~0ull);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&n_valid[b], __popcll(bal));
}

// ----------------------------------------------------------------------------
// 2. batched bitonic sort of u64 keys (ascending).  Chunks of SORT_CHUNK keys
//    are sorted / merged in LDS; strides >= SORT_CHUNK go through global.
// ----------------------------------------------------------------------------
#define SORT_CHUNK 4096
#define SORT_THREADS 512

__device__ __forceinline__ void cmp_swap(uint64_t& a, uint64_t& b, bool asc) {
  if ((a > b) == asc) { uint64_t t = a; a = b; b = t; }
}

// LDS phases of the sort (round 4).  Rounds 1-3 ran one compare-exchange pass per barrier — 78 barriers for a 4096-key
// chunk, each pass two LDS reads + two writes per pair — and filled LDS with a load -> wait -> ds_write loop (8 serial
// round trips per thread).  Now
//   * a chunk is loaded with all of a thread's (<= 8) global reads in flight together;
//   * up to THREE consecutive passes of a merge step (strides j, j/2, j/4) run on registers: a thread owns the 8 keys
//     that differ in exactly those three index bits (the scheme of k_sort_global_multi, inside LDS): 30 barriers for
//     the local sort, 4 for a merge tail;
//   * key i sits at slot i + i / 32: with 8-byte keys and power-of-two strides the plain layout puts a wave's accesses
//     on a few banks (8 consecutive keys per thread at the smallest strides: 8-way), the skew spreads them.
#define SORT_SLOT(i_) ((i_) + ((i_) >> 5))
#define SORT_LDS_KEYS(chunk_) ((chunk_) + ((chunk_) >> 5) + 1)

template <int NT>
__device__ __forceinline__ void sort_chunk_load(uint64_t* s, const uint64_t* __restrict__ src, int chunk) {
  uint64_t v[SORT_CHUNK / NT];
#pragma unroll
  for (int u = 0; u < SORT_CHUNK / NT; ++u) {
    const int i = threadIdx.x + u * NT;
    v[u] = src[min(i, chunk - 1)];
  }
#pragma unroll
  for (int u = 0; u < SORT_CHUNK / NT; ++u) {
    const int i = threadIdx.x + u * NT;
    if (i < chunk) s[SORT_SLOT(i)] = v[u];
  }
}

template <int NT>
__device__ __forceinline__ void sort_chunk_store(const uint64_t* s, uint64_t* __restrict__ dst, int chunk) {
#pragma unroll
  for (int u = 0; u < SORT_CHUNK / NT; ++u) {
    const int i = threadIdx.x + u * NT;
    if (i < chunk) dst[i] = s[SORT_SLOT(i)];
  }
}

// S passes (strides j, j / 2, ..., j >> (S - 1)) of merge step k over the chunk in LDS; ascending where bit k of the
// GLOBAL index (gbase + index in the chunk) is clear.  Ends with a barrier.
template <int S>
__device__ __forceinline__ void sort_lds_passes(uint64_t* s, int chunk, int gbase, int k, int j) {
  constexpr int E = 1 << S;
  const int jl = j >> (S - 1);
  for (int g = threadIdx.x; g < chunk / E; g += blockDim.x) {
    const int low = g & (jl - 1);
    const int base = ((g - low) << S) | low;             // index with the S stride bits clear
    const bool asc = (((gbase + base) & k) == 0);          // bit k lies above every stride of the step
    uint64_t v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = s[SORT_SLOT(base + e * jl)];
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const int bit = 1 << (S - 1 - q);
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (!(e & bit)) cmp_swap(v[e], v[e | bit], asc);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) s[SORT_SLOT(base + e * jl)] = v[e];
  }
  __syncthreads();
}

// every pass of merge step k with stride <= j0 (j0 < chunk), three at a time
__device__ __forceinline__ void sort_lds_step(uint64_t* s, int chunk, int gbase, int k, int j0) {
  int j = j0;
  while (j > 0) {
    if (j >= 4) { sort_lds_passes<3>(s, chunk, gbase, k, j); j >>= 3; }
    else if (j == 2) { sort_lds_passes<2>(s, chunk, gbase, k, j); j = 0; }
    else { sort_lds_passes<1>(s, chunk, gbase, k, j); j = 0; }
  }
}

// Sort (full network up to k = min(n_pad, SORT_CHUNK)) each chunk in LDS.
__global__ void __launch_bounds__(SORT_THREADS)
k_sort_local(uint64_t* __restrict__ keys, int n_pad, int chunk) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain beside MFMA kernels of other streams: win the issue arbitration
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t* s = reinterpret_cast<uint64_t*>(smem_raw);
  const size_t base = (size_t)blockIdx.y * n_pad + (size_t)blockIdx.x * chunk;
  sort_chunk_load<SORT_THREADS>(s, keys + base, chunk);
  __syncthreads();
  const int gbase = blockIdx.x * chunk;
  for (int k = 2; k <= chunk; k <<= 1) sort_lds_step(s, chunk, gbase, k, k >> 1);
  sort_chunk_store<SORT_THREADS>(s, keys + base, chunk);
}

// One global compare-exchange pass (stride j >= chunk) of merge step k.
__global__ void __launch_bounds__(256)
k_sort_global(uint64_t* __restrict__ keys, int n_pad, int k, int j) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain beside MFMA kernels of other streams: win the issue arbitration
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_pad / 2) return;
  uint64_t* kb = keys + (size_t)blockIdx.y * n_pad;
  const int i = 2 * t - (t & (j - 1));
  uint64_t a = kb[i], b = kb[i + j];
  const bool asc = ((i & k) == 0);
  if ((a > b) == asc) { kb[i] = b; kb[i + j] = a; }
}

// S consecutive global compare-exchange passes of merge step k (strides j, j/2, ..., j >> (S-1), all >= chunk) in ONE launch:
// a thread owns the 2^S keys that differ in exactly those S index bits and runs the S passes on them in registers.  The
// proposal sort of 65 536 keys needs 10 global passes; with up to 4 per launch they are 4 launches instead of 10 on the
// latency chain the main stream waits for.
template <int S>
__global__ void __launch_bounds__(256)
k_sort_global_multi(uint64_t* __restrict__ keys, int n_pad, int k, int j) {
  __builtin_amdgcn_s_setprio(3);
  constexpr int E = 1 << S;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_pad / E) return;
  uint64_t* kb = keys + (size_t)blockIdx.y * n_pad;
  const int jl = j >> (S - 1);                          // lowest stride of the group
  const int low = t & (jl - 1);
  const int base = ((t - low) << S) | low;              // index with the S bits log2(jl) .. log2(j) clear
  const bool asc = ((base & k) == 0);                   // bit k lies above every stride of the step
  uint64_t v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) v[e] = kb[base + e * jl];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int bit = 1 << (S - 1 - s);                   // element-index bit of stride j >> s
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (!(e & bit)) cmp_swap(v[e], v[e | bit], asc);
  }
#pragma unroll
  for (int e = 0; e < E; ++e) kb[base + e * jl] = v[e];
}

// Finish merge step k inside each chunk (strides chunk/2 .. 1) in LDS.
__global__ void __launch_bounds__(SORT_THREADS)
k_sort_merge_local(uint64_t* __restrict__ keys, int n_pad, int chunk, int k) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain beside MFMA kernels of other streams: win the issue arbitration
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t* s = reinterpret_cast<uint64_t*>(smem_raw);
  const size_t base = (size_t)blockIdx.y * n_pad + (size_t)blockIdx.x * chunk;
  sort_chunk_load<SORT_THREADS>(s, keys + base, chunk);
  __syncthreads();
  sort_lds_step(s, chunk, blockIdx.x * chunk, k, chunk >> 1);       // k > chunk: one direction for the whole chunk
  sort_chunk_store<SORT_THREADS>(s, keys + base, chunk);
}

int lmh_sort_u64_impl(uint64_t* keys, int B, int n_pad, hipStream_t st) {
  if (n_pad <= 1) return LMH_OK;
  const int chunk = n_pad < SORT_CHUNK ? n_pad : SORT_CHUNK;
  const size_t lds = (size_t)SORT_LDS_KEYS(chunk) * sizeof(uint64_t);
  dim3 gl(n_pad / chunk, B);
  lmh_launch(k_sort_local, gl, dim3(SORT_THREADS), lds, st, keys, n_pad, chunk);
  for (int k = chunk * 2; k <= n_pad; k <<= 1) {
    int j = k >> 1;
    while (j >= chunk) {
      int ns = 0;                                       // passes left in this step: strides j, j/2, ..., chunk
      for (int q = j; q >= chunk; q >>= 1) ++ns;
      const int S = ns >= 4 ? 4 : ns;
      dim3 gg((n_pad / (1 << S) + 255) / 256, B);
      if (S == 4) lmh_launch(k_sort_global_multi<4>, gg, dim3(256), 0, st, keys, n_pad, k, j);
      else if (S == 3) lmh_launch(k_sort_global_multi<3>, gg, dim3(256), 0, st, keys, n_pad, k, j);
      else if (S == 2) lmh_launch(k_sort_global_multi<2>, gg, dim3(256), 0, st, keys, n_pad, k, j);
      else lmh_launch(k_sort_global, gg, dim3(256), 0, st, keys, n_pad, k, j);
      j >>= S;
    }
    lmh_launch(k_sort_merge_local, gl, dim3(SORT_THREADS), lds, st, keys, n_pad, chunk, k);
  }
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

extern "C" int lmh_sort_u64(uint64_t* keys, int B, int n_pad, lmh_stream_t stream) {
  LMH_CHECK_ARG(keys != nullptr && B > 0 && n_pad > 0);
  LMH_CHECK_ARG((n_pad & (n_pad - 1)) == 0);
  return lmh_sort_u64_impl(keys, B, n_pad, (hipStream_t)stream);
}

// ----------------------------------------------------------------------------
// 3. gather the top-k (sorted) boxes + scores
// ----------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_gather_topk(const uint64_t* __restrict__ keys, const float4* __restrict__ boxes,
              const float* __restrict__ cls_prob, const int32_t* __restrict__ n_valid, int N,
              int Npad, int K, float4* __restrict__ top_boxes, float* __restrict__ top_scores,
              int32_t* __restrict__ top_count) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain beside MFMA kernels of other streams: win the issue arbitration
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int cnt = min(n_valid[b], K);
  if (i == 0) top_count[b] = cnt;
  if (i >= K) return;
  float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
  float sc = 0.f;
  if (i < cnt) {
    const uint32_t n = (uint32_t)(keys[(size_t)b * Npad + i] & 0xFFFFFFFFull);
    bx = boxes[(size_t)b * N + n];
    sc = cls_prob[((size_t)b * N + n) * 2 + 1];
  }
  top_boxes[(size_t)b * K + i] = bx;
  top_scores[(size_t)b * K + i] = sc;
}

// ----------------------------------------------------------------------------
// 4. NMS (tf.image.non_max_suppression: greedy over the score-sorted list, stops after max_out keeps).
//    TF IOUGreaterThanThreshold: min/max-normalised corners, continuous areas (no +1), IoU := inter/(a_i+a_j-inter),
//    suppressed iff IoU > thr (strict), never when either area <= 0.
//    Phase 1: 64x64 suppression bit tiles (upper triangle); phase 2: greedy reduce, one block per image.
//
//    Optional two STAGES (round 4; lmh_set_option("nms_stage_mult", m), default 0 = one stage).  The scan stops after
//    max_out keeps and never looks at the rest of the K x K mask (72 M IoUs per image at 12 000 candidates, ~120 us of
//    vector-ALU work on the whole chip).  Stage A = mask + scan of the first m x max_out candidates, stage B = the rest;
//    both launches of B return at once when A ended with `done`.  Same decisions, bit for bit.  On anchor-like boxes
//    (scripts/bench_nms.py: 2000 keeps inside the first 5 000 candidates) m = 3 takes the pair from 254 to 183 us alone;
//    inside the train step of a randomly initialised network the scan needs MORE than 6 000 candidates, both stages run,
//    and the step is 0.05-0.07 ms SLOWER (fp32 6.95 -> 7.00-7.02 ms, f16 4.21 -> 4.23, A/B on one box,
//    scripts/ab.sh "" "LMH_OPT_NMS_STAGE_MULT=2") — hence off by default; a trained RPN (sharper scores) is the case it is kept for.
//    (A finer version, one pair of launches per 1024-candidate super-chunk computing only the kept rows x the columns
//    reached — 10x fewer IoUs — put 24 small dependent launches on the proposal stream, each queued behind the resident
//    MFMA grids of the other streams: the chain got 0.67 ms LONGER inside the step; deleted.)
// ----------------------------------------------------------------------------
struct nms_state { int32_t total, done; };

// v_max_f32 / v_min_f32 as they are: fmaxf / fminf on a value that comes out of LDS compile to a canonicalising
// v_max_f32 x, x, x in front of the operation (four extra instructions per pair in k_nms_mask's loop; ISA check)
__device__ __forceinline__ float nms_vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float nms_vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__global__ void __launch_bounds__(64)
k_nms_mask(const float4* __restrict__ boxes, const int32_t* __restrict__ counts, int K, int W,
           float thr, int cb0, const nms_state* __restrict__ state, uint64_t* __restrict__ mask) {
  // column blocks cb0 .. cb0 + gridDim.x - 1, row blocks 0 .. gridDim.y - 1; `state` (stage B): nothing to do when done
  const int b = blockIdx.z, rb = blockIdx.y, cb = cb0 + blockIdx.x;
  if (cb < rb) return;
  if (state && state[b].done) return;
  const int cnt = counts[b];
  if (rb * 64 >= cnt || cb * 64 >= cnt) return;
  __shared__ float4 cbox[64];
  __shared__ float carea[64];
  const int lane = threadIdx.x;
  const float4* bb = boxes + (size_t)b * K;
  {
    const int c = cb * 64 + lane;
    float4 v = (c < cnt) ? bb[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 nrm = make_float4(fminf(v.x, v.z), fminf(v.y, v.w), fmaxf(v.x, v.z), fmaxf(v.y, v.w));
    cbox[lane] = nrm;
    // a column that can never be suppressed (past the list, or area <= 0) carries a NaN area: its union, and with it
    // every comparison below, is then false whatever the threshold — no test of its own inside the loop
    const float a = (nrm.w - nrm.y) * (nrm.z - nrm.x);
    carea[lane] = (c < cnt && a > 0.f) ? a : __builtin_nanf("");
  }
  __syncthreads();
  const int r = rb * 64 + lane;
  uint32_t wlo = 0, whi = 0;
  if (r < cnt) {
    const float4 v = bb[r];
    const float x1 = fminf(v.x, v.z), y1 = fminf(v.y, v.w), x2 = fmaxf(v.x, v.z), y2 = fmaxf(v.y, v.w);
    const float area_r = (y2 - y1) * (x2 - x1);
    if (area_r > 0.f) {
      // TF decides on fl(inter / uni) > thr.  The correctly rounded division is most of the vector-ALU work of a pair
      // (72 M pairs per image at 12 000 candidates) and only matters within a few ulp of the threshold: pairs clearly on
      // one side are decided by a product, and the division sits behind a branch that is taken only when some lane of the
      // wave is that close — practically never.  (Round 3 wrote it as `if (sure) .. else if (close && division)`, which
      // compiled to the division for every lane that was not `sure`, i.e. in every iteration; ISA check, round 4.)
      // Same decisions, bit for bit.  Diagonal tiles carry both triangles (IoU is symmetric): lane i of the scan reads
      // "who suppresses me" (bits j < i) and "whom I suppress" (bits j > i) from the same word; the self pair is cleared
      // after the loop.
#pragma unroll 16
      for (int j = 0; j < 64; ++j) {
        const float4 c = cbox[j];
        const float area_c = carea[j];
        const float iy1 = nms_vmax(y1, c.y), ix1 = nms_vmax(x1, c.x);
        const float iy2 = nms_vmin(y2, c.w), ix2 = nms_vmin(x2, c.z);
        const float inter = fmaxf(iy2 - iy1, 0.f) * fmaxf(ix2 - ix1, 0.f);
        const float uni = (area_r + area_c) - inter;      // > 0 for a real column: both areas are, and inter <= min(area)
        const float p = thr * uni;
        bool bit = inter > p * 1.000001f;
        if (__builtin_expect(!bit && inter >= p * 0.999999f, 0)) bit = inter / uni > thr;
        const uint32_t m = bit ? (1u << (j & 31)) : 0u;
        if (j < 32) wlo |= m; else whi |= m;
      }
      if (cb == rb) { if (lane < 32) wlo &= ~(1u << lane); else whi &= ~(1u << (lane - 32)); }
    }
  }
  if (r < K) mask[((size_t)b * K + r) * W + cb] = ((uint64_t)whi << 32) | wlo;
}

//    Phase 2: greedy reduce, one 1024-thread block per image, in SUPER-CHUNKS of 16 mask words (1024 candidates).
//    The greedy scan is a dependent chain over the candidates; what round 1 paid for was not that chain but a global
//    round trip per 64-candidate chunk: after every chunk the kept rows were OR-ed into ALL remaining words of the
//    removed-bitmap (188 words at 12 000 candidates) although the scan stops after max_out (2000) keeps — 377 us, the
//    longest kernel of the proposal chain the main stream waits for.  Rounds 2-3 (k_nms_reduce, deleted in round 4; git
//    history): (1) thread t owns row r0 + t of the current super-chunk and reads that row's 16 diagonal-block words;
//    (2) the removed-words of the super-chunk start as the OR of those 16 words over every row kept so far (one parallel
//    gather per super-chunk, not per chunk); (3) the 16 chunks are resolved back to back without touching global memory by
//    ONE wave that reads the rows' words out of LDS (word-major), the greedy keep inside a chunk being a parallel fixed
//    point on the symmetric diagonal word (2-4 ballots instead of a 64-step serial scan).
#define NMS_RED_THREADS 1024
#define NMS_SC_WORDS 16
#define NMS_LDS_KEEP 2048
#define NMS_MAX_K (64 * 65535)  // grid.y of k_nms_mask; the mask itself is K*K/8 bytes of the caller's workspace
//    Round 4: the same scan as a PIPELINE over the super-chunks (k_nms_reduce_p).
//    What bounded k_nms_reduce at the train-step size (12 super-chunks, all of them scanned when the RPN is untrained:
//    215 us on an idle chip) was not the greedy chain but, per super-chunk, three global round trips issued one after
//    the other by ONE block — the 1024 row reads (a lane per row: 64 different cache lines per wave instruction), the
//    gather of every kept row's 16 words (<= 2000 rows, 31 dependent-free loads per thread in batches of 8), and only
//    then the 16 chunk hand-shakes.  Here
//      (a) the rows of super-chunk s + 1 are requested right before super-chunk s is resolved and stay in registers until
//          it is (8 lanes per row: a wave instruction reads 8 whole 128-byte rows);
//      (b) the words of super-chunk s + 1 in the rows kept BEFORE super-chunk s — known when s starts, and almost all of
//          the gather — are OR-ed by eleven waves that have no part in resolving s, into a second bitmap (remN);
//          what is left for the critical path is the gather over the rows kept IN s (a few hundred at most: one round trip);
//      (c) wave 0 folds the kept rows' word c + 1 into rem[c + 1] itself (LDS atomics of one wave run in order with its
//          later read: no hand-shake between a chunk and the next); four waves fold the words from c + 2 on and publish
//          `word j complete up to chunk j - 2`, which wave 0 finds set when it gets there.
//    Same decisions, bit for bit (tests/test_gpu_kernels.py, the reference fixtures of tests/test_gpu_ref_tf_golden.py).
// OR of a 64-bit value over the lanes of a wave into one LDS word: one ds_or_b64 per lane that has something to add (the
// LDS unit serialises the lanes of a same-address atomic itself, a handful of cycles for the ~10 kept rows of a chunk).
// The offset is an opaque zero in a vector register: with a provably wave-uniform address the compiler's atomic optimizer
// replaces the instruction by a scalar loop over all 64 lanes (v_readlane + s_or per lane: ISA check), several times slower.
__device__ __forceinline__ void nms_wave_or(unsigned long long* dst, uint64_t v) {
  int z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  if (v) atomicOr(dst + z, (unsigned long long)v);
}

#define NMS_SDT_STRIDE (NMS_RED_THREADS + 1)    // u64 per word plane: the (row, word-pair) writes of (1b) hit 16 bank pairs
#define NMS_FOLD_WAVES 4

template <bool WIDE>   // WIDE: W even, so a row's word pairs are 16-byte aligned and read as one dwordx4
__device__ __forceinline__ void nms_load2(const uint64_t* __restrict__ rowp, int pair, int last, uint64_t& a, uint64_t& b) {
  // words 2 pair, 2 pair + 1 of the 16-word window at rowp; indices past `last` (the last word of the window that exists)
  // are clamped: their values are discarded by the caller
  if (WIDE) {
    const int pc = min(pair, last >> 1);
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(rowp + 2 * pc);
    a = v.x; b = v.y;
  } else {
    a = rowp[min(2 * pair, last)];
    b = rowp[min(2 * pair + 1, last)];
  }
}

// OR of words [wbase, wbase + nwin) over the keep-list rows [lo, hi): thread `t` of `nt` (a multiple of 8) takes word pair
// t & 7 of every (nt / 8)-th row, eight independent 16-byte loads in flight; lanes with the same pair are folded and lanes
// 0..7 of each wave add the result to dst[0..15] (LDS atomics).  Clamped row indices repeat a row: harmless under OR.
template <bool WIDE>
__device__ __forceinline__ void nms_gather_or(const uint64_t* __restrict__ mb, int W, int wbase, int nwin,
                                              const int32_t* __restrict__ kl, int lo, int hi, int t, int nt,
                                              unsigned long long* dst) {
  if (lo >= hi) return;                          // block-uniform
  const int p = t & 7, stride = nt >> 3, lane = threadIdx.x & 63;
  uint64_t a0 = 0ull, a1 = 0ull;
  for (int base = lo + (t >> 3); base < hi; base += 8 * stride) {
    uint64_t x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kr = kl[min(base + u * stride, hi - 1)];
      nms_load2<WIDE>(mb + (size_t)kr * W + wbase, p, nwin - 1, x[u], y[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { a0 |= x[u]; a1 |= y[u]; }
  }
  if (2 * p >= nwin) a0 = 0ull;
  if (2 * p + 1 >= nwin) a1 = 0ull;
  uint32_t v0 = (uint32_t)a0, v1 = (uint32_t)(a0 >> 32), v2 = (uint32_t)a1, v3 = (uint32_t)(a1 >> 32);
#pragma unroll
  for (int m = 8; m < 64; m <<= 1) {
    v0 |= __shfl_xor(v0, m); v1 |= __shfl_xor(v1, m); v2 |= __shfl_xor(v2, m); v3 |= __shfl_xor(v3, m);
  }
  if (lane < 8) {
    a0 = ((uint64_t)v1 << 32) | v0;
    a1 = ((uint64_t)v3 << 32) | v2;
    if (a0) atomicOr(&dst[2 * p], (unsigned long long)a0);
    if (a1) atomicOr(&dst[2 * p + 1], (unsigned long long)a1);
  }
}

// LDS flags of the scan's hand-shakes: relaxed atomics on the __shared__ objects themselves compile to ds_read / ds_write
// (a `volatile` access through a cast pointer becomes a FLAT load with a vmcnt(0) wait, which would also wait for the
// row prefetch in flight); a wave's LDS operations execute in order, so a flag written after the data is seen after it
__device__ __forceinline__ int nms_lds_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned long long nms_lds_ld(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void nms_lds_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void nms_lds_st(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <bool WIDE>
__global__ void __launch_bounds__(NMS_RED_THREADS)
k_nms_reduce_p(const uint64_t* __restrict__ mask, const int32_t* __restrict__ counts, int K, int W,
               int max_out, int sc_begin, int sc_end, nms_state* __restrict__ state, int32_t* __restrict__ keep_idx,
               int32_t* __restrict__ keep_count) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ __attribute__((aligned(16))) unsigned long long rem[NMS_SC_WORDS];    // removed-bits of the current super-chunk
  __shared__ __attribute__((aligned(16))) unsigned long long remN[NMS_SC_WORDS];   // ... of the next one, from the rows kept before this one
  __shared__ unsigned long long s_keptm[NMS_SC_WORDS];
  __shared__ int s_ready, s_folded[NMS_SC_WORDS], s_tot;
  __shared__ int32_t s_kidx[NMS_LDS_KEEP];
  __shared__ __attribute__((aligned(16))) unsigned long long sdT[NMS_SC_WORDS * NMS_SDT_STRIDE];   // [word][row]
  const bool lds_keep = max_out <= NMS_LDS_KEEP;
  const int b = blockIdx.x;
  if (sc_begin > 0 && state[b].done) return;
  const int cnt = min(counts[b], K);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t* mb = mask + (size_t)b * K * W;
  int32_t* kidx = keep_idx + (size_t)b * max_out;
  int total = 0;
  if (sc_begin == 0) {
    for (int i = tid; i < max_out; i += NMS_RED_THREADS) kidx[i] = -1;
  } else {
    total = state[b].total;
    if (lds_keep)
      for (int i = tid; i < total; i += NMS_RED_THREADS) s_kidx[i] = kidx[i];
  }
  if (tid < NMS_SC_WORDS) remN[tid] = 0ull;
  const int nchunks = (cnt + 63) / 64;
  const int nsc_all = (nchunks + NMS_SC_WORDS - 1) / NMS_SC_WORDS;
  const int nsc = min(sc_end, nsc_all);
  // (1) row loads: instruction i of a wave covers rows 64 wave + 8 i .. + 7 of the super-chunk, 8 lanes (word pairs) per row
  const int lrow = lane >> 3, lp = lane & 7;
  uint64_t d0[8], d1[8];
#define NMS_LOAD_ROWS(sc_)                                                                          \
  do {                                                                                              \
    const int w0_ = (sc_) * NMS_SC_WORDS, nw_ = min(NMS_SC_WORDS, nchunks - w0_);                   \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                 \
      const int row_ = min((sc_) * 64 * NMS_SC_WORDS + wave * 64 + 8 * i + lrow, cnt - 1);          \
      nms_load2<WIDE>(mb + (size_t)row_ * W + w0_, lp, nw_ - 1, d0[i], d1[i]);                      \
    }                                                                                               \
  } while (0)
  if (sc_begin < nsc) NMS_LOAD_ROWS(sc_begin);
  int pre_hi = 0;            // keep-list rows [0, pre_hi) are already folded into remN for the super-chunk about to start
  for (int sc = sc_begin; sc < nsc; ++sc) {
    const int r0 = sc * 64 * NMS_SC_WORDS, w0 = sc * NMS_SC_WORDS;
    const int nw = min(NMS_SC_WORDS, nchunks - w0);
    // (1b) rows -> LDS, word-major; words that do not exist (past the list, below the diagonal block) as zeros
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rl = wave * 64 + 8 * i + lrow, row = r0 + rl;
      const bool ok0 = row < cnt && 2 * lp < nw && (w0 + 2 * lp) >= (row >> 6);
      const bool ok1 = row < cnt && 2 * lp + 1 < nw && (w0 + 2 * lp + 1) >= (row >> 6);
      sdT[(2 * lp) * NMS_SDT_STRIDE + rl] = ok0 ? d0[i] : 0ull;
      sdT[(2 * lp + 1) * NMS_SDT_STRIDE + rl] = ok1 ? d1[i] : 0ull;
    }
    if (tid < NMS_SC_WORDS) { rem[tid] = remN[tid]; remN[tid] = 0ull; s_folded[tid] = 0; }
    if (tid == 0) s_ready = 0;
    __syncthreads();
    // (2) what the look-ahead of the previous round could not know: the rows kept since then
    if (lds_keep) nms_gather_or<WIDE>(mb, W, w0, nw, s_kidx, pre_hi, total, tid, NMS_RED_THREADS, rem);
    else nms_gather_or<WIDE>(mb, W, w0, nw, kidx, pre_hi, total, tid, NMS_RED_THREADS, rem);
    const bool more = sc + 1 < nsc;
    if (more) NMS_LOAD_ROWS(sc + 1);          // (a) in flight while this super-chunk is resolved
    __syncthreads();
    // (3) the 16 chunks
    if (wave == 0) {
      int tot = total;
      uint64_t diag = sdT[lane];                                                  // chunk 0: word 0 of row (0, lane)
      uint64_t nxt = nw > 1 ? sdT[NMS_SDT_STRIDE + lane] : 0ull;                  // ... and its word 1
      for (int c = 0; c < nw; ++c) {
        uint64_t diag_n = 0ull, nxt_n = 0ull;          // the next chunk's operands do not depend on this chunk's outcome
        if (c + 1 < nw) diag_n = sdT[(c + 1) * NMS_SDT_STRIDE + (c + 1) * 64 + lane];
        if (c + 2 < nw) nxt_n = sdT[(c + 2) * NMS_SDT_STRIDE + (c + 1) * 64 + lane];
        if (c >= 2)                                    // chunks 0 .. c-2 folded into word c by its owner; c-1 by this wave
          while (nms_lds_ld(&s_folded[c]) == 0) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");                 // rem[c] is read after the flag, not before
        const int nin = min(64, cnt - (w0 + c) * 64);
        uint64_t alive_v = ~nms_lds_ld(&rem[c]);
        if (nin < 64) alive_v &= ((1ull << nin) - 1ull);
        const uint64_t below = (1ull << lane) - 1ull;
        const bool me_alive = (alive_v >> lane) & 1ull;
        uint64_t kept = __ballot(me_alive);            // greedy keep inside the chunk as a parallel fixed point (above)
        for (int it = 0; it < 64; ++it) {
          const uint64_t next = __ballot(me_alive && (diag & below & kept) == 0ull);
          if (next == kept) break;
          kept = next;
        }
        const int room = max_out - tot;
        if (__popcll(kept) > room)
          kept = __ballot(((kept >> lane) & 1ull) && __popcll(kept & below) < room);
        const bool mine = (kept >> lane) & 1ull;
        if (c + 1 < nw) nms_wave_or(&rem[c + 1], mine ? nxt : 0ull);      // in order with this wave's next read of it
        if (lane == 0) {
          nms_lds_st(&s_keptm[c], (unsigned long long)kept);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          nms_lds_st(&s_ready, c + 1);
        }
        if (mine) {
          const int slot = tot + __popcll(kept & below);
          kidx[slot] = (w0 + c) * 64 + lane;
          if (lds_keep) s_kidx[slot] = (w0 + c) * 64 + lane;
        }
        tot += __popcll(kept);
        if (tot >= max_out) break;
        diag = diag_n; nxt = nxt_n;
      }
      if (!lds_keep) __threadfence();     // the gathers of the next rounds read the keep list from global memory
      if (lane == 0) {
        s_tot = tot;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        nms_lds_st(&s_ready, -1);
      }
    } else if (wave <= NMS_FOLD_WAVES) {
      // wave f owns words 1 + f, 1 + f + 4, ...: after chunk c is published it folds the kept rows' owned words >= c + 2,
      // the one wave 0 needs first (c + 2) first, and then says so
      for (int c = 0; c + 2 < nw; ++c) {
        int jf = 1 + wave;
        while (jf < c + 2) jf += NMS_FOLD_WAVES;
        uint64_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int j = jf + NMS_FOLD_WAVES * k;
          v[k] = j < nw ? sdT[j * NMS_SDT_STRIDE + c * 64 + lane] : 0ull;
        }
        int rdy;
        while ((rdy = nms_lds_ld(&s_ready)) >= 0 && rdy <= c) __builtin_amdgcn_s_sleep(1);
        if (rdy < 0) break;
        asm volatile("" ::: "memory");
        const uint64_t k = nms_lds_ld(&s_keptm[c]);
        const bool mine = (k >> lane) & 1ull;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = jf + NMS_FOLD_WAVES * q;
          if (j < nw) nms_wave_or(&rem[j], mine ? v[q] : 0ull);
        }
        if (jf == c + 2) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) nms_lds_st(&s_folded[jf], 1);
        }
      }
    } else if (more) {
      // (b) look-ahead: the next super-chunk's words in every row kept before this one
      const int w0n = w0 + NMS_SC_WORDS;
      const int nwn = min(NMS_SC_WORDS, nchunks - w0n), pt = tid - 64 * (NMS_FOLD_WAVES + 1);
      constexpr int NP = NMS_RED_THREADS - 64 * (NMS_FOLD_WAVES + 1);
      if (lds_keep) nms_gather_or<WIDE>(mb, W, w0n, nwn, s_kidx, 0, total, pt, NP, remN);
      else nms_gather_or<WIDE>(mb, W, w0n, nwn, kidx, 0, total, pt, NP, remN);
    }
    pre_hi = total;
    __syncthreads();
    total = s_tot;
    if (total >= max_out) break;
  }
#undef NMS_LOAD_ROWS
  if (tid == 0) {
    keep_count[b] = total;
    state[b].total = total;
    state[b].done = (total >= max_out || nsc >= nsc_all) ? 1 : 0;
  }
}

extern "C" size_t lmh_nms_workspace_bytes(int B, int K) {
  const size_t W = (size_t)(K + 63) / 64;
  return lmh_align_up((size_t)B * K * W * sizeof(uint64_t), 256) + lmh_align_up(sizeof(nms_state) * (size_t)B, 256);
}

static void nms_reduce_launch(int B, hipStream_t st, const uint64_t* mask, const int32_t* counts, int K, int W, int max_out,
                              int sc_begin, int sc_end, nms_state* state, int32_t* keep_idx, int32_t* keep_count) {
  if ((W & 1) || ((uintptr_t)mask & 15))      // odd row length or a mask that is not 16-byte aligned: 8-byte loads
    lmh_launch(k_nms_reduce_p<false>, dim3(B), dim3(NMS_RED_THREADS), 0, st, mask, counts, K, W, max_out, sc_begin, sc_end,
               state, keep_idx, keep_count);
  else
    lmh_launch(k_nms_reduce_p<true>, dim3(B), dim3(NMS_RED_THREADS), 0, st, mask, counts, K, W, max_out, sc_begin, sc_end,
               state, keep_idx, keep_count);
}

int lmh_nms_impl(const float* boxes, const int32_t* counts, int B, int K, float thr, int max_out,
                    int32_t* keep_idx, int32_t* keep_count, void* ws, hipStream_t st) {
  LMH_CHECK_ARG(K > 0 && K <= NMS_MAX_K);
  const int W = (K + 63) / 64;
  uint64_t* mask = reinterpret_cast<uint64_t*>(ws);
  nms_state* state = reinterpret_cast<nms_state*>(reinterpret_cast<char*>(ws) +
                                                  lmh_align_up((size_t)B * K * W * sizeof(uint64_t), 256));
  const float4* b4 = reinterpret_cast<const float4*>(boxes);
  // stage A: the first R1 candidates (whole super-chunks); stage B: the rest, skipped on the device when A was enough
  const int nsc_all = (W + NMS_SC_WORDS - 1) / NMS_SC_WORDS;
  const int mult = lmh_opt("nms_stage_mult");        // stage A covers mult x max_out candidates; 0: one stage (rounds 1-3)
  int sc1 = mult > 0 ? (mult * max_out + 64 * NMS_SC_WORDS - 1) / (64 * NMS_SC_WORDS) : nsc_all;
  if (sc1 < 1) sc1 = 1;
  if (sc1 > nsc_all) sc1 = nsc_all;
  const int W1 = sc1 * NMS_SC_WORDS < W ? sc1 * NMS_SC_WORDS : W;
  lmh_launch(k_nms_mask, dim3(W1, W1, B), dim3(64), 0, st, b4, counts, K, W, thr, 0, (const nms_state*)nullptr, mask);
  nms_reduce_launch(B, st, (const uint64_t*)mask, counts, K, W, max_out, 0, sc1, state, keep_idx, keep_count);
  if (W1 < W) {
    lmh_launch(k_nms_mask, dim3(W - W1, W, B), dim3(64), 0, st, b4, counts, K, W, thr, W1, (const nms_state*)state, mask);
    nms_reduce_launch(B, st, (const uint64_t*)mask, counts, K, W, max_out, sc1, nsc_all, state, keep_idx, keep_count);
  }
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

extern "C" int lmh_nms(const float* boxes, const int32_t* counts, int B, int K, float iou_threshold,
                       int max_out, int32_t* keep_idx, int32_t* keep_count, void* ws,
                       size_t ws_bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(boxes && counts && keep_idx && keep_count && ws);
  LMH_CHECK_ARG(B > 0 && K > 0 && max_out > 0 && K <= NMS_MAX_K);
  if (ws_bytes < lmh_nms_workspace_bytes(B, K)) {
    lmh_set_error("lmh_nms: workspace %zu < %zu", ws_bytes, lmh_nms_workspace_bytes(B, K));
    return LMH_ERR_WORKSPACE;
  }
  return lmh_nms_impl(boxes, counts, B, K, iou_threshold, max_out, keep_idx, keep_count, ws,
                  (hipStream_t)stream);
}

// ----------------------------------------------------------------------------
// 5. final gather of kept proposals (+ optional clip after NMS)
// ----------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_gather_keep(const float4* __restrict__ top_boxes, const float* __restrict__ top_scores,
              const int32_t* __restrict__ keep_idx, const int32_t* __restrict__ keep_count, int K,
              int max_out, int clip_after, float im_h, float im_w, float4* __restrict__ proposals,
              float* __restrict__ scores, int32_t* __restrict__ num) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound chain beside MFMA kernels of other streams: win the issue arbitration
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) num[b] = keep_count[b];
  if (i >= max_out) return;
  float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
  float sc = 0.f;
  if (i < keep_count[b]) {
    const int k = keep_idx[(size_t)b * max_out + i];
    bx = top_boxes[(size_t)b * K + k];
    sc = top_scores[(size_t)b * K + k];
    if (clip_after) {
      lmh_box c = lmh_clip(lmh_box{bx.x, bx.y, bx.z, bx.w}, im_h, im_w);
      bx = make_float4(c.x1, c.y1, c.x2, c.y2);
    }
  }
  proposals[(size_t)b * max_out + i] = bx;
  scores[(size_t)b * max_out + i] = sc;
}

__global__ void k_iota_keep(const int32_t* __restrict__ top_count, int max_out,
                            int32_t* __restrict__ keep_idx, int32_t* __restrict__ keep_count) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = min(top_count[b], max_out);
  if (i == 0) keep_count[b] = c;
  if (i < max_out) keep_idx[(size_t)b * max_out + i] = (i < c) ? i : -1;
}

struct rpn_prop_ws {
  float4* boxes;      // B*N
  uint64_t* keys;     // B*Npad
  int32_t* n_valid;   // B
  float4* top_boxes;  // B*K
  float* top_scores;  // B*K
  int32_t* top_count; // B
  int32_t* keep_idx;  // B*post
  int32_t* keep_count;// B
  void* nms_ws;
  size_t total;
};

static rpn_prop_ws rpn_prop_layout(const lmh_rpn_proposal_desc* d, void* base) {
  const size_t B = d->B, N = (size_t)d->feat_h * d->feat_w * d->A;
  const size_t Npad = lmh_next_pow2((int)N), K = d->pre_nms_top_n;
  const size_t P = d->apply_nms ? d->post_nms_top_n : d->pre_nms_top_n;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += lmh_align_up(bytes, 256); return o; };
  rpn_prop_ws w;
  char* p = reinterpret_cast<char*>(base);
  w.boxes = reinterpret_cast<float4*>(p + take(B * N * 16));
  w.keys = reinterpret_cast<uint64_t*>(p + take(B * Npad * 8));
  w.n_valid = reinterpret_cast<int32_t*>(p + take(B * 4));
  w.top_boxes = reinterpret_cast<float4*>(p + take(B * K * 16));
  w.top_scores = reinterpret_cast<float*>(p + take(B * K * 4));
  w.top_count = reinterpret_cast<int32_t*>(p + take(B * 4));
  w.keep_idx = reinterpret_cast<int32_t*>(p + take(B * P * 4));
  w.keep_count = reinterpret_cast<int32_t*>(p + take(B * 4));
  w.nms_ws = p + take(lmh_nms_workspace_bytes((int)B, (int)K));
  w.total = off;
  return w;
}

extern "C" size_t lmh_rpn_proposal_workspace_bytes(const lmh_rpn_proposal_desc* d) {
  if (!d) return 0;
  return rpn_prop_layout(d, nullptr).total;
}

extern "C" int lmh_rpn_proposal(const lmh_rpn_proposal_desc* d, const float* cls_score,
                                const float* bbox_pred, const int32_t* anchor_ref, float* cls_prob,
                                float* proposals, float* scores, int32_t* num_proposals, void* ws,
                                size_t ws_bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(d && cls_score && bbox_pred && anchor_ref && cls_prob && proposals && scores &&
                num_proposals && ws);
  LMH_CHECK_ARG(d->B > 0 && d->feat_h > 0 && d->feat_w > 0 && d->A > 0);
  LMH_CHECK_ARG(d->pre_nms_top_n > 0 && d->post_nms_top_n > 0 && d->pre_nms_top_n <= NMS_MAX_K);
  const int N = d->feat_h * d->feat_w * d->A;
  const int Npad = lmh_next_pow2(N);
  rpn_prop_ws w = rpn_prop_layout(d, ws);
  if (ws_bytes < w.total) {
    lmh_set_error("lmh_rpn_proposal: workspace %zu < %zu", ws_bytes, w.total);
    return LMH_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int B = d->B, K = d->pre_nms_top_n;
  const int P = d->apply_nms ? d->post_nms_top_n : d->pre_nms_top_n;  // no post cap without NMS (rpn_proposal.py:172-174)
  LMH_CHECK_HIP(lmh_memset_async(w.n_valid, 0, sizeof(int32_t) * B, st));
  lmh_launch(k_rpn_decode, dim3((Npad + 255) / 256, B), dim3(256), 0, st, *d, N, Npad,
                     cls_score, bbox_pred, anchor_ref, cls_prob, w.boxes, w.keys, w.n_valid);
  int rc = lmh_sort_u64_impl(w.keys, B, Npad, st);
  if (rc) return rc;
  lmh_launch(k_gather_topk, dim3((K + 255) / 256, B), dim3(256), 0, st, w.keys, w.boxes,
                     cls_prob, w.n_valid, N, Npad, K, w.top_boxes, w.top_scores, w.top_count);
  if (d->apply_nms) {
    rc = lmh_nms_impl(reinterpret_cast<const float*>(w.top_boxes), w.top_count, B, K, d->nms_threshold,
                  P, w.keep_idx, w.keep_count, w.nms_ws, st);
    if (rc) return rc;
  } else {
    lmh_launch(k_iota_keep, dim3((P + 255) / 256, B), dim3(256), 0, st, w.top_count, P,
                       w.keep_idx, w.keep_count);
  }
  lmh_launch(k_gather_keep, dim3((P + 255) / 256, B), dim3(256), 0, st, w.top_boxes,
                     w.top_scores, w.keep_idx, w.keep_count, K, P, d->clip_after_nms, d->im_h,
                     d->im_w, reinterpret_cast<float4*>(proposals), scores, num_proposals);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
