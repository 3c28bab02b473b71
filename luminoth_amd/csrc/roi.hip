// ROI crop-and-resize + 2x2 max pool, fused (gfx950).
//
// Reference: luminoth/models/fasterrcnn/roi_pool.py:37-95 —
// tf.image.crop_and_resize(feat, boxes/[H,W], crop=[2*pw, 2*ph], bilinear,
// extrapolation 0) followed by tf.nn.max_pool 2x2/2 VALID.  The 2ph x 2pw
// intermediate (205 MB at R=256, C=1024) is never written: each output element
// evaluates its four bilinear samples in registers and keeps the max plus a
// 2-bit argmax for the backward.  NHWC: lanes run along C with float4 loads, so
// every corner fetch is a fully coalesced 1 KiB wave request and the backward's
// scatter-add atomics hit distinct addresses per lane.
#include "lmh_common.h"
int lmh_opt(const char* name);   // api.hip: the lmh_set_option registry
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __attribute__((aligned(16))) const float roi_zero_page[4] = {0.f, 0.f, 0.f, 0.f};   // target of unconditional loads

struct roi_geom {
  float y1n, x1n, hs, ws;  // normalised top-left * (dim-1), per-sample scale
};

// TF 1.x crop_and_resize_op.cc coordinate arithmetic (oracle/tfops.py twin).
__device__ __forceinline__ roi_geom roi_geometry(const float4 roi, float im_h, float im_w, int FH,
                                                 int FW, int ch, int cw) {
  const float y1 = roi.y / im_h, x1 = roi.x / im_w, y2 = roi.w / im_h, x2 = roi.z / im_w;
  roi_geom g;
  g.hs = (ch > 1) ? (y2 - y1) * (float)(FH - 1) / (float)(ch - 1) : 0.f;
  g.ws = (cw > 1) ? (x2 - x1) * (float)(FW - 1) / (float)(cw - 1) : 0.f;
  g.y1n = (ch > 1) ? y1 * (float)(FH - 1) : .5f * (y1 + y2) * (float)(FH - 1);
  g.x1n = (cw > 1) ? x1 * (float)(FW - 1) : .5f * (x1 + x2) * (float)(FW - 1);
  return g;
}

struct roi_sample {
  bool valid;
  int top, bot, left, right;
  float ylerp, xlerp;
};

__device__ __forceinline__ roi_sample roi_sample_at(const roi_geom& g, int y, int x, int FH, int FW,
                                                    int ch, int cw) {
  roi_sample s;
  const float in_y = (ch > 1) ? g.y1n + (float)y * g.hs : g.y1n;
  const float in_x = (cw > 1) ? g.x1n + (float)x * g.ws : g.x1n;
  s.valid = !(in_y < 0.f || in_y > (float)(FH - 1) || in_x < 0.f || in_x > (float)(FW - 1));
  s.top = (int)floorf(in_y);
  s.bot = (int)ceilf(in_y);
  s.left = (int)floorf(in_x);
  s.right = (int)ceilf(in_x);
  s.ylerp = in_y - (float)s.top;
  s.xlerp = in_x - (float)s.left;
  return s;
}

__device__ __forceinline__ float bilerp(float tl, float tr, float bl, float br, float xl, float yl) {
  const float top = tl + (tr - tl) * xl;
  const float bot = bl + (br - bl) * xl;
  return top + (bot - top) * yl;
}

// grid: B*R*ph blocks, one per (ROI, cell row); block: min(C/4, 256) threads, each a float4 of channels,
// looping over the pw cells of the row.  Block ids are mapped so that the ph rows of one ROI land on the same
// XCD (ids 8 apart): neighbouring cells share their bilinear corner rows, which then hit in that XCD's L2 / the
// CU's L1 (the first version launched one block per (ROI, cell) in launch order: 701 MB fetched for a 33.5 MB
// feature map — PMC — and 0.48 ms when running beside the convolution streams).  All 16 corner loads of a cell
// are unconditional (clamped coordinates, value selected afterwards) so they are in flight together.
__global__ void __launch_bounds__(256)
k_roi_pool_fwd(const float* __restrict__ feat, const float4* __restrict__ rois,
               const int32_t* __restrict__ roi_count, int R, int BR, int FH, int FW, int C, float im_h,
               float im_w, int ph, int pw, float* __restrict__ out, uint8_t* __restrict__ argmax) {
  int rr, py;
  {
    const int id = blockIdx.x, n8 = (BR >> 3) << 3;
    if (id < n8 * ph) {
      const int g8 = id >> 3;
      rr = (id & 7) + 8 * (g8 / ph);
      py = g8 % ph;
    } else {
      const int t = id - n8 * ph;
      rr = n8 + t / ph;
      py = t % ph;
    }
  }
  const int b = rr / R, r = rr % R;
  const bool live = r < roi_count[b];
  const int C4 = C >> 2;
  const size_t obase0 = ((size_t)rr * ph + py) * pw * C;
  if (!live) {
    for (int px = 0; px < pw; ++px)
      for (int c4 = threadIdx.x; c4 < C4; c4 += blockDim.x) {
        reinterpret_cast<float4*>(out + obase0 + (size_t)px * C)[c4] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (argmax) reinterpret_cast<uint32_t*>(argmax + obase0 + (size_t)px * C)[c4] = 0u;
      }
    return;
  }
  const int ch = 2 * ph, cw = 2 * pw;  // crop size is passed (w*2, h*2): square in practice
  const roi_geom g = roi_geometry(rois[rr], im_h, im_w, FH, FW, ch, cw);
  const float* fb = feat + (size_t)b * FH * FW * C;
  for (int c4 = threadIdx.x; c4 < C4; c4 += blockDim.x) {
    const float4* fc = reinterpret_cast<const float4*>(fb) + c4;
#pragma unroll 2
    for (int px = 0; px < pw; ++px) {
      const size_t obase = obase0 + (size_t)px * C;
      f32x4 tl[4], tr[4], bl[4], br[4];
      roi_sample s[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s[q] = roi_sample_at(g, 2 * py + (q >> 1), 2 * px + (q & 1), FH, FW, ch, cw);
        const int t = min(max(s[q].top, 0), FH - 1), bo = min(max(s[q].bot, 0), FH - 1);
        const int l = min(max(s[q].left, 0), FW - 1), ri = min(max(s[q].right, 0), FW - 1);
        tl[q] = *reinterpret_cast<const f32x4*>(fc + ((size_t)t * FW + l) * C4);
        tr[q] = *reinterpret_cast<const f32x4*>(fc + ((size_t)t * FW + ri) * C4);
        bl[q] = *reinterpret_cast<const f32x4*>(fc + ((size_t)bo * FW + l) * C4);
        br[q] = *reinterpret_cast<const f32x4*>(fc + ((size_t)bo * FW + ri) * C4);
      }
      float best[4];
      uint32_t am[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = s[q].valid ? bilerp(tl[q][e], tr[q][e], bl[q][e], br[q][e], s[q].xlerp, s[q].ylerp) : 0.f;
          if (q == 0 || v > best[e]) { best[e] = v; am[e] = q; }  // first max wins
        }
      }
      reinterpret_cast<float4*>(out + obase)[c4] = make_float4(best[0], best[1], best[2], best[3]);
      if (argmax)
        reinterpret_cast<uint32_t*>(argmax + obase)[c4] = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24);
    }
  }
}

// Backward: max-pool routes dout to the arg-max sample; CropAndResizeGradImage
// scatters it to the four corners (same (1-lerp) operation order as TF).
__global__ void __launch_bounds__(256)
k_roi_pool_bwd(const float* __restrict__ dout, const uint8_t* __restrict__ argmax,
               const float4* __restrict__ rois, const int32_t* __restrict__ roi_count, int R, int FH,
               int FW, int C, float im_h, float im_w, int ph, int pw, float* __restrict__ dfeat) {
  const int rr = blockIdx.y;
  const int b = rr / R, r = rr % R;
  if (r >= roi_count[b]) return;
  const int cell = blockIdx.x;
  const int py = cell / pw, px = cell % pw;
  const size_t obase = ((size_t)rr * ph * pw + cell) * C;
  const int ch = 2 * ph, cw = 2 * pw;
  const roi_geom g = roi_geometry(rois[rr], im_h, im_w, FH, FW, ch, cw);
  roi_sample s[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) s[q] = roi_sample_at(g, 2 * py + (q >> 1), 2 * px + (q & 1), FH, FW, ch, cw);
  float* fb = dfeat + (size_t)b * FH * FW * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float go = dout[obase + c];
    const int q = argmax[obase + c] & 3;
    // select sample q without dynamic register indexing
    roi_sample sq = s[0];
    if (q == 1) sq = s[1];
    if (q == 2) sq = s[2];
    if (q == 3) sq = s[3];
    if (!sq.valid || go == 0.f) continue;
    const float dtop = (1.f - sq.ylerp) * go;
    const float dbot = sq.ylerp * go;
    unsafeAtomicAdd(fb + ((size_t)sq.top * FW + sq.left) * C + c, (1.f - sq.xlerp) * dtop);
    unsafeAtomicAdd(fb + ((size_t)sq.top * FW + sq.right) * C + c, sq.xlerp * dtop);
    unsafeAtomicAdd(fb + ((size_t)sq.bot * FW + sq.left) * C + c, (1.f - sq.xlerp) * dbot);
    unsafeAtomicAdd(fb + ((size_t)sq.bot * FW + sq.right) * C + c, sq.xlerp * dbot);
  }
}

extern "C" int lmh_roi_pool_fwd(const float* feat, const float* rois, const int32_t* roi_count, int B,
                                int R, int FH, int FW, int C, float im_h, float im_w, int ph, int pw,
                                float* out, uint8_t* argmax, lmh_stream_t stream) {
  LMH_CHECK_ARG(feat && rois && roi_count && out);
  LMH_CHECK_ARG(B > 0 && R > 0 && FH > 0 && FW > 0 && C > 0 && (C % 4) == 0 && ph > 0 && pw > 0);
  const int threads = (C / 4) < 256 ? ((C / 4 + 63) / 64 * 64) : 256;
  lmh_launch(k_roi_pool_fwd, dim3(B * R * ph), dim3(threads), 0, (hipStream_t)stream, feat,
                     reinterpret_cast<const float4*>(rois), roi_count, R, B * R, FH, FW, C, im_h, im_w,
                     ph, pw, out, argmax);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// ---- backward, slab variant (default) --------------------------------------------------------------
// Step 1 (k_roi_sample_table): the four bilinear samples of every (roi, cell) are evaluated ONCE.
// Step 2 (k_roi_pool_bwd_slab): one 1024-thread block owns CS channels of one image's whole feature map
// in LDS ([pixel][CS] 64-bit cells, 128 KiB at 64x64x4).  A thread is one (roi-cell, channel):
// consecutive lanes are the CS channels of a pair (distinct, bank-consecutive addresses).  No global
// atomics, no pre-zeroing: the slab is written once as float4s.
struct __attribute__((aligned(16))) roi_sample_rec {
  int16_t top, left, bot, right;   // top < 0: sample outside the feature map (contributes nothing)
  float ylerp, xlerp;
};

__global__ void __launch_bounds__(256)
k_roi_sample_table(const float4* __restrict__ rois, const int32_t* __restrict__ roi_count, int B, int R, int FH,
                   int FW, float im_h, float im_w, int ph, int pw, roi_sample_rec* __restrict__ table) {
  const int cells = ph * pw;
  const int i = blockIdx.x * 256 + threadIdx.x;      // (b, r, cell)
  if (i >= B * R * cells) return;
  const int rr = i / cells, cell = i - rr * cells;
  const int b = rr / R, r = rr - b * R;
  const int py = cell / pw, px = cell - py * pw;
  roi_sample_rec rec[4];
  if (r < roi_count[b]) {
    const roi_geom g = roi_geometry(rois[rr], im_h, im_w, FH, FW, 2 * ph, 2 * pw);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const roi_sample sq = roi_sample_at(g, 2 * py + (q >> 1), 2 * px + (q & 1), FH, FW, 2 * ph, 2 * pw);
      rec[q].top = sq.valid ? (int16_t)sq.top : (int16_t)-1;
      rec[q].left = (int16_t)sq.left; rec[q].bot = (int16_t)sq.bot; rec[q].right = (int16_t)sq.right;
      rec[q].ylerp = sq.ylerp; rec[q].xlerp = sq.xlerp;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) { rec[q].top = -1; rec[q].left = rec[q].bot = rec[q].right = 0; rec[q].ylerp = rec[q].xlerp = 0.f; }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) table[(size_t)i * 4 + q] = rec[q];
}

// Accumulation is 64-bit FIXED POINT (value * 2^38, ds_add_u64): on gfx950 an LDS float atomic retires
// ~0.33 lane-ops/clk/CU while the integer ones retire >3.5 (scripts/probes/lds_atomic_rate.hip), and integer
// addition is associative, so the result is the exactly-rounded sum of the fp32 terms, bit-identical run to
// run (the float-atomic version depended on arrival order).  Range: |sum| < 2^25, resolution 2^-38.
#define ROI_FX_SCALE 274877906944.0f          /* 2^38 */
#define ROI_FX_INV (1.0 / 274877906944.0)

__device__ __forceinline__ void roi_fx_add(unsigned long long* cell, float v) {
  atomicAdd(cell, (unsigned long long)(long long)__float2ll_rn(v * ROI_FX_SCALE));
}

// Channel slabs that share a 128-byte line of the NHWC feature map are given to blocks of the SAME XCD (ids 8 apart
// round-robin over the XCDs), so the 32-byte pieces they read / write meet in that XCD's L2.
__device__ __forceinline__ int roi_slab_of_block(int id, int nslabs) {
  const int per = nslabs >> 3;
  return (nslabs & 7) ? id : (id & 7) * per + (id >> 3);
}

// MEAN: the pooled ROIs were reduced by tf.reduce_mean over the cells (rcnn.py:185-188 use_mean) — the incoming
// gradient is dy[roi][c] / cells for every cell, read from the (roi, c) tensor instead of a 49x larger broadcast.
template <int CS, bool MEAN>
__global__ void __launch_bounds__(1024)
k_roi_pool_bwd_slab(const float* __restrict__ dout, const uint8_t* __restrict__ argmax,
                    const roi_sample_rec* __restrict__ table, const int32_t* __restrict__ roi_count, int R,
                    int FH, int FW, int C, int cells, const float* __restrict__ addend, float* __restrict__ dfeat) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long slab[];   // [npix][CS] fixed point
  const int b = blockIdx.y, c0 = roi_slab_of_block(blockIdx.x, gridDim.x) * CS;
  const int npix = FH * FW;
  for (int i = threadIdx.x; i < npix * CS; i += 1024) slab[i] = 0ull;
  __syncthreads();
  const int nroi = min(roi_count[b], R);
  const int pairs = nroi * cells;
  const int cc = threadIdx.x % CS;
  // FOUR (roi-cell, channel) pairs per trip: a pair costs two DEPENDENT global reads (arg-max byte -> sample record) and a
  // thread walks ~49 of them; one at a time that was ~100 serial round trips of memory latency per thread — most of the
  // kernel (135 us alone, 240-290 us beside the convolution streams).  Here the four gradient / arg-max reads go out
  // together (clamped index past the end), then the four record reads, then the LDS atomics (round 4).
  constexpr int STEP = 1024 / CS;
  for (int pair0 = threadIdx.x / CS; pair0 < pairs; pair0 += 4 * STEP) {
    float go[4];
    int q[4];
    size_t gp[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pair = min(pair0 + u * STEP, pairs - 1);
      gp[u] = (size_t)b * R * cells + pair;
      const size_t o = gp[u] * C + c0 + cc;
      go[u] = MEAN ? dout[((size_t)b * R + pair / cells) * C + c0 + cc] : dout[o];
      q[u] = argmax[o];
    }
    roi_sample_rec s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = table[gp[u] * 4 + (q[u] & 3)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float g = MEAN ? go[u] / (float)cells : go[u];
      if (pair0 + u * STEP >= pairs || s[u].top < 0 || g == 0.f) continue;
      const float dtop = (1.f - s[u].ylerp) * g;
      const float dbot = s[u].ylerp * g;
      roi_fx_add(&slab[(s[u].top * FW + s[u].left) * CS + cc], (1.f - s[u].xlerp) * dtop);
      roi_fx_add(&slab[(s[u].top * FW + s[u].right) * CS + cc], s[u].xlerp * dtop);
      roi_fx_add(&slab[(s[u].bot * FW + s[u].left) * CS + cc], (1.f - s[u].xlerp) * dbot);
      roi_fx_add(&slab[(s[u].bot * FW + s[u].right) * CS + cc], s[u].xlerp * dbot);
    }
  }
  __syncthreads();
  float* fb = dfeat + (size_t)b * npix * C + c0;
  const float* ab = addend ? addend + (size_t)b * npix * C + c0 : nullptr;
  const int nvec = npix * CS / 4;
  for (int i0 = threadIdx.x; i0 < nvec; i0 += 4 * 1024) {      // four rows per trip, their addend reads issued together
    f32x4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 1024;
      const int pix = i / (CS / 4), part = i - pix * (CS / 4);
      const float* ap = (ab && i < nvec) ? ab + (size_t)pix * C + 4 * part : roi_zero_page;
      a[u] = *reinterpret_cast<const f32x4*>(ap);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 1024;
      if (i >= nvec) break;
      const int pix = i / (CS / 4), part = i - pix * (CS / 4);
      const unsigned long long* p4 = slab + (size_t)pix * CS + 4 * part;
      float4 v;
      v.x = (float)((double)(long long)p4[0] * ROI_FX_INV);
      v.y = (float)((double)(long long)p4[1] * ROI_FX_INV);
      v.z = (float)((double)(long long)p4[2] * ROI_FX_INV);
      v.w = (float)((double)(long long)p4[3] * ROI_FX_INV);
      if (addend) {       // the other branch's gradient of the same feature map (RPN): the sum leaves in this one store
        v.x = a[u].x + v.x; v.y = a[u].y + v.y; v.z = a[u].z + v.z; v.w = a[u].w + v.w;
      }
      *reinterpret_cast<float4*>(fb + (size_t)pix * C + 4 * part) = v;
    }
  }
}

extern "C" size_t lmh_roi_pool_bwd_workspace_bytes(int B, int R, int ph, int pw) {
  return lmh_align_up((size_t)B * R * ph * pw * 4 * sizeof(roi_sample_rec), 256);
}

template <int CS, bool MEAN>
static int roi_bwd_slab_launch(const float* dout, const uint8_t* argmax, const roi_sample_rec* table,
                               const int32_t* roi_count, int B, int R, int FH, int FW, int C, int cells,
                               const float* addend, float* dfeat, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    LMH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_roi_pool_bwd_slab<CS, MEAN>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  lmh_launch((k_roi_pool_bwd_slab<CS, MEAN>), dim3(C / CS, B), dim3(1024),
                     (size_t)FH * FW * CS * sizeof(unsigned long long), st, dout, argmax, table, roi_count, R, FH, FW, C,
                     cells, addend, dfeat);
  return LMH_OK;
}

static int roi_slab_width(int FH, int FW, int C) {      // channels per LDS slab of the backward; 0: does not fit
  const size_t npix = (size_t)FH * FW, lds_cap = 160 * 1024;
  const int force_cs = lmh_opt("roi_cs");   // diagnostics (lmh_set_option)
  if (FH >= 32768 || FW >= 32768) return 0;
  if (force_cs != 4 && (C % 8) == 0 && npix * 8 * sizeof(unsigned long long) <= lds_cap) return 8;
  if ((C % 4) == 0 && npix * 4 * sizeof(unsigned long long) <= lds_cap) return 4;
  return 0;
}

// dfeat is OVERWRITTEN (it does not need to be zeroed by the caller) with the gradient, plus `addend` (same shape,
// may be NULL; must not alias dfeat) — the gradient another branch left for the same feature map.
extern "C" int lmh_roi_pool_bwd(const float* dout, const uint8_t* argmax, const float* rois,
                                const int32_t* roi_count, int B, int R, int FH, int FW, int C,
                                float im_h, float im_w, int ph, int pw, const float* addend, float* dfeat, void* ws,
                                size_t ws_bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(dout && argmax && rois && roi_count && dfeat);
  LMH_CHECK_ARG(B > 0 && R > 0 && FH > 0 && FW > 0 && C > 0 && ph > 0 && pw > 0);
  hipStream_t st = (hipStream_t)stream;
  const size_t npix = (size_t)FH * FW;
  const int cs = roi_slab_width(FH, FW, C);
  if (cs) {
    if (!ws || ws_bytes < lmh_roi_pool_bwd_workspace_bytes(B, R, ph, pw)) {
      lmh_set_error("lmh_roi_pool_bwd: workspace too small");
      return LMH_ERR_WORKSPACE;
    }
    roi_sample_rec* table = reinterpret_cast<roi_sample_rec*>(ws);
    const int cells = ph * pw;
    lmh_launch(k_roi_sample_table, dim3((B * R * cells + 255) / 256), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(rois), roi_count, B, R, FH, FW, im_h, im_w, ph, pw, table);
    const int rc = cs == 8 ? roi_bwd_slab_launch<8, false>(dout, argmax, table, roi_count, B, R, FH, FW, C, cells, addend, dfeat, st)
                           : roi_bwd_slab_launch<4, false>(dout, argmax, table, roi_count, B, R, FH, FW, C, cells, addend, dfeat, st);
    if (rc != LMH_OK) return rc;
  } else {   // very large feature maps: global scatter-add
    if (addend) LMH_CHECK_HIP(lmh_memcpy_d2d_async(dfeat, addend, (size_t)B * npix * C * sizeof(float), st));
    else LMH_CHECK_HIP(lmh_memset_async(dfeat, 0, (size_t)B * npix * C * sizeof(float), st));
    const int threads = C < 256 ? ((C + 63) / 64 * 64) : 256;
    lmh_launch(k_roi_pool_bwd, dim3(ph * pw, B * R), dim3(threads), 0, st, dout, argmax,
                       reinterpret_cast<const float4*>(rois), roi_count, R, FH, FW, C, im_h, im_w, ph, pw, dfeat);
  }
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// ---- ROI pooling fused with the spatial mean (rcnn.py:185-188, use_mean without a pooled tail) ----------------
// When the RCNN head averages the pooled cells straight away (ResNet-50 / VGG configurations), the (B*R, ph, pw, C)
// pooled tensor (51 MB at R=256, C=1024) only exists to be reduced 49:1 — and its gradient is a 49-fold broadcast.
// k_roi_pool_mean_fwd never writes it: a 1024-thread block stages CS channels of one image's whole feature map in
// LDS ([pixel][CS] floats, 128 KiB at 64x64x8) with one coalesced pass, then a thread is one (ROI, channel): it walks
// the ph*pw cells in order, evaluates the four bilinear samples of each from LDS (same arithmetic as k_roi_pool_fwd,
// bit for bit), keeps the 2-bit arg-max for the backward and accumulates the cell maxima sequentially — the order
// k_spatial_mean_fwd uses.  HBM traffic: the feature map once (33.5 MB) + arg-max bytes (12.8 MB) + the means (2 MB),
// instead of ~485 MB of corner gathers that depend on L2 hit rates — which is what made the gather kernel 3x slower
// whenever the convolution streams were thrashing the L2 next to it.
// CS = 8 / 4: 1024 threads, the slab of a 64 x 64 map is 128 / 64 KB — a block then needs (half) a compute unit to itself and,
// beside the convolution streams of the train step, WAITS for one to drain (round 4: 116 us alone, 141 us in the fp32 step,
// 396 us in the f16 step whose GEMM blocks hold 2 x 64 KB per CU).  CS = 2 / 1 (round 5): 32 / 16 KB slabs in blocks of
// 256 * CS threads that fit beside resident GEMM blocks; the slab is then filled with 4-byte pieces (a pixel's CS channels).
template <int CS>
__global__ void __launch_bounds__(CS >= 4 ? 1024 : 256 * CS)
k_roi_pool_mean_fwd(const float* __restrict__ feat, const float4* __restrict__ rois,
                    const int32_t* __restrict__ roi_count, int R, int FH, int FW, int C, float im_h, float im_w,
                    int ph, int pw, float* __restrict__ mean, uint8_t* __restrict__ argmax) {
  extern __shared__ __attribute__((aligned(16))) float fslab[];      // [npix][CS]
  constexpr int NT = CS >= 4 ? 1024 : 256 * CS;
  __builtin_amdgcn_s_setprio(3);
  const int b = blockIdx.y, c0 = roi_slab_of_block(blockIdx.x, gridDim.x) * CS;
  const int npix = FH * FW;
  const float* fb = feat + (size_t)b * npix * C + c0;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  if (CS >= 4) {
    // the slab goes from global memory straight into LDS (global_load_lds_dwordx4: a wave instruction fills 1 KB = PPI
    // pixels x CS channels, lane-linear, which IS the [pixel][CS] layout), every instruction of a wave issued back to
    // back and awaited once.  The register-staged loop it replaces was load -> wait -> ds_write per float4: eight
    // serial round trips of memory latency per thread before the first ROI (round 4; ISA check).
    constexpr int LPP = CS >= 4 ? CS / 4 : 1, PPI = 64 / LPP;          // lanes per pixel, pixels per wave instruction
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nins = npix / PPI;
    const float* src = fb + (size_t)(lane / LPP) * C + 4 * (lane % LPP);
    for (int k = wave; k < nins; k += NT / 64)
      __builtin_amdgcn_global_load_lds((glb_ptr)(src + (size_t)k * PPI * C), (lds_ptr)(fslab + (size_t)k * PPI * CS), 16, 0, 0);
    for (int i = nins * PPI * LPP + threadIdx.x; i < npix * LPP; i += NT) {      // pixels past the last whole instruction
      const int pix = i / LPP, part = i - pix * LPP;
      *reinterpret_cast<float4*>(fslab + (size_t)pix * CS + 4 * part) =
          *reinterpret_cast<const float4*>(fb + (size_t)pix * C + 4 * part);
    }
  } else {
    // dword pieces: a wave instruction fills 256 B = 64 / CS pixels x CS channels, lane-linear = the [pixel][CS] layout
    constexpr int PPI = 64 / CS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nins = npix / PPI;
    const float* src = fb + (size_t)(lane / CS) * C + (lane % CS);
    for (int k = wave; k < nins; k += NT / 64)
      __builtin_amdgcn_global_load_lds((glb_ptr)(src + (size_t)k * PPI * C), (lds_ptr)(fslab + (size_t)k * PPI * CS), 4, 0, 0);
    for (int i = nins * PPI * CS + threadIdx.x; i < npix * CS; i += NT) {
      const int pix = i / CS, part = i - pix * CS;
      fslab[(size_t)pix * CS + part] = fb[(size_t)pix * C + part];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int nroi = min(roi_count[b], R);
  const int cells = ph * pw, ch = 2 * ph, cw = 2 * pw;
  const int cc = threadIdx.x % CS;
  for (int r = threadIdx.x / CS; r < R; r += NT / CS) {
    const int rr = b * R + r;
    uint8_t* am = argmax + (size_t)rr * cells * C + c0 + cc;
    if (r >= nroi) {                                   // dead ROI: zeros, like k_roi_pool_fwd
      mean[(size_t)rr * C + c0 + cc] = 0.f;
      for (int s = 0; s < cells; ++s) am[(size_t)s * C] = 0;
      continue;
    }
    const roi_geom g = roi_geometry(rois[rr], im_h, im_w, FH, FW, ch, cw);
    float acc = 0.f;
    for (int py = 0; py < ph; ++py) {
#pragma unroll 2
      for (int px = 0; px < pw; ++px) {
        float best = 0.f;
        int bq = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const roi_sample s = roi_sample_at(g, 2 * py + (q >> 1), 2 * px + (q & 1), FH, FW, ch, cw);
          const int t = min(max(s.top, 0), FH - 1), bo = min(max(s.bot, 0), FH - 1);
          const int l = min(max(s.left, 0), FW - 1), ri = min(max(s.right, 0), FW - 1);
          const float tl = fslab[(t * FW + l) * CS + cc], tr = fslab[(t * FW + ri) * CS + cc];
          const float bl = fslab[(bo * FW + l) * CS + cc], br = fslab[(bo * FW + ri) * CS + cc];
          const float v = s.valid ? bilerp(tl, tr, bl, br, s.xlerp, s.ylerp) : 0.f;
          if (q == 0 || v > best) { best = v; bq = q; }        // first max wins
        }
        acc += best;
        am[(size_t)(py * pw + px) * C] = (uint8_t)bq;
      }
    }
    mean[(size_t)rr * C + c0 + cc] = acc / (float)cells;
  }
}

static int roi_mean_fwd_width(int FH, int FW, int C) {
  const size_t npix = (size_t)FH * FW, lds_cap = 160 * 1024;
  const int force = lmh_opt("roi_mean_cs");   // 0: report "unsupported"; 1, 2, 4: force that slab width (lmh_set_option)
  if (force == 0) return 0;
  if ((force == 1 || force == 2) && (C % force) == 0 && npix * force * sizeof(float) <= lds_cap) return force;
  if (force != 4 && (C % 8) == 0 && npix * 8 * sizeof(float) <= lds_cap) return 8;
  if ((C % 4) == 0 && npix * 4 * sizeof(float) <= lds_cap) return 4;
  return 0;
}

// 1 when the fused pool+mean kernels can take this feature map (it has to fit the LDS slabs of both directions)
extern "C" int lmh_roi_pool_mean_supported(int FH, int FW, int C) {
  return roi_mean_fwd_width(FH, FW, C) != 0 && roi_slab_width(FH, FW, C) != 0;
}

// mean (B*R, C) = reduce_mean over the ph*pw cells of ROIPoolingLayer's output; argmax (B*R, ph, pw, C) for the backward
extern "C" int lmh_roi_pool_mean_fwd(const float* feat, const float* rois, const int32_t* roi_count, int B, int R,
                                     int FH, int FW, int C, float im_h, float im_w, int ph, int pw, float* mean,
                                     uint8_t* argmax, lmh_stream_t stream) {
  LMH_CHECK_ARG(feat && rois && roi_count && mean && argmax);
  LMH_CHECK_ARG(B > 0 && R > 0 && FH > 0 && FW > 0 && C > 0 && ph > 0 && pw > 0);
  const int cs = roi_mean_fwd_width(FH, FW, C);
  if (!cs) {
    lmh_set_error("lmh_roi_pool_mean_fwd: feature map does not fit the LDS slab (use lmh_roi_pool_fwd + lmh_spatial_mean_fwd)");
    return LMH_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)FH * FW * cs * sizeof(float);
#define ROI_MEAN_LAUNCH(CS_)                                                                                         \
  do {                                                                                                               \
    static bool attr = false;                                                                                        \
    if (!attr) {                                                                                                     \
      LMH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_roi_pool_mean_fwd<CS_>),                    \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                    \
      attr = true;                                                                                                   \
    }                                                                                                                \
    lmh_launch((k_roi_pool_mean_fwd<CS_>), dim3(C / CS_, B), dim3(CS_ >= 4 ? 1024 : 256 * CS_), lds, st, feat,        \
               reinterpret_cast<const float4*>(rois), roi_count, R, FH, FW, C, im_h, im_w, ph, pw, mean, argmax);    \
  } while (0)
  if (cs == 8) ROI_MEAN_LAUNCH(8);
  else if (cs == 4) ROI_MEAN_LAUNCH(4);
  else if (cs == 2) ROI_MEAN_LAUNCH(2);
  else ROI_MEAN_LAUNCH(1);
#undef ROI_MEAN_LAUNCH
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// dmean (B*R, C): gradient of the means; dfeat is OVERWRITTEN.
extern "C" int lmh_roi_pool_mean_bwd(const float* dmean, const uint8_t* argmax, const float* rois,
                                     const int32_t* roi_count, int B, int R, int FH, int FW, int C, float im_h,
                                     float im_w, int ph, int pw, const float* addend, float* dfeat, void* ws,
                                     size_t ws_bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(dmean && argmax && rois && roi_count && dfeat);
  LMH_CHECK_ARG(B > 0 && R > 0 && FH > 0 && FW > 0 && C > 0 && ph > 0 && pw > 0);
  const int cs = roi_slab_width(FH, FW, C);
  if (!cs) {
    lmh_set_error("lmh_roi_pool_mean_bwd: feature map does not fit the LDS slab");
    return LMH_ERR_UNSUPPORTED;
  }
  if (!ws || ws_bytes < lmh_roi_pool_bwd_workspace_bytes(B, R, ph, pw)) {
    lmh_set_error("lmh_roi_pool_mean_bwd: workspace too small");
    return LMH_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  roi_sample_rec* table = reinterpret_cast<roi_sample_rec*>(ws);
  const int cells = ph * pw;
  lmh_launch(k_roi_sample_table, dim3((B * R * cells + 255) / 256), dim3(256), 0, st,
                     reinterpret_cast<const float4*>(rois), roi_count, B, R, FH, FW, im_h, im_w, ph, pw, table);
  const int rc = cs == 8 ? roi_bwd_slab_launch<8, true>(dmean, argmax, table, roi_count, B, R, FH, FW, C, cells, addend, dfeat, st)
                         : roi_bwd_slab_launch<4, true>(dmean, argmax, table, roi_count, B, R, FH, FW, C, cells, addend, dfeat, st);
  if (rc != LMH_OK) return rc;
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// tf.reduce_mean(x, [1,2]): x (M,S,C) -> y (M,C); sequential over S (deterministic).
__global__ void __launch_bounds__(256)
k_spatial_mean_fwd(const float* __restrict__ x, int S, int C, float* __restrict__ y) {
  const size_t m = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float* xp = x + m * S * C + c;
  float acc = 0.f;
  for (int s = 0; s < S; ++s) acc += xp[(size_t)s * C];
  y[m * C + c] = acc / (float)S;
}
__global__ void __launch_bounds__(256)
k_spatial_mean_bwd(const float* __restrict__ dy, int S, int C, float* __restrict__ dx) {
  const size_t m = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float g = dy[m * C + c] / (float)S;
  float* xp = dx + m * S * C + c;
  for (int s = 0; s < S; ++s) xp[(size_t)s * C] = g;
}
extern "C" int lmh_spatial_mean_fwd(const float* x, int64_t M, int S, int C, float* y, lmh_stream_t stream) {
  LMH_CHECK_ARG(x && y && M > 0 && S > 0 && C > 0);
  lmh_launch(k_spatial_mean_fwd, dim3((C + 255) / 256, (unsigned)M), dim3(256), 0,
                     (hipStream_t)stream, x, S, C, y);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
extern "C" int lmh_spatial_mean_bwd(const float* dy, int64_t M, int S, int C, float* dx, lmh_stream_t stream) {
  LMH_CHECK_ARG(dy && dx && M > 0 && S > 0 && C > 0);
  lmh_launch(k_spatial_mean_bwd, dim3((C + 255) / 256, (unsigned)M), dim3(256), 0,
                     (hipStream_t)stream, dy, S, C, dx);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
