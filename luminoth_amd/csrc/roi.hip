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

// grid: (ph*pw, B*R); block: min(C/4, 256) threads, each a float4 of channels.
__global__ void __launch_bounds__(256)
k_roi_pool_fwd(const float* __restrict__ feat, const float4* __restrict__ rois,
               const int32_t* __restrict__ roi_count, int R, int FH, int FW, int C, float im_h,
               float im_w, int ph, int pw, float* __restrict__ out, uint8_t* __restrict__ argmax) {
  const int rr = blockIdx.y;
  const int b = rr / R, r = rr % R;
  const int cell = blockIdx.x;
  const int py = cell / pw, px = cell % pw;
  const size_t obase = ((size_t)rr * ph * pw + cell) * C;
  const bool live = r < roi_count[b];
  const int C4 = C >> 2;
  if (!live) {
    for (int c4 = threadIdx.x; c4 < C4; c4 += blockDim.x) {
      reinterpret_cast<float4*>(out + obase)[c4] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (argmax) reinterpret_cast<uint32_t*>(argmax + obase)[c4] = 0u;
    }
    return;
  }
  const int ch = 2 * ph, cw = 2 * pw;  // crop size is passed (w*2, h*2): square in practice
  const roi_geom g = roi_geometry(rois[rr], im_h, im_w, FH, FW, ch, cw);
  roi_sample s[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) s[q] = roi_sample_at(g, 2 * py + (q >> 1), 2 * px + (q & 1), FH, FW, ch, cw);
  const float* fb = feat + (size_t)b * FH * FW * C;
  for (int c4 = threadIdx.x; c4 < C4; c4 += blockDim.x) {
    float best[4];
    uint32_t am[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (s[q].valid) {
        const float4 tl = reinterpret_cast<const float4*>(fb + ((size_t)s[q].top * FW + s[q].left) * C)[c4];
        const float4 tr = reinterpret_cast<const float4*>(fb + ((size_t)s[q].top * FW + s[q].right) * C)[c4];
        const float4 bl = reinterpret_cast<const float4*>(fb + ((size_t)s[q].bot * FW + s[q].left) * C)[c4];
        const float4 br = reinterpret_cast<const float4*>(fb + ((size_t)s[q].bot * FW + s[q].right) * C)[c4];
        v[0] = bilerp(tl.x, tr.x, bl.x, br.x, s[q].xlerp, s[q].ylerp);
        v[1] = bilerp(tl.y, tr.y, bl.y, br.y, s[q].xlerp, s[q].ylerp);
        v[2] = bilerp(tl.z, tr.z, bl.z, br.z, s[q].xlerp, s[q].ylerp);
        v[3] = bilerp(tl.w, tr.w, bl.w, br.w, s[q].xlerp, s[q].ylerp);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (q == 0 || v[e] > best[e]) { best[e] = v[e]; am[e] = q; }  // first max wins
      }
    }
    reinterpret_cast<float4*>(out + obase)[c4] = make_float4(best[0], best[1], best[2], best[3]);
    if (argmax)
      reinterpret_cast<uint32_t*>(argmax + obase)[c4] = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24);
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
  hipLaunchKernelGGL(k_roi_pool_fwd, dim3(ph * pw, B * R), dim3(threads), 0, (hipStream_t)stream, feat,
                     reinterpret_cast<const float4*>(rois), roi_count, R, FH, FW, C, im_h, im_w, ph, pw,
                     out, argmax);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

// Slab variant (default): one 1024-thread block owns CS channels of one image's whole feature map in
// LDS (CS channel planes of FH*FW floats, 128 KiB at 64x64x8), loops over every (roi, cell) of that image and scatter-adds
// with LDS atomics, then writes its slab once.  No global atomics, no pre-zeroing, and the 4-way corner
// contention of overlapping ROIs stays inside the CU (the global-atomic kernel above took 1.1 ms at
// R=256, C=1024 because clustered foreground ROIs serialise in L2).
template <int CS, int DBG = 0>
__global__ void __launch_bounds__(1024)
k_roi_pool_bwd_slab(const float* __restrict__ dout, const uint8_t* __restrict__ argmax,
                    const float4* __restrict__ rois, const int32_t* __restrict__ roi_count, int R, int FH,
                    int FW, int C, float im_h, float im_w, int ph, int pw, float* __restrict__ dfeat) {
  extern __shared__ __attribute__((aligned(16))) float slab[];
  const int b = blockIdx.y, c0 = blockIdx.x * CS;
  const int npix = FH * FW;
  for (int i = threadIdx.x; i < npix * CS / 4; i += 1024)
    reinterpret_cast<float4*>(slab)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const int cells = ph * pw, ch = 2 * ph, cw = 2 * pw;
  const int nroi = min(roi_count[b], R);
  const int pairs = nroi * cells;
  for (int pair = threadIdx.x; pair < pairs; pair += 1024) {
    const int r = pair / cells, cell = pair - r * cells;
    const int py = cell / pw, px = cell - py * pw;
    const int rr = b * R + r;
    const roi_geom g = roi_geometry(rois[rr], im_h, im_w, FH, FW, ch, cw);
    roi_sample s[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] = roi_sample_at(g, 2 * py + (q >> 1), 2 * px + (q & 1), FH, FW, ch, cw);
    const size_t obase = ((size_t)rr * cells + cell) * C + c0;
    float go[CS];
    uint8_t am[CS];
#pragma unroll
    for (int v = 0; v < CS / 4; ++v) {
      const float4 t = reinterpret_cast<const float4*>(dout + obase)[v];
      go[4 * v] = t.x; go[4 * v + 1] = t.y; go[4 * v + 2] = t.z; go[4 * v + 3] = t.w;
      const uint32_t a = reinterpret_cast<const uint32_t*>(argmax + obase)[v];
      am[4 * v] = a & 3; am[4 * v + 1] = (a >> 8) & 3; am[4 * v + 2] = (a >> 16) & 3; am[4 * v + 3] = (a >> 24) & 3;
    }
#pragma unroll
    for (int cc = 0; cc < CS; ++cc) {
      const int q = am[cc];
      roi_sample sq = s[0];
      if (q == 1) sq = s[1];
      if (q == 2) sq = s[2];
      if (q == 3) sq = s[3];
      if (!sq.valid || go[cc] == 0.f) continue;
      const float dtop = (1.f - sq.ylerp) * go[cc];
      const float dbot = sq.ylerp * go[cc];
      if (DBG == 1) continue;
      float* pl = slab + cc * npix;   // channel-major planes: lanes (different pixels) hit different banks
      atomicAdd(&pl[sq.top * FW + sq.left], (1.f - sq.xlerp) * dtop);
      atomicAdd(&pl[sq.top * FW + sq.right], sq.xlerp * dtop);
      atomicAdd(&pl[sq.bot * FW + sq.left], (1.f - sq.xlerp) * dbot);
      atomicAdd(&pl[sq.bot * FW + sq.right], sq.xlerp * dbot);
    }
  }
  __syncthreads();
  float* fb = dfeat + (size_t)b * npix * C + c0;
  if (DBG == 2) return;
  for (int i = threadIdx.x; i < npix * (CS / 4); i += 1024) {
    const int part = i / npix, pix = i - part * npix;   // consecutive lanes = consecutive pixels (conflict-free)
    const float* pl = slab + 4 * part * npix + pix;
    *reinterpret_cast<float4*>(fb + (size_t)pix * C + 4 * part) = make_float4(pl[0], pl[npix], pl[2 * npix], pl[3 * npix]);
  }
}

static int g_roi_dbg = 0;
extern "C" void lmh_roi_dbg(int v) { g_roi_dbg = v; }
// dfeat is OVERWRITTEN (it does not need to be zeroed by the caller).
extern "C" int lmh_roi_pool_bwd(const float* dout, const uint8_t* argmax, const float* rois,
                                const int32_t* roi_count, int B, int R, int FH, int FW, int C,
                                float im_h, float im_w, int ph, int pw, float* dfeat,
                                lmh_stream_t stream) {
  LMH_CHECK_ARG(dout && argmax && rois && roi_count && dfeat);
  LMH_CHECK_ARG(B > 0 && R > 0 && FH > 0 && FW > 0 && C > 0 && ph > 0 && pw > 0);
  hipStream_t st = (hipStream_t)stream;
  const size_t npix = (size_t)FH * FW;
  const size_t lds_cap = 160 * 1024;
  if ((C % 8) == 0 && npix * 8 * sizeof(float) <= lds_cap) {
    static bool attr8 = false;
    if (!attr8) {
      LMH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_roi_pool_bwd_slab<8>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
      attr8 = true;
    }
    if (g_roi_dbg == 1) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_roi_pool_bwd_slab<8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
      hipLaunchKernelGGL((k_roi_pool_bwd_slab<8, 1>), dim3(C / 8, B), dim3(1024), npix * 8 * sizeof(float), st, dout,
                       argmax, reinterpret_cast<const float4*>(rois), roi_count, R, FH, FW, C, im_h, im_w, ph, pw, dfeat);
    } else if (g_roi_dbg == 2) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_roi_pool_bwd_slab<8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
      hipLaunchKernelGGL((k_roi_pool_bwd_slab<8, 2>), dim3(C / 8, B), dim3(1024), npix * 8 * sizeof(float), st, dout,
                       argmax, reinterpret_cast<const float4*>(rois), roi_count, R, FH, FW, C, im_h, im_w, ph, pw, dfeat);
    } else
    hipLaunchKernelGGL((k_roi_pool_bwd_slab<8>), dim3(C / 8, B), dim3(1024), npix * 8 * sizeof(float), st, dout,
                       argmax, reinterpret_cast<const float4*>(rois), roi_count, R, FH, FW, C, im_h, im_w, ph, pw,
                       dfeat);
  } else if ((C % 4) == 0 && npix * 4 * sizeof(float) <= lds_cap) {
    static bool attr4 = false;
    if (!attr4) {
      LMH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_roi_pool_bwd_slab<4>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
      attr4 = true;
    }
    hipLaunchKernelGGL((k_roi_pool_bwd_slab<4>), dim3(C / 4, B), dim3(1024), npix * 4 * sizeof(float), st, dout,
                       argmax, reinterpret_cast<const float4*>(rois), roi_count, R, FH, FW, C, im_h, im_w, ph, pw,
                       dfeat);
  } else {   // very large feature maps: global scatter-add
    LMH_CHECK_HIP(hipMemsetAsync(dfeat, 0, (size_t)B * npix * C * sizeof(float), st));
    const int threads = C < 256 ? ((C + 63) / 64 * 64) : 256;
    hipLaunchKernelGGL(k_roi_pool_bwd, dim3(ph * pw, B * R), dim3(threads), 0, st, dout, argmax,
                       reinterpret_cast<const float4*>(rois), roi_count, R, FH, FW, C, im_h, im_w, ph, pw, dfeat);
  }
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
  hipLaunchKernelGGL(k_spatial_mean_fwd, dim3((C + 255) / 256, (unsigned)M), dim3(256), 0,
                     (hipStream_t)stream, x, S, C, y);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
extern "C" int lmh_spatial_mean_bwd(const float* dy, int64_t M, int S, int C, float* dx, lmh_stream_t stream) {
  LMH_CHECK_ARG(dy && dx && M > 0 && S > 0 && C > 0);
  hipLaunchKernelGGL(k_spatial_mean_bwd, dim3((C + 255) / 256, (unsigned)M), dim3(256), 0,
                     (hipStream_t)stream, dy, S, C, dx);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
