// Final detection stage: per-class decode + clip + filter + NMS, then top-k
// over all classes (gfx950).  Batched over (image, class) pairs so the C
// per-class NMS graphs of the reference become ONE launch sequence.
//
// Reference: RCNNProposal._build luminoth/models/fasterrcnn/rcnn_proposal.py:46-164
// (also the shape of SSDProposal, models/ssd/proposal.py:41-171).
#include "lmh_common.h"

// one thread per (image, class, proposal)
__global__ void __launch_bounds__(256)
k_class_decode(lmh_rcnn_proposal_desc d, int Rpad, const float4* __restrict__ proposals,
               const int32_t* __restrict__ prop_count, const float* __restrict__ bbox_pred,
               const float* __restrict__ cls_prob, float4* __restrict__ boxes, uint64_t* __restrict__ keys,
               int32_t* __restrict__ n_valid) {
  const int bc = blockIdx.y;  // b * C + c
  const int b = bc / d.C, c = bc % d.C;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= Rpad) return;
  uint64_t key = ~0ull;
  if (r < min(prop_count[b], d.R)) {
    const size_t row = (size_t)b * d.R + r;
    const float4 p = proposals[row];
    const float prob = cls_prob[row * (d.C + 1) + c + 1];  // 0 is background (rcnn_proposal.py:80)
    const float4 dl = d.class_agnostic_boxes
                          ? reinterpret_cast<const float4*>(bbox_pred)[row]
                          : *reinterpret_cast<const float4*>(bbox_pred + row * 4 * d.C + 4 * c);
    lmh_box o = lmh_decode(lmh_box{p.x, p.y, p.z, p.w}, dl.x, dl.y, dl.z, dl.w, d.variance_xy, d.variance_wh);
    o = lmh_clip(o, d.im_h, d.im_w);
    const bool ok = (prob >= d.min_prob_threshold) &&
                    (fmaxf(o.x2 - o.x1, 0.f) * fmaxf(o.y2 - o.y1, 0.f) > 0.f);
    boxes[(size_t)bc * d.R + r] = make_float4(o.x1, o.y1, o.x2, o.y2);
    if (ok) key = ((uint64_t)(~lmh_float_orderable(prob)) << 32) | (uint32_t)r;
  }
  keys[(size_t)bc * Rpad + r] = key;
  const unsigned long long bal = __ballot(key != ~0ull);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&n_valid[bc], __popcll(bal));
}

__global__ void __launch_bounds__(256)
k_class_gather(lmh_rcnn_proposal_desc d, int Rpad, const uint64_t* __restrict__ keys,
               const float4* __restrict__ boxes, const int32_t* __restrict__ n_valid,
               float4* __restrict__ sorted_boxes, int32_t* __restrict__ sorted_src) {
  const int bc = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.R) return;
  float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
  int src = -1;
  if (i < n_valid[bc]) {
    src = (int)(keys[(size_t)bc * Rpad + i] & 0xFFFFFFFFull);
    bx = boxes[(size_t)bc * d.R + src];
  }
  sorted_boxes[(size_t)bc * d.R + i] = bx;
  sorted_src[(size_t)bc * d.R + i] = src;
}

// keys over the (class, kept slot) grid of one image: prob desc, concat order asc
__global__ void __launch_bounds__(256)
k_final_keys(lmh_rcnn_proposal_desc d, int Tpad, const int32_t* __restrict__ keep_idx,
             const int32_t* __restrict__ keep_count, const int32_t* __restrict__ sorted_src,
             const float* __restrict__ cls_prob, uint64_t* __restrict__ fkeys, int32_t* __restrict__ n_total) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Tpad) return;
  uint64_t key = ~0ull;
  const int cmax = d.class_max_detections;
  if (t < d.C * cmax) {
    const int c = t / cmax, i = t % cmax;
    const int bc = b * d.C + c;
    if (i < keep_count[bc]) {
      const int r = sorted_src[(size_t)bc * d.R + keep_idx[(size_t)bc * cmax + i]];
      const float prob = cls_prob[((size_t)b * d.R + r) * (d.C + 1) + c + 1];
      key = ((uint64_t)(~lmh_float_orderable(prob)) << 32) | (uint32_t)t;
    }
  }
  fkeys[(size_t)b * Tpad + t] = key;
  const unsigned long long bal = __ballot(key != ~0ull);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&n_total[b], __popcll(bal));
}

__global__ void __launch_bounds__(256)
k_final_gather(lmh_rcnn_proposal_desc d, int Tpad, const uint64_t* __restrict__ fkeys,
               const int32_t* __restrict__ n_total, const int32_t* __restrict__ keep_idx,
               const int32_t* __restrict__ sorted_src, const float4* __restrict__ sorted_boxes,
               const float* __restrict__ cls_prob, float4* __restrict__ objects, int32_t* __restrict__ labels,
               float* __restrict__ probs, int32_t* __restrict__ num) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int T = d.total_max_detections;
  const int cnt = min(n_total[b], T);
  if (i == 0) num[b] = cnt;
  if (i >= T) return;
  float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
  int lab = -1;
  float pr = 0.f;
  if (i < cnt) {
    const int t = (int)(fkeys[(size_t)b * Tpad + i] & 0xFFFFFFFFull);
    const int cmax = d.class_max_detections;
    const int c = t / cmax, slot = t % cmax;
    const int bc = b * d.C + c;
    const int k = keep_idx[(size_t)bc * cmax + slot];
    bx = sorted_boxes[(size_t)bc * d.R + k];
    const int r = sorted_src[(size_t)bc * d.R + k];
    pr = cls_prob[((size_t)b * d.R + r) * (d.C + 1) + c + 1];
    lab = c;
  }
  objects[(size_t)b * T + i] = bx;
  labels[(size_t)b * T + i] = lab;
  probs[(size_t)b * T + i] = pr;
}

struct det_ws {
  float4* boxes; uint64_t* keys; int32_t* n_valid; float4* sorted_boxes; int32_t* sorted_src;
  int32_t* keep_idx; int32_t* keep_count; uint64_t* fkeys; int32_t* n_total; void* nms_ws; size_t total;
};

static det_ws det_layout(const lmh_rcnn_proposal_desc* d, void* base) {
  const size_t BC = (size_t)d->B * d->C, R = d->R, Rpad = lmh_next_pow2(d->R);
  const size_t Tpad = lmh_next_pow2(d->C * d->class_max_detections);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += lmh_align_up(bytes, 256); return o; };
  char* p = reinterpret_cast<char*>(base);
  det_ws w;
  w.boxes = reinterpret_cast<float4*>(p + take(BC * R * 16));
  w.keys = reinterpret_cast<uint64_t*>(p + take(BC * Rpad * 8));
  w.n_valid = reinterpret_cast<int32_t*>(p + take(BC * 4));
  w.sorted_boxes = reinterpret_cast<float4*>(p + take(BC * R * 16));
  w.sorted_src = reinterpret_cast<int32_t*>(p + take(BC * R * 4));
  w.keep_idx = reinterpret_cast<int32_t*>(p + take(BC * d->class_max_detections * 4));
  w.keep_count = reinterpret_cast<int32_t*>(p + take(BC * 4));
  w.fkeys = reinterpret_cast<uint64_t*>(p + take((size_t)d->B * Tpad * 8));
  w.n_total = reinterpret_cast<int32_t*>(p + take((size_t)d->B * 4));
  w.nms_ws = p + take(lmh_nms_workspace_bytes((int)BC, (int)R));
  w.total = off;
  return w;
}

extern "C" size_t lmh_rcnn_proposal_workspace_bytes(const lmh_rcnn_proposal_desc* d) {
  if (!d) return 0;
  return det_layout(d, nullptr).total;
}

static int det_run(const lmh_rcnn_proposal_desc* d, const float* proposals,
                   const int32_t* prop_count, const float* bbox_pred, const float* cls_prob,
                   float* objects, int32_t* labels, float* probs, int32_t* num_objects,
                   void* ws, size_t ws_bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(d && proposals && prop_count && bbox_pred && cls_prob && objects && labels && probs &&
                num_objects && ws);
  LMH_CHECK_ARG(d->B > 0 && d->R > 0 && d->C > 0 && d->class_max_detections > 0 && d->total_max_detections > 0);
  det_ws w = det_layout(d, ws);
  if (ws_bytes < w.total) {
    lmh_set_error("lmh_rcnn_proposal: workspace %zu < %zu", ws_bytes, w.total);
    return LMH_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int BC = d->B * d->C, Rpad = lmh_next_pow2(d->R);
  const int Tpad = lmh_next_pow2(d->C * d->class_max_detections);
  LMH_CHECK_HIP(lmh_memset_async(w.n_valid, 0, sizeof(int32_t) * BC, st));
  LMH_CHECK_HIP(lmh_memset_async(w.n_total, 0, sizeof(int32_t) * d->B, st));
  lmh_launch(k_class_decode, dim3((Rpad + 255) / 256, BC), dim3(256), 0, st, *d, Rpad,
                     reinterpret_cast<const float4*>(proposals), prop_count, bbox_pred, cls_prob, w.boxes,
                     w.keys, w.n_valid);
  int rc = lmh_sort_u64_impl(w.keys, BC, Rpad, st);
  if (rc) return rc;
  lmh_launch(k_class_gather, dim3((d->R + 255) / 256, BC), dim3(256), 0, st, *d, Rpad, w.keys,
                     w.boxes, w.n_valid, w.sorted_boxes, w.sorted_src);
  rc = lmh_nms_impl(reinterpret_cast<const float*>(w.sorted_boxes), w.n_valid, BC, d->R,
                    d->class_nms_threshold, d->class_max_detections, w.keep_idx, w.keep_count, w.nms_ws, st);
  if (rc) return rc;
  lmh_launch(k_final_keys, dim3((Tpad + 255) / 256, d->B), dim3(256), 0, st, *d, Tpad, w.keep_idx,
                     w.keep_count, w.sorted_src, cls_prob, w.fkeys, w.n_total);
  rc = lmh_sort_u64_impl(w.fkeys, d->B, Tpad, st);
  if (rc) return rc;
  lmh_launch(k_final_gather, dim3((d->total_max_detections + 255) / 256, d->B), dim3(256), 0, st, *d,
                     Tpad, w.fkeys, w.n_total, w.keep_idx, w.sorted_src, w.sorted_boxes, cls_prob,
                     reinterpret_cast<float4*>(objects), labels, probs, num_objects);
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}

extern "C" int lmh_rcnn_proposal(const lmh_rcnn_proposal_desc* d, const float* proposals,
                                 const int32_t* prop_count, const float* bbox_pred, const float* cls_prob,
                                 float* objects, int32_t* labels, float* probs, int32_t* num_objects,
                                 void* ws, size_t ws_bytes, lmh_stream_t stream) {
  return det_run(d, proposals, prop_count, bbox_pred, cls_prob, objects, labels, probs, num_objects, ws, ws_bytes,
                 stream);
}

// ---- SSDProposal's two debug outputs (models/ssd/proposal.py:143,160-171) --------------------------------------
// 'raw_proposals': the unclipped decode of the LAST class's probability-filtered anchors (the loop variable leaks out
// of the class loop, proposal.py:83,167), order preserved.  One 1024-thread block per image, ballot compaction.
__global__ void __launch_bounds__(1024)
k_ssd_raw_proposals(lmh_rcnn_proposal_desc d, const float4* __restrict__ anchors, const int32_t* __restrict__ prop_count,
                    const float4* __restrict__ loc_pred, const float* __restrict__ cls_prob,
                    float4* __restrict__ raw, int32_t* __restrict__ raw_count) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = min(prop_count[b], d.R);
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int r0 = 0; r0 < d.R; r0 += 1024) {
    const int r = r0 + tid;
    const size_t row = (size_t)b * d.R + r;
    bool ok = false;
    if (r < n) ok = cls_prob[row * (d.C + 1) + d.C] >= d.min_prob_threshold;   // class C-1 -> column C
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    if (ok) {
      const float4 a = anchors[row], dl = loc_pred[row];
      const lmh_box o = lmh_decode(lmh_box{a.x, a.y, a.z, a.w}, dl.x, dl.y, dl.z, dl.w, d.variance_xy, d.variance_wh);
      raw[(size_t)b * d.R + off + __popcll(bal & ((1ull << lane) - 1ull))] = make_float4(o.x1, o.y1, o.x2, o.y2);
    }
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < 16; ++w) t += s_wave[w];
      s_base += t;
    }
    __syncthreads();
  }
  for (int r = s_base + tid; r < d.R; r += 1024) raw[(size_t)b * d.R + r] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid == 0) raw_count[b] = s_base;
}

// 'anchors': the reference gathers with the top-k indices — positions in the concatenation of the per-class NMS-KEPT
// lists — from the concatenation of the per-class FILTERED anchor lists (prob >= thr and area > 0, anchor order),
// which is a longer list (proposal.py:143 appends `proposal_anchors` un-gathered).  Restated as is: detection i with
// concat position j receives element j of that longer list.  One wave per detection: j is located by walking the
// classes' n_valid counts, then the q-th valid row of that class by ballot-counting 64 rows at a time.
__global__ void __launch_bounds__(64)
k_ssd_det_anchors(lmh_rcnn_proposal_desc d, int Tpad, const uint64_t* __restrict__ fkeys,
                  const int32_t* __restrict__ n_total, const int32_t* __restrict__ keep_count,
                  const int32_t* __restrict__ n_valid, const float4* __restrict__ anchors,
                  const int32_t* __restrict__ prop_count, const float4* __restrict__ loc_pred,
                  const float* __restrict__ cls_prob, float4* __restrict__ det_anchors) {
  const int b = blockIdx.y, i = blockIdx.x, lane = threadIdx.x;
  const int T = d.total_max_detections;
  const int cnt = min(n_total[b], T);
  float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < cnt) {
    const int t = (int)(fkeys[(size_t)b * Tpad + i] & 0xFFFFFFFFull);
    const int cmax = d.class_max_detections;
    const int c = t / cmax, slot = t % cmax;
    int j = slot;
    for (int cc = 0; cc < c; ++cc) j += keep_count[b * d.C + cc];
    int cls = 0;
    while (cls < d.C - 1 && j >= n_valid[b * d.C + cls]) { j -= n_valid[b * d.C + cls]; ++cls; }
    const int n = min(prop_count[b], d.R);
    for (int r0 = 0; r0 < n; r0 += 64) {       // uniform across the wave
      const int r = r0 + lane;
      bool ok = false;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < n) {
        const size_t row = (size_t)b * d.R + r;
        a = anchors[row];
        const float4 dl = loc_pred[row];
        lmh_box o = lmh_decode(lmh_box{a.x, a.y, a.z, a.w}, dl.x, dl.y, dl.z, dl.w, d.variance_xy, d.variance_wh);
        o = lmh_clip(o, d.im_h, d.im_w);
        ok = (cls_prob[row * (d.C + 1) + cls + 1] >= d.min_prob_threshold) &&
             (fmaxf(o.x2 - o.x1, 0.f) * fmaxf(o.y2 - o.y1, 0.f) > 0.f);
      }
      const unsigned long long bal = __ballot(ok);
      const int here = __popcll(bal);
      if (j < here) {
        const int rank = __popcll(bal & ((1ull << lane) - 1ull));
        if (ok && rank == j) det_anchors[(size_t)b * T + i] = a;
        return;
      }
      j -= here;
    }
    // not reachable while this kernel's validity test agrees with the one that produced n_valid / keep_count; if it
    // ever did not (a changed filter, a NaN probability) the row gets the zero box, not whatever the buffer held
    if (lane == 0) det_anchors[(size_t)b * T + i] = out;
    return;
  }
  if (lane == 0) det_anchors[(size_t)b * T + i] = out;
}

extern "C" size_t lmh_ssd_proposal_workspace_bytes(const lmh_rcnn_proposal_desc* d) {
  return lmh_rcnn_proposal_workspace_bytes(d);
}

extern "C" int lmh_ssd_proposal(const lmh_rcnn_proposal_desc* d, const float* anchors, const int32_t* anchor_count,
                                const float* loc_pred, const float* cls_prob, float* objects, int32_t* labels,
                                float* probs, int32_t* num_objects, float* raw_proposals, int32_t* raw_count,
                                float* det_anchors, void* ws, size_t ws_bytes, lmh_stream_t stream) {
  LMH_CHECK_ARG(d && d->class_agnostic_boxes == 1 && raw_proposals && raw_count && det_anchors);
  int rc = det_run(d, anchors, anchor_count, loc_pred, cls_prob, objects, labels, probs, num_objects, ws, ws_bytes,
                   stream);
  if (rc) return rc;
  det_ws w = det_layout(d, ws);
  hipStream_t st = (hipStream_t)stream;
  const int Tpad = lmh_next_pow2(d->C * d->class_max_detections);
  lmh_launch(k_ssd_raw_proposals, dim3(d->B), dim3(1024), 0, st, *d, reinterpret_cast<const float4*>(anchors),
                     anchor_count, reinterpret_cast<const float4*>(loc_pred), cls_prob,
                     reinterpret_cast<float4*>(raw_proposals), raw_count);
  lmh_launch(k_ssd_det_anchors, dim3(d->total_max_detections, d->B), dim3(64), 0, st, *d, Tpad, w.fkeys,
                     w.n_total, w.keep_count, w.n_valid, reinterpret_cast<const float4*>(anchors), anchor_count,
                     reinterpret_cast<const float4*>(loc_pred), cls_prob, reinterpret_cast<float4*>(det_anchors));
  LMH_CHECK_LAUNCH();
  return LMH_OK;
}
