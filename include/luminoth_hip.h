/* luminoth_hip.h — C ABI of libluminoth_hip.so (gfx950 / MI355X).
 *
 * The reference (tryolabs/luminoth) has NO FFI: its hot path is a Python
 * module protocol (luminoth/models/models.py:6-17 get_model -> FasterRCNN /
 * SSD Sonnet modules) whose "kernels" are TensorFlow ops.  This header is the
 * flat C boundary placed BELOW that protocol: one entry point per group of TF
 * ops the reference calls on the path.  Each declaration
 * cites the reference call site(s) it replaces (paths relative to
 * /root/reference/luminoth/).
 *
 * Conventions
 *  - extern "C", plain device pointers + sizes; no allocation inside; every
 *    launch goes to the caller's `stream` (hipStream_t passed as void*);
 *    returns 0 or a negative lmh_status, message via lmh_last_error()
 *    (thread local).
 *  - State: the data path keeps none — every result is a function of the
 *    arguments of the call.  What IS thread-global (nothing is process-global
 *    but the option DEFAULTS) is confined to four documented entry points, all
 *    per calling thread: tuning options (lmh_set_option: which
 *    kernel variant / tile runs — equal results up to fp32 summation order,
 *    EXCEPT `wino_m`, which selects Winograd F(2x2,3x3) or F(4x4,3x3) and
 *    with it the rounding of every Winograd layer: ~1e-6 vs ~2e-5 of the
 *    output scale, both inside the 1e-4 contract; the library reads no
 *    environment variable), lmh_conv2d_force_config (sweeps),
 *    lmh_tail_defer (per thread: where a weight-gradient tail is finished) and
 *    lmh_conv2d_profile_next (event timing of the next convolution launch).
 *  - boxes are (x1,y1,x2,y2) fp32, inclusive-pixel convention; gt boxes are
 *    (G,5) with the 0-based class in column 4; image shape is (H, W).
 *  - Batched: leading B dimension, ragged per-image counts in int32 device
 *    arrays, fixed-capacity outputs (no host synchronisation anywhere).
 *  - "ws" = caller-provided scratch, size from the matching *_workspace_bytes.
 */
#ifndef LUMINOTH_HIP_H_
#define LUMINOTH_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lmh_stream_t; /* hipStream_t */

/* Tuning options (defaults are the measured best on MI355X).  Names: bd_parity_small, half_pf, x3_tile_slots, x3_pf,
 * x3_pf_fwd, x3_pf_gb, x3_pf_bd, x3_pf_bw, x3_new, x3_wg_plain, x3_pipe, x3_stagger, x3_bw_slots, bd_slots, bw_slots, wgrad_glds, wg_slots, wino_m,
 * hs_slab_cap, hs_wg_tile, hs_wg_rs, hs_bg, nms_stage_mult, head_gemm, conv_pp, roi_cs, roi_mean_cs (csrc/api.hip documents each).  Unknown name:
 * LMH_ERR_INVALID.
 * Re-entrancy (round 6): lmh_set_option sets the value for the CALLING THREAD only — two threads that drive two models
 * with different options do not see each other's settings; lmh_set_default_option sets the process default, which is what
 * a thread sees for every option it has not set itself (the host forwards LMH_OPT_<NAME> variables through it when it
 * loads the library); lmh_get_option returns what the calling thread's launches will use. */
int lmh_set_option(const char* name, int value);
int lmh_set_default_option(const char* name, int value);
int lmh_get_option(const char* name, int* value);

enum lmh_status {
  LMH_OK = 0,
  LMH_ERR_INVALID = -1,   /* bad argument */
  LMH_ERR_LAUNCH = -2,    /* HIP launch / runtime error */
  LMH_ERR_WORKSPACE = -3, /* workspace too small */
  LMH_ERR_UNSUPPORTED = -4
};

int lmh_version(void);
const char* lmh_last_error(void);
/* Number of HIP devices visible (used by the host side to fail loudly). */
int lmh_device_count(void);

/* ------------------------------------------------------------------ conv --
 * Implicit-GEMM NHWC fp32 convolution on v_mfma_f32_32x32x2_f32 with a fused
 * epilogue  y = act(conv(x,w) * scale[k] + shift[k] + residual).
 * Replaces: tf.contrib.slim conv2d / conv2d_same + frozen batch_norm + relu of
 * resnet_v1 (models/base/base_network.py:82-93, truncated_base_network.py:
 * 39-95), slim vgg conv (base_network.py:70-80, truncated_vgg.py:99-121),
 * sonnet Conv2D of the RPN (models/fasterrcnn/rpn.py:69-90,148-153) and the
 * SSD extra layers / heads (models/ssd/feature_extractor.py:27-37,
 * ssd.py:83-96), sonnet Linear of RCNN (models/fasterrcnn/rcnn.py:73-98) as a
 * 1x1 convolution.
 *   x (N,H,W,C)  w (R,S,C,K) [TF HWIO]  y (N,OH,OW,K);  pad_* = leading pads
 *   (TF SAME / conv2d_same pads are computed by the host), OH/OW given.
 *   scale/shift/residual may be NULL.  act: 0 none, 1 relu, 2 relu6.
 *   in_sub (3 floats or NULL) is subtracted from in-bounds x (channel means,
 *   base_network.py:14-16,159-177) — padding stays exactly 0.
 */
typedef struct lmh_conv_desc {
  int32_t N, H, W, C;      /* input  */
  int32_t K, R, S;         /* filter */
  int32_t OH, OW;          /* output */
  int32_t stride, dilation;
  int32_t pad_top, pad_left;
  int32_t act;
  /* Arithmetic of the MFMA operands: 0 = fp32 (v_mfma_f32_32x32x2_f32, the parity dtype), 1 = f16, 2 = bf16
   * (v_mfma_f32_32x32x16_*: operands rounded to half precision on their way into LDS, fp32 accumulation, fp32
   * tensors in memory — BASELINE configs[4]); 3 = bf16x3: fp32 ARITHMETIC on the bf16 pipe — every fp32 operand split
   * exactly into three bf16 pieces, six v_mfma_f32_32x32x16_bf16 per fp32 product (bit-exact with mode 0 on integer
   * data, same error against float64; also accepted by the *_winograd entry points, whose 16 stacked GEMMs then run
   * in bf16x3).  Shapes the half kernels do not cover run in fp32. */
  int32_t compute;
} lmh_conv_desc;

/* act_bits (may be NULL; needs act != 0 and K % 32 == 0): the ACTIVATION BIT MASK of y — N*OH*OW rows of K/32
 * words, bit (k % 32) of word (k / 32) set iff act'(y) != 0 (relu: y > 0; relu6: 0 < y < 6).  It is what the
 * backward pass needs of tf.nn.relu's output (TF keeps the whole tensor for ReluGrad): 1/32 of the bytes, written by
 * the epilogue that already holds the values. */
int lmh_conv2d_fwd(const lmh_conv_desc* d, const float* x, const float* w,
                   const float* scale, const float* shift, const float* residual,
                   const float* in_sub, float* y, uint32_t* act_bits, lmh_stream_t stream);
/* dx (N,H,W,C) = sum_{r,s,k} dy[..] * kscale[k] * w[r,s,c,k] + addend
 * (kscale, addend may be NULL; addend may alias dx: residual-branch accumulate). */
int lmh_conv2d_bwd_data(const lmh_conv_desc* d, const float* dy, const float* w,
                        const float* kscale, const float* addend, const float* yact,
                        const uint32_t* xbits, float* dx, lmh_stream_t stream);
/* The mask on its own (layers whose producer / consumer is not a convolution of this library):
 * bits <- act'(y) != 0 over (rows, K) with K % 32 == 0;  dx <- dx where bit else 0, in place. */
int lmh_act_bits(const float* y, int act, int64_t rows, int K, uint32_t* bits, lmh_stream_t stream);
int lmh_apply_act_bits(float* dx, const uint32_t* bits, int64_t rows, int C, lmh_stream_t stream);
/* dw (R,S,C,K) = sum_{n,oh,ow} x[..] * g[..]; split-K partials are reduced deterministically through `ws`.
 * Fused activation backward (both bwd entry points): when `yact` (the layer output y, same shape as dy) is
 * given, g = dy * act'(y) with act = d->act is applied while the operand is loaded (replaces a separate
 * lmh_act_bwd pass); otherwise g = dy.  `colsum` (K floats, may be NULL) receives sum_rows g (dbeta/dbias).
 * bwd_data only: `xbits` (the activation bit mask of the layer INPUT x = output of the layer below, C % 32 == 0, may
 * be NULL) makes the epilogue emit dx * act'(x): the layer below then receives its pre-activation gradient and needs
 * no lmh_act_bwd pass (ReluGrad of the reference graph, fused into the producer of its operand). */
size_t lmh_conv2d_bwd_weight_workspace_bytes(const lmh_conv_desc* d);
int lmh_conv2d_bwd_weight(const lmh_conv_desc* d, const float* x, const float* dy, const float* yact,
                          float* dw, float* colsum, void* ws, size_t ws_bytes, lmh_stream_t stream);
/* Which kernel instantiation a launch will use: op 0 fwd, 1 bwd_data, 2 bwd_weight.
 * Returns BM*1000 + BN (+1000000 for the generic C%32 != 0 forward gather).  Used by
 * bench.py to attribute per-kernel time / algorithmic FLOPs (roofline). */
int lmh_conv2d_kernel_id(const lmh_conv_desc* d, int op);
/* Diagnostics: force the block tile (bm,bn in {64,128}) and the bwd-weight split count; 0 = automatic.  Per calling thread. */
void lmh_conv2d_force_config(int bm, int bn, int splits);
/* Probes (round 5; only in a library built with LMH_PROBES=1, otherwise LMH_ERR_UNSUPPORTED for units != 0): low 8 bits =
 * start the co-resident blocks of the fp32 forward / backward-data kernels `units` x ~1 us apart (block slot
 * (blockIdx >> 8) % resident-blocks-per-CU); bits 8..10 = timing decomposition of the forward kernel (wrong results). */
int lmh_conv_set_stagger(int units);
/* Diagnostics: 1x1 weight-gradient kernel variant: 0 automatic, -1 register-staged kernel only, 2..4 LDS ring depth
 * of the direct-to-LDS GEMM kernel (conv_wgrad1x1.h). */
void lmh_conv2d_force_wgrad_variant(int v);
/* 1 when lmh_conv2d_bwd_weight(d) can emit `colsum` itself (fp32 fast paths); 0 when the caller should take the
 * per-channel sums of g from lmh_act_bwd (generic and half-precision kernels). */
int lmh_conv2d_bwd_weight_fuses_colsum(const lmh_conv_desc* d);
/* Per-launch timing (bench.py roofline leg): arms the NEXT lmh_conv2d_* call of this thread — `ev_start` /
 * `ev_stop` (hipEvent_t, e.g. from lmh_event_create) are recorded on the launch stream immediately before / after
 * its MFMA kernel (not the split-K reduce or Winograd transforms around it).  lmh_conv2d_profile_last then returns
 * that kernel's name as rocprofv3 prints it and the FLOPs the launch executed. */
int lmh_conv2d_profile_next(void* ev_start, void* ev_stop);
const char* lmh_conv2d_profile_last(double* flops);
/* Compulsory HBM bytes of that launch: every operand tensor read once + the result written once (fp32). */
double lmh_conv2d_profile_last_bytes(void);
/* HIP events for hosts without a HIP binding (ctypes): create / destroy / elapsed ms (synchronises on e1). */
/* `waiter` waits for everything enqueued on `signaler` so far (event record + stream wait on an internal event ring). */
int lmh_stream_wait_stream(lmh_stream_t waiter, lmh_stream_t signaler);
/* Record a caller-owned event (lmh_event_create) on a stream / make a stream wait for it; async memset and
 * device-to-device copy.  The forms of these HIP calls that a launch plan (below) can record. */
int lmh_event_record(void* event, lmh_stream_t stream);
int lmh_stream_wait_event(lmh_stream_t stream, void* event);
int lmh_memset(void* ptr, int value, size_t bytes, lmh_stream_t stream);
int lmh_memcpy_d2d(void* dst, const void* src, size_t bytes, lmh_stream_t stream);
/* Launch plans (csrc/plan.hip) — the counterpart of the reference's ONE host call per train step
 * (`sess.run(train_op)`, train.py:235-247, over a graph built once).  Between lmh_plan_begin and lmh_plan_end every
 * kernel launch, memset, copy, event record and stream wait this library issues on the calling thread is executed as
 * usual AND recorded with a private copy of its argument values; lmh_plan_run(plan, first, last) re-issues nodes
 * [first, last) (last < 0: to the end) — the same kernels with the same arguments on the same streams — from C.
 * The caller keeps every recorded device pointer valid and meaning the same thing (luminoth_amd/plan.py).
 * lmh_plan_position: nodes recorded so far (-1: not recording), for callers that interleave host work (collectives)
 * between two parts of a plan. */
int lmh_plan_begin(void);
void* lmh_plan_end(void);
void lmh_plan_abort(void);
void lmh_plan_destroy(void* plan);
int lmh_plan_recording(void);
int lmh_plan_position(void);
int lmh_plan_size(void* plan);
int lmh_plan_kernel_count(void* plan, int first, int last);
int lmh_plan_run(void* plan, int first, int last);
/* A stream restricted to the compute units {i : i % period < keep} (hipExtStreamCreateWithCUMask) — the
 * backward-overlap experiment of DESIGN.md §4; NULL on failure. */
lmh_stream_t lmh_stream_create_cu_mask(int period, int keep);
/* ... those with lo <= c % period < hi (complementary ranges partition the chip between two streams). */
lmh_stream_t lmh_stream_create_cu_range(int period, int lo, int hi);
void lmh_stream_destroy(lmh_stream_t stream);
/* scale = gamma * rstd, shift = beta - mean * scale over n channels: the frozen-statistics BatchNorm of every layer
 * (base_network.py:84-89) folded into per-channel scale / shift for the convolution epilogues, refreshed once a step. */
int lmh_bn_refresh(const float* gamma, const float* beta, const float* mean, const float* rstd, int64_t n,
                   float* scale, float* shift, lmh_stream_t stream);
/* BatchNorm in TRAINING mode — `train_batch_norm: True` (base_network.py:82-93, truncated_base_network.py:56-95:
 * slim.batch_norm(is_training=True), epsilon 1e-5, decay 0.997; update ops: train.py:87-88).  z (rows, K) is the RAW
 * convolution output.  fwd: mean / rstd (K) of the batch are written (kept for the backward), y = act(z * gamma * rstd +
 * beta - mean * gamma * rstd (+ residual)); with update_moving the moving statistics are advanced in place (variance with
 * Bessel's correction, like tf.nn.fused_batch_norm).  bwd: g = gradient of the pre-activation sum; dgamma / dbeta (K) are
 * WRITTEN, dz = gradient of the raw convolution output (then lmh_conv2d_bwd_data / _bwd_weight with no BatchNorm scale). */
size_t lmh_bn_train_workspace_bytes(int64_t rows, int K);
int lmh_bn_train_fwd(const float* z, int64_t rows, int K, const float* gamma, const float* beta, float eps, float decay,
                     float* moving_mean, float* moving_var, int update_moving, const float* residual, int act, float* y,
                     float* mean, float* rstd, void* ws, size_t ws_bytes, lmh_stream_t stream);
int lmh_bn_train_bwd(const float* g, const float* z, const float* mean, const float* rstd, const float* gamma,
                     int64_t rows, int K, const float* addend, int frozen_statistics, float* dgamma, float* dbeta,
                     float* dz, void* ws, size_t ws_bytes, lmh_stream_t stream);
/* ... `addend` (rows, K; may be NULL) is added to dz (a second consumer's gradient of the same tensor);
 * frozen_statistics != 0: mean / rstd are constants (the moving statistics): dz = gamma * rstd * g, dgamma / dbeta as
 * above — the backward of a stand-alone inference-mode BatchNorm with trainable gamma / beta; dz may be NULL (parameter
 * gradients only).  lmh_bn_apply: that layer's forward, y = act(z * scale + shift (+ residual)) — the `preact` BatchNorm
 * of slim's resnet_v2 units (base_network.py:94-101), which does not follow a convolution. */
int lmh_bn_apply(const float* z, int64_t rows, int K, const float* scale, const float* shift, const float* residual,
                 int act, float* y, lmh_stream_t stream);
/* total_loss bookkeeping of fasterrcnn.py:203-259 on the device: out[1] = no_reg = sum of the n (<= 8) weighted loss
 * scalars in order, out[2] = regularization = reg_a + reg_b (NULL = 0), out[0] = total = no_reg + regularization. */
int lmh_loss_sums(const float* const* terms, int n, const float* reg_a, const float* reg_b, float* out,
                  lmh_stream_t stream);
void* lmh_event_create(void);
void lmh_event_destroy(void* e);
float lmh_event_elapsed_ms(void* e0, void* e1);
/* Median interval of an event pair around an EMPTY kernel on `stream` (synchronises it): the dispatch latency an event
 * pair adds to a kernel's own duration.  bench.py subtracts it from its per-launch timings (comparable with rocprofv3). */
float lmh_event_pair_overhead_ms(int reps, lmh_stream_t stream);
/* Activations that no convolution epilogue fuses (luminoth/utils/vars.py:80-88 hands any tf.nn.<name> to the RPN
 * convolution and the RCNN fully connected layers, rpn.py:57-59, rcnn.py:73-74): codes 3 elu, 4 selu, 5 softplus,
 * 6 softsign, 7 sigmoid, 8 tanh, 9 leaky_relu (alpha 0.2) besides 0 none, 1 relu, 2 relu6.  lmh_act_fwd: y <- act(z) over
 * n floats (both 16-byte aligned; y may be z); the convolution descriptors take codes 0..2 only. */
#define LMH_ACT_MAX 9
int lmh_act_fwd(const float* z, float* y, int act, int64_t n, lmh_stream_t stream);
/* g = dy * act'(.) with the derivative expressed the way TF's gradient op takes it: in the OUTPUT y for relu / relu6 (the
 * select y > 0 [&& y < 6]), elu, selu, sigmoid, tanh, leaky_relu — and in the INPUT z for softplus / softsign (5, 6), for
 * which `y` must be the pre-activation (csrc/elementwise.hip has the table).  g may be NULL; colsum[k] = sum_rows g
 * (may be NULL; written, not accumulated; two-stage deterministic reduction through ws). */
size_t lmh_act_bwd_workspace_bytes(int64_t rows, int K);
int lmh_act_bwd(const float* dy, const float* y, int act, int64_t rows, int K,
                float* g, float* colsum, void* ws, size_t ws_bytes, lmh_stream_t stream);
/* BN (frozen) parameter gradients from the raw weight gradient:
 * dgamma[k] = rstd[k]*(sum_{rsc} w*dw_raw - mean[k]*dbeta[k]); dw = dw_raw*scale[k] (in place). */
size_t lmh_bn_param_grads_workspace_bytes(int64_t rsc, int K);
int lmh_bn_param_grads(const float* w, float* dw_raw_inout, const float* dbeta,
                       const float* mean, const float* rstd, const float* scale,
                       int64_t rsc, int K, float* dgamma, void* ws, size_t ws_bytes,
                       lmh_stream_t stream);
/* Winograd F(2x2,3x3) variants of lmh_conv2d_fwd / lmh_conv2d_bwd_data for stride-1, dilation-1, pad-1 3x3
 * convolutions with C % 32 == 0 and K % 32 == 0 (lmh_conv2d_winograd_ok): same results up to fp32
 * rounding of the transforms, 2.25x fewer matrix FLOPs.  Same call sites as the direct entry points (the sonnet
 * Conv2D of rpn.py:69-75, slim conv2d of vgg / resnet 3x3 layers).  ws: lmh_conv2d_winograd_workspace_bytes(d). */
int lmh_conv2d_winograd_ok(const lmh_conv_desc* d);
size_t lmh_conv2d_winograd_workspace_bytes(const lmh_conv_desc* d);
/* u: 16*C*K floats of transformed weights from lmh_conv2d_winograd_transform_weights (forward: G w G^T as
 * [16][C][K]; backward: of w[2-r][2-s][c][k]*kscale[k] as [16][K][C]), or NULL to transform inside the call. */
int lmh_conv2d_winograd_transform_weights(const lmh_conv_desc* d, const float* w, const float* kscale,
                                          int backward, float* u, lmh_stream_t stream);
/* act_bits / xbits: as for lmh_conv2d_fwd / lmh_conv2d_bwd_data (emitted / applied by the output transform). */
/* bf16x3 with PRE-SPLIT weights (round 6; csrc/conv_x3.h).  The weights of a layer are split once per step into their three
 * exact bf16 pieces, laid out in the fragment order of v_mfma_f32_32x32x16_bf16, and the convolution loads them straight
 * from global memory — instead of every row tile of every launch splitting the same slab of the weight matrix again.
 * lmh_x3_weights_bytes: bytes of one layer's planes (0 when C or K is not a multiple of 32); `backward`: 0 = the forward
 * arrangement (GEMM-k = (tap, c), columns = output channels), 1 = the backward-data one (GEMM-k = (tap, k), columns = input
 * channels).  lmh_x3_split_weights_batch: every job in one launch (w: (rs, C, K) fp32 = HWIO with rs = R * S; out: 16-byte
 * aligned).  lmh_conv2d_fwd_x3w: lmh_conv2d_fwd for compute bf16x3 with `w3` from a forward split of the layer's w;
 * bit-identical to it (same pieces, same order of products). */
typedef struct lmh_x3_weight_job {
  const float* w;
  void* out;
  int32_t rs, C, K;
} lmh_x3_weight_job;
size_t lmh_x3_weights_bytes(int rs, int c, int k, int backward);
int lmh_x3_split_weights_batch(const lmh_x3_weight_job* jobs, int n, int backward, lmh_stream_t stream);
int lmh_conv2d_fwd_x3w_supported(const lmh_conv_desc* d);
int lmh_conv2d_fwd_x3w(const lmh_conv_desc* d, const float* x, const void* w3, const float* scale,
                       const float* shift, const float* residual, float* y, uint32_t* act_bits, lmh_stream_t stream);
/* lmh_conv2d_bwd_data (no `yact`) with `w3` from a BACKWARD split of the layer's w. */
int lmh_conv2d_bwd_data_x3w_supported(const lmh_conv_desc* d);
int lmh_conv2d_bwd_data_x3w(const lmh_conv_desc* d, const float* dy, const void* w3, const float* kscale,
                            const float* addend, const uint32_t* xbits, float* dx, lmh_stream_t stream);
/* Transformed weights ahead of the convolution calls: every job of a step in ONE launch (`u` of job i:
 * lmh_winograd_u_bytes(C, K) bytes; forward: G g G^T of w (R,S,C,K); backward: of w[2-r][2-s][c][k] * kscale[k], kscale may
 * be NULL).  Pass the result as `u` to lmh_conv2d_fwd_winograd / lmh_conv2d_bwd_data_winograd (same "wino_m"). */
typedef struct lmh_wino_weight_job {
  const float* w;
  const float* kscale;
  float* u;
  int32_t C, K;
} lmh_wino_weight_job;
size_t lmh_winograd_u_bytes(int C, int K);
int lmh_winograd_transform_weights_batch(const lmh_wino_weight_job* jobs, int n, int backward, lmh_stream_t stream);
/* v_keep (may be NULL; lmh_conv2d_winograd_v_bytes(d) bytes): receives the transformed input planes B^T x B, which
 * lmh_conv2d_bwd_weight_winograd(v_cached) needs again for the same x — the training forward keeps them instead of
 * transforming x twice. */
size_t lmh_conv2d_winograd_v_bytes(const lmh_conv_desc* d);
int lmh_conv2d_fwd_winograd(const lmh_conv_desc* d, const float* x, const float* w, const float* u,
                            const float* scale, const float* shift, const float* residual, float* y,
                            uint32_t* act_bits, float* v_keep, void* ws, size_t ws_bytes, lmh_stream_t stream);
int lmh_conv2d_bwd_data_winograd(const lmh_conv_desc* d, const float* dy, const float* w, const float* u,
                                 const float* kscale, const float* addend, const uint32_t* xbits, float* dx,
                                 void* ws, size_t ws_bytes, lmh_stream_t stream);
/* dw (RAW, like lmh_conv2d_bwd_weight) = G^T [ sum_tiles (B^T x B)^T (A dy A^T) ] G. */
size_t lmh_conv2d_bwd_weight_winograd_workspace_bytes(const lmh_conv_desc* d);
/* v_cached (may be NULL): see v_keep above (then x may be NULL).  colsum (may be NULL, K floats): WRITTEN with the
 * per-channel sums of dy (dbeta / dbias), taken from the (1,1) plane of the transformed gradient. */
int lmh_conv2d_bwd_weight_winograd(const lmh_conv_desc* d, const float* x, const float* dy, float* dw,
                                   const float* v_cached, float* colsum, void* ws, size_t ws_bytes,
                                   lmh_stream_t stream);
/* tf.nn.max_pool NHWC (slim resnet pool1 3x3/2 SAME; vgg 2x2/2 VALID; SSD 3x3/1 SAME). */
int lmh_maxpool_fwd(const float* x, int N, int H, int W, int C, int ksize, int stride,
                    int pad_top, int pad_left, int OH, int OW, float* y, lmh_stream_t stream);
int lmh_maxpool_bwd(const float* x, const float* y, const float* dy, int N, int H, int W, int C,
                    int ksize, int stride, int pad_top, int pad_left, int OH, int OW, float* dx,
                    lmh_stream_t stream);

/* ---- Half-STORAGE convolution path (BASELINE.json configs[4] "fp16 MFMA path"; SURVEY.md 8(d): "fp16 activations/weights
 * with fp32 accumulate + fp32 master weights") for the conv stack the reference builds in
 * luminoth/models/base/base_network.py:82-93 (slim resnet_v1 bottlenecks).  Every `void*` tensor below holds 16-bit
 * elements: IEEE f16 when d->compute == 1, bfloat16 when d->compute == 2 (other values: LMH_ERR_UNSUPPORTED); needs
 * C % 64 == 0 and K % 64 == 0 (lmh_conv2d_hs_supported).  Master weights, weight gradients, BatchNorm vectors stay fp32.
 *   w_fwd [K][R][S][C] = q(w[r][s][c][k])                 w_bwd [R][S][C][K] = q(w[r][s][c][k] * kscale[k])
 * are the working copies lmh_half_weights_batch writes from the fp32 HWIO master weights (once per optimizer step).
 *   forward    y  = q( act( conv(x, w_fwd) * scale + shift + residual ) )        (y fp32 and unrounded when y_is_f32)
 *              act_bits (may be NULL): activation mask of the STORED y, layout of lmh_act_bits
 *   backward   dx = q( ( conv^T(g, w_bwd) * mul + addend ) * act'(x) )          act'(x) from xbits (may be NULL); dx fp32 and
 *              unrounded when dx_is_f32 (a half-storage layer whose input is an fp32 tensor: mul = 1 / loss scale)
 *   weights    dw = inv_scale * corr(x, g)   (fp32, RAW like lmh_conv2d_bwd_weight);  colsum[k] = inv_scale * sum_p g[p][k]
 * g carries the caller's loss scale (a power of two; 1 for bf16): inv_scale removes it.  Workspace of the weight
 * gradient: lmh_conv2d_bwd_weight_workspace_bytes(d); deferred tails (lmh_tail_defer) work as for lmh_conv2d_bwd_weight. */
int lmh_conv2d_hs_supported(const lmh_conv_desc* d);
int lmh_conv2d_fwd_hs(const lmh_conv_desc* d, const void* x, const void* w_fwd, const float* scale, const float* shift,
                      const void* residual, void* y, int y_is_f32, uint32_t* act_bits, lmh_stream_t stream);
int lmh_conv2d_bwd_data_hs(const lmh_conv_desc* d, const void* g, const void* w_bwd, const void* addend,
                           const uint32_t* xbits, void* dx, int dx_is_f32, float mul, lmh_stream_t stream);
int lmh_conv2d_bwd_weight_hs(const lmh_conv_desc* d, const void* x, const void* g, float inv_scale, float* dw,
                             float* colsum, void* ws, size_t ws_bytes, lmh_stream_t stream);
typedef struct lmh_half_weight_job {
  const float* w;       /* (R*S, C, K) fp32 master weights (HWIO) */
  const float* kscale;  /* K floats folded into w_bwd (frozen-BatchNorm scale), or NULL */
  void* w_fwd;          /* K * R*S*C halfs, or NULL: q(w) as B[n = k][q = tap * C + c] */
  void* w_bwd;          /* R*S*C * K halfs, or NULL: q(w * kscale[k]) as B[n = c][q = tap * K + k] */
  /* Layout of both copies when C % 64 == 0 and K % 64 == 0 (the shapes lmh_conv2d_hs_supported accepts): the fragment order
   * of the MFMA's B operand, element (n, q) at
   *     ((((n / 32) * (Q / 64) + q / 64) * 4 + (q % 64) / 16) * 64 + 32 * ((q % 16) / 8) + n % 32) * 8 + q % 8      (Q = q's range)
   * — opaque working copies: lmh_conv2d_fwd_hs / lmh_conv2d_bwd_data_hs are their only readers.  Other shapes: row-major
   * [n][q] for w_fwd, [q-major HWIO] i.e. [R*S*C][K] for w_bwd (rounds 3-5's layout; nothing reads them). */
  int32_t RS, C, K;
} lmh_half_weight_job;
int lmh_half_weights_batch(const lmh_half_weight_job* jobs, int n, int dtype, lmh_stream_t stream);
/* y = q(x * mul * mask): fp32 (rows, C) -> half; bits (may be NULL): activation mask [rows][C / 32] — the entry of the
 * trunk backward (gradient of the fp32 feature map times the loss scale, masked by the top layer's ReLU).  C % 8 == 0. */
int lmh_cast_to_half(const float* x, int64_t rows, int C, float mul, const uint32_t* bits, void* y, int dtype,
                     lmh_stream_t stream);
int lmh_cast_to_f32(const void* x, int64_t n, float mul, float* y, int dtype, lmh_stream_t stream);   /* n % 8 == 0 */
/* tf.nn.max_pool on NHWC with a half result; x fp32 (x_is_f32: the fp32 stem output) or half.  C % 8 == 0. */
int lmh_maxpool_fwd_hs(const void* x, int x_is_f32, int N, int H, int W, int C, int ksize, int stride, int pad_top,
                       int pad_left, int OH, int OW, void* y, int dtype, lmh_stream_t stream);
/* backward of resnet_utils.subsample (1x1 max pool, stride s) on 16-bit tensors: dx (N,H,W,C) fully written. */
int lmh_subsample_bwd_hs(const void* dy, int N, int H, int W, int C, int stride, int OH, int OW, void* dx,
                         lmh_stream_t stream);

/* tf.image.resize_images(BILINEAR) as called by luminoth/utils/image.py:92-95 (resize_image) and :126-129
 * (resize_image_fixed) — the first op of the `lumi predict` path (utils/predicting.py:43-47).  TF 1.x legacy
 * sampling (align_corners=False, no half-pixel centres).  src is (H,W,C) uint8 or float32, dst (OH,OW,C) f32.
 * flip_lr / flip_ud: resample tf.image.flip_left_right / flip_up_down of src (the flip augmentation,
 * utils/image.py:318-370, runs before the resize: object_detection_dataset.py:77-78) without materialising it. */
int lmh_resize_bilinear(const void* src, int src_is_u8, int H, int W, int C, float* dst, int OH,
                        int OW, int flip_lr, int flip_ud, lmh_stream_t stream);

/* ------------------------------------------------------------ proposals --
 * RPNProposal._build (models/fasterrcnn/rpn_proposal.py:41-197) for a batch:
 * softmax(2) (rpn.py:163) -> anchors generated on the fly with the int32
 * truncation quirk (fasterrcnn.py:261-308) -> decode (utils/
 * bbox_transform_tf.py:41-66) -> area>0 & score>=min_prob filter -> clip ->
 * top_k(pre_nms_top_n) (stable: score desc, index asc) -> NMS(thr, strict >,
 * TF continuous-area IoU) -> gather.
 *   cls_score (B,N,2)  bbox_pred (B,N,4)  anchor_ref (A,4) int32
 *   out: cls_prob (B,N,2), proposals (B,cap,4), scores (B,cap), num_proposals
 *        (B) int32, cap = apply_nms ? post_nms_top_n : pre_nms_top_n.  Rows
 *        beyond num_proposals[b] are zero.
 */
typedef struct lmh_rpn_proposal_desc {
  int32_t B, feat_h, feat_w, A, anchor_stride;
  float im_h, im_w;
  int32_t pre_nms_top_n, post_nms_top_n;
  float nms_threshold, min_prob_threshold;
  int32_t apply_nms, clip_after_nms, filter_outside_anchors;
} lmh_rpn_proposal_desc;

size_t lmh_rpn_proposal_workspace_bytes(const lmh_rpn_proposal_desc* d);
int lmh_rpn_proposal(const lmh_rpn_proposal_desc* d, const float* cls_score,
                     const float* bbox_pred, const int32_t* anchor_ref,
                     float* cls_prob, float* proposals, float* scores,
                     int32_t* num_proposals, void* ws, size_t ws_bytes,
                     lmh_stream_t stream);

/* Batched in-place ascending sort of u64 keys, n_pad a power of two (LDS
 * bitonic).  Building block of top_k (tf.nn.top_k call sites:
 * rpn_proposal.py:140, rcnn_proposal.py:152, ssd/target.py:143, ssd/proposal.py:159). */
int lmh_sort_u64(uint64_t* keys, int B, int n_pad, lmh_stream_t stream);

/* tf.image.non_max_suppression (rpn_proposal.py:152-157, rcnn_proposal.py:
 * 114-117, ssd/proposal.py:123-126) on boxes ALREADY sorted by descending
 * score: boxes (B,K,4) xyxy, counts (B).  keep_idx (B,max_out) indices into
 * the sorted order (-1 padded), keep_count (B). */
size_t lmh_nms_workspace_bytes(int B, int K);
int lmh_nms(const float* boxes, const int32_t* counts, int B, int K, float iou_threshold,
            int max_out, int32_t* keep_idx, int32_t* keep_count, void* ws, size_t ws_bytes,
            lmh_stream_t stream);

/* -------------------------------------------------------------- targets --
 * RPNTarget._build (models/fasterrcnn/rpn_target.py:73-335) for a batch.
 *   gt (B,Gmax,5), gt_count (B); seeds (B) per-image u32 for the shared
 *   counter-based sampler (oracle/rng.py).
 *   out: labels (B,N) f32 in {-1,0,1}; bbox_targets (B,N,4); max_overlaps
 *   (B,N); labels_pre (B,N) labels before subsampling (may be NULL).
 */
typedef struct lmh_rpn_target_desc {
  int32_t B, feat_h, feat_w, A, anchor_stride, Gmax;
  int32_t im_h, im_w;
  int32_t allowed_border, clobber_positives;
  float foreground_threshold, background_threshold_high, foreground_fraction;
  int32_t minibatch_size;
} lmh_rpn_target_desc;

size_t lmh_rpn_target_workspace_bytes(const lmh_rpn_target_desc* d);
int lmh_rpn_target(const lmh_rpn_target_desc* d, const int32_t* anchor_ref, const float* gt,
                   const int32_t* gt_count, const uint32_t* seeds, float* labels,
                   float* bbox_targets, float* max_overlaps, float* labels_pre, void* ws,
                   size_t ws_bytes, lmh_stream_t stream);

/* RCNNTarget._build (models/fasterrcnn/rcnn_target.py:48-299) + the
 * is_training batch compaction of RCNN._build (rcnn.py:156-167).
 *   proposals (B,P,4), prop_count (B), gt (B,Gmax,5), gt_count (B)
 *   out: labels (B,P), bbox_targets (B,P,4) [full, as the reference returns];
 *   compacted rows with label >= 0, proposal order preserved:
 *   rois (B,R,4), roi_labels (B,R), roi_targets (B,R,4), roi_count (B)
 *   with R = minibatch_size (rows beyond roi_count: boxes 0, label -1).
 * Neither P nor Gmax is bounded (the reference bounds neither): the per-proposal state sits in LDS up to 4096
 * proposals and in `ws` beyond, gt boxes stream through LDS 128 at a time (Gmax <= 32767: int16 indices).
 */
typedef struct lmh_rcnn_target_desc {
  int32_t B, P, Gmax, minibatch_size;
  float foreground_fraction, foreground_threshold, background_threshold_high,
      background_threshold_low;
  float variance_xy, variance_wh;
} lmh_rcnn_target_desc;

size_t lmh_rcnn_target_workspace_bytes(const lmh_rcnn_target_desc* d);
int lmh_rcnn_target(const lmh_rcnn_target_desc* d, const float* proposals,
                    const int32_t* prop_count, const float* gt, const int32_t* gt_count,
                    const uint32_t* seeds, float* labels, float* bbox_targets, float* labels_pre,
                    float* rois, float* roi_labels, float* roi_targets, int32_t* roi_count,
                    void* ws, size_t ws_bytes, lmh_stream_t stream);

/* RCNNProposal._build (models/fasterrcnn/rcnn_proposal.py:46-164): per class
 * decode(variances) -> clip -> (prob >= min_prob & area > 0) -> NMS(thr, <=
 * class_max) ; concat in class order ; top_k(total_max) by prob.  One batched
 * launch sequence over (image, class).  Also SSDProposal
 * (models/ssd/proposal.py:41-171) with class_agnostic_boxes = 1.
 *   proposals (B,R,4), prop_count (B), bbox_pred (B,R,4C) [(B,R,4) if
 *   class_agnostic_boxes], cls_prob (B,R,C+1)
 *   out: objects (B,T,4), labels (B,T) int32 (-1 padded), probs (B,T), num (B).
 */
typedef struct lmh_rcnn_proposal_desc {
  int32_t B, R, C;
  float im_h, im_w;
  float variance_xy, variance_wh;
  int32_t class_max_detections;
  float class_nms_threshold;
  int32_t total_max_detections;
  float min_prob_threshold;
  int32_t class_agnostic_boxes;
} lmh_rcnn_proposal_desc;

size_t lmh_rcnn_proposal_workspace_bytes(const lmh_rcnn_proposal_desc* d);
int lmh_rcnn_proposal(const lmh_rcnn_proposal_desc* d, const float* proposals,
                      const int32_t* prop_count, const float* bbox_pred, const float* cls_prob,
                      float* objects, int32_t* labels, float* probs, int32_t* num_objects,
                      void* ws, size_t ws_bytes, lmh_stream_t stream);

/* SSDProposal._build (models/ssd/proposal.py:41-171) with ALL five keys of its return dict (:165-171): the
 * class-agnostic form of lmh_rcnn_proposal (d->class_agnostic_boxes must be 1; proposals = anchors (B,R,4),
 * bbox_pred = loc_pred (B,R,4)) plus the two debug outputs:
 *   raw_proposals (B,R,4), raw_count (B): the UNCLIPPED decode of the anchors that pass the probability filter of
 *     the LAST class — the loop variable the reference returns after its class loop (:83,167); order preserved;
 *   det_anchors (B,T,4): `tf.gather(proposal_anchors, top_k.indices)` (:143,162) — the top-k indices address the
 *     concatenation of the per-class NMS-selected boxes but are applied to the concatenation of every class's
 *     FILTERED anchors (a longer list): misaligned in the reference, restated as is. */
size_t lmh_ssd_proposal_workspace_bytes(const lmh_rcnn_proposal_desc* d);
int lmh_ssd_proposal(const lmh_rcnn_proposal_desc* d, const float* anchors, const int32_t* anchor_count,
                     const float* loc_pred, const float* cls_prob, float* objects, int32_t* labels,
                     float* probs, int32_t* num_objects, float* raw_proposals, int32_t* raw_count,
                     float* det_anchors, void* ws, size_t ws_bytes, lmh_stream_t stream);

/* ------------------------------------------------------------------ ROI --
 * ROIPoolingLayer._roi_crop (models/fasterrcnn/roi_pool.py:37-95):
 * tf.image.crop_and_resize(2ph x 2pw, bilinear, extrapolation 0) fused with
 * the 2x2/2 VALID max pool.  feat (B,FH,FW,C); rois (B,R,4) image coords;
 * roi_count (B) (rows beyond -> zeros).  out (B*R,ph,pw,C); argmax (same
 * shape, u8: which of the 4 samples) for the backward.
 */
int lmh_roi_pool_fwd(const float* feat, const float* rois, const int32_t* roi_count, int B, int R,
                     int FH, int FW, int C, float im_h, float im_w, int ph, int pw, float* out,
                     uint8_t* argmax, lmh_stream_t stream);
/* dfeat is OVERWRITTEN with the full gradient (CropAndResizeGradImage scatter-add, done in LDS slabs) PLUS `addend`
 * (same shape, may be NULL, must not alias dfeat): the gradient the RPN branch left for the same feature map — the
 * sum TF's autodiff forms for a tensor with two consumers (fasterrcnn.py:129-147) leaves in the slab's one store. */
size_t lmh_roi_pool_bwd_workspace_bytes(int B, int R, int ph, int pw);
int lmh_roi_pool_bwd(const float* dout, const uint8_t* argmax, const float* rois,
                     const int32_t* roi_count, int B, int R, int FH, int FW, int C, float im_h,
                     float im_w, int ph, int pw, const float* addend, float* dfeat, void* ws, size_t ws_bytes,
                     lmh_stream_t stream);
/* ROI pooling fused with tf.reduce_mean(pooled, [1, 2]) (rcnn.py:185-188, `use_mean` with no pooled tail between:
 * ResNet-50 / VGG configurations): mean (B*R, C) is what lmh_roi_pool_fwd + lmh_spatial_mean_fwd return, bit for
 * bit, without the (B*R, ph, pw, C) intermediate; the backward takes dmean (B*R, C).  The feature map of one image
 * has to fit the kernels' LDS slabs (FH*FW <= 5120 at C % 8 == 0): ask lmh_roi_pool_mean_supported first, the
 * unfused pair is the general path (LMH_ERR_UNSUPPORTED otherwise).  Workspace of the backward:
 * lmh_roi_pool_bwd_workspace_bytes. */
int lmh_roi_pool_mean_supported(int FH, int FW, int C);
int lmh_roi_pool_mean_fwd(const float* feat, const float* rois, const int32_t* roi_count, int B, int R, int FH,
                          int FW, int C, float im_h, float im_w, int ph, int pw, float* mean, uint8_t* argmax,
                          lmh_stream_t stream);
int lmh_roi_pool_mean_bwd(const float* dmean, const uint8_t* argmax, const float* rois, const int32_t* roi_count,
                          int B, int R, int FH, int FW, int C, float im_h, float im_w, int ph, int pw,
                          const float* addend, float* dfeat, void* ws, size_t ws_bytes, lmh_stream_t stream);
/* tf.reduce_mean(features, [1,2]) (rcnn.py:185-188): x (M,S,C) -> y (M,C). */
int lmh_spatial_mean_fwd(const float* x, int64_t M, int S, int C, float* y, lmh_stream_t stream);
int lmh_spatial_mean_bwd(const float* dy, int64_t M, int S, int C, float* dx, lmh_stream_t stream);

/* --------------------------------------------------------------- losses --
 * RPN.loss (models/fasterrcnn/rpn.py:219-309) + smooth_l1_loss
 * (utils/losses.py:4-32): per image  cls = mean_{label!=-1} CE,
 * reg = mean_{label==1} sum_4 SL1_sigma; batch loss = mean over images.
 * losses (2) = {w_cls*rpn_cls_loss, w_reg*rpn_reg_loss}; per_image (B,4) =
 * {cls, reg, #labelled, #positive} per image (required scratch/output);
 * gradients of (w_cls*cls + w_reg*reg) wrt cls_score / bbox_pred.
 */
int lmh_rpn_loss(const float* cls_score, const float* bbox_pred, const float* labels,
                 const float* bbox_targets, int B, int N, float sigma, float w_cls, float w_reg,
                 float* losses, float* per_image, float* d_cls_score, float* d_bbox_pred,
                 lmh_stream_t stream);
/* RCNN.loss (models/fasterrcnn/rcnn.py:255-411): rows (B,R), num_classes C:
 * cls_score (B,R,C+1), bbox_offsets (B,R,4C), labels (B,R), targets (B,R,4). */
int lmh_rcnn_loss(const float* cls_score, const float* bbox_offsets, const float* labels,
                  const float* targets, int B, int R, int C, float sigma, float w_cls, float w_reg,
                  float* losses, float* per_image, float* d_cls_score, float* d_bbox_offsets,
                  lmh_stream_t stream);
/* The gradient half of lmh_rcnn_loss alone (models/fasterrcnn/rcnn.py:255-411 under TF autodiff, train.py:80): d(w_cls*cls +
 * w_reg*reg) / d(cls_score, bbox_offsets).  It counts its two normalisers (#labelled, #positive rows per image) from
 * `labels` itself, so the train step can issue it where the loss sits and the sums (lmh_rcnn_loss with NULL gradient
 * pointers: reported values only) anywhere behind the RCNN backward.  Same bits as lmh_rcnn_loss's gradients. */
int lmh_rcnn_loss_grad(const float* cls_score, const float* bbox_offsets, const float* labels,
                       const float* targets, int B, int R, int C, float sigma, float w_cls, float w_reg,
                       float* d_cls_score, float* d_bbox_offsets, lmh_stream_t stream);
/* tf.nn.softmax over the last axis (rcnn.py:206, ssd.py:109). */
int lmh_softmax(const float* x, int64_t rows, int C, float* y, lmh_stream_t stream);

/* ------------------------------------------------------------------ SSD --
 * conv4_3 normalisation (models/ssd/feature_extractor.py:75-89):
 * y = tf.nn.l2_normalize(x, axis=3, epsilon) * gamma;  x (P, C) NHWC rows, gamma (C). */
int lmh_l2norm_scale_fwd(const float* x, const float* gamma, int64_t P, int C, float eps, float* y,
                         lmh_stream_t stream);
size_t lmh_l2norm_scale_bwd_workspace_bytes(int64_t P, int C);
int lmh_l2norm_scale_bwd(const float* x, const float* dy, const float* gamma, int64_t P, int C, float eps,
                         float* dx, float* dgamma, void* ws, size_t ws_bytes, lmh_stream_t stream);

/* SSDTarget._build (models/ssd/target.py:35-200) for a batch: IoU(+1) labels (>= fg threshold -> gt
 * label + 1), best anchor per gt (override, last gt wins on duplicates), hard-negative mining
 * (top_k(int(#fg * ratio)) of max_c>=1 prob over rows with max IoU <= bg_high and label <= 0; those rows
 * become 0 — whatever they were), encode(variances) targets for label > 0.
 *   anchors (N,4) shared by all images; gt (B,Gmax,5), gt_count (B); probs (B,N,C+1) softmax
 *   out: labels (B,N) f32 in {-1,0,1..C}; bbox_targets (B,N,4); max_overlaps (B,N). */
typedef struct lmh_ssd_target_desc {
  int32_t B, N, C, Gmax;
  float foreground_threshold, background_threshold_high, hard_negative_ratio;
  float variance_xy, variance_wh;
} lmh_ssd_target_desc;
size_t lmh_ssd_target_workspace_bytes(const lmh_ssd_target_desc* d);
int lmh_ssd_target(const lmh_ssd_target_desc* d, const float* anchors, const float* gt,
                   const int32_t* gt_count, const float* probs, float* labels, float* bbox_targets,
                   float* max_overlaps, void* ws, size_t ws_bytes, lmh_stream_t stream);

/* SSD.loss (models/ssd/ssd.py:197-300) + smooth_l1 (utils/losses.py:4-32, sigma 3): rows with label -1 are
 * ignored.  losses (3) = mean over images of [final, cls_sum, bbox_sum]; per_image (B,4) =
 * [cls_sum, bbox_sum, #pos, final]; gradients of losses[0] (may be NULL). */
int lmh_ssd_loss(const float* cls_pred, const float* loc_pred, const float* labels, const float* targets,
                 int B, int N, int C, float sigma, float w_loc, float* losses, float* per_image,
                 float* d_cls_pred, float* d_loc_pred, lmh_stream_t stream);

/* ------------------------------------------------------------ optimizer --
 * tf.train.MomentumOptimizer (utils/training.py:64-81, train.py:79-91) over a
 * flat parameter buffer split in `nseg` segments, with the L2 regulariser
 * (rpn.py:54-56, rcnn.py:59-60, slim weight_decay) folded in:
 *   g' = g*gscale + wd[s]*w ; [per-tensor clip_by_norm, training.py:84-120, if clip>0]
 *   v = momentum*v + g' ; w -= lr*v
 * seg_offset (nseg+1) int64, seg_wd (nseg) float: device arrays.
 */
int lmh_sgd_momentum(float* w, const float* g, float* v, int64_t n, const int64_t* seg_offset,
                     const float* seg_wd, int nseg, float lr, float momentum, float gscale,
                     lmh_stream_t stream);
/* The same update over the range [lo, hi) of the flat buffer (lo % 4 == 0; hi % 4 == 0 or hi == n) with the learning rate read
 * from device memory (`lr_dev`, one float): recordable in a launch plan, issued per gradient range under the backward pass
 * (replaces one slice of MomentumOptimizer.apply_gradients, luminoth/train.py:79-91).  `early` only selects the kernel NAME
 * (k_sgd_early_range / k_sgd_momentum_range: trace tools find the end of a step by the latter). */
int lmh_sgd_momentum_range(float* w, const float* g, float* v, int64_t n, int64_t lo, int64_t hi,
                           const int64_t* seg_offset, const float* seg_wd, int nseg, const float* lr_dev,
                           float momentum, float gscale, int early, lmh_stream_t stream);
/* Deferred weight-gradient tails.  While lmh_tail_defer(1) is in effect on the calling thread, lmh_conv2d_bwd_weight and
 * lmh_act_bwd launch their main kernel only: the split-K slabs stay in the caller's `ws`, the per-channel partial sums
 * of g stay in theirs, and lmh_tail_last_plan reports where (slabs NULL / splits 0: `dw` already holds the raw
 * gradient).  The caller keeps those workspaces untouched, queues one lmh_wgrad_tail per layer and finishes the whole
 * backlog with lmh_wgrad_tail_batch (two launches per <= 20 layers): split-K reduction in fixed order, frozen-BatchNorm
 * scaling dW = dW_raw * scale[k] and dgamma[k] = rstd[k] * (sum_i w[i,k] dW_raw[i,k] - mean[k] dbeta[k]) (slim
 * batch_norm in inference mode, base_network.py:84-89), dbeta / dbias = column sums of the partial rows. */
typedef struct lmh_wgrad_tail {
  const float* slabs;   /* [splits][n] or NULL */
  float* dw;            /* (n) out; in when splits == 0 */
  const float* w;       /* BN only */
  const float* scale;   /* BN only: gamma * rstd */
  const float* mean;    /* BN only */
  const float* rstd;    /* BN only */
  float* dgamma;        /* BN only (NULL: plain / bias layer) */
  const float* colpart; /* [colrows][K] or NULL */
  float* colsum;        /* dbeta / dbias (out when colpart != NULL, else read for dgamma) */
  int64_t n;            /* R*S*C*K */
  int32_t splits, K, colrows, reserved;
} lmh_wgrad_tail;
void lmh_tail_defer(int on);
void lmh_tail_last_plan(const float** slabs, int* splits, const float** colpart, int* colrows);
size_t lmh_wgrad_tail_batch_workspace_bytes(const lmh_wgrad_tail* tails, int count);
int lmh_wgrad_tail_batch(const lmh_wgrad_tail* tails, int count, void* ws, size_t ws_bytes, lmh_stream_t stream);

/* The non-default rest of utils/training.py.  lmh_grad_clip_factors: factors[s] = clip / max(||g'_s||, clip) per
 * segment (clip_gradients_by_norm: tf.clip_by_norm(g', 10), training.py:84-120; g' includes the L2 term like TF's
 * gradient of total_loss).  lmh_optimizer_step: kind 0 momentum (p1 = momentum; 0 = GradientDescentOptimizer),
 * 1 Adam (p1, p2 = beta1, beta2; lr = bias-corrected lr_t), 2 RMSProp (p1 = decay, p2 = momentum), 3 momentum with
 * use_nesterov=True (TF ApplyMomentum: w -= g'*lr + v*p1*lr), 4 RMSProp with centered=True (slot3 = the mean-gradient
 * slot mg; TF ApplyCenteredRMSProp) — the keyword arguments training.py:64-81 forwards to the TF optimizers; OPTIMIZERS
 * table training.py:6-11.  slot2 may be NULL for kinds 0 / 3, slot3 for all but 4; seg_factor may be NULL (no clipping). */
size_t lmh_grad_clip_workspace_bytes(int nseg);
int lmh_grad_clip_factors(const float* w, const float* g, int64_t n, const int64_t* seg_offset, const float* seg_wd,
                          int nseg, float gscale, float clip_norm, float* factors, void* ws, size_t ws_bytes,
                          lmh_stream_t stream);
int lmh_optimizer_step(int kind, float* w, const float* g, float* slot1, float* slot2, float* slot3, int64_t n,
                       const int64_t* seg_offset, const float* seg_wd, const float* seg_factor, int nseg, float lr,
                       float p1, float p2, float eps, float gscale, lmh_stream_t stream);
/* tf.nn.dropout of the RCNN head (models/fasterrcnn/rcnn.py:196,218): y = x * keep / keep_prob, keep a pure function
 * of (seed, element index) — call it again on dy with the same seed for the backward pass. */
int lmh_dropout(const float* x, int64_t n, float keep_prob, uint32_t seed, float* y, lmh_stream_t stream);
/* regularization_loss = sum_s wd[s] * sum(w_s^2)/2 (tf l2_regularizer); out (1) is WRITTEN.  Two launches through `ws`
 * (lmh_l2_reg_workspace_bytes): per-block partial sums, then their sum in block order — the same bits every call. */
size_t lmh_l2_reg_workspace_bytes(void);
int lmh_l2_reg_loss(const float* w, int64_t n, const int64_t* seg_offset, const float* seg_wd,
                    int nseg, float* out, void* ws, size_t ws_bytes, lmh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LUMINOTH_HIP_H_ */
