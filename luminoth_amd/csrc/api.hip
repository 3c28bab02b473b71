// Error plumbing + misc entry points of libluminoth_hip.so.
#include <stdarg.h>
#include <string.h>

#include "lmh_common.h"

static thread_local char g_err[512] = "";

void lmh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- tuning options (none changes a result beyond fp32 summation order, except `wino_m`: luminoth_hip.h).  Round 6
// (the ABI is specified re-entrant): an option has a PROCESS DEFAULT (lmh_set_default_option: what the
// host forwards from LMH_OPT_<NAME>=<int> at load time, and what a thread sees until it sets the option itself) and a
// PER-THREAD value (lmh_set_option: the calling thread only) — two threads driving two models with different options do
// not see each other's settings.  No environment variable is read anywhere in this library.
#include <atomic>
struct lmh_option { const char* name; int value; };
static const lmh_option g_option_defaults[] = {
    {"bd_parity_small", 1},   // stride-2 3x3 backward data: 64x64 tiles when the parity classes are unbalanced
    {"half_pf", 1},           // f16/bf16 kernels: 1, 2 register sets; 3, 4 warp-specialised 512-thread blocks
    {"x3_tile_slots", 512},   // bf16x3 fwd / bwd_data tile by pick_tile(slots): 512 = two resident blocks per CU (round 6: 5.47 ms per
                              // step against 5.54 with 256, 5.51 with 1024); 0: half_tile
    {"x3_bw_slots", 256},     // bf16x3 weight gradient: resident-block slots its split count fills (bw_slots for the native kernels)
    {"x3_pf", -1},            // round-2 bf16x3 kernels (x3_new = 0): pipeline for every pass (-1: per-pass values below)
    {"x3_pf_fwd", 0}, {"x3_pf_gb", 0}, {"x3_pf_bd", 0}, {"x3_pf_bw", 0},
    {"x3_pipe", 0},           // round-6 bf16x3 kernels (conv_x3.h): 0 = phase by phase, two blocks per CU (default: 5.59 ms per step
                              // against 6.60 — the pipelined blocks need a whole CU and shut the other streams' kernels out);
                              // 1 = software-pipelined, one block per CU
    {"x3_stagger", 0},        // phase-by-phase schedule: the block in a CU's second LDS slot starts this many x 64 cycles late (no effect measured)
    {"x3_wg_plain", 1},       // bf16x3 weight gradient of 1x1 / stride-1 layers and stacked Winograd GEMMs: no pixel decode in the loader
    {"x3_new", 1},            // bf16x3: the kernels of round 6 (conv_x3.h: fused mask epilogues); 0: the round-2 kernels (conv_half.h)
    {"bd_slots", 256},        // resident-block slots the backward-data tile choice fills
    {"bw_slots", 512},        // ... the split-K weight gradient
    {"wgrad_glds", 1},        // 1x1 weight gradient: operands straight into LDS (conv_wgrad1x1.h)
    {"wg_slots", 512},        // ... its block count target
    {"wino_m", 4},            // Winograd output tile: 4 = F(4x4,3x3) (round 3 default), 2 = F(2x2,3x3)
    {"hs_slab_cap", 2},       // half-storage weight gradient: split-K slabs stay within this multiple of the operand bytes (0: no cap)
    {"hs_wg_tile", 0},        // half-storage weight gradient tile of the 1x1 layers: 0 / 64 = 64 x 64 (round 6, with hs_wg_rs: half the split-K slab bytes of a layer and
                              // faster, profiles/r06_ab.md); 128 = 128 x 128 when both channel counts reach it (rounds 3-5)
    {"hs_bg", 1},             // half-storage forward / backward data: B fragments straight from global memory into registers (conv_hs.h, BG); 0: through
                              // the LDS ring like A
    {"hs_wg_rs", 1},          // half-storage weight gradient: tiles through registers + ds_write_b128 (k_wgrad_hs_tr RS = 4); 0: LDS-DMA instructions
    {"nms_stage_mult", 0},    // > 0: NMS in two stages, A = this many x max_out candidates (mask + scan), the rest only if needed; 0: one stage
    {"head_gemm", 1},         // Linear heads on <= 4096 rows: the split-reduction 32x32 kernel (conv_generic.h k_head_fwd); 0: the tiled / skinny kernels
    {"roi_cs", 0},            // ROI backward slab width (0: automatic, 4: force the 4-channel slab)
    {"conv_pp", 1},           // 1x1 forward with >= 2 tiles of 128 x 128 per compute unit: the persistent pipelined kernel (conv_pp.h)
    {"roi_mean_cs", -1},      // fused ROI pool+mean (-1: automatic, 0: report unsupported, 4: force 4 channels)
};
enum { LMH_NOPT = sizeof(g_option_defaults) / sizeof(g_option_defaults[0]) };
static std::atomic<int> g_option_process[LMH_NOPT];          // process defaults (initialised below on first use)
static std::atomic<int> g_option_init{0};
struct lmh_thread_options { bool set[LMH_NOPT]; int value[LMH_NOPT]; };
static thread_local lmh_thread_options g_option_thread = {};
static void lmh_options_init() {
  if (g_option_init.load(std::memory_order_acquire) == 2) return;
  int expect = 0;
  if (g_option_init.compare_exchange_strong(expect, 1)) {
    for (int i = 0; i < LMH_NOPT; ++i) g_option_process[i].store(g_option_defaults[i].value, std::memory_order_relaxed);
    g_option_init.store(2, std::memory_order_release);
  } else {
    while (g_option_init.load(std::memory_order_acquire) != 2) {}
  }
}
static int lmh_option_index(const char* name) {
  for (int i = 0; i < LMH_NOPT; ++i)
    if (!strcmp(g_option_defaults[i].name, name)) return i;
  return -1;
}
extern "C" int lmh_set_option(const char* name, int value) {          // the calling thread only
  if (!name) return LMH_ERR_INVALID;
  const int i = lmh_option_index(name);
  if (i < 0) { lmh_set_error("lmh_set_option: unknown option '%s'", name); return LMH_ERR_INVALID; }
  g_option_thread.set[i] = true;
  g_option_thread.value[i] = value;
  return LMH_OK;
}
extern "C" int lmh_set_default_option(const char* name, int value) {  // every thread that has not set the option itself
  if (!name) return LMH_ERR_INVALID;
  const int i = lmh_option_index(name);
  if (i < 0) { lmh_set_error("lmh_set_default_option: unknown option '%s'", name); return LMH_ERR_INVALID; }
  lmh_options_init();
  g_option_process[i].store(value, std::memory_order_relaxed);
  return LMH_OK;
}
int lmh_opt(const char* name) {   // internal reader (a dozen strcmp per convolution launch: nanoseconds)
  const int i = lmh_option_index(name);
  if (i < 0) return 0;
  if (g_option_thread.set[i]) return g_option_thread.value[i];
  lmh_options_init();
  return g_option_process[i].load(std::memory_order_relaxed);
}
extern "C" int lmh_get_option(const char* name, int* value) {         // what the calling thread's launches see
  if (!name || !value) return LMH_ERR_INVALID;
  if (lmh_option_index(name) < 0) { lmh_set_error("lmh_get_option: unknown option '%s'", name); return LMH_ERR_INVALID; }
  *value = lmh_opt(name);
  return LMH_OK;
}

extern "C" int lmh_version(void) { return 101; }
extern "C" const char* lmh_last_error(void) { return g_err; }
extern "C" int lmh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---- HIP events for callers without a HIP binding (bench.py times single kernels with them) --------------------
extern "C" void* lmh_event_create(void) {
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return (void*)e;
}
extern "C" void lmh_event_destroy(void* e) {
  if (e) (void)hipEventDestroy((hipEvent_t)e);
}
extern "C" float lmh_event_elapsed_ms(void* e0, void* e1) {
  float ms = -1.f;
  if (hipEventSynchronize((hipEvent_t)e1) != hipSuccess) return -1.f;
  if (hipEventElapsedTime(&ms, (hipEvent_t)e0, (hipEvent_t)e1) != hipSuccess) return -1.f;
  return ms;
}

// What an event pair around ONE kernel measures on top of the kernel's own duration: the dispatch latency between the
// start event's completion and the kernel's first wave, and the end event's own processing.  Measured with an empty
// kernel (median of `reps` pairs on `stream`, which is synchronised); bench.py subtracts it from its per-launch event
// timings so that they are comparable with rocprofv3's kernel durations (begin / end timestamps of the dispatch itself).
__global__ void k_noop() {}
extern "C" float lmh_event_pair_overhead_ms(int reps, lmh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (reps < 1) reps = 1;
  if (reps > 64) reps = 64;
  hipEvent_t e0[64], e1[64];
  float ms[64];
  for (int i = 0; i < reps; ++i) {
    if (hipEventCreate(&e0[i]) != hipSuccess || hipEventCreate(&e1[i]) != hipSuccess) return -1.f;
  }
  for (int i = 0; i < reps; ++i) {
    lmh_launch(k_noop, dim3(1), dim3(64), 0, st);       // something in front, like in a real step
    (void)hipEventRecord(e0[i], st);
    lmh_launch(k_noop, dim3(1), dim3(64), 0, st);
    (void)hipEventRecord(e1[i], st);
  }
  if (hipStreamSynchronize(st) != hipSuccess) return -1.f;
  for (int i = 0; i < reps; ++i) {
    if (hipEventElapsedTime(&ms[i], e0[i], e1[i]) != hipSuccess) ms[i] = 0.f;
    (void)hipEventDestroy(e0[i]);
    (void)hipEventDestroy(e1[i]);
  }
  for (int i = 1; i < reps; ++i)          // insertion sort: median
    for (int j = i; j > 0 && ms[j] < ms[j - 1]; --j) { const float t = ms[j]; ms[j] = ms[j - 1]; ms[j - 1] = t; }
  return ms[reps / 2];
}

// ---- deferred weight-gradient tails: per-thread switch + the plan of the last deferred call ----------------------
thread_local int g_lmh_defer_tail = 0;
thread_local lmh_tail_plan g_lmh_last_plan = {nullptr, 0, nullptr, 0};
extern "C" void lmh_tail_defer(int on) {
  g_lmh_defer_tail = on;
  g_lmh_last_plan = lmh_tail_plan{nullptr, 0, nullptr, 0};
}
extern "C" void lmh_tail_last_plan(const float** slabs, int* splits, const float** colpart, int* colrows) {
  if (slabs) *slabs = g_lmh_last_plan.slabs;
  if (splits) *splits = g_lmh_last_plan.splits;
  if (colpart) *colpart = g_lmh_last_plan.colpart;
  if (colrows) *colrows = g_lmh_last_plan.colrows;
}

// lmh_stream_wait_stream, lmh_event_record, lmh_stream_wait_event, lmh_memset and the launch plans: plan.hip
