// Error plumbing + misc entry points of libluminoth_hip.so.
#include <stdarg.h>

#include "lmh_common.h"

static thread_local char g_err[512] = "";

void lmh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int lmh_version(void) { return 100; }
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
