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
