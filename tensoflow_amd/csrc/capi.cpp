// Error plumbing and version of libtensoflow_hip.so (no exceptions cross the C ABI).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/tensoflow_hip.h"

static thread_local char g_err[512] = "";

void tf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* tf_last_error(void) { return g_err; }
extern "C" int tf_version(void) { return 100; /* 0.1.0 */ }
