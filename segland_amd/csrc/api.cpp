// Error string + version of libsegland_hip.so
#include <stdarg.h>
#include <stdio.h>
#include "common.h"

static thread_local char g_err[512] = "";

void sl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int sl_version(void) { return 100; }
extern "C" const char* sl_last_error_string(void) { return g_err; }
// the runtime's sticky "last error" of this thread, read and reset: after a failed stream capture the next SL_LAUNCH_CHECK would otherwise report THAT error for a
// launch that went through (the kernel-by-kernel step right after a failed capture attempt: graph_step.py / bucket_step.py call this in their except branch)
extern "C" int sl_hip_clear_error(void) { return (int)hipGetLastError(); }
