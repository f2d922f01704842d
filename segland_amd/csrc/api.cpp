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

// ---- test / tuning hooks (include/segland_hip_debug.h).  They write ONE process-wide record; not thread-safe, not for production callers.
#include "../../include/segland_hip_debug.h"
SlDebugState g_sl_debug;
extern "C" void sl_debug_reset(void) { g_sl_debug = SlDebugState(); }
extern "C" void sl_debug_conv_affine(int v) { g_sl_debug.conv_affine = v ? 1 : 0; }
extern "C" void sl_debug_conv_p9(int v) { g_sl_debug.conv_p9 = (v & 1) ? 1 : 0; }
extern "C" void sl_debug_conv_ring192(int v) { g_sl_debug.conv_ring192 = v ? 1 : 0; }
extern "C" void sl_debug_conv_ringn64(int v) { g_sl_debug.conv_ringn64 = v ? 1 : 0; }
extern "C" void sl_debug_conv_rows_small(int v) { g_sl_debug.conv_rows_small = v ? 1 : 0; }
extern "C" void sl_debug_conv_parity(int v) { g_sl_debug.conv_parity = v ? 1 : 0; }
extern "C" void sl_debug_ppm_fact_walk(int v) { g_sl_debug.ppm_fact_walk = v ? 1 : 0; }
extern "C" void sl_debug_ring64_max_tiles(int v) { g_sl_debug.ring64_max_tiles = v; }
extern "C" void sl_debug_ring_small_k(int v) { g_sl_debug.ring_small_k = v; }
extern "C" void sl_debug_wgrad3(int v) { g_sl_debug.wgrad3 = v ? 1 : 0; }
extern "C" void sl_debug_wgrad_bias(int v) { g_sl_debug.wgrad_bias = v ? 1 : 0; }
extern "C" void sl_debug_wgrad_tr(int v) { g_sl_debug.wgrad_tr = v ? 1 : 0; }
extern "C" void sl_debug_wgrad_pair_min(int rows) { g_sl_debug.wgrad_pair_min_rows = rows; }
extern "C" void sl_debug_attn_valu(int v) { g_sl_debug.attn_valu = v; }
extern "C" void sl_debug_p8_trace(void* buf) { g_sl_debug.p8_trace = (unsigned long long*)buf; }
extern "C" void sl_debug_wgrad_trace(void* buf) { g_sl_debug.wgrad_trace = (unsigned long long*)buf; }
extern "C" void sl_debug_wgrad3_trace(void* buf) { g_sl_debug.wgrad3_trace = (unsigned long long*)buf; }
extern "C" void sl_debug_attn_trace(void* buf) { g_sl_debug.attn_trace = (unsigned long long*)buf; }
