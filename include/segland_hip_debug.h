/* Test and tuning hooks of libsegland_hip.so.  NOT part of the product ABI (include/segland_hip.h): a deployment never calls these.
 *
 * Every hook writes one process-wide record (SlDebugState, segland_amd/csrc/common.h) that the dispatch reads; the record is at its defaults unless a hook
 * was called, and sl_debug_reset() puts it back.  The hooks are NOT thread-safe and must not be called while launches of other threads are being issued.
 * Users: tests/ (route A vs route B bit-identity tests; tests/conftest.py resets the record after every GPU test, also when the test failed), the tools/ trace scripts
 * (s_memtime phase stamps), tools/with_hook.py (same-box A/B of a whole bench run with one route switched).
 */
#ifndef SEGLAND_HIP_DEBUG_H
#define SEGLAND_HIP_DEBUG_H
#include "segland_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

void sl_debug_reset(void);                 /* every override back to its default, every trace buffer detached */

/* dispatch overrides (1 = default route, 0 = the route it replaced) */
void sl_debug_conv_affine(int on);         /* branch-free affine store phase of biased / folded-BN epilogues vs the generic store phase */
void sl_debug_conv_p9(int on);             /* 3x3 patch kernel (conv_gemm_p9_kernel) vs the half-tile / ring kernels */
void sl_debug_conv_ring192(int on);        /* 128 x 192 ring tiles for 192-multiple output widths vs the two-stage 256 x 64 kernel */
void sl_debug_conv_ringn64(int on);        /* 128 x 64 ring tiles for 64-column inference layers vs the two-stage kernel */
void sl_debug_conv_rows_small(int on);     /* <= 32-row launches on conv_rows_small_kernel vs the tile kernels */
void sl_debug_ppm_fact_walk(int on);       /* factorised PPM prior path: sliding-window scatter / gather kernels vs the general two-stage kernels */
void sl_debug_conv_parity(int on);         /* stride-2 3x3 data gradients as four parity-plane launches vs one launch over all nine taps */
void sl_debug_ring_small_k(int k);        /* big-M layers with N % 128 == 0 and at most k reduction elements on 128 x 128 ring tiles, two blocks per CU (default 128; 0: 256 x 128 tiles) */
void sl_debug_ring64_max_tiles(int tiles); /* 64 x 128 ring tiles up to this many 128 x 128 tiles (default 256, 0: never) */
void sl_debug_wgrad3(int on);              /* nine-tap 3x3 weight gradient (conv_wgrad3_kernel) vs the per-tap kernels */
void sl_debug_wgrad_bias(int on);          /* bias-gradient column sums inside the weight-gradient kernel vs in the slab-reduce launch */
void sl_debug_wgrad_tr(int on);            /* ds_read_b64_tr_b16 fragment reads vs scalar LDS reads (bit-identical) */
void sl_debug_wgrad_pair_min(int rows);    /* pixel-pair weight gradients from this many rows (0: the built-in rule) */
void sl_debug_attn_valu(int v);            /* 1: window attention on the VALU reference kernels, -1: default (MFMA for bf16) */

/* s_memtime phase stamps: a device buffer the named kernel fills, NULL detaches (production) */
void sl_debug_p8_trace(void* buf);         /* conv_gemm_p8_kernel: [blocks][8] u64 */
void sl_debug_wgrad_trace(void* buf);      /* conv_wgrad_glds_kernel: [blocks][8] u64 */
void sl_debug_wgrad3_trace(void* buf);     /* conv_wgrad3_kernel: [blocks][8] u64 */
void sl_debug_attn_trace(void* buf);       /* window_attention_bwd_mfma2_kernel: [blocks][16] u64 */

/* the nine-tap weight gradient's plan for a shape (host logic only): out[8] = {eligible, splits, pieces per block, ...}; returns 1 when the shape is served */
int sl_debug_wgrad3_plan(const SlConvDesc* d, int* out);

#ifdef __cplusplus
}
#endif
#endif
