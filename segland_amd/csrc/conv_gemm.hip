// Implicit-GEMM convolution on MFMA for gfx950, forward (gather) and data-gradient (transposed gather): the dispatch (ONE predicate chain, choose_kernel) and the C entry
// points.  The kernels live in conv_gemm_tiles.hip / conv_gemm_patch.hip / conv_gemm_sk.hip, their shared pieces in conv_gemm_common.h.
#include "conv_gemm_common.h"

using namespace slconv;
namespace {

// rows per block of the kernel that will run for an M-row problem (also the granularity of the BN partial statistics): 256-row tiles need enough row blocks
// to fill 256 CUs; the 4-stage ring replaces the 2-stage kernel from 16 blocks of 128 rows (Swin-T stage 3 / 4 GEMMs of 8 192 / 2 048 tokens: +4 %)
constexpr int MIN_TILES256 = 96, RING128_MIN = 16;
#ifndef SL_C64K3_GATE
#define SL_C64K3_GATE 1       // 0: the gated 64 -> 64 3x3 data gradient back on the two-stage 256 x 64 tile kernel (A/B builds, tools/build_variant.sh)
#endif
static int block_rows(long long M) { return M >= 256LL * MIN_TILES256 ? 256 : 128; }
// (Measured and dropped, tools/ft_shapes.py: 256 x 256 tiles by TILE count on short M -- 8 192 rows x 1024 / 2048 channels are 128 / 256 tiles -- lose to the 128 x 128
// ring kernel with two blocks per CU on three of four shapes: 256 -> 1024 25.0 vs 11.2 us, 512 -> 1024 29.4 vs 16.3, 512 -> 2048 35.1 vs 32.7, 1024 -> 2048 44.9 vs 51.0.)

// Split-K plan of an inference conv (ConvGemmParams::ksplit): parts > 1 for a 3x3 layer of the patch kernel's kind with too few 16 x 16-pixel tiles for the chip and at least
// 1024 input channels to cut by 64-channel chunks: the pyramid conv of a fine-tune pair (8 192 rows, 2048 -> 512: 64 tiles) runs in 152 instead of 251 us on the 128 x 128
// ring kernel.  Measured and left unsplit (tools/ft_shapes.py, us split / unsplit): 3x3 512 -> 512 d4 68.4 / 67.1, 3x3 256 -> 256 d2 44.8 / 36.8, and every 1x1 layer
// (half-tile kernel by K-tiles: 2048 -> 512 52 / 33, 1024 -> 256 35 / 19) -- the partial tiles' round trip costs what the extra blocks gain.
static int splitk_parts(const ConvGemmParams& p, int dtype) {
  if (dtype != SL_BF16 || p.stat_partial || p.gate || p.mask_src || p.addend_mask || p.out2 || p.row_scale || p.C2 || p.N % 256 || p.M % 256 || p.C1 % 64) return 1;
  const long long tiles = (long long)(p.M / 256) * (p.N / 256);
  if (tiles >= 128 || p.M >= 32768 || p.C1 < 1024) return 1;
  if (!(p9_on() && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == p.dil && (p.dil == 1 || p.dil == 2 || p.dil == 4) && p.Hs == p.Hd && p.Ws == p.Wd && p.Hs % 16 == 0 && p.Ws % 16 == 0))
    return 1;
  const int units = p.C1 / 64;
  int s = 1;
  while (s < 8 && units / (2 * s) >= 4 && tiles * 2 * s <= 256) s *= 2;
  return s;
}

// route overrides of the tests (include/segland_hip_debug.h); all on in production
static bool ring192_on() { return g_sl_debug.conv_ring192 != 0; }          // the 128 x 192 ring tile
static bool rows_small_on() { return g_sl_debug.conv_rows_small != 0; }    // <= 32-row launches on conv_rows_small_kernel
static bool ringn64_on() { return g_sl_debug.conv_ringn64 != 0; }          // 64-column inference layers on 128 x 64 ring tiles
// Which kernel a launch runs on: 1000000 * family + 1000 * BM + BN (family 8 = 3x3 patch (+ 10000000: split-K), 7 = 64 -> 64 patch,
// 6 = pixel-stationary K <= 256, 5 = half-tile, 4 = ring, 2 = two-stage glds).  The ONE predicate chain: launch_gemm switches on it, sl_conv2d_tile_config(_ex) and
// sl_conv2d_stat_rows answer from it (round-4 advisor: the query had drifted from the dispatch).
static int choose_kernel(const ConvGemmParams& p, int dtype) {
  const bool n128 = (p.N % 128 == 0), n256 = (p.N % 256 == 0);
  const bool big = block_rows(p.M) == 256;           // tiny problems (PPM stages, prototype rows) stay on 128-row tiles
  if (dtype == SL_BF16) {
    // <= 32 rows of a 1x1 layer (prototype rows of the POP head's MLP): one wave per 32 columns, operands straight from global memory (conv_rows_small_kernel)
    if (rows_small_on() && p.M <= 32 && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.C2 == 0 && p.Hs == p.Hd && p.Ws == p.Wd && p.N % 32 == 0 && p.ksplit <= 1 &&
        !(p.bias || p.scale || p.addend || p.pre_addend || p.row_scale || p.out2 || p.stat_partial || p.gate || p.addend_mask || (p.flags & 4)))
      return 3032032;
    if (p.ksplit > 1) return 18256256;                                                // planned by splitk_parts: the shape is served by the patch kernel
    if (c64k3_shape(SL_BF16, p.KH, p.KW, p.stride, p.pad, p.dil, p.C1 + p.C2, p.C1, p.N, p.M) && p.Hs == p.Hd && p.Ws == p.Wd &&
        !(p.bias || p.scale || p.relu || p.addend || p.mask_src || p.pre_addend || p.row_scale || p.out2) && (!p.gate || (SL_C64K3_GATE && p.stat_partial && p.bn_x)))
      return 7016016;                                                                  // (round 6: also the gated data gradient with BN-backward column sums)
    if (sk_shape(SL_BF16, p.KH, p.KW, p.stride, p.pad, p.C1 + p.C2, p.C1, p.N, p.M) && p.Hs == p.Hd && p.Ws == p.Wd &&
        !(p.bias || p.scale || p.relu || p.mask_src || p.pre_addend || p.row_scale || p.out2) && (p.gate || !(p.addend && p.stat_partial)) && (p.addend || !p.addend_mask) &&
        (!p.gate || (p.addend && !p.addend_mask && p.stat_partial)) && (!(p.addend_mask || p.gate) || (p.N % 128 == 0 && p.N <= 1024)))
      return 6256064;
    if (p9_shape(p) && !(p.flags & 4)) return 8256256;
    // half-tile kernel: needs the affine row -> pixel map (forward, or data gradient of a stride-1 conv) and <= 32 taps in the mask
    if (big && n256 && (p.mode == 0 || p.stride == 1) && !(p.flags & 4)) return 5256256;      // (flags bit 2, a GELU behind the data gradient: store phases of the tile kernels only)
  }
  // 192-column tiles (N = 192, 576, ...: Swin-T stage 2 -- qkv / proj / fc2 and their data gradients on 32 768 tokens): these widths are 64- but not 128-multiples and
  // ran on the two-stage 256 x 64 kernel
  // (a data gradient behind a GELU -- flags bit 2 -- takes them for every 192-multiple: the persistent half-tile kernel has no fast store phase for it; Swin-T stage 2: 768)
  if (dtype == SL_BF16 && ring192_on() && p.N % 192 == 0 && (!n128 || (p.flags & 4)) && p.M >= 128LL * RING128_MIN) return 4128192;
  // 64-column inference layers (a fine-tune pair's frozen layer1: 256 -> 64 and 3x3 64 -> 64 on 32 768 rows): 128 x 64 tiles of the ring kernel, four waves of 32 rows x 64
  // columns, three blocks per CU -- 256 blocks where the two-stage 256 x 64 kernel has 128.  Launches without BN statistic partials only (the partial rows' granularity).
  if (dtype == SL_BF16 && ringn64_on() && !n128 && p.N % 192 != 0 && !p.stat_partial && !p.gate && p.M >= 128LL * RING128_MIN) return 4128064;
  // short reductions on many rows (Swin stage 1: 128 -> 384 on 131 072 tokens): a 256 x 128 tile's ring holds the whole reduction (96 KiB, one block per CU), so load, MFMA
  // and store phases of a CU run one after the other; two 64 KiB blocks of 128 x 128 tiles overlap one block's store phase with the other's loads (hook: sl_debug_ring_small_k)
  if (big && dtype == SL_BF16 && n128 && !n256 && g_sl_debug.ring_small_k > 0 && (long long)p.KH * p.KW * (p.C1 + p.C2) <= g_sl_debug.ring_small_k && !p.stat_partial && !p.gate)
    return 4128128;
  if (big) {
    // fp32 stages hold 16 (64-byte rows) or 32 K elements; channel counts are multiples of 64, so both divide
    if (n256) return 4256256;
    if (n128) return 4256128;
    return 2256064;                                     // N = 64 layers: too few weight rows for a 64-byte-row ring
  }
  // 128 x 128 tiles (few rows: Swin stage 3 / 4 token maps, the fine-tune pair's 8 192-row layers): 64-byte rows, 4 stages, 64 KiB, two blocks per CU.  Round 4 measured
  // the other stage geometries of this template end to end (Swin-T POP tiles/s / fine-tune pairs/s, one box): 128-byte rows x 3 stages (96 KiB, one block per CU)
  // 696.6 / 379.5, x 4 stages 712.3 / 408.1, 128-byte rows x 3 stages on eight waves of 64 x 32 705.7 / 392.0 -- against 720.2 / 419.1 for this one
  if (dtype == SL_BF16) {
    // few 128 x 128 tiles (the fine-tune pair's 8 192-row layers with 128 / 256 output channels: 64 / 128 tiles on 256 CUs; Swin stage 4 projections): 64-row tiles, bit-identical
    // results.  tools/ring64_check.py (us, 128 x 128 -> 64 x 128): 1024 -> 256 19.0 -> 15.4, 3x3 256 -> 256 d2 37.4 -> 30.0, 512 -> 128 11.6 -> 9.3, 3x3 128 -> 128 21.5 -> 16.7;
    // from 256 tiles on the smaller tile loses (2048 -> 512 36.1 -> 38.9, 256 -> 1024 11.1 -> 13.3).  End to end, one box: 440.9 -> 454.3 pairs/s (ResNet-50), 389 -> 406 (Swin-T),
    // Swin-T training step 733.9 -> 736.8 tiles/s; a limit of 200 / 300 tiles: 453.1 / 447.5 pairs/s.  Launches without BN statistic partials only (a training conv's
    // partials keep the 128-row granularity sl_conv2d_stat_rows promises).
    // Round 5 (profiles/r5_ab_ring64_geom.txt): the 64-row tiles stream 128-byte stage rows (conv_gemm_tiles.hip: launch_tile): 1024 -> 256 16.0 -> 13.6 us, 3x3 256 -> 256 d2
    // 30.7 -> 26.0, 2048 -> 512 (256 tiles) 36.7 -> 32.6 where it lost before; 459.5 -> 473 pairs/s (ResNet-50), 400 -> 413 (Swin-T) with the limit at 256 tiles
    // (320: 469, 512: 465).  1x1 layers stay bit-identical to the 128 x 128 tiles; 3x3 layers sum (64-channel chunk, tap) instead of (32-channel chunk, tap).
    if (n128 && !p.stat_partial && !p.gate && p.M >= 128LL * RING128_MIN && (long long)cdiv(p.M, 128) * (p.N / 128) <= g_sl_debug.ring64_max_tiles) return 4064128;
  }
  if (n128 && p.M >= 128LL * RING128_MIN) return 4128128;
  if (n128) return 2128128;
  return 2128064;
}

// Data gradient of a stride-2 3x3 conv (resnet.py:46, a stage entry's conv2): a destination pixel of parity (py, px) receives only the taps with ky = py + pad, kx = px + pad
// (mod 2) -- one, two, two or four of the nine.  The undecomposed launch multiplies all nine per pixel, three quarters of them against the zero page (layer2.0.conv2 at the
// bench shape: 177 us at 109 TFLOP/s; its forward takes 41).  Here the four parity classes run as four launches of the ring kernel's SUBP instantiation, each over the
// half-resolution grid with its own tap list, writing its pixels in place (the store phases address 2 N elements apart).  Same store phases, same partial-row count.
static int parity_cfg(const ConvGemmParams& p, int dtype) {
  if (!g_sl_debug.conv_parity || dtype != SL_BF16 || p.mode != 1 || p.stride != 2 || p.KH != 3 || p.KW != 3 || p.C2 || (p.C1 % 32)) return 0;
  if (p.Hd != 2 * p.Hs || p.Wd != 2 * p.Ws || (p.N % 128)) return 0;
  if (p.bias || p.scale || p.relu || p.mask_src || p.pre_addend || p.row_scale || p.out2 || p.addend_mask || p.ksplit > 1 || (p.flags & 4) || p.addend_half) return 0;
  if (p.gate ? (p.addend || !p.stat_partial) : (p.addend && p.stat_partial)) return 0;             // the three fast store phases: gated statistics, plain (+ statistics), + addend
  const int Wh = p.Wd / 2, hw = (p.Hd / 2) * Wh;
  const long long Ms = (long long)p.B * hw;
  if ((Wh & (Wh - 1)) || Wh % 32 || hw % 256 || Ms < 256LL * MIN_TILES256) return 0;                // a sweep inside one image row, a tile inside one image, enough tiles per plane
  return p.N % 256 == 0 ? 4256256 : 4256128;
}
static int launch_parity_planes(int cfg, int dtype, const ConvGemmParams& p, hipStream_t st) {
  const long long Ms = (long long)p.B * (p.Hd / 2) * (p.Wd / 2);
  for (int k = 0; k < 4; ++k) {
    ConvGemmParams q = p;
    q.sub = 1; q.sub_py = k >> 1; q.sub_px = k & 1; q.M = (int)Ms;
    if (q.stat_partial) q.stat_partial = p.stat_partial + (size_t)k * (Ms / 256) * 2 * p.N;       // plane k's row blocks: partial rows [k Ms / 256, (k + 1) Ms / 256)
    if (int e = launch_tile(cfg, dtype, q, st)) return e;
  }
  return 0;
}

int launch_gemm(int dtype, ConvGemmParams& p, hipStream_t st) {
  if (const int pc = parity_cfg(p, dtype)) return launch_parity_planes(pc, dtype, p, st);
  const int cfg = choose_kernel(p, dtype);
  if (dtype == SL_BF16) {
    switch (cfg) {
      case 18256256: case 8256256: return launch_p9(p, st);
      case 7016016: return launch_c64k3(p, st);
      case 6256064: return launch_sk(p, st);
      case 5256256: return launch_p8(p, st);
      default: break;
    }
  }
  return launch_tile(cfg, dtype, p, st);       // families 4 (ring) and 2 (two-stage): conv_gemm_tiles.hip
}

int run_gemm(int dtype, ConvGemmParams& p, hipStream_t st) {
  if (!g_sl_debug.conv_affine) p.flags |= 2;      // test hook: the generic store phase instead of the branch-free affine one (bit-identity test)
  const int bke = dtype == SL_BF16 ? 64 : 32;
  SL_REQUIRE(dtype == SL_BF16 || dtype == SL_F32, "conv: bad dtype %d", dtype);
  SL_REQUIRE(p.C1 > 0 && p.C1 % bke == 0 && p.C2 % bke == 0, "conv: source channels (%d,%d) must be multiples of %d", p.C1, p.C2, bke);
  SL_REQUIRE(p.N > 0 && p.N % 64 == 0, "conv: output channels %d must be a multiple of 64", p.N);
  SL_REQUIRE(p.M > 0, "conv: empty output");
  return launch_gemm(dtype, p, st);
}

int check_desc(const SlConvDesc* d) {
  SL_REQUIRE(d != nullptr, "conv: null descriptor");
  SL_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "conv: bad sizes");
  SL_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0, "conv: bad window");
  const int ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
  const int wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
  SL_REQUIRE(ho == d->Ho && wo == d->Wo, "conv: Ho/Wo (%d,%d) inconsistent with input (expected %d,%d)", d->Ho, d->Wo, ho, wo);
  SL_REQUIRE(d->C1 > 0 && d->C1 <= d->Cin, "conv: bad C1");
  return 0;
}

}  // namespace

// Which kernel a launch of this shape runs on (codes: choose_kernel): the SAME function launch_gemm switches on, applied to the parameter block the entry points would
// build.  mode 0: forward, 1: data gradient.  epi (sl_conv2d_tile_config_ex): what the launch carries besides the raw result -- SL_EPI_STATS (BN statistic partials),
// SL_EPI_AFFINE (bias / folded BN / ReLU / residual: the inference forms), SL_EPI_ADDEND (data gradient + shortcut gradient), SL_EPI_ADDEND_BITS (gated by ReLU bits),
// SL_EPI_GATE (gated result + BN-backward column sums, sl_conv2d_bwd_data_bnstat), SL_EPI_SPLITK (the split-K plan of sl_conv2d_affine_fwd_ex applies).
// sl_conv2d_tile_config(d, mode) = the training forms: forward with statistics, plain data gradient.
static unsigned char g_cfg_dummy[16];
extern "C" int sl_conv2d_tile_config_ex(const SlConvDesc* d, int mode, int epi) {
  if (!d) return SL_EINVAL;
  ConvGemmParams p{};
  void* dm = (void*)g_cfg_dummy;
  p.src1 = dm; p.wt = dm; p.out = dm;
  if (mode == 0) {
    p.C1 = d->C1; p.C2 = d->Cin - d->C1; p.B = d->B; p.Hs = d->H; p.Ws = d->W; p.Hd = d->Ho; p.Wd = d->Wo; p.N = d->Cout; p.M = d->B * d->Ho * d->Wo;
  } else {
    p.C1 = d->Cout; p.C2 = 0; p.B = d->B; p.Hs = d->Ho; p.Ws = d->Wo; p.Hd = d->H; p.Wd = d->W; p.N = d->Cin; p.M = d->B * d->H * d->W;
  }
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.mode = mode;
  if (epi & SL_EPI_STATS) p.stat_partial = (float*)dm;
  if (epi & SL_EPI_AFFINE) { p.scale = (const float*)dm; p.bias = (const float*)dm; p.relu = 1; }
  if (epi & (SL_EPI_ADDEND | SL_EPI_ADDEND_BITS)) p.addend = dm;
  if (epi & SL_EPI_ADDEND_BITS) p.addend_mask = (const unsigned char*)dm;
  if (epi & SL_EPI_GATE) { p.gate = (const unsigned char*)dm; p.bn_x = dm; p.bn_mean = (const float*)dm; p.bn_invstd = (const float*)dm; p.stat_partial = (float*)dm; }
  if (epi & SL_EPI_GELU) { p.mask_src = dm; p.flags = 4; }
  p.ksplit = (epi & SL_EPI_SPLITK) ? splitk_parts(p, d->dtype) : 1;
  return choose_kernel(p, d->dtype);
}
extern "C" int sl_conv2d_tile_config(const SlConvDesc* d, int mode) { return sl_conv2d_tile_config_ex(d, mode, mode == 0 ? SL_EPI_STATS : 0); }

extern "C" int sl_conv2d_tile_config(const SlConvDesc* d, int mode);
// Rows of the BN statistic partials a forward launch writes: derived from the SAME predicate chain as launch_gemm (sl_conv2d_tile_config), so a
// change of the dispatch thresholds can never make the caller allocate rows the kernel does not write.
extern "C" int sl_conv2d_stat_rows(const SlConvDesc* d) {
  if (!d) return SL_EINVAL;
  const long long M = (long long)d->B * d->Ho * d->Wo;
  const int cfg = sl_conv2d_tile_config(d, 0);
  if (cfg == 7016016) return d->B * cdiv(d->H, C64_T) * cdiv(d->W, C64_T);    // conv_c64k3_kernel: one row per 16 x 16 tile
  return (int)cdiv(M, (long long)((cfg / 1000) % 1000));
}

extern "C" int sl_conv2d_fwd_ex(const SlConvDesc* d, const void* x, const void* x2, const void* w, const void* pre_addend,
                                const float* bias, int relu, void* y, float* stat_partial, sl_stream_t stream);

extern "C" int sl_conv2d_fwd(const SlConvDesc* d, const void* x, const void* x2, const void* w, const float* bias,
                             int relu, void* y, float* stat_partial, sl_stream_t stream) {
  return sl_conv2d_fwd_ex(d, x, x2, w, nullptr, bias, relu, y, stat_partial, stream);
}

extern "C" int sl_conv2d_fwd_ex(const SlConvDesc* d, const void* x, const void* x2, const void* w, const void* pre_addend,
                                const float* bias, int relu, void* y, float* stat_partial, sl_stream_t stream) {
  if (int e = check_desc(d)) return e;
  SL_REQUIRE(x && w && y, "conv fwd: null buffer");
  SL_REQUIRE(d->C1 == d->Cin || x2, "conv fwd: x2 missing for a concat input");
  SL_REQUIRE(!(stat_partial && (bias || relu)), "conv fwd: statistics are defined on the raw conv output");
  ConvGemmParams p{};
  p.src1 = x; p.src2 = x2; p.C1 = d->C1; p.C2 = d->Cin - d->C1; p.wt = w; p.out = y;
  p.B = d->B; p.Hs = d->H; p.Ws = d->W; p.Hd = d->Ho; p.Wd = d->Wo;
  p.N = d->Cout; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.mode = 0;
  p.bias = bias; p.relu = relu; p.stat_partial = stat_partial; p.pre_addend = pre_addend;
  p.M = d->B * d->Ho * d->Wo;
  return run_gemm(d->dtype, p, (hipStream_t)stream);
}

// nn.Linear as a 1x1 conv with the elementwise tail of a transformer block in the epilogue:
//   y = row_scale[b] * (x w^T + bias) + residual          (attention proj / Mlp fc2 with DropPath, swintransformer.py:246-249)
//   y = x w^T + bias,  gelu_out = GELU(y)                  (Mlp fc1 + act, swintransformer.py:36)
extern "C" int sl_linear_fwd(const SlConvDesc* d, const void* x, const void* w, const float* bias, const float* row_scale, const void* residual,
                             void* y, void* gelu_out, sl_stream_t stream) {
  if (int e = check_desc(d)) return e;
  SL_REQUIRE(x && w && y, "linear fwd: null buffer");
  SL_REQUIRE(d->C1 == d->Cin, "linear fwd: single input tensor");
  ConvGemmParams p{};
  p.src1 = x; p.src2 = nullptr; p.C1 = d->C1; p.C2 = 0; p.wt = w; p.out = y;
  p.B = d->B; p.Hs = d->H; p.Ws = d->W; p.Hd = d->Ho; p.Wd = d->Wo;
  p.N = d->Cout; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.mode = 0;
  p.bias = bias; p.row_scale = row_scale; p.addend = residual; p.out2 = gelu_out;
  p.M = d->B * d->Ho * d->Wo;
  return run_gemm(d->dtype, p, (hipStream_t)stream);
}

static void affine_params(ConvGemmParams& p, const SlConvDesc* d, const void* x, const void* x2, const void* w, const void* pre_addend, const float* scale,
                          const float* shift, const void* residual, int relu, void* y) {
  p.src1 = x; p.src2 = x2; p.C1 = d->C1; p.C2 = d->Cin - d->C1; p.wt = w; p.out = y;
  p.B = d->B; p.Hs = d->H; p.Ws = d->W; p.Hd = d->Ho; p.Wd = d->Wo;
  p.N = d->Cout; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.mode = 0;
  p.scale = scale; p.bias = shift; p.relu = relu; p.addend = residual; p.pre_addend = pre_addend;
  p.M = d->B * d->Ho * d->Wo;
}

extern "C" int sl_conv2d_affine_fwd_ex(const SlConvDesc* d, const void* x, const void* x2, const void* w, const void* pre_addend, const float* scale,
                                       const float* shift, const void* residual, int relu, void* y, void* workspace, size_t workspace_bytes, sl_stream_t stream);

extern "C" int sl_conv2d_affine_fwd(const SlConvDesc* d, const void* x, const void* x2, const void* w, const float* scale,
                                    const float* shift, const void* residual, int relu, void* y, sl_stream_t stream) {
  return sl_conv2d_affine_fwd_ex(d, x, x2, w, nullptr, scale, shift, residual, relu, y, nullptr, 0, stream);
}

// bytes of split-K workspace sl_conv2d_affine_fwd_ex can use for this layer (0: the layer is not split)
extern "C" size_t sl_conv2d_affine_fwd_workspace(const SlConvDesc* d) {
  if (!d || check_desc(d)) return 0;
  ConvGemmParams p{};
  affine_params(p, d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
  const int parts = splitk_parts(p, d->dtype);
  return parts > 1 ? (size_t)parts * p.M * p.N * sizeof(float) : 0;
}

extern "C" int sl_conv2d_affine_fwd_ex(const SlConvDesc* d, const void* x, const void* x2, const void* w, const void* pre_addend, const float* scale,
                                       const float* shift, const void* residual, int relu, void* y, void* workspace, size_t workspace_bytes, sl_stream_t stream) {
  if (int e = check_desc(d)) return e;
  SL_REQUIRE(x && w && y && scale && shift, "conv affine fwd: null buffer");
  SL_REQUIRE(d->C1 == d->Cin || x2, "conv affine fwd: x2 missing for a concat input");
  ConvGemmParams p{};
  affine_params(p, d, x, x2, w, pre_addend, scale, shift, residual, relu, y);
  const int parts = workspace ? splitk_parts(p, d->dtype) : 1;
  if (parts > 1 && workspace_bytes >= (size_t)parts * p.M * p.N * sizeof(float) && ((size_t)workspace & 15) == 0) { p.ws = (float*)workspace; p.ksplit = parts; }
  return run_gemm(d->dtype, p, (hipStream_t)stream);
}

extern "C" int sl_conv2d_bwd_data(const SlConvDesc* d, const void* dy, const void* wt, const void* addend, const uint8_t* addend_mask,
                                  const void* mask_src, void* dx, sl_stream_t stream) {
  if (int e = check_desc(d)) return e;
  SL_REQUIRE(dy && wt && dx, "conv bwd_data: null buffer");
  ConvGemmParams p{};
  p.src1 = dy; p.src2 = nullptr; p.C1 = d->Cout; p.C2 = 0; p.wt = wt; p.out = dx;
  p.B = d->B; p.Hs = d->Ho; p.Ws = d->Wo; p.Hd = d->H; p.Wd = d->W;
  p.N = d->Cin; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.mode = 1;
  p.addend = addend; p.mask_src = mask_src; p.addend_mask = addend ? addend_mask : nullptr;
  p.M = d->B * d->H * d->W;
  return run_gemm(d->dtype, p, (hipStream_t)stream);
}

// dx = (data gradient) * GELU'(h): the Mlp of a transformer block (swintransformer.py:26-31: fc2(act(fc1(x)))) -- the data gradient of fc2 lands behind the activation in
// one launch; h = the stored pre-activation fc1(x), [B][H][W][Cin].  Replaces sl_conv2d_bwd_data + sl_gelu_bwd (one write and two reads of the block's widest tensor less).
extern "C" int sl_conv2d_bwd_data_gelu(const SlConvDesc* d, const void* dy, const void* wt, const void* h, void* dx, sl_stream_t stream) {
  if (int e = check_desc(d)) return e;
  SL_REQUIRE(dy && wt && h && dx, "conv bwd_data_gelu: null buffer");
  ConvGemmParams p{};
  p.src1 = dy; p.src2 = nullptr; p.C1 = d->Cout; p.C2 = 0; p.wt = wt; p.out = dx;
  p.B = d->B; p.Hs = d->Ho; p.Ws = d->Wo; p.Hd = d->H; p.Wd = d->W;
  p.N = d->Cin; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.mode = 1;
  p.mask_src = h; p.flags = 4;
  p.M = d->B * d->H * d->W;
  return run_gemm(d->dtype, p, (hipStream_t)stream);
}

// Data gradient whose result is gated with the ReLU bits of its own positions and reduced for the BatchNorm backward of the layer below (see ConvGemmParams::gate).
// rows of stat_partial: sl_conv2d_bwd_data_bnstat_rows(d), 0 = this shape is not served (the caller runs sl_conv2d_bwd_data + sl_bn_bwd_reduce instead): served are the
// shapes that the tile kernels with the LDS-staged store phase take (half-tile, 3x3 patch, ring, two-stage) when every row block is full.
extern "C" int sl_conv2d_bwd_data_bnstat_rows(const SlConvDesc* d) {
  if (!d) return 0;
  const int cfg = sl_conv2d_tile_config_ex(d, 1, SL_EPI_GATE);
  const int fam = cfg / 1000000, bm = (cfg / 1000) % 1000;
  const long long M = (long long)d->B * d->H * d->W;
  if (cfg == 7016016) return d->B * cdiv(d->H, C64_T) * cdiv(d->W, C64_T);      // conv_c64k3_kernel: one row per 16 x 16 tile (as the forward's statistics)
  if (!(fam == 5 || fam == 8 || fam == 4 || fam == 2) || bm <= 0 || M % bm != 0) return 0;
  if (cfg % 1000 == 192) return 0;      // the 128 x 192 ring tile's gated-statistics store phase is not exercised by any BatchNorm model (192-multiple widths are Swin's LayerNorm layers): not offered
  return (int)(M / bm);
}

extern "C" int sl_conv2d_bwd_data_bnstat(const SlConvDesc* d, const void* dy, const void* wt, const uint8_t* gate, const void* bn_x, const float* bn_mean,
                                         const float* bn_invstd, void* dx, float* stat_partial, sl_stream_t stream) {
  if (int e = check_desc(d)) return e;
  SL_REQUIRE(dy && wt && dx && gate && bn_x && bn_mean && bn_invstd && stat_partial, "conv bwd_data_bnstat: null buffer");
  SL_REQUIRE(sl_conv2d_bwd_data_bnstat_rows(d) > 0, "conv bwd_data_bnstat: shape not served (sl_conv2d_bwd_data_bnstat_rows == 0)");
  ConvGemmParams p{};
  p.src1 = dy; p.src2 = nullptr; p.C1 = d->Cout; p.C2 = 0; p.wt = wt; p.out = dx;
  p.B = d->B; p.Hs = d->Ho; p.Ws = d->Wo; p.Hd = d->H; p.Wd = d->W;
  p.N = d->Cin; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.mode = 1;
  p.gate = gate; p.bn_x = bn_x; p.bn_mean = bn_mean; p.bn_invstd = bn_invstd; p.stat_partial = stat_partial;
  p.M = d->B * d->H * d->W;
  return run_gemm(d->dtype, p, (hipStream_t)stream);
}

// The same with the BatchNorm-backward APPLY pass of the conv's own output folded into the data gradient (resnet.py:66-70 backward of conv3 -> bn3, 1x1 convs):
//   dc3 = cA g + cB (c3 - mean) + cC  and  c3 = a2 W3^T   =>   da2 = dc3 W3 = [g | a2] [diag(cA) W3 ; W3^T diag(cB) W3] + (cC - cB mean) W3
// i.e. the data gradient reads the GATED incoming gradient g (Cout channels) and the conv's own input a2 (Cin channels) as one virtual concat against an extended
// weight wt_ext [Cin][Cout + Cin] (sl_bn_fold_weights) plus a per-column bias; the apply pass over (g, c3) -- 6 bytes per element of the Cout-channel tensor -- and
// the tensor dc3 disappear.  Served: the half-tile kernel's shapes (Cin % 256 == 0, whole 256-row tiles); rows of stat_partial as sl_conv2d_bwd_data_bnstat_rows(d).
extern "C" int sl_conv2d_bwd_data_bnstat_folded_rows(const SlConvDesc* d) {
  if (!d || d->dtype != SL_BF16 || d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->H != d->Ho || d->W != d->Wo) return 0;
  const long long M = (long long)d->B * d->H * d->W;
  if (d->Cin % 256 || d->Cout % 64 || M % 256 || block_rows((int)M) != 256) return 0;
  ConvGemmParams p{};
  p.src1 = p.src2 = p.wt = p.out = (void*)g_cfg_dummy;
  p.C1 = d->Cout; p.C2 = d->Cin; p.B = d->B; p.Hs = d->Ho; p.Ws = d->Wo; p.Hd = d->H; p.Wd = d->W; p.N = d->Cin; p.M = (int)M;
  p.KH = p.KW = 1; p.stride = 1; p.pad = 0; p.dil = 1; p.mode = 1; p.ksplit = 1;
  p.gate = (const unsigned char*)g_cfg_dummy; p.bn_x = g_cfg_dummy; p.bn_mean = p.bn_invstd = p.bias = (const float*)g_cfg_dummy; p.stat_partial = (float*)g_cfg_dummy;
  return choose_kernel(p, d->dtype) == 5256256 ? (int)(M / 256) : 0;
}

extern "C" int sl_conv2d_bwd_data_bnstat_folded(const SlConvDesc* d, const void* g, const void* x, const void* wt_ext, const float* bias, const uint8_t* gate, const void* bn_x,
                                                const float* bn_mean, const float* bn_invstd, void* dx, float* stat_partial, sl_stream_t stream) {
  if (int e = check_desc(d)) return e;
  SL_REQUIRE(g && x && wt_ext && bias && dx && gate && bn_x && bn_mean && bn_invstd && stat_partial, "conv bwd_data_bnstat_folded: null buffer");
  SL_REQUIRE(sl_conv2d_bwd_data_bnstat_folded_rows(d) > 0, "conv bwd_data_bnstat_folded: shape not served (sl_conv2d_bwd_data_bnstat_folded_rows == 0)");
  ConvGemmParams p{};
  p.src1 = g; p.src2 = x; p.C1 = d->Cout; p.C2 = d->Cin; p.wt = wt_ext; p.out = dx;
  p.B = d->B; p.Hs = d->Ho; p.Ws = d->Wo; p.Hd = d->H; p.Wd = d->W;
  p.N = d->Cin; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.dil = 1; p.mode = 1;
  p.bias = bias; p.gate = gate; p.bn_x = bn_x; p.bn_mean = bn_mean; p.bn_invstd = bn_invstd; p.stat_partial = stat_partial;
  p.M = d->B * d->H * d->W;
  return run_gemm(d->dtype, p, (hipStream_t)stream);
}

// The same across a block boundary (resnet.py:71-78 backward): the data gradient of conv1 plus the shortcut gradient `addend` IS the gradient wrt the previous block's
// output relu(bn3(c3) + res); gated with that ReLU's bits and reduced against c3 it hands the previous block its bn3 backward column sums -- its reduce pass over
// (dout, c3) disappears, and dout arrives gated.  Served: the shapes of the pixel-stationary kernel (1x1, K = 64 / 128 / 256, N % 128 == 0, N <= 1024, M % 256 == 0).
extern "C" int sl_conv2d_bwd_data_addend_bnstat_rows(const SlConvDesc* d) {
  if (!d) return 0;
  const long long M = (long long)d->B * d->H * d->W;
  if (d->dtype != SL_BF16 || d->H != d->Ho || d->W != d->Wo) return 0;
  if (!sk_shape(d->dtype, d->KH, d->KW, d->stride, d->pad, d->Cout, d->Cout, d->Cin, M) || d->Cin % 128 != 0 || d->Cin > 1024) return 0;
  return (int)(M / 256);
}

extern "C" int sl_conv2d_bwd_data_addend_bnstat(const SlConvDesc* d, const void* dy, const void* wt, const void* addend, const uint8_t* gate, const void* bn_x,
                                                const float* bn_mean, const float* bn_invstd, void* dx, float* stat_partial, sl_stream_t stream) {
  if (int e = check_desc(d)) return e;
  SL_REQUIRE(dy && wt && dx && addend && gate && bn_x && bn_mean && bn_invstd && stat_partial, "conv bwd_data_addend_bnstat: null buffer");
  SL_REQUIRE(sl_conv2d_bwd_data_addend_bnstat_rows(d) > 0, "conv bwd_data_addend_bnstat: shape not served (sl_conv2d_bwd_data_addend_bnstat_rows == 0)");
  ConvGemmParams p{};
  p.src1 = dy; p.src2 = nullptr; p.C1 = d->Cout; p.C2 = 0; p.wt = wt; p.out = dx;
  p.B = d->B; p.Hs = d->Ho; p.Ws = d->Wo; p.Hd = d->H; p.Wd = d->W;
  p.N = d->Cin; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.mode = 1;
  p.addend = addend; p.gate = gate; p.bn_x = bn_x; p.bn_mean = bn_mean; p.bn_invstd = bn_invstd; p.stat_partial = stat_partial;
  p.M = d->B * d->H * d->W;
  return run_gemm(d->dtype, p, (hipStream_t)stream);
}

// Data gradient + a HALF-RESOLUTION addend at the even positions (resnet.py:109-110, 71-76 backward of a stride-2 stage entry: the downsample branch is a 1x1 stride-2
// conv, whose data gradient is non-zero at the even positions only): addend_half [B][H/2][W/2][Cin] is the DENSE data gradient of that conv on its own output grid
// (sl_conv2d_bwd_data of the stride-1 form), added where (y, x) are both even -- the zero-filled full-resolution tensor (3/4 zeros, written and read back as an addend)
// never exists.  Served: the pixel-stationary kernel's shapes with even H, W (sl_conv2d_bwd_data_addend_half_ok); optional cross-block statistics as in
// sl_conv2d_bwd_data_addend_bnstat (gate / bn_x / bn_mean / bn_invstd / stat_partial all NULL: plain).
extern "C" int sl_conv2d_bwd_data_addend_half_ok(const SlConvDesc* d) {
  if (!d || d->dtype != SL_BF16 || d->H != d->Ho || d->W != d->Wo || (d->H & 1) || (d->W & 1)) return 0;
  return sk_shape(d->dtype, d->KH, d->KW, d->stride, d->pad, d->Cout, d->Cout, d->Cin, (long long)d->B * d->H * d->W) ? 1 : 0;
}
extern "C" int sl_conv2d_bwd_data_addend_half(const SlConvDesc* d, const void* dy, const void* wt, const void* addend_half, const uint8_t* gate, const void* bn_x,
                                              const float* bn_mean, const float* bn_invstd, void* dx, float* stat_partial, sl_stream_t stream) {
  if (int e = check_desc(d)) return e;
  SL_REQUIRE(dy && wt && dx && addend_half, "conv bwd_data_addend_half: null buffer");
  SL_REQUIRE(sl_conv2d_bwd_data_addend_half_ok(d), "conv bwd_data_addend_half: shape not served (sl_conv2d_bwd_data_addend_half_ok == 0)");
  SL_REQUIRE(!gate || (bn_x && bn_mean && bn_invstd && stat_partial && sl_conv2d_bwd_data_addend_bnstat_rows(d) > 0), "conv bwd_data_addend_half: statistics not served for this shape");
  ConvGemmParams p{};
  p.src1 = dy; p.src2 = nullptr; p.C1 = d->Cout; p.C2 = 0; p.wt = wt; p.out = dx;
  p.B = d->B; p.Hs = d->Ho; p.Ws = d->Wo; p.Hd = d->H; p.Wd = d->W;
  p.N = d->Cin; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.mode = 1;
  p.addend = addend_half; p.addend_half = 1;
  if (gate) { p.gate = gate; p.bn_x = bn_x; p.bn_mean = bn_mean; p.bn_invstd = bn_invstd; p.stat_partial = stat_partial; }
  p.M = d->B * d->H * d->W;
  return run_gemm(d->dtype, p, (hipStream_t)stream);
}

// The dual form (resnet.py:71-76 backward of a stage's FIRST bottleneck): that block's output ReLU sits behind bn3 AND the downsample BatchNorm, so the gated gradient is
// reduced against both inputs in the same store loop: stat_partial <- (sum g, sum g * xhat(bn_x)), stat_partial2 <- (sum g, sum g * xhat(bn_x2)); the separate dual
// reduce pass (sl_bn_bwd_reduce2: three tensor reads) disappears.  Same shapes as sl_conv2d_bwd_data_addend_bnstat.
extern "C" int sl_conv2d_bwd_data_addend_bnstat2(const SlConvDesc* d, const void* dy, const void* wt, const void* addend, const uint8_t* gate, const void* bn_x,
                                                 const float* bn_mean, const float* bn_invstd, const void* bn_x2, const float* bn_mean2, const float* bn_invstd2, void* dx,
                                                 float* stat_partial, float* stat_partial2, sl_stream_t stream) {
  if (int e = check_desc(d)) return e;
  SL_REQUIRE(dy && wt && dx && addend && gate && bn_x && bn_mean && bn_invstd && bn_x2 && bn_mean2 && bn_invstd2 && stat_partial && stat_partial2, "conv bwd_data_addend_bnstat2: null buffer");
  SL_REQUIRE(sl_conv2d_bwd_data_addend_bnstat_rows(d) > 0, "conv bwd_data_addend_bnstat2: shape not served (sl_conv2d_bwd_data_addend_bnstat_rows == 0)");
  ConvGemmParams p{};
  p.src1 = dy; p.src2 = nullptr; p.C1 = d->Cout; p.C2 = 0; p.wt = wt; p.out = dx;
  p.B = d->B; p.Hs = d->Ho; p.Ws = d->Wo; p.Hd = d->H; p.Wd = d->W;
  p.N = d->Cin; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.mode = 1;
  p.addend = addend; p.gate = gate; p.bn_x = bn_x; p.bn_mean = bn_mean; p.bn_invstd = bn_invstd; p.stat_partial = stat_partial;
  p.bn_x2 = bn_x2; p.bn_mean2 = bn_mean2; p.bn_invstd2 = bn_invstd2; p.stat_partial2 = stat_partial2;
  p.M = d->B * d->H * d->W;
  return run_gemm(d->dtype, p, (hipStream_t)stream);
}

