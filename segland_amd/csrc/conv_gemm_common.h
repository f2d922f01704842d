// Shared pieces of the implicit-GEMM convolution kernels (conv_gemm*.hip): the launch parameter block, the MFMA wrappers, the LDS geometry helpers, the LDS-DMA
// wrappers and the store phases (epilogues).  Device code here is TU-local (anonymous namespace); what crosses translation units is declared in namespace slconv below.
//
//   out[m][n] = sum_{tap, c} src[pix(m, tap)][c] * wt[n][tap][c]
//
// rows m   = destination pixels (b, yd, xd), NHWC
// cols n   = destination channels
// K        = taps x source channels, one K-tile = 128 bytes of channels of one tap (64 bf16 / 32 f32)
// Both element types share the LDS geometry: tiles of [rows][8 x 16-byte chunks], chunk index XOR-swizzled with
// (row>>1)&7 so that the ds_read_b128 fragment reads (row = lane&31) are bank-conflict free.  A bf16 k-step is one
// v_mfma_f32_32x32x16_bf16 per chunk pair; an f32 k-step is four v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).
#pragma once
#include <stdlib.h>
#include "common.h"

#ifndef SL_CH3_SPLIT
#define SL_CH3_SPLIT 2
#endif


struct ConvGemmParams {
  const void* src1; const void* src2;   // source tensors (virtual channel concat), NHWC
  int C1, C2;                           // channels in src1 / src2
  const void* wt;                       // [N][taps][C1+C2]
  void* out;                            // [M][N]
  int B, Hs, Ws;                        // source spatial
  int Hd, Wd;                           // destination spatial
  int N, KH, KW, stride, pad, dil;
  int mode;                             // 0: forward gather, 1: data-gradient gather
  const float* scale;                   // per-output-channel multiplier applied before bias (folded eval-mode BN) or null
  const float* bias; int relu;
  const void* addend; const void* mask_src;
  const void* pre_addend;               // [M][N] added to the accumulators BEFORE statistics / bias (factorised PPM priors) or null
  const unsigned char* addend_mask;     // relu bits (1 byte per 16-byte vector) gating the addend, or null
  float* stat_partial;                  // [gridM][2][N] or null
  // BatchNorm-backward statistics in a data-gradient epilogue (MODE 3 of the fast store phase): the result is the gradient wrt the ACTIVATION a = relu(bn(c)) of the
  // previous layer; it is gated with that ReLU's bits, stored, and its column sums (sum g, sum g * (c - mean) * invstd) go to stat_partial -- what bn_bwd_reduce would
  // compute in a pass of its own over g and c
  const unsigned char* gate;            // relu bits of the OUTPUT positions (1 byte per 16-byte vector) or null
  const void* bn_x;                     // [M][N] the BN input c
  const float* bn_mean; const float* bn_invstd;   // [N]
  // pixel-stationary kernel MODE 5, DUAL: a second BatchNorm behind the same ReLU (bn3 + the downsample BN of a stage's first bottleneck, resnet.py:71-76): the gated result is
  // also reduced against bn_x2; stat_partial2 [gridM][2][N] receives (sum g, sum g * xhat2)
  const void* bn_x2; const float* bn_mean2; const float* bn_invstd2; float* stat_partial2;
  int addend_half;                      // pixel-stationary kernel: the addend is [B][Hd/2][Wd/2][N] and enters at the EVEN positions only (the data gradient of a 1x1 stride-2 conv, never scattered)
  const float* row_scale;               // [B] per-sample multiplier of (acc * scale + bias), applied before the addend (DropPath), or null
  void* out2;                           // [M][N] or null: GELU of the stored (rounded) acc * scale + bias, written next to `out` (Mlp fc1)
  int M;
  int gridM, gridN;
  int tile16;                           // patch kernel: row block bm is a 16 x 16-pixel tile (b, y0 / 16, x0 / 16), its 256 rows are 16 segments of 16 pixels
  int flags;                            // p8: bit 0 = counted first wait (always set); bit 1 = generic store phase instead of conv_epilogue_affine (test hook sl_debug_conv_affine: the bit-identity test)
                                        // bit 2: mask_src holds the pre-activation of a GELU and the result is multiplied by GELU'(mask_src) instead of gated by mask_src > 0
  // split-K (inference convs on few row tiles: the fine-tune pair's 8 192-row layers): the K range is cut into `ksplit` parts, block (tile, part) writes its raw fp32
  // accumulators to ws [part][M][N] and conv_splitk_finish_kernel sums the parts in a fixed order and applies pre_addend / scale / bias / addend / ReLU
  float* ws; int ksplit;
  // parity planes of a stride-2 data gradient (conv_gemm.hip: launch_parity_planes; SUBP instantiations of the ring kernel): the rows of this launch are the positions
  // (b, i, j) of the half-resolution grid [Hd/2][Wd/2]; row (b, i, j) is destination pixel (2i + sub_py, 2j + sub_px), and only the taps whose source index is an
  // integer for that parity are multiplied (4 / 2 / 2 / 1 of a 3x3 window's nine)
  int sub, sub_py, sub_px;
  unsigned long long* trace;            // debug (tools/p8_trace.py): per block {s_memtime at entry, after the prologue, after the main loop, at the end, HW_ID}; null in production
};

// ---- what crosses the translation units of the conv library (conv_gemm.hip = dispatch + C entry points; conv_gemm_tiles.hip = two-stage and ring tile kernels;
// conv_gemm_patch.hip = half-tile (p8) and 3x3 patch (p9) kernels + split-K; conv_gemm_sk.hip = 64 -> 64 patch and pixel-stationary kernels)
namespace slconv __attribute__((visibility("hidden"))) {
constexpr int C64_T = 16;                                              // conv_c64k3_kernel: tile edge (one BN statistic partial row per tile)
int launch_tile(int cfg, int dtype, ConvGemmParams& p, hipStream_t st);     // families 4 (ring) and 2 (two-stage) by kernel code
int launch_p8(ConvGemmParams& p, hipStream_t st);
int launch_p9(ConvGemmParams& p, hipStream_t st);
bool p9_on();
bool p9_shape(const ConvGemmParams& p);
int launch_c64k3(ConvGemmParams& p, hipStream_t st);
bool c64k3_shape(int dtype, int KH, int KW, int stride, int pad, int dil, int Cin, int C1, int Cout, long long M);
int launch_sk(ConvGemmParams& p, hipStream_t st);
bool sk_shape(int dtype, int KH, int KW, int stride, int pad, int Cin, int C1, int N, long long M);
}  // namespace slconv

namespace {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
    // lanes 0-31 hold k = 4j..4j+3 of the chunk pair, lanes 32-63 the next four: any pairing of k is valid as long
    // as A and B use the same one.
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// ---------------------------------------------------------------------------------------------------------------
// Epilogue of the glds kernels.  The MFMAs are issued with the operand roles swapped (A = weight rows, B = pixel rows),
// so a lane holds, for ONE pixel (col = lane&31), four consecutive output channels per register quad
// (n = 8*(r>>2) + 4*(lane>>5) + (r&3)).  The tile is staged through LDS ([pixel][channel], 16-byte padded rows) with
// 8/16-byte writes and then streamed out with 16-byte coalesced global stores; bias, ReLU, the residual addend and the
// ReLU mask are applied on those vectors (16-byte coalesced loads), and the per-channel BN statistics (sum, sum of
// squares of the stored values) are accumulated in the same pass.
// streamed-out tile store: non-temporal 16-byte stores (the tile is next touched by another kernel, after > L2 of traffic)
template <typename T>
__device__ __forceinline__ void st16(T* dst, const uint4& v) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, v), (u32x4_t*)dst);
}

// Barrier for LDS hand-offs inside the epilogues.  __syncthreads() carries a workgroup release fence: with global stores in flight it becomes
// s_waitcnt vmcnt(0) + s_barrier, i.e. every pass (and the end of the tile) would wait for its stores to be acknowledged by the L2.  The hand-offs
// here are LDS only (staged tile <-> store loop <-> statistic partials), so only the LDS counter is drained.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T, int BM, int BN, int WM, int WN, bool SPLIT = false>
struct EpiGeom {
  static constexpr int PITCH = BN * (int)sizeof(T) + 16;
  static constexpr int TILE_BYTES = BM * PITCH;
  // SPLIT (conv_gemm_p8_kernel): always two passes of BM / 2 rows = the two operand halves, 66 KiB of staging: the persistent kernel keeps the next tile's first K-tile
  // (four 16 KiB slots) in flight across the epilogue
  static constexpr int NPASS = (SPLIT || TILE_BYTES > 150 * 1024) ? 2 : 1;
  static constexpr int MAIN_BYTES = 2 * (BM + BN) * 128;
  static constexpr int LDS_BYTES = (TILE_BYTES / NPASS) > MAIN_BYTES ? (TILE_BYTES / NPASS) : MAIN_BYTES;
  static_assert(SPLIT || WM % NPASS == 0, "passes split whole wave rows");
};

// accumulators of the wave rows belonging to `pass` -> LDS staging tile (rounded to T), in the row-major layout the store phase reads
// SPLIT: the half-tile kernel's wave layout (wave rows/columns interleaved over the two operand halves, see conv_gemm_p8_kernel).
template <typename T, int BM, int BN, int WM, int WN, bool SPLIT = false>
__device__ __forceinline__ void epi_stage_acc(f32x16_t (&acc)[BM / WM / 32][BN / WN / 32], int pass, int wm, int wn, int lane, unsigned char* smem) {
  using G = EpiGeom<T, BM, BN, WM, WN, SPLIT>;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(!SPLIT || (G::NPASS == 2 && TM == 4 && TN == 2), "split layout: two passes (operand halves), 4x2 accumulator blocks per wave");
  const int fhalf = lane >> 5;
  if constexpr (!SPLIT) { if (wm / (WM / G::NPASS) != pass) return; }
  const int lrow = SPLIT ? 0 : (wm % (WM / G::NPASS)) * (BM / WM) + (lane & 31);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (SPLIT && (i >> 1) != pass) continue;                       // pass h = accumulator blocks of operand half h: every wave stages in both passes
        const int row = SPLIT ? wm * (BM / 4) + (i & 1) * 32 + (lane & 31) : lrow + i * 32;
        const int col = SPLIT ? j * (BN / 2) + wn * (BN / 8) + 8 * q + 4 * fhalf : wn * (BN / WN) + j * 32 + 8 * q + 4 * fhalf;
        unsigned char* dst = smem + row * G::PITCH + col * (int)sizeof(T);
        if constexpr (sizeof(T) == 2) {
          typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
          typedef __attribute__((ext_vector_type(2))) float f32x2_t;
          uint2 v;
          v.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[i][j][4 * q + 0], acc[i][j][4 * q + 1]}, bf16x2_t));
          v.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]}, bf16x2_t));
          *(uint2*)dst = v;
        } else {
          *(float4*)dst = make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        }
      }
}

// Branch-free store phase for the two shapes of epilogue that carry almost all of the traffic, on tiles that lie completely
// inside M:  MODE 1 = store the staged tile as is (+ BN statistic partials when requested),  MODE 2 = add the (optionally
// bit-gated) addend and store.  No per-row predicates, one pointer increment per row, operand loads batched CH rows deep.
template <typename T, int BM, int BN, int WM, int WN, int MODE, bool SPLIT = false, bool SUBP = false>
__device__ __forceinline__ void conv_epilogue_fast(const ConvGemmParams& p, f32x16_t (&acc)[BM / WM / 32][BN / WN / 32], int bm, int bn, int wm, int wn,
                                                   int lane, int tid, unsigned char* smem) {
  using G = EpiGeom<T, BM, BN, WM, WN, SPLIT>;
  constexpr int EPC = 16 / sizeof(T);
  constexpr int NT = 64 * WM * WN;
  constexpr int CPR = BN / EPC;
  constexpr int RS = NT / CPR;
  constexpr int ROWS = BM / G::NPASS;
  constexpr int NIT = ROWS / RS, CH = SPLIT ? 4 : (NIT < 8 ? NIT : 8);     // SPLIT: half of the accumulators and the next tile's row map are live across the first pass
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  const int cc = tid % CPR, r0 = tid / CPR;
  const int ncol = bn * BN + cc * EPC;
  const size_t rstep = (p.tile16 ? (size_t)p.Wd : (size_t)RS) * p.N * sizeof(T);          // tile16: a sweep of RS = 16 rows is one 16-pixel segment, the next sweep is the next image row
  // byte offset of sweep `it` from the thread's first row.  tile16 with RS == 8 (the four-wave patch kernel: 256 threads): two sweeps per 16-pixel segment
  // SUBP (parity plane of a stride-2 data gradient): tile rows are positions of the half-resolution grid; a sweep of RS rows lies inside one half-resolution image row
  // (Wd / 2 is a power of two and a multiple of RS, a tile never crosses an image: launch_parity_planes checks), its pixels are 2 N elements apart in the destination
  const int sub_lw = SUBP ? 31 - __builtin_clz((unsigned)(p.Wd >> 1)) : 0;
  auto soff = [&](int it) -> size_t {
    if constexpr (SUBP) {
      const int q = it * RS;                                    // rows behind the pass's first sweep: q >> lw image rows further, q & (Wh - 1) pixels to the right
      return ((size_t)(q >> sub_lw) * (2 * p.Wd) + (size_t)(q & ((p.Wd >> 1) - 1)) * 2) * p.N * sizeof(T);
    }
    if constexpr (RS == 8) { if (p.tile16) return (size_t)(it >> 1) * rstep + (size_t)(it & 1) * (8 * p.N * sizeof(T)); }
    return (size_t)it * rstep;
  };
  f2_t ssum[EPC / 2], ssq[EPC / 2];
#pragma unroll
  for (int e = 0; e < EPC / 2; ++e) { ssum[e] = (f2_t){0.f, 0.f}; ssq[e] = (f2_t){0.f, 0.f}; }
  auto unpack2 = [](const uint4& raw, f2_t* v) {
    if constexpr (sizeof(T) == 2) {
      const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (f2_t){__uint_as_float(w[k] << 16), __uint_as_float(w[k] & 0xffff0000u)};
    } else {
      v[0] = (f2_t){__uint_as_float(raw.x), __uint_as_float(raw.y)}; v[1] = (f2_t){__uint_as_float(raw.z), __uint_as_float(raw.w)};
    }
  };
#pragma unroll
  for (int pass = 0; pass < G::NPASS; ++pass) {
    epi_stage_acc<T, BM, BN, WM, WN, SPLIT>(acc, pass, wm, wn, lane, smem);
    lds_barrier();
    size_t grow = (size_t)(bm * BM + pass * ROWS + r0);
    if constexpr (SUBP) {
      const int Wh = p.Wd >> 1, hw = (p.Hd >> 1) * Wh;
      const int u = bm * BM + pass * ROWS;                      // first half-resolution position of this pass (a multiple of RS)
      const int b = u / hw, rem = u - b * hw, i = rem >> sub_lw, j = rem & (Wh - 1);
      grow = ((size_t)b * p.Hd + 2 * i + p.sub_py) * p.Wd + 2 * (j + r0) + p.sub_px;
    } else if (p.tile16) {
      const int tx = p.Wd >> 4, ty = p.Hd >> 4;
      const int bx = bm % tx, by = (bm / tx) % ty, b = bm / (tx * ty);
      grow = ((size_t)b * p.Hd + by * 16 + pass * (ROWS / 16)) * p.Wd + bx * 16 + r0;
    }
    const size_t goff = (grow * p.N + ncol) * sizeof(T);
    unsigned char* o = (unsigned char*)p.out + goff;
    const unsigned char* l = smem + r0 * G::PITCH + cc * 16;
    if constexpr (MODE == 3) {
      // gate with the ReLU bits of these positions, store, accumulate (sum g, sum g * xhat) per column with xhat = (c - mean) * invstd exactly as bn_bwd_reduce forms it
      const unsigned char* xs = (const unsigned char*)p.bn_x + goff;
      const unsigned char* gb = p.gate + goff / 16;
      f2_t mu[EPC / 2], is[EPC / 2];
#pragma unroll
      for (int e = 0; e < EPC / 2; ++e) {
        mu[e] = (f2_t){p.bn_mean[ncol + 2 * e], p.bn_mean[ncol + 2 * e + 1]};
        is[e] = (f2_t){p.bn_invstd[ncol + 2 * e], p.bn_invstd[ncol + 2 * e + 1]};
      }
      constexpr int CH3 = SPLIT ? SL_CH3_SPLIT : CH;          // the persistent half-tile kernel is at its register limit: few rows in flight there
#pragma unroll 1
      for (int it0 = 0; it0 < NIT; it0 += CH3) {
        uint4 xv[CH3]; unsigned bits[CH3];
#pragma unroll
        for (int u = 0; u < CH3; ++u) { xv[u] = *(const uint4*)(xs + soff(it0 + u)); bits[u] = gb[soff(it0 + u) / 16]; }
#pragma unroll
        for (int u = 0; u < CH3; ++u) {
          uint4 raw = *(const uint4*)(l + (it0 + u) * (RS * G::PITCH));
          const unsigned b = bits[u];
          if constexpr (sizeof(T) == 2) {
            raw.x &= ((unsigned)__builtin_amdgcn_sbfe(b, 0, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 1, 1) & 0xffff0000u);
            raw.y &= ((unsigned)__builtin_amdgcn_sbfe(b, 2, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 3, 1) & 0xffff0000u);
            raw.z &= ((unsigned)__builtin_amdgcn_sbfe(b, 4, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 5, 1) & 0xffff0000u);
            raw.w &= ((unsigned)__builtin_amdgcn_sbfe(b, 6, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 7, 1) & 0xffff0000u);
          } else {
            raw.x &= (unsigned)__builtin_amdgcn_sbfe(b, 0, 1); raw.y &= (unsigned)__builtin_amdgcn_sbfe(b, 1, 1);
            raw.z &= (unsigned)__builtin_amdgcn_sbfe(b, 2, 1); raw.w &= (unsigned)__builtin_amdgcn_sbfe(b, 3, 1);
          }
          st16(o + soff(it0 + u), raw);
          f2_t v[EPC / 2], w[EPC / 2];
          unpack2(raw, v); unpack2(xv[u], w);
#pragma unroll
          for (int e = 0; e < EPC / 2; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * ((w[e] - mu[e]) * is[e]); }
        }
      }
    } else if constexpr (MODE == 1) {
      if (p.stat_partial) {
#pragma unroll 4
        for (int it = 0; it < NIT; ++it) {
          const uint4 raw = *(const uint4*)(l + it * (RS * G::PITCH));
          st16(o + soff(it), raw);
          f2_t v[EPC / 2];
          unpack2(raw, v);
#pragma unroll
          for (int e = 0; e < EPC / 2; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
        }
      } else {
#pragma unroll 8
        for (int it = 0; it < NIT; ++it) st16(o + soff(it), *(const uint4*)(l + it * (RS * G::PITCH)));
      }
    } else {
      // MODE 6: the second operand is the pre-activation of a GELU (mask_src) and multiplies the result by GELU'(.) instead of being added
      const unsigned char* ad = (const unsigned char*)(MODE == 6 ? p.mask_src : p.addend) + goff;
      const unsigned char* ab = (MODE != 6 && p.addend_mask) ? p.addend_mask + goff / 16 : nullptr;
#pragma unroll 1
      for (int it0 = 0; it0 < NIT; it0 += CH) {
        uint4 addv[CH]; unsigned bits[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) addv[u] = *(const uint4*)(ad + soff(it0 + u));
        if (ab) {
#pragma unroll
          for (int u = 0; u < CH; ++u) bits[u] = ab[soff(it0 + u) / 16];
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const uint4 raw = *(const uint4*)(l + (it0 + u) * (RS * G::PITCH));
          uint4 a = addv[u];
          if (ab) {
            const unsigned b = bits[u];
            if constexpr (sizeof(T) == 2) {
              // bit e gates element e: build a 16-bit-lane mask per packed pair from sign-extended single bits
              a.x &= ((unsigned)__builtin_amdgcn_sbfe(b, 0, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 1, 1) & 0xffff0000u);
              a.y &= ((unsigned)__builtin_amdgcn_sbfe(b, 2, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 3, 1) & 0xffff0000u);
              a.z &= ((unsigned)__builtin_amdgcn_sbfe(b, 4, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 5, 1) & 0xffff0000u);
              a.w &= ((unsigned)__builtin_amdgcn_sbfe(b, 6, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 7, 1) & 0xffff0000u);
            } else {
              a.x &= (unsigned)__builtin_amdgcn_sbfe(b, 0, 1); a.y &= (unsigned)__builtin_amdgcn_sbfe(b, 1, 1);
              a.z &= (unsigned)__builtin_amdgcn_sbfe(b, 2, 1); a.w &= (unsigned)__builtin_amdgcn_sbfe(b, 3, 1);
            }
          }
          f2_t v[EPC / 2], w[EPC / 2];
          unpack2(raw, v); unpack2(a, w);
          if constexpr (MODE == 6) {
#pragma unroll
            for (int e = 0; e < EPC / 2; ++e) { v[e].x *= sl_gelu_grad<T>(w[e].x); v[e].y *= sl_gelu_grad<T>(w[e].y); }
          } else {
#pragma unroll
            for (int e = 0; e < EPC / 2; ++e) v[e] += w[e];
          }
          uint4 r;
          if constexpr (sizeof(T) == 2) {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
            r.x = __builtin_bit_cast(unsigned, __builtin_convertvector(v[0], bf16x2_t)); r.y = __builtin_bit_cast(unsigned, __builtin_convertvector(v[1], bf16x2_t));
            r.z = __builtin_bit_cast(unsigned, __builtin_convertvector(v[2], bf16x2_t)); r.w = __builtin_bit_cast(unsigned, __builtin_convertvector(v[3], bf16x2_t));
          } else {
            r = make_uint4(__float_as_uint(v[0].x), __float_as_uint(v[0].y), __float_as_uint(v[1].x), __float_as_uint(v[1].y));
          }
          st16(o + soff(it0 + u), r);
        }
      }
    }
    lds_barrier();
  }
  if constexpr (MODE == 1 || MODE == 3) {
    if (p.stat_partial) {
      float fs[EPC], fq[EPC];
#pragma unroll
      for (int e = 0; e < EPC / 2; ++e) { fs[2 * e] = ssum[e].x; fs[2 * e + 1] = ssum[e].y; fq[2 * e] = ssq[e].x; fq[2 * e + 1] = ssq[e].y; }
#pragma unroll
      for (int off = CPR; off < 64; off <<= 1) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) { fs[e] += __shfl_xor(fs[e], off, 64); fq[e] += __shfl_xor(fq[e], off, 64); }
      }
      float* red = (float*)smem;                       // [NW][2][BN]
      const int wave = tid >> 6;
      if (lane < CPR) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          red[(wave * 2 + 0) * BN + cc * EPC + e] = fs[e];
          red[(wave * 2 + 1) * BN + cc * EPC + e] = fq[e];
        }
      }
      lds_barrier();
      for (int e = tid; e < 2 * BN; e += NT) {
        const int which = e / BN, col = e % BN;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WM * WN; ++w) t += red[(w * 2 + which) * BN + col];
        p.stat_partial[((size_t)bm * 2 + which) * p.N + bn * BN + col] = t;
      }
    }
  }
}

// Branch-free store phase of the affine epilogues on tiles inside M (nn.Linear over token maps, eval-mode BatchNorm folded into the conv):
//   v = acc * scale[n] + bias[n];  GELU: out2 = gelu(round(v));  RS: v *= row_scale[image of the row];  ADD: v += addend;  relu (runtime floor);  out = round(v)
// -- the generic store phase evaluates every optional operand per row behind run-time tests and always carries the statistic sums: 59 us against 25 us for the plain store
// on the 131 072 x 128 -> 384 qkv GEMM of Swin-T stage 1 (tools/gemm_time.py).  Same operation order as conv_epilogue_generic (bit-identical results).
// SPLIT: the half-tile kernel's staging (two passes = the two operand halves, every wave stages in both; conv_gemm_p8_kernel<2>).
template <typename T, int BM, int BN, int WM, int WN, bool ADD, bool RS_, bool GELU, bool SCRELU, bool SPLIT = false>
__device__ __forceinline__ void conv_epilogue_affine(const ConvGemmParams& p, f32x16_t (&acc)[BM / WM / 32][BN / WN / 32], int bm, int bn, int wm, int wn,
                                                     int lane, int tid, unsigned char* smem) {
  using G = EpiGeom<T, BM, BN, WM, WN, SPLIT>;
  constexpr int EPC = 16 / sizeof(T), EP2 = EPC / 2;
  constexpr int NT = 64 * WM * WN;
  constexpr int CPR = BN / EPC;
  constexpr int RS = NT / CPR;
  constexpr int ROWS = BM / G::NPASS;
  constexpr int NIT = ROWS / RS, CH = SPLIT ? 4 : (NIT < 8 ? NIT : 8);
  typedef __attribute__((ext_vector_type(2))) float f2_t;       // pairs: v_pk_mul_f32 / v_pk_add_f32, one v_cvt_pk_bf16_f32 per pair
  const int cc = tid % CPR, r0 = tid / CPR;
  const int ncol = bn * BN + cc * EPC;
  f2_t bias[EP2], scl[EP2];
#pragma unroll
  for (int e = 0; e < EP2; ++e) {
    bias[e] = p.bias ? (f2_t){p.bias[ncol + 2 * e], p.bias[ncol + 2 * e + 1]} : (f2_t){0.f, 0.f};
    scl[e] = (SCRELU && p.scale) ? (f2_t){p.scale[ncol + 2 * e], p.scale[ncol + 2 * e + 1]} : (f2_t){1.f, 1.f};
  }
  const float lo = p.relu ? 0.f : -INFINITY;
  const size_t rstep = (size_t)RS * p.N * sizeof(T);
  const int hw = p.Hd * p.Wd;
  auto unpack2 = [](const uint4& raw, f2_t* v) {
    if constexpr (sizeof(T) == 2) {
      const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (f2_t){__uint_as_float(w[k] << 16), __uint_as_float(w[k] & 0xffff0000u)};
    } else {
      v[0] = (f2_t){__uint_as_float(raw.x), __uint_as_float(raw.y)}; v[1] = (f2_t){__uint_as_float(raw.z), __uint_as_float(raw.w)};
    }
  };
  auto pack2 = [](const f2_t* v) -> uint4 {
    if constexpr (sizeof(T) == 2) {
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
      return make_uint4(__builtin_bit_cast(unsigned, __builtin_convertvector(v[0], bf16x2_t)), __builtin_bit_cast(unsigned, __builtin_convertvector(v[1], bf16x2_t)),
                        __builtin_bit_cast(unsigned, __builtin_convertvector(v[2], bf16x2_t)), __builtin_bit_cast(unsigned, __builtin_convertvector(v[3], bf16x2_t)));
    } else {
      return make_uint4(__float_as_uint(v[0].x), __float_as_uint(v[0].y), __float_as_uint(v[1].x), __float_as_uint(v[1].y));
    }
  };
#pragma unroll
  for (int pass = 0; pass < G::NPASS; ++pass) {
    epi_stage_acc<T, BM, BN, WM, WN, SPLIT>(acc, pass, wm, wn, lane, smem);
    lds_barrier();
    const int m0 = bm * BM + pass * ROWS + r0;
    const size_t goff = ((size_t)m0 * p.N + ncol) * sizeof(T);
    unsigned char* o = (unsigned char*)p.out + goff;
    unsigned char* o2 = GELU ? (unsigned char*)p.out2 + goff : nullptr;
    const unsigned char* ad = ADD ? (const unsigned char*)p.addend + goff : nullptr;
    const unsigned char* l = smem + r0 * G::PITCH + cc * 16;
    int img = 0, rem = 0;
    if constexpr (RS_) { img = m0 / hw; rem = m0 - img * hw; }
#pragma unroll 1
    for (int it0 = 0; it0 < NIT; it0 += CH) {
      uint4 addv[CH]; float rsv[CH];
      if constexpr (ADD) {
#pragma unroll
        for (int u = 0; u < CH; ++u) addv[u] = *(const uint4*)(ad + (size_t)(it0 + u) * rstep);
      }
      if constexpr (RS_) {
#pragma unroll
        for (int u = 0; u < CH; ++u) { rsv[u] = p.row_scale[img]; rem += RS; while (rem >= hw) { rem -= hw; ++img; } }
      }
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        f2_t v[EP2];
        unpack2(*(const uint4*)(l + (it0 + u) * (RS * G::PITCH)), v);
#pragma unroll
        for (int e = 0; e < EP2; ++e) { if constexpr (SCRELU) v[e] = v[e] * scl[e]; v[e] = v[e] + bias[e]; }
        if constexpr (GELU) {
          f2_t g[EP2];
          unpack2(pack2(v), g);
#pragma unroll
          for (int e = 0; e < EP2; ++e) g[e] = (f2_t){sl_gelu<T>(g[e].x), sl_gelu<T>(g[e].y)};
          st16((T*)(o2 + (size_t)(it0 + u) * rstep), pack2(g));
        }
        if constexpr (RS_) {
#pragma unroll
          for (int e = 0; e < EP2; ++e) v[e] = v[e] * rsv[u];
        }
        if constexpr (ADD) {
          f2_t a[EP2];
          unpack2(addv[u], a);
#pragma unroll
          for (int e = 0; e < EP2; ++e) v[e] = v[e] + a[e];
        }
        if constexpr (SCRELU) {
#pragma unroll
          for (int e = 0; e < EP2; ++e) v[e] = (f2_t){v[e].x > lo ? v[e].x : lo, v[e].y > lo ? v[e].y : lo};
        }
        st16((T*)(o + (size_t)(it0 + u) * rstep), pack2(v));
      }
    }
    lds_barrier();
  }
}

template <typename T, int BM, int BN, int WM, int WN, bool SPLIT = false>
__device__ __forceinline__ void conv_epilogue_generic(const ConvGemmParams& p, f32x16_t (&acc)[BM / WM / 32][BN / WN / 32], int bm, int bn, int wm, int wn,
                                                      int lane, int tid, unsigned char* smem);

// GATE: this instantiation carries the gated-statistics store phase (MODE 3).  The two persistent 512-thread kernels are compiled once with and once without it: they sit
// at the register limit (the half-tile kernel: 12 -> 19 spilled VGPRs with MODE 3 inlined), and every launch paid for it, gated or not: 25.98 vs 25.78 ms per ResNet-50
// step, A/B/A/B on one box (profiles/r3_ab_gate_split.txt).
// AFF: this instantiation carries the branch-free affine store phases (bias / folded BatchNorm / residual / GELU side output); the tile kernels always, the half-tile
// kernel in an instantiation of its own (conv_gemm_p8_kernel<2>: inference convs and biased Linears with N % 256 == 0 on >= 24 576 rows).
template <typename T, int BM, int BN, int WM, int WN, bool SPLIT = false, bool GATE = true, bool AFF = !SPLIT, bool SUBP = false, bool GATEB = false>
__device__ __forceinline__ int conv_epilogue_lds(const ConvGemmParams& p, f32x16_t (&acc)[BM / WM / 32][BN / WN / 32], int bm, int bn, int wm, int wn,
                                                  int lane, int tid, unsigned char* smem) {
  const bool full = (bm + 1) * BM <= p.M;
  const bool shaped = p.bias || p.scale || p.relu || p.mask_src || p.pre_addend || p.row_scale || p.out2;
  // Returns the number of vector-memory instructions the wave issued (loads + stores; every path below issues the same count in every wave), or -1 when that is
  // not a compile-time fact of the path: the persistent half-tile kernel uses it to wait for loads that are OLDER than these instructions without waiting for them.
  if constexpr (!SPLIT) {
    // the data gradient behind a GELU (sl_conv2d_bwd_data_gelu): the fast store phase with a multiplicative second operand (not in the persistent kernels: register limit)
    if (full && (p.flags & 4) && p.mask_src && !(p.bias || p.scale || p.relu || p.pre_addend || p.row_scale || p.out2 || p.addend || p.stat_partial)) {
      conv_epilogue_fast<T, BM, BN, WM, WN, 6, SPLIT>(p, acc, bm, bn, wm, wn, lane, tid, smem);
      return -1;
    }
  }
  if constexpr (SUBP) {
    // a parity plane of a stride-2 data gradient: launch_parity_planes (conv_gemm.hip) only launches whole tiles with one of the three fast store phases
    if constexpr (GATE) conv_epilogue_fast<T, BM, BN, WM, WN, 3, SPLIT, true>(p, acc, bm, bn, wm, wn, lane, tid, smem);
    else if (p.addend) conv_epilogue_fast<T, BM, BN, WM, WN, 2, SPLIT, true>(p, acc, bm, bn, wm, wn, lane, tid, smem);
    else conv_epilogue_fast<T, BM, BN, WM, WN, 1, SPLIT, true>(p, acc, bm, bn, wm, wn, lane, tid, smem);
    return -1;
  }
  if constexpr (GATEB) {
    // conv_gemm_p8_kernel<3>: launched only for a gated data gradient with a bias (sl_conv2d_bwd_data_bnstat_folded) on whole tiles; the kernel has added the bias to the
    // accumulators (one rounding of the centred value), the store phase is the plain gated-statistics one
    conv_epilogue_fast<T, BM, BN, WM, WN, 3, SPLIT>(p, acc, bm, bn, wm, wn, lane, tid, smem);
    return SPLIT ? 49 : -1;
  }
  if constexpr (GATE) {
    if (full && !shaped && !p.addend && p.gate) {
      conv_epilogue_fast<T, BM, BN, WM, WN, 3, SPLIT>(p, acc, bm, bn, wm, wn, lane, tid, smem);
      return SPLIT ? 49 : -1;                                           // 2 passes x 8 sweeps x (BN input + gate byte + store), + the statistic partial
    }
  }
  if constexpr (SPLIT && AFF) {
    // conv_gemm_p8_kernel<2> is launched for biased / folded-BatchNorm launches only: the plain store phases are not compiled into it (register limit)
  } else if (full && !shaped && !p.addend) {
    if constexpr (SPLIT) asm volatile("; EPI_BEGIN mode1");
    conv_epilogue_fast<T, BM, BN, WM, WN, 1, SPLIT>(p, acc, bm, bn, wm, wn, lane, tid, smem);
    if constexpr (SPLIT) asm volatile("; EPI_END mode1");
    return SPLIT ? 16 + (p.stat_partial ? 1 : 0) : -1;                  // 2 passes x 8 row sweeps, + the statistic partial
  } else if (full && !shaped && !p.stat_partial) {
    if constexpr (SPLIT) asm volatile("; EPI_BEGIN mode2");
    conv_epilogue_fast<T, BM, BN, WM, WN, 2, SPLIT>(p, acc, bm, bn, wm, wn, lane, tid, smem);
    if constexpr (SPLIT) asm volatile("; EPI_END mode2");
    return SPLIT ? (p.addend_mask ? 48 : 32) : -1;                      // per sweep: addend load (+ gate byte) + store
  }
  if constexpr (AFF) {
    if (full && (p.bias || p.scale) && !p.mask_src && !p.pre_addend && !p.addend_mask && !p.stat_partial && !p.tile16 && !(p.flags & 2)) {
      const bool sr = p.scale || p.relu;              // eval-mode BatchNorm folded into the conv (+ residual + ReLU); the Swin linears have neither
      if (p.out2) {
        if (!p.addend && !p.row_scale && !sr) { conv_epilogue_affine<T, BM, BN, WM, WN, false, false, true, false, SPLIT>(p, acc, bm, bn, wm, wn, lane, tid, smem); return -1; }
      } else if (p.addend) {
        if (p.row_scale) { if constexpr (!SPLIT) { if (!sr) { conv_epilogue_affine<T, BM, BN, WM, WN, true, true, false, false, SPLIT>(p, acc, bm, bn, wm, wn, lane, tid, smem); return -1; } } }
        else if (sr) { conv_epilogue_affine<T, BM, BN, WM, WN, true, false, false, true, SPLIT>(p, acc, bm, bn, wm, wn, lane, tid, smem); return -1; }
        else { conv_epilogue_affine<T, BM, BN, WM, WN, true, false, false, false, SPLIT>(p, acc, bm, bn, wm, wn, lane, tid, smem); return -1; }
      } else if (!p.row_scale) {
        if (sr) conv_epilogue_affine<T, BM, BN, WM, WN, false, false, false, true, SPLIT>(p, acc, bm, bn, wm, wn, lane, tid, smem);
        else conv_epilogue_affine<T, BM, BN, WM, WN, false, false, false, false, SPLIT>(p, acc, bm, bn, wm, wn, lane, tid, smem);
        return -1;
      }
    }
  }
  conv_epilogue_generic<T, BM, BN, WM, WN, SPLIT>(p, acc, bm, bn, wm, wn, lane, tid, smem);
  return -1;
}

template <typename T, int BM, int BN, int WM, int WN, bool SPLIT>
__device__ __forceinline__ void conv_epilogue_generic(const ConvGemmParams& p, f32x16_t (&acc)[BM / WM / 32][BN / WN / 32], int bm, int bn, int wm, int wn,
                                                      int lane, int tid, unsigned char* smem) {
  using G = EpiGeom<T, BM, BN, WM, WN, SPLIT>;
  constexpr int EPC = 16 / sizeof(T);
  constexpr int NT = 64 * WM * WN;
  constexpr int CPR = BN / EPC;              // 16-byte chunks per tile row
  constexpr int RS = NT / CPR;               // rows covered per sweep of the block
  constexpr int ROWS = BM / G::NPASS;
  static_assert(NT % CPR == 0 && CPR <= 64, "store-phase mapping");
  const int cc = tid % CPR, r0 = tid / CPR;
  const int ncol = bn * BN + cc * EPC;
  float bias[EPC], scl[EPC], ssum[EPC], ssq[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { bias[e] = p.bias ? p.bias[ncol + e] : 0.f; scl[e] = p.scale ? p.scale[ncol + e] : 1.f; ssum[e] = 0.f; ssq[e] = 0.f; }
  T* out = (T*)p.out;
#pragma unroll
  for (int pass = 0; pass < G::NPASS; ++pass) {
    epi_stage_acc<T, BM, BN, WM, WN, SPLIT>(acc, pass, wm, wn, lane, smem);
    lds_barrier();
    // the residual / mask operands are fetched CH rows at a time BEFORE they are consumed: 16-byte loads of CH rows are in
    // flight together instead of one load -> use -> store latency chain per row
    const bool plain = !(p.bias || p.scale || p.addend || p.relu || p.mask_src || p.pre_addend || p.row_scale || p.out2);
    constexpr int NIT = ROWS / RS, CH = SPLIT ? 2 : (NIT < 8 ? NIT : 8);
    static_assert(ROWS % RS == 0 && NIT % CH == 0, "store-phase chunking");
    int mrow0 = bm * BM + pass * ROWS + r0, mstep = RS;                 // global row of the thread's first sweep, rows between sweeps
    if (p.tile16) {                                                     // 16 x 16-pixel tile: a sweep of RS = 16 rows is one 16-pixel segment, the next sweep is the next image row
      const int tx = p.Wd >> 4, ty = p.Hd >> 4;
      const int bx = bm % tx, by = (bm / tx) % ty, b = bm / (tx * ty);
      mrow0 = (b * p.Hd + by * 16 + pass * (ROWS / 16)) * p.Wd + bx * 16 + r0; mstep = p.Wd;
    }
    auto mrow = [&](int it) { if constexpr (RS == 8) { if (p.tile16) return mrow0 + (it >> 1) * mstep + (it & 1) * 8; } return mrow0 + it * mstep; };
#pragma unroll 1
    for (int it0 = 0; it0 < NIT; it0 += CH) {
      uint4 addv[CH], mskv[CH], prev[CH];
      if (p.pre_addend) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int m = mrow(it0 + u);
          if (m < p.M) prev[u] = *(const uint4*)((const T*)p.pre_addend + (size_t)m * p.N + ncol);
        }
      }
      unsigned abit[CH];
      if (p.addend) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int m = mrow(it0 + u);
          if (m < p.M) addv[u] = *(const uint4*)((const T*)p.addend + (size_t)m * p.N + ncol);
        }
        if (p.addend_mask) {                    // the gate bytes ride in the same batch (one exposed latency, not one per row)
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            const int m = mrow(it0 + u);
            if (m < p.M) abit[u] = p.addend_mask[((size_t)m * p.N + ncol) / EPC];
          }
        }
      }
      if (p.mask_src) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int m = mrow(it0 + u);
          if (m < p.M) mskv[u] = *(const uint4*)((const T*)p.mask_src + (size_t)m * p.N + ncol);
        }
      }
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int row = r0 + (it0 + u) * RS;
        const int m = mrow(it0 + u);
        if (m < p.M) {
          float v[EPC];
          const uint4 raw = *(const uint4*)(smem + row * G::PITCH + cc * 16);
          unpack16<T>(raw, v);
          if (p.pre_addend) {
            float a[EPC];
            unpack16<T>(prev[u], a);
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] += a[e];
          }
#pragma unroll
          for (int e = 0; e < EPC; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
          if (plain) { st16(out + (size_t)m * p.N + ncol, raw); continue; }      // nothing to apply: stream the staged chunk
          if (p.bias || p.scale) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] = v[e] * scl[e] + bias[e];
          }
          if (p.out2) {                                   // the activation sees what `out` stores (the rounded pre-activation)
            float g[EPC];
            unpack16<T>(pack16<T>(v), g);
#pragma unroll
            for (int e = 0; e < EPC; ++e) g[e] = sl_gelu<T>(g[e]);
            st16((T*)p.out2 + (size_t)m * p.N + ncol, pack16<T>(g));
          }
          if (p.row_scale) {
            const float rs = p.row_scale[m / (p.Hd * p.Wd)];
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] *= rs;
          }
          if (p.addend) {
            float a[EPC];
            unpack16<T>(addv[u], a);
            if (p.addend_mask) {
              const unsigned bits = abit[u];
#pragma unroll
              for (int e = 0; e < EPC; ++e) a[e] = (bits >> e) & 1u ? a[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] += a[e];
          }
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
          }
          if (p.mask_src) {
            float k[EPC];
            unpack16<T>(mskv[u], k);
            bool gelu = false;
            if constexpr (!SPLIT) gelu = (p.flags & 4) != 0;      // the persistent kernels never see such a launch (choose_kernel) and stay free of the code
            if (gelu) {                                     // k = the pre-activation of a GELU: the result is the gradient behind it (Mlp fc2 data gradient, swintransformer.py:26-31)
#pragma unroll
              for (int e = 0; e < EPC; ++e) v[e] *= sl_gelu_grad<T>(k[e]);
            } else {
#pragma unroll
              for (int e = 0; e < EPC; ++e) v[e] = k[e] > 0.f ? v[e] : 0.f;
            }
          }
          st16(out + (size_t)m * p.N + ncol, pack16<T>(v));
        }
      }
    }
    lds_barrier();
  }
  if (p.stat_partial) {
    // lanes l and l + CPR*k of a wave own the same chunk column
#pragma unroll
    for (int off = CPR; off < 64; off <<= 1) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { ssum[e] += __shfl_xor(ssum[e], off, 64); ssq[e] += __shfl_xor(ssq[e], off, 64); }
    }
    float* red = (float*)smem;                       // [NW][2][BN]
    const int wave = tid >> 6;
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        red[(wave * 2 + 0) * BN + cc * EPC + e] = ssum[e];
        red[(wave * 2 + 1) * BN + cc * EPC + e] = ssq[e];
      }
    }
    lds_barrier();
    for (int e = tid; e < 2 * BN; e += NT) {
      const int which = e / BN, col = e % BN;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WM * WN; ++w) t += red[(w * 2 + which) * BN + col];
      p.stat_partial[((size_t)bm * 2 + which) * p.N + bn * BN + col] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// v2: operands go HBM -> LDS directly (global_load_lds, 16 B per lane, no VGPR staging, no ds_write pass).
// The LDS destination of one wave-instruction is lane-linear (base + lane*16 = 8 rows x 128 B), so the bank swizzle is
// applied to the SOURCE chunk index: LDS position p of row r receives global chunk p ^ ((r>>1)&7) -- still inside the
// same 128-byte line of that row, so coalescing is untouched.  Lanes whose tap falls into the padding (or rows >= M)
// read a zero page instead.
__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ __forceinline__ void glds16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)l, 16, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// v4 "ring": NST LDS stages; the loads of stage i+NST-1 are issued while stage i is computed, and a wave only waits
// (counted s_waitcnt vmcnt(N), raw s_barrier -- never vmcnt(0) inside the loop) for the stage it is about to read, so
// the LDS-DMA of NST-2 stages stays in flight across every barrier.  A stage row holds RBYTES (64 or 128) bytes of K.
template <int RBYTES> __device__ __forceinline__ int ring_off(int row, int chunk);
template <> __device__ __forceinline__ int ring_off<128>(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
template <> __device__ __forceinline__ int ring_off<64>(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }
template <int RBYTES> __device__ __forceinline__ int ring_swz(int row);
template <> __device__ __forceinline__ int ring_swz<128>(int row) { return (row >> 1) & 7; }
template <> __device__ __forceinline__ int ring_swz<64>(int row) { return (row >> 2) & 3; }

// LDS-DMA issued from inline asm: hipcc does not count it, so it inserts no vmcnt(0) in front of the ds_reads; ordering
// is ours (counted wait + barrier before the stage is read).  M0 carries the wave-uniform LDS byte address.
__device__ __forceinline__ void glds16_asm(const void* g, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const unsigned char* p) {
  return (unsigned)(size_t)((const __attribute__((address_space(3))) unsigned char*)p);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }


}  // namespace
