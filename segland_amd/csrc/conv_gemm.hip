// Implicit-GEMM convolution on MFMA for gfx950: forward (gather) and data-gradient (transposed gather).
//
//   out[m][n] = sum_{tap, c} src[pix(m, tap)][c] * wt[n][tap][c]
//
// rows m   = destination pixels (b, yd, xd), NHWC
// cols n   = destination channels
// K        = taps x source channels, one K-tile = 128 bytes of channels of one tap (64 bf16 / 32 f32)
// Both element types share the LDS geometry: tiles of [rows][8 x 16-byte chunks], chunk index XOR-swizzled with
// (row>>1)&7 so that the ds_read_b128 fragment reads (row = lane&31) are bank-conflict free.  A bf16 k-step is one
// v_mfma_f32_32x32x16_bf16 per chunk pair; an f32 k-step is four v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).
#include <stdlib.h>
#include "common.h"

#ifndef SL_SK512_PF
#define SL_SK512_PF 4
#endif
#ifndef SL_CH3_SPLIT
#define SL_CH3_SPLIT 2
#endif
namespace {

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
  // split-K (inference convs on few row tiles: the fine-tune pair's 8 192-row layers): the K range is cut into `ksplit` parts, block (tile, part) writes its raw fp32
  // accumulators to ws [part][M][N] and conv_splitk_finish_kernel sums the parts in a fixed order and applies pre_addend / scale / bias / addend / ReLU
  float* ws; int ksplit;
  unsigned long long* trace;            // debug (tools/p8_trace.py): per block {s_memtime at entry, after the prologue, after the main loop, at the end, HW_ID}; null in production
};

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
template <typename T, int BM, int BN, int WM, int WN, int MODE, bool SPLIT = false>
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
  auto soff = [&](int it) -> size_t {
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
    if (p.tile16) {
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
      const unsigned char* ad = (const unsigned char*)p.addend + goff;
      const unsigned char* ab = p.addend_mask ? p.addend_mask + goff / 16 : nullptr;
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
#pragma unroll
          for (int e = 0; e < EPC / 2; ++e) v[e] += w[e];
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
template <typename T, int BM, int BN, int WM, int WN, bool ADD, bool RS_, bool GELU, bool SCRELU>
__device__ __forceinline__ void conv_epilogue_affine(const ConvGemmParams& p, f32x16_t (&acc)[BM / WM / 32][BN / WN / 32], int bm, int bn, int wm, int wn,
                                                     int lane, int tid, unsigned char* smem) {
  using G = EpiGeom<T, BM, BN, WM, WN, false>;
  constexpr int EPC = 16 / sizeof(T), EP2 = EPC / 2;
  constexpr int NT = 64 * WM * WN;
  constexpr int CPR = BN / EPC;
  constexpr int RS = NT / CPR;
  constexpr int ROWS = BM / G::NPASS;
  constexpr int NIT = ROWS / RS, CH = NIT < 8 ? NIT : 8;
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
    epi_stage_acc<T, BM, BN, WM, WN, false>(acc, pass, wm, wn, lane, smem);
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
template <typename T, int BM, int BN, int WM, int WN, bool SPLIT = false, bool GATE = true>
__device__ __forceinline__ int conv_epilogue_lds(const ConvGemmParams& p, f32x16_t (&acc)[BM / WM / 32][BN / WN / 32], int bm, int bn, int wm, int wn,
                                                  int lane, int tid, unsigned char* smem) {
  const bool full = (bm + 1) * BM <= p.M;
  const bool shaped = p.bias || p.scale || p.relu || p.mask_src || p.pre_addend || p.row_scale || p.out2;
  // Returns the number of vector-memory instructions the wave issued (loads + stores; every path below issues the same count in every wave), or -1 when that is
  // not a compile-time fact of the path: the persistent half-tile kernel uses it to wait for loads that are OLDER than these instructions without waiting for them.
  if constexpr (GATE) {
    if (full && !shaped && !p.addend && p.gate) {
      conv_epilogue_fast<T, BM, BN, WM, WN, 3, SPLIT>(p, acc, bm, bn, wm, wn, lane, tid, smem);
      return SPLIT ? 49 : -1;                                           // 2 passes x 8 sweeps x (BN input + gate byte + store), + the statistic partial
    }
  }
  if (full && !shaped && !p.addend) {
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
  if constexpr (!SPLIT) {
    if (full && (p.bias || p.scale) && !p.mask_src && !p.pre_addend && !p.addend_mask && !p.stat_partial && !p.tile16 && !(p.flags & 2)) {
      const bool sr = p.scale || p.relu;              // eval-mode BatchNorm folded into the conv (+ residual + ReLU); the Swin linears have neither
      if (p.out2) {
        if (!p.addend && !p.row_scale && !sr) { conv_epilogue_affine<T, BM, BN, WM, WN, false, false, true, false>(p, acc, bm, bn, wm, wn, lane, tid, smem); return -1; }
      } else if (p.addend) {
        if (p.row_scale) { if (!sr) { conv_epilogue_affine<T, BM, BN, WM, WN, true, true, false, false>(p, acc, bm, bn, wm, wn, lane, tid, smem); return -1; } }
        else if (sr) { conv_epilogue_affine<T, BM, BN, WM, WN, true, false, false, true>(p, acc, bm, bn, wm, wn, lane, tid, smem); return -1; }
        else { conv_epilogue_affine<T, BM, BN, WM, WN, true, false, false, false>(p, acc, bm, bn, wm, wn, lane, tid, smem); return -1; }
      } else if (!p.row_scale) {
        if (sr) conv_epilogue_affine<T, BM, BN, WM, WN, false, false, false, true>(p, acc, bm, bn, wm, wn, lane, tid, smem);
        else conv_epilogue_affine<T, BM, BN, WM, WN, false, false, false, false>(p, acc, bm, bn, wm, wn, lane, tid, smem);
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
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] = k[e] > 0.f ? v[e] : 0.f;
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

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_gemm_glds_kernel(ConvGemmParams p) {
  constexpr int EPC = 16 / sizeof(T);
  constexpr int BKE = 8 * EPC;
  constexpr int NW = WM * WN;                   // waves per block (4 or 8)
  constexpr int AR = BM / 8 / NW, BR = BN / 8 / NW;     // 8-row (1 KiB) groups loaded per wave
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(AR >= 1 && BR >= 1 && TM >= 1 && TN >= 1, "tile/wave shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* lds_a = smem;
  unsigned char* lds_b = smem + 2 * BM * 128;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  int bid = blockIdx.x;
  {
    const int nwg = p.gridM * p.gridN, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bm = bid / p.gridN, bn = bid % p.gridN;

  const int lrow = lane >> 3, lpos = lane & 7;
  // rows handled by this lane: A: (wave*AR + j)*8 + lrow, B: (wave*BR + j)*8 + lrow
  int rb[AR], ry[AR], rx[AR], rsw[AR];
#pragma unroll
  for (int j = 0; j < AR; ++j) {
    const int row = (wave * AR + j) * 8 + lrow;
    rsw[j] = (lpos ^ ((row >> 1) & 7)) * EPC;      // source chunk (element offset) feeding this LDS position
    const int m = bm * BM + row;
    if (m < p.M) {
      const int b = m / (p.Hd * p.Wd), rem = m - b * (p.Hd * p.Wd);
      const int yd = rem / p.Wd, xd = rem - yd * p.Wd;
      rb[j] = b;
      if (p.mode == 0) { ry[j] = yd * p.stride - p.pad; rx[j] = xd * p.stride - p.pad; }
      else             { ry[j] = yd + p.pad;            rx[j] = xd + p.pad; }
    } else { rb[j] = -1; ry[j] = 0; rx[j] = 0; }
  }
  const int CT = p.C1 + p.C2;
  const int ctiles = CT / BKE;
  const int taps = p.KH * p.KW;
  const int nk = taps * ctiles;
  const T* wrow[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    const int row = (wave * BR + j) * 8 + lrow;
    wrow[j] = (const T*)p.wt + (size_t)(bn * BN + row) * taps * CT + (lpos ^ ((row >> 1) & 7)) * EPC;
  }

  // K-tile order: channel tile OUTER, tap INNER -- the taps of one 64-channel slice touch the same cache lines (3x3
  // neighbourhoods overlap), so they hit in L2 instead of re-streaming the slab once per tap (measured: FETCH_SIZE was
  // 7.6x the algorithmic bytes with the tap loop outside).  Per row we keep the source pixel index of tap (0,0) and a
  // validity bit per tap; the per-tap displacement is wave-uniform.  (dgrad through a stride > 1 is not affine in
  // the tap: that rare case recomputes the row per K-tile.)
  const bool affine = (p.mode == 0) || (p.stride == 1);
  const int sgn = p.mode == 0 ? 1 : -1;
  int rbase[AR]; unsigned vmask[AR];
#pragma unroll
  for (int j = 0; j < AR; ++j) {
    rbase[j] = rb[j] >= 0 ? (rb[j] * p.Hs + ry[j]) * p.Ws + rx[j] : 0;     // may be "outside": only used with a valid bit
    unsigned m = 0;
    if (rb[j] >= 0 && affine)
      for (int t = 0; t < taps; ++t) {
        const int ky = t / p.KW, kx = t - ky * p.KW;
        const int ys = ry[j] + sgn * ky * p.dil, xs = rx[j] + sgn * kx * p.dil;
        if ((unsigned)ys < (unsigned)p.Hs && (unsigned)xs < (unsigned)p.Ws) m |= 1u << t;
      }
    vmask[j] = m;
  }
  auto slow_pix = [&](int j, int tap_) -> int {        // dgrad, stride > 1
    const int ky = tap_ / p.KW, kx = tap_ - ky * p.KW;
    const int ty = ry[j] - ky * p.dil, tx = rx[j] - kx * p.dil;
    if (rb[j] < 0 || ty < 0 || tx < 0) return -1;
    const int ys = ty / p.stride, xs = tx / p.stride;
    if (ys * p.stride != ty || xs * p.stride != tx || ys >= p.Hs || xs >= p.Ws) return -1;
    return (rb[j] * p.Hs + ys) * p.Ws + xs;
  };
  int tap = 0, ct = 0;
  const unsigned char* zsrc = g_zero_page + lpos * 16;
  auto issue = [&](int buf) {
    const int c0 = ct * BKE;
    const unsigned char* base; unsigned pitchb;
    if (c0 < p.C1) { base = (const unsigned char*)p.src1 + (size_t)c0 * sizeof(T); pitchb = p.C1 * (unsigned)sizeof(T); }
    else           { base = (const unsigned char*)p.src2 + (size_t)(c0 - p.C1) * sizeof(T); pitchb = p.C2 * (unsigned)sizeof(T); }
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int delta = sgn * (ky * p.dil * p.Ws + kx * p.dil);
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      int pix; bool ok;
      if (affine) { pix = rbase[j] + delta; ok = (vmask[j] >> tap) & 1u; }
      else { pix = slow_pix(j, tap); ok = pix >= 0; }
      const unsigned char* src = base + (size_t)((unsigned)pix) * pitchb + rsw[j] * (int)sizeof(T);
      src = ok ? src : zsrc;
      glds16(src, lds_a + buf * BM * 128 + (wave * AR + j) * 1024);
    }
    const size_t koff = (size_t)tap * CT + c0;
#pragma unroll
    for (int j = 0; j < BR; ++j) glds16(wrow[j] + koff, lds_b + buf * BN * 128 + (wave * BR + j) * 1024);
    if (++tap == taps) { tap = 0; ++ct; }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issue(0);
  __syncthreads();      // the barrier's release waits for the LDS-DMA (vmcnt) of every wave

  const int frow = lane & 31, fhalf = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) issue(buf ^ 1);
    const unsigned char* la = lds_a + buf * BM * 128;
    const unsigned char* lb = lds_b + buf * BN * 128;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *(const uint4*)(la + lds_off(wm * (BM / WM) + i * 32 + frow, 2 * s + fhalf));
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *(const uint4*)(lb + lds_off(wn * (BN / WN) + j * 32 + frow, 2 * s + fhalf));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(bf[j], af[i], acc[i][j]);     // swapped roles: D[n][m]
    }
    __syncthreads();
  }

  conv_epilogue_lds<T, BM, BN, WM, WN>(p, acc, bm, bn, wm, wn, lane, tid, smem);
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

template <typename T, int BM, int BN, int WM, int WN, int RBYTES, int NST>
struct RingGeom {
  static constexpr int STAGE = (BM + BN) * RBYTES;
  static constexpr int RING_BYTES = NST * STAGE;
  static constexpr int EPI = EpiGeom<T, BM, BN, WM, WN>::TILE_BYTES / EpiGeom<T, BM, BN, WM, WN>::NPASS;
  static constexpr int LDS_BYTES = RING_BYTES > EPI ? RING_BYTES : EPI;
};

template <typename T, int BM, int BN, int WM, int WN, int RBYTES, int NST, bool GATE = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_gemm_ring_kernel(ConvGemmParams p) {
  constexpr int EPC = 16 / sizeof(T);
  constexpr int CPRW = RBYTES / 16;             // 16-byte chunks per stage row
  constexpr int BKE = CPRW * EPC;               // K elements per stage
  constexpr int RPI = 1024 / RBYTES;            // rows per 1 KiB wave-instruction
  constexpr int NW = WM * WN;
  constexpr int AR = BM / RPI / NW, BR = BN / RPI / NW;
  constexpr int L = AR + BR;                    // LDS-DMA instructions per wave per stage
  constexpr int KS = CPRW / 2;                  // MFMA k-steps per stage
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  using RG = RingGeom<T, BM, BN, WM, WN, RBYTES, NST>;
  static_assert(AR >= 1 && BR >= 1, "ring shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  int bid = blockIdx.x;
  {
    const int nwg = p.gridM * p.gridN, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bm = bid / p.gridN, bn = bid % p.gridN;
  const int lrow = lane / CPRW, lpos = lane % CPRW;
  const int taps = p.KH * p.KW;
  const int CT = p.C1 + p.C2;
  const int ctiles = CT / BKE;
  const int nk = taps * ctiles;
  const bool affine = (p.mode == 0) || (p.stride == 1);
  const int sgn = p.mode == 0 ? 1 : -1;

  int rb[AR], ry[AR], rx[AR], rbase[AR], rsw[AR]; unsigned vmask[AR];
#pragma unroll
  for (int j = 0; j < AR; ++j) {
    const int row = (wave * AR + j) * RPI + lrow;
    rsw[j] = (lpos ^ ring_swz<RBYTES>(row)) * 16;             // byte offset of the source chunk inside the K slice
    const int m = bm * BM + row;
    rb[j] = -1; ry[j] = 0; rx[j] = 0;
    if (m < p.M) {
      const int b = m / (p.Hd * p.Wd), rem = m - b * (p.Hd * p.Wd);
      const int yd = rem / p.Wd, xd = rem - yd * p.Wd;
      rb[j] = b;
      if (p.mode == 0) { ry[j] = yd * p.stride - p.pad; rx[j] = xd * p.stride - p.pad; }
      else             { ry[j] = yd + p.pad;            rx[j] = xd + p.pad; }
    }
    rbase[j] = rb[j] >= 0 ? (rb[j] * p.Hs + ry[j]) * p.Ws + rx[j] : 0;
    unsigned mk = 0;
    if (rb[j] >= 0 && affine)
      for (int t = 0; t < taps; ++t) {
        const int ky = t / p.KW, kx = t - ky * p.KW;
        const int ys = ry[j] + sgn * ky * p.dil, xs = rx[j] + sgn * kx * p.dil;
        if ((unsigned)ys < (unsigned)p.Hs && (unsigned)xs < (unsigned)p.Ws) mk |= 1u << t;
      }
    vmask[j] = mk;
  }
  auto slow_pix = [&](int j, int tap_) -> int {
    const int ky = tap_ / p.KW, kx = tap_ - ky * p.KW;
    const int ty = ry[j] - ky * p.dil, tx = rx[j] - kx * p.dil;
    if (rb[j] < 0 || ty < 0 || tx < 0) return -1;
    const int ys = ty / p.stride, xs = tx / p.stride;
    if (ys * p.stride != ty || xs * p.stride != tx || ys >= p.Hs || xs >= p.Ws) return -1;
    return (rb[j] * p.Hs + ys) * p.Ws + xs;
  };
  const unsigned char* wrow[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    const int row = (wave * BR + j) * RPI + lrow;
    wrow[j] = (const unsigned char*)p.wt + ((size_t)(bn * BN + row) * taps * CT) * sizeof(T) + (lpos ^ ring_swz<RBYTES>(row)) * 16;
  }
  int tap = 0, ct = 0;
  const unsigned char* zsrc = g_zero_page + lpos * 16;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  auto issue = [&](int slot) {
    const unsigned la = lds_base + slot * RG::STAGE;
    const unsigned lb = la + BM * RBYTES;
    const int c0 = ct * BKE;
    const unsigned char* base; unsigned pitchb;
    if (c0 < p.C1) { base = (const unsigned char*)p.src1 + (size_t)c0 * sizeof(T); pitchb = p.C1 * (unsigned)sizeof(T); }
    else           { base = (const unsigned char*)p.src2 + (size_t)(c0 - p.C1) * sizeof(T); pitchb = p.C2 * (unsigned)sizeof(T); }
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int delta = sgn * (ky * p.dil * p.Ws + kx * p.dil);
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      int pix; bool ok;
      if (affine) { pix = rbase[j] + delta; ok = (vmask[j] >> tap) & 1u; }
      else { pix = slow_pix(j, tap); ok = pix >= 0; }
      const unsigned char* src = base + (size_t)((unsigned)pix) * pitchb + rsw[j];
      glds16_asm(ok ? src : zsrc, la + (wave * AR + j) * 1024);
    }
    const size_t koff = ((size_t)tap * CT + c0) * sizeof(T);
#pragma unroll
    for (int j = 0; j < BR; ++j) glds16_asm(wrow[j] + koff, lb + (wave * BR + j) * 1024);
    if (++tap == taps) { tap = 0; ++ct; }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Pipeline invariant at the top of iteration i: stages i and i+1 are complete and visible to every wave, and the
  // fragments of (stage i, k-step 0) are already in registers.  Inside the iteration the fragment loads of the NEXT
  // k-step (the last one reaches into stage i+1) are issued before the MFMAs of the current one, so LDS latency hides
  // behind the matrix pipe; the LDS-DMA of stages i+2 / i+3 stays in flight across the barrier.
  static_assert(NST >= 3 && NST <= 6 && (KS % 2) == 0, "ring schedule: 3..6 stages and an even number of k-steps");
  constexpr int D = NST - 1;                    // stages issued ahead of the one being consumed
  // wait until at most `fl` of the most recently issued stages are still in flight (fl is block-uniform)
  auto wait_stages = [&](int fl) {
    if constexpr (D >= 5) { if (fl >= 3) { wait_vmcnt<3 * L>(); return; } }
    if constexpr (D >= 4) { if (fl == 2) { wait_vmcnt<2 * L>(); return; } }
    if (fl >= 1) wait_vmcnt<L>(); else wait_vmcnt<0>();
  };
  const int frow = lane & 31, fhalf = lane >> 5;
  auto ldfrag = [&](uint4* af, uint4* bf, int slot_, int s2) {
    const unsigned char* la = smem + slot_ * RG::STAGE;
    const unsigned char* lb = la + BM * RBYTES;
#pragma unroll
    for (int ii = 0; ii < TM; ++ii) af[ii] = *(const uint4*)(la + ring_off<RBYTES>(wm * (BM / WM) + ii * 32 + frow, 2 * s2 + fhalf));
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) bf[jj] = *(const uint4*)(lb + ring_off<RBYTES>(wn * (BN / WN) + jj * 32 + frow, 2 * s2 + fhalf));
  };
  auto mma = [&](const uint4* af, const uint4* bf) {
#pragma unroll
    for (int ii = 0; ii < TM; ++ii)
#pragma unroll
      for (int jj = 0; jj < TN; ++jj) Mma<T>::run(bf[jj], af[ii], acc[ii][jj]);
  };
  uint4 afA[TM], bfA[TN], afB[TM], bfB[TN];

#pragma unroll
  for (int st = 0; st < D; ++st)
    if (st < nk) issue(st);
  wait_stages(min(nk, D) - 2);                            // stages 0 and 1 landed (later ones may still fly)
  __builtin_amdgcn_s_barrier();
  ldfrag(afA, bfA, 0, 0);

  int slot = 0;
  for (int i = 0; i < nk; ++i) {
    if (i + D < nk) { int ns = slot + D; if (ns >= NST) ns -= NST; issue(ns); }
    int nslot = slot + 1; if (nslot == NST) nslot = 0;
#pragma unroll
    for (int s2 = 0; s2 < KS; s2 += 2) {
      ldfrag(afB, bfB, slot, s2 + 1);
      mma(afA, bfA);
      if (s2 + 2 < KS) ldfrag(afA, bfA, slot, s2 + 2);
      else             ldfrag(afA, bfA, nslot, 0);          // first k-step of the next stage (complete by the invariant)
      mma(afB, bfB);
    }
    // make stage i+2 complete before anyone starts iteration i+1; stages i+3 .. i+D (already issued) may stay in flight
    wait_stages(min(i + D, nk - 1) - (i + 2));
    __builtin_amdgcn_s_barrier();
    slot = nslot;
  }
  __syncthreads();
  conv_epilogue_lds<T, BM, BN, WM, WN, false, GATE>(p, acc, bm, bn, wm, wn, lane, tid, smem);
}

template <typename T, int BM, int BN, int WM, int WN, int RBYTES, int NST>
int launch_ring(ConvGemmParams& p, hipStream_t st) {
  using RG = RingGeom<T, BM, BN, WM, WN, RBYTES, NST>;
  p.gridM = cdiv(p.M, BM);
  p.gridN = p.N / BN;
  const size_t lds = RG::LDS_BYTES;
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_gemm_ring_kernel<T, BM, BN, WM, WN, RBYTES, NST, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)conv_gemm_ring_kernel<T, BM, BN, WM, WN, RBYTES, NST, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  // like the half-tile and patch kernels: the gated-statistics store phase lives in an instantiation of its own
  if (p.gate) hipLaunchKernelGGL((conv_gemm_ring_kernel<T, BM, BN, WM, WN, RBYTES, NST, true>), dim3(p.gridM * p.gridN), dim3(64 * WM * WN), lds, st, p);
  else        hipLaunchKernelGGL((conv_gemm_ring_kernel<T, BM, BN, WM, WN, RBYTES, NST, false>), dim3(p.gridM * p.gridN), dim3(64 * WM * WN), lds, st, p);
  SL_LAUNCH_CHECK("conv_gemm_ring_kernel");
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------
// v5 (bf16, N % 256 == 0): 256x256 tile, K-tile of 64 elements = 128-BYTE operand rows, LDS = 2 K-tiles x 4 half-tile slots
// {A0, A1, B0, B1} of 16 KiB (128 rows x 128 B).  tools/micro/glds_bw.hip: the LDS-DMA path delivers 30-50 % more bytes/s when
// each row request is a full 128-byte line than with the 64-byte rows of the v4 ring, and v4 sits exactly on that limit.
// A K-tile is computed as 4 phases, one output quadrant each -- (A0,B0) (A0,B1) (A1,B1) (A1,B0) -- so a slot is free again
// after at most two phases and is refilled with the same half of the K-tile two steps ahead:
//   P0: issue B0(i+2)            P1: issue A0(i+2)            P3: issue A1(i+2), B1(i+2)       (A0 and B0 have three slots, mod 3)
// The A fragments of a half (8 x 16 B per lane) stay in registers for its two phases and are refilled in place (A1 during P1, the
// next K-tile's A0 during P3); B fragments stream through a 4-deep register ring, three k-steps ahead (P3 re-uses the B0 fragments
// and P2 the B1 fragments of P1 from registers: 24 LDS fragment reads per 32 MFMAs).  Two counted waits and two barriers per K-tile.
// Wave (wm, wn) of the 2 x 4 grid owns rows {h*128 + wm*64 ..+63} and columns {h*128 + wn*32 ..+31} of both halves h.
// split-K: the wave's accumulators (half-tile layout: tile row = half*128 + wm*64 + i2*32 + lane&31, column = j*128 + wn*32 + 8q + 4*(lane>>5) .. +3) as fp32 to
// ws [part][M][N]; 32 16-byte stores per lane.  tile16: row block bm is a 16 x 16-pixel tile.
__device__ __forceinline__ void conv_store_partial(const ConvGemmParams& p, f32x16_t (&acc)[4][2], int part, int bm, int bn, int wm, int wn, int lane) {
  const int l31 = lane & 31, fh = lane >> 5;
  float* base = p.ws + (size_t)part * p.M * p.N + bn * 256 + wn * 32 + 4 * fh;
  int tbase = 0;
  if (p.tile16) { const int tx = p.Wd >> 4, ty = p.Hd >> 4; tbase = ((bm / (tx * ty)) * p.Hd + ((bm / tx) % ty) * 16) * p.Wd + (bm % tx) * 16; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int R = (i >> 1) * 128 + wm * 64 + (i & 1) * 32 + l31;
    const int m = p.tile16 ? tbase + (R >> 4) * p.Wd + (R & 15) : bm * 256 + R;
    if (m < p.M) {
      float* row = base + (size_t)m * p.N;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(float4*)(row + j * 128 + 8 * q) = make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
    }
  }
}

// out = act((sum over parts + pre_addend) * scale + bias + addend): the epilogue of the generic store phase (same operation order) behind a split-K launch
template <typename T>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(ConvGemmParams p) {
  constexpr int EPC = 16 / sizeof(T);
  const size_t nvec = (size_t)p.M * p.N / EPC, slab = (size_t)p.M * p.N;
  const int nvc = p.N / EPC;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (size_t)gridDim.x * 256) {
    float a[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) a[e] = 0.f;
    for (int s = 0; s < p.ksplit; ++s) {
      const float4* src = (const float4*)(p.ws + s * slab + v * EPC);
#pragma unroll
      for (int h = 0; h < EPC / 4; ++h) { const float4 t = src[h]; a[4 * h + 0] += t.x; a[4 * h + 1] += t.y; a[4 * h + 2] += t.z; a[4 * h + 3] += t.w; }
    }
    // the tile kernels round the accumulators to T when they stage the tile and apply the epilogue to the rounded values: the same here, so that a layer gives the
    // same result whichever way it is dispatched (up to the order of the K sum)
    float r[EPC];
    unpack16<T>(pack16<T>(a), r);
    const int c = (int)(v % nvc) * EPC;
    if (p.pre_addend) {
      float t[EPC];
      unpack16<T>(((const uint4*)p.pre_addend)[v], t);
#pragma unroll
      for (int e = 0; e < EPC; ++e) r[e] += t[e];
    }
    if (p.bias || p.scale) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) r[e] = r[e] * (p.scale ? p.scale[c + e] : 1.f) + (p.bias ? p.bias[c + e] : 0.f);
    }
    if (p.addend) {
      float t[EPC];
      unpack16<T>(((const uint4*)p.addend)[v], t);
#pragma unroll
      for (int e = 0; e < EPC; ++e) r[e] += t[e];
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) r[e] = r[e] > 0.f ? r[e] : 0.f;
    }
    ((uint4*)p.out)[v] = pack16<T>(r);
  }
}

constexpr int P8_SLOT = 128 * 128;
constexpr int P8_RING = 10 * P8_SLOT;         // A0 x3, A1 x2, B1 x2, B0 x3 = the whole 160 KiB
constexpr int P8_LDS = P8_RING;
static_assert(EpiGeom<bf16_t, 256, 256, 2, 4, true>::TILE_BYTES / 2 <= 6 * P8_SLOT, "the half-tile staging area must fit below the four slots of a prefetched K-tile");

// PERSISTENT: the grid is min(tiles, 256) blocks (one per CU, 160 KiB of LDS each) and a block walks over tiles b, b + grid, ...  tools/p8_trace.py (s_memtime per
// block) showed where a one-tile block spends its time on the short-K layers: 512 -> 2048 forward 15 % prologue (address set-up + the HBM latency of the first K-tile)
// / 61 % main loop / 24 % epilogue, 2048 -> 512 data gradient with gated addend 15 / 46 / 40, 1024 -> 2048 10 / 75 / 15, 3x3 512 -> 512 6 / 90 / 4, plus ~800 ticks
// between two blocks on a CU.  Here the NEXT tile's set-up and first K-tile (LDS-DMA into four slots) are issued right after the main loop, so that latency runs
// under the epilogue; the epilogue stages the tile in two half-tile passes in the six slots the prefetch leaves free.  Physical slot order (16 KiB each):
//   0,1 = A0 of K-tiles 1,2 (mod 3)   2 = A1 odd   3 = B1 odd   4,5 = B0 of K-tiles 1,2   | 6 = A0 of K-tile 0   7 = A1 even   8 = B1 even   9 = B0 of K-tile 0
// vmcnt is in order over loads AND stores: the first wait of the next tile (all but the 8 LDS-DMA of its second K-tile) also covers the epilogue's stores, which by
// then have had the set-up of the second K-tile to drain.
__device__ __forceinline__ int p8_slot_a0(int j) { return (j == 0 ? 6 : j - 1) * P8_SLOT; }
__device__ __forceinline__ int p8_slot_a1(int par) { return (par ? 2 : 7) * P8_SLOT; }
__device__ __forceinline__ int p8_slot_b1(int par) { return (par ? 3 : 8) * P8_SLOT; }
__device__ __forceinline__ int p8_slot_b0(int j) { return (j == 0 ? 9 : j + 3) * P8_SLOT; }

template <int EPI>       // 0: the store phases without MODE 3, 1: with the gated-statistics store phase (MODE 3)
__global__ __launch_bounds__(512) void conv_gemm_p8_kernel(ConvGemmParams p) {
  constexpr bool GATE = EPI == 1;
  using T = bf16_t;
  constexpr int BM = 256, BN = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
  if (p.trace) tr0 = __builtin_amdgcn_s_memtime();
  const int ntiles = p.gridM * p.gridN;
  const int taps = p.KH * p.KW;
  const int CT = p.C1 + p.C2;
  const int nk = taps * (CT / 64);
  const int sgn = p.mode == 0 ? 1 : -1;

  // ---- load side: instruction j of this wave fills rows wave*16 + j*8 + (lane>>3) of a half-tile slot, 16 B per lane
  const int lr = lane >> 3, lpos = lane & 7;
  int rbase[2][2]; unsigned vmask[2][2]; int rsw[2];
  const unsigned char* wptr[2][2];
  const size_t wpitch = (size_t)taps * CT * sizeof(T);
#pragma unroll
  for (int j = 0; j < 2; ++j) rsw[j] = (lpos ^ (((j * 8 + lr) >> 1) & 7)) * 16;
  auto tile_of = [&](int t, int& bm, int& bn) {                        // XCD-aware order: the tiles a CU group of one XCD works on concurrently share their A rows
    const int q = ntiles >> 3, r = ntiles & 7, xcd = t & 7, idx = t >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bm = bid / p.gridN; bn = bid % p.gridN;
  };

  auto setup = [&](int bm, int bn) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = h * 128 + wave * 16 + j * 8 + lr;
        const int m = bm * BM + row;
        rbase[h][j] = 0; vmask[h][j] = 0;
        if (m < p.M) {
          const int b = m / (p.Hd * p.Wd), rem = m - b * (p.Hd * p.Wd);
          const int yd = rem / p.Wd, xd = rem - yd * p.Wd;
          const int ry = p.mode == 0 ? yd * p.stride - p.pad : yd + p.pad;
          const int rx = p.mode == 0 ? xd * p.stride - p.pad : xd + p.pad;
          rbase[h][j] = (b * p.Hs + ry) * p.Ws + rx;
          unsigned mk = 0;
          for (int t = 0; t < taps; ++t) {
            const int ky = t / p.KW, kx = t - ky * p.KW;
            const int ys = ry + sgn * ky * p.dil, xs = rx + sgn * kx * p.dil;
            if ((unsigned)ys < (unsigned)p.Hs && (unsigned)xs < (unsigned)p.Ws) mk |= 1u << t;
          }
          vmask[h][j] = mk;
        }
        wptr[h][j] = (const unsigned char*)p.wt + (size_t)(bn * BN + row) * wpitch + rsw[j];
      }
  };
  const unsigned char* zsrc = g_zero_page + lpos * 16;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  auto issueA = [&](int h, int tap, int ct, int slot_off) {
    const unsigned dst = lds_base + slot_off + wave * 2048;
    const int c0 = ct * 64;
    const unsigned char* base; unsigned pitchb;
    if (c0 < p.C1) { base = (const unsigned char*)p.src1 + (size_t)c0 * sizeof(T); pitchb = p.C1 * (unsigned)sizeof(T); }
    else           { base = (const unsigned char*)p.src2 + (size_t)(c0 - p.C1) * sizeof(T); pitchb = p.C2 * (unsigned)sizeof(T); }
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int delta = sgn * (ky * p.dil * p.Ws + kx * p.dil);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = (vmask[h][j] >> tap) & 1u;
      const unsigned char* src = base + (size_t)((unsigned)(rbase[h][j] + delta)) * pitchb + rsw[j];
      glds16_asm(ok ? src : zsrc, dst + j * 1024);
    }
  };
  auto issueB = [&](int h, int tap, int ct, int slot_off) {
    const unsigned dst = lds_base + slot_off + wave * 2048;
    const size_t koff = ((size_t)tap * CT + ct * 64) * sizeof(T);
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16_asm(wptr[h][j] + koff, dst + j * 1024);
  };
  auto adv = [&](int& tap, int& ct) { if (++tap == taps) { tap = 0; ++ct; } };
  auto issue_first = [&]() { issueB(0, 0, 0, p8_slot_b0(0)); issueA(0, 0, 0, p8_slot_a0(0)); issueA(1, 0, 0, p8_slot_a1(0)); issueB(1, 0, 0, p8_slot_b1(0)); };

  // ---- fragment side: lane (l31, fh) reads row base + l31, 16-byte chunk 2*ks + fh (swizzled) of a slot
  const int l31 = lane & 31, fh = lane >> 5;
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = l31 * 128 + (((2 * ks + fh) ^ ((l31 >> 1) & 7)) << 4);
  const unsigned char* fa = smem + wm * (64 * 128);
  const unsigned char* fb = smem + wn * (32 * 128);
  auto ldA = [&](int slot_off, int i2, int ks) { return *(const uint4*)(fa + slot_off + i2 * 4096 + foff[ks]); };
  auto ldB = [&](int slot_off, int ks) { return *(const uint4*)(fb + slot_off + foff[ks]); };

  int bm, bn;
  tile_of(blockIdx.x, bm, bn);
  setup(bm, bn);
  issue_first();
  int younger = 0;                               // vector-memory instructions issued after the current K-tile 0 was requested (see the first wait)
  const bool counted = p.flags & 1;
#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    f32x16_t acc[4][2];                         // [half*2 + 32-row block][column half]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue.  Global issue order is K-tile by K-tile: B0(s), A0(s), A1(s), B1(s) (K-tile 0 is already in flight); the loop continues it with
    // B0(i+2) in P0(i), A0(i+2) in P1(i), A1(i+2) and B1(i+2) in P3(i).
    int tap2 = 0, ct2 = 0;                      // K-tile i+2 (after the prologue)
    adv(tap2, ct2);
    if (nk > 1) {
      issueB(0, tap2, ct2, p8_slot_b0(1)); issueA(0, tap2, ct2, p8_slot_a0(1)); issueA(1, tap2, ct2, p8_slot_a1(1)); issueB(1, tap2, ct2, p8_slot_b1(1));
      // K-tile 0 must have landed.  Younger than its loads: the 8 LDS-DMA just issued and the `younger` loads / stores of the previous tile's epilogue, which need not be waited for
      switch (younger) { case 16: wait_vmcnt<24>(); break; case 17: wait_vmcnt<25>(); break; case 32: wait_vmcnt<40>(); break; case 48: wait_vmcnt<56>(); break; case 49: wait_vmcnt<57>(); break; default: wait_vmcnt<8>(); }
    } else wait_vmcnt<0>();
    adv(tap2, ct2);
    __builtin_amdgcn_s_barrier();
    if (p.trace && tile == (int)blockIdx.x) tr1 = __builtin_amdgcn_s_memtime();
    // B fragments live in two register sets that are loaded IN PLACE four k-steps before their first use: b0k (B0 half: used in P0 and P3, reloaded for the next K-tile
    // right after its P3 use) and b1k (B1 half: loaded in P0, used in P1 and P2).  (An earlier form streamed them through a 4-deep ring and copied them into keep registers:
    // 32 v_mov per K-tile next to 32 MFMAs.)
    uint4 a[4][2], b0k[4], b1k[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { a[ks][0] = ldA(p8_slot_a0(0), 0, ks); a[ks][1] = ldA(p8_slot_a0(0), 1, ks); b0k[ks] = ldB(p8_slot_b0(0), ks); }

    int s3 = 0;                                 // i mod 3
    for (int i = 0; i < nk; ++i) {
      const int par = i & 1;
      const int s3n = s3 == 2 ? 0 : s3 + 1, s3nn = s3 == 0 ? 2 : s3 - 1;                                   // (i+1) % 3, (i+2) % 3
      const int b0nxt = p8_slot_b0(s3n), b0nn = p8_slot_b0(s3nn);                                         // B0 slots of K-tiles i+1, i+2
      const int a1cur = p8_slot_a1(par), b1cur = p8_slot_b1(par), a0nxt = p8_slot_a0(s3n), a0nn = p8_slot_a0(s3nn);
      const bool more2 = i + 2 < nk;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int ph = q >> 2, ks = q & 3;
        const int ih = ph >> 1, jh = (ph == 1 || ph == 2) ? 1 : 0;
        // the two waves of a SIMD (w, w + 4) run the same phase; their address/issue sections are placed two k-steps apart so that
        // one wave's MFMAs cover the other's VALU + LDS-DMA issue (same per-wave issue order, so the vmcnt arithmetic is unchanged)
        if ((ks == 0 || ks == 2) && more2 && (ks == 2) == (wm == 1)) {
          if (ph == 0) issueB(0, tap2, ct2, b0nn);
          if (ph == 1) issueA(0, tap2, ct2, a0nn);
          if (ph == 3) { issueA(1, tap2, ct2, a1cur); issueB(1, tap2, ct2, b1cur); }
        }
        if (ph == 0) b1k[ks] = ldB(b1cur, ks);                                           // B1 of this K-tile, used from P1 on
        const uint4 bq = (ph == 0 || ph == 3) ? b0k[ks] : b1k[ks];
        Mma<T>::run(bq, a[ks][0], acc[ih * 2 + 0][jh]);
        Mma<T>::run(bq, a[ks][1], acc[ih * 2 + 1][jh]);
        if (ph == 1) { a[ks][0] = ldA(a1cur, 0, ks); a[ks][1] = ldA(a1cur, 1, ks); }   // A1 of this K-tile
        if (ph == 3) { a[ks][0] = ldA(a0nxt, 0, ks); a[ks][1] = ldA(a0nxt, 1, ks); b0k[ks] = ldB(b0nxt, ks); }   // A0 and B0 of the next K-tile
        if (ks == 3) {
          // end of P2: B0(i+1), A0(i+1) must have landed (read in P3); end of P3: A1(i+1), B1(i+1) (read from P0(i+1) on)
          if (ph == 2) { if (more2) wait_vmcnt<8>(); else if (i + 1 < nk) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
          if (ph == 3) { if (more2) wait_vmcnt<8>(); else wait_vmcnt<0>(); }
          if (ph >= 2) __builtin_amdgcn_s_barrier();
        }
      }
      adv(tap2, ct2);
      s3 = s3 == 2 ? 0 : s3 + 1;
    }
    if (p.trace && tile == (int)blockIdx.x) tr2 = __builtin_amdgcn_s_memtime();
    lds_barrier();
    // every slot is idle: the next tile's row map, weight rows and first K-tile go out now and land under the epilogue (slots 6-9; the staging passes use 0-5)
    const int cbm = bm, cbn = bn;
    if (tile + (int)gridDim.x < ntiles) {
      tile_of(tile + gridDim.x, bm, bn);
      setup(bm, bn);
      issue_first();
    }
    younger = conv_epilogue_lds<T, BM, BN, 2, 4, true, GATE>(p, acc, cbm, cbn, wm, wn, lane, tid, smem);
    if (!counted) younger = 0;
    lds_barrier();                              // statistic partials are read from the staging area: the next tile's second K-tile goes to slots inside it
  }
  if (p.trace && tid == 0) {
    unsigned long long* t = p.trace + (size_t)blockIdx.x * 8;
    t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = __builtin_amdgcn_s_memtime(); t[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); t[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));      // HW_ID, XCC_ID
  }
}

int launch_splitk_finish(ConvGemmParams& p, hipStream_t st) {
  const size_t nvec = (size_t)p.M * p.N / 8;
  const int blocks = (int)((nvec + 255) / 256 < 4096 ? (nvec + 255) / 256 : 4096);
  hipLaunchKernelGGL(conv_splitk_finish_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, p);
  SL_LAUNCH_CHECK("conv_splitk_finish_kernel");
  return 0;
}

unsigned long long* g_p8_trace = nullptr;
int launch_p8(ConvGemmParams& p, hipStream_t st) {
  p.gridM = cdiv(p.M, 256);
  p.gridN = p.N / 256;
  p.trace = g_p8_trace;
  p.flags = 1;                                  // the next tile's first wait is counted past the epilogue's own loads and stores
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_gemm_p8_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8_LDS);
    (void)hipFuncSetAttribute((const void*)conv_gemm_p8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8_LDS);
    attr_set = true;
  }
  const int ntiles = p.gridM * p.gridN;
  // persistent: min(tiles, 256) blocks walk over the tiles; the instantiation with the gated-statistics store phase only where it is used
  if (p.gate) hipLaunchKernelGGL(conv_gemm_p8_kernel<1>, dim3(ntiles > 256 ? 256 : ntiles), dim3(512), P8_LDS, st, p);
  else        hipLaunchKernelGGL(conv_gemm_p8_kernel<0>, dim3(ntiles > 256 ? 256 : ntiles), dim3(512), P8_LDS, st, p);
  SL_LAUNCH_CHECK("conv_gemm_p8_kernel");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3, stride 1 layers of the dilated trunk (pad = dilation, N % 256 == 0, H and W multiples of 16): the half-tile kernel above fetches the pixel operand once per TAP --
// nine shifted copies of nearly the same rows -- and its main loop is co-limited by the LDS fill rate.  Here a block owns a 16 x 16-pixel output tile:
// per 64-channel chunk the input patch WITH its dilation halo ((16 + 2d)^2 pixels x 128 B: 41 / 50 / 72 KiB for d = 1 / 2 / 4) goes to the LDS once and the nine taps read
// their fragments from shifted patch rows; only the weight rows (two 16 KiB halves per tap) stream through a two-K-tile ring.  Same wave grid, accumulator layout, phases
// (A0,B0) (A0,B1) (A1,B1) (A1,B0) and epilogues as the half-tile kernel (A0 / A1 = image rows 0-7 / 8-15 of the tile); one barrier per tap.
constexpr int P9_PATCH = 576 * 128;                                     // largest patch (d = 4)
constexpr int P9_LDS = P9_PATCH + 4 * P8_SLOT;                          // + B0 / B1 of two K-tiles = 136 KiB
constexpr int P9_PATCH1 = 42 * 1024;                                    // d = 1: 324 rows -> two patch buffers (the next chunk's patch lands under the current chunk's taps)
constexpr int P9_LDS1 = 2 * P9_PATCH1 + 4 * P8_SLOT;                    // 148 KiB
template <int EPI>       // 0 / 1 / 2 as in conv_gemm_p8_kernel; split-K parts are ranges of 64-channel chunks (all nine taps of a chunk stay together: one patch per chunk)
__global__ __launch_bounds__(512) void conv_gemm_p9_kernel(ConvGemmParams p) {
  using T = bf16_t;
  constexpr int BM = 256, BN = 256;
  constexpr bool GATE = EPI == 1, SPLITK = EPI == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int KS = SPLITK ? p.ksplit : 1;
  const int part = SPLITK ? (int)(blockIdx.x % KS) : 0;               // neighbouring blocks share the tile: the same patch rows and weight rows pass through the L2 together
  int bid = blockIdx.x / KS;
  {
    const int nwg = p.gridM * p.gridN, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // block order: column tiles in groups of two, row tiles inside a group -- with many column tiles (PPM data gradient: 8 x 2.4 MB of weight rows) an XCD's contiguous share of
  // the blocks then covers ONE group, whose weights stay in its L2, instead of streaming all 18.9 MB once per round of its 32 CUs
  const int GN = p.gridN > 2 && p.gridN % 2 == 0 && !(p.flags & 16) ? 2 : p.gridN;
  const int grp = bid / (p.gridM * GN), rem = bid - grp * (p.gridM * GN);
  const int bm = rem / GN, bn = grp * GN + rem % GN;
  const int d = p.dil, PW = 16 + 2 * d, PP = PW * PW;
  const int CT = p.C1;
  const int cbeg = SPLITK ? (CT / 64) * part / KS : 0, nchunk = SPLITK ? (CT / 64) * (part + 1) / KS : CT / 64;      // chunks [cbeg, nchunk)
  const int tx = p.Ws >> 4, ty = p.Hs >> 4;
  const int bx = bm % tx, by = (bm / tx) % ty, bb = bm / (tx * ty);
  const int y0 = by * 16 - d, x0 = bx * 16 - d;                          // image position of patch pixel (0, 0)
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  const int lr = lane >> 3, lpos = lane & 7;
  const unsigned char* zsrc = g_zero_page + lpos * 16;

  // ---- patch fill: wave-instruction g = j * 8 + wave covers patch rows g * 8 .. + 7 (lane -> row lr, 16-byte position lpos; the swizzle is applied to the source piece)
  constexpr int NPI = 9;                                                  // instructions per wave: 9 x 8 waves x 8 rows = 576 rows (rows >= PP are skipped)
  const unsigned char* psrc[NPI];
#pragma unroll
  for (int j = 0; j < NPI; ++j) {
    const int pr = (j * 8 + wave) * 8 + lr;
    const int py = pr / PW, px = pr - py * PW;
    const int iy = y0 + py, ix = x0 + px;
    const bool ok = pr < PP && (unsigned)iy < (unsigned)p.Hs && (unsigned)ix < (unsigned)p.Ws;
    psrc[j] = ok ? (const unsigned char*)p.src1 + ((size_t)(bb * p.Hs + iy) * p.Ws + ix) * CT * sizeof(T) + ((lpos ^ ((px >> 1) & 7)) << 4) : nullptr;      // swizzle by the patch COLUMN (see ldA)
  }
  const bool dbuf = d == 1 && !(p.flags & 8);                             // two patch buffers fit (flags bit 3 forces one: unused since round 4)
  const int boff = dbuf ? 2 * P9_PATCH1 : P9_PATCH;                       // weight ring behind the patch area
  auto issue_patch = [&](int chunk) {
    const unsigned dst = lds_base + (dbuf && (chunk & 1) ? P9_PATCH1 : 0);
#pragma unroll
    for (int j = 0; j < NPI; ++j) {
      if ((j * 8 + wave) * 8 < PP)                                        // wave-uniform
        glds16_asm(psrc[j] ? psrc[j] + (size_t)chunk * 128 : zsrc, dst + (j * 8 + wave) * 1024);
    }
  };
  // ---- weight rows: half h, rows h*128 + wave*16 + j*8 + lr of the block's 256 output channels; K-tile (tap, chunk) at byte offset (tap * CT + chunk * 64) * 2
  const size_t wpitch = (size_t)9 * CT * sizeof(T);
  int rsw[2];
  const unsigned char* wptr[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) rsw[j] = (lpos ^ (((j * 8 + lr) >> 1) & 7)) * 16;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) wptr[h][j] = (const unsigned char*)p.wt + (size_t)(bn * BN + h * 128 + wave * 16 + j * 8 + lr) * wpitch + rsw[j];
  auto issueB = [&](int tap, int chunk, int par) {
    const size_t koff = ((size_t)tap * CT + chunk * 64) * sizeof(T);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned dst = lds_base + boff + (par * 2 + h) * P8_SLOT + wave * 2048;
#pragma unroll
      for (int j = 0; j < 2; ++j) glds16_asm(wptr[h][j] + koff, dst + j * 1024);
    }
  };
  // ---- fragments.  B: as in the half-tile kernel.  A: lane (l31, fh) of row block (h, i2) is tile pixel (ty, tx) = (h*8 + wm*4 + i2*2 + (l31 >> 4), l31 & 15) -> patch row
  // (ty + ky d) PW + tx + kx d, 16-byte piece (2 ks + fh) ^ ((row >> 1) & 7)
  const int l31 = lane & 31, fh = lane >> 5;
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = l31 * 128 + (((2 * ks + fh) ^ ((l31 >> 1) & 7)) << 4);
  const unsigned char* fb = smem + boff + wn * (32 * 128);
  auto ldB = [&](int slot_off, int ks) { return *(const uint4*)(fb + slot_off + foff[ks]); };
  int prow[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) prow[h][i2] = (h * 8 + wm * 4 + i2 * 2 + (l31 >> 4)) * PW + (l31 & 15);
  // The XOR swizzle is a function of the patch COLUMN px, not of the LDS row: a ds_read_b128 is serviced in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... --
  // with lane = (image row l31 >> 4, pixel l31 & 15) a group holds 16 DIFFERENT columns of two image rows, and since the patch width is even the row parity (address bit 7)
  // is the column parity: (px & 1, (px >> 1) & 7) is distinct for 16 consecutive columns, whatever the tap shift.  (Swizzled by the LDS row, the second image row of a group
  // landed on the first one's banks: SQ_LDS_BANK_CONFLICT = 40 % of the LDS cycles.)
  auto ldA = [&](int h, int i2, int toff, int ks, int pbase = 0) {
    const int pr = prow[h][i2] + (toff >> 8);                             // toff = ((ky PW + kx) d) << 8 | kx d
    const int px = (l31 & 15) + (toff & 255);
    return *(const uint4*)(smem + pbase + pr * 128 + (((2 * ks + fh) ^ ((px >> 1) & 7)) << 4));
  };

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // K-tile k = (chunk, tap), weight slot pair k & 1.  One barrier per K-tile, at the end of P2: by then every wave has read both halves of slot pair k & 1 (B0(k) in P3 of
  // K-tile k - 1, B1(k) in P0), so B(k + 2) is issued into it right there and has a whole K-tile to land; B(k + 1) is waited for at the same point and P3 already loads the next
  // K-tile's first fragments (B0 from the other slot pair, the first eight image rows of the patch at the next tap's offset), so no K-tile starts with an empty pipeline.
  const int NK = 9 * (nchunk - cbeg);
  auto toff_of = [&](int tap) { const int t2 = p.mode ? 8 - tap : tap; return ((((t2 / 3) * PW + (t2 % 3)) * d) << 8) | ((t2 % 3) * d); };      // patch row offset << 8 | column offset      // data gradient: the correlation with the flipped window
  issue_patch(cbeg);
  issueB(0, cbeg, 0);
  issueB(1, cbeg, 1);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  uint4 a[4][2], b0k[4], b1k[4];
  {
    const int toff = toff_of(0), pb0 = dbuf && (cbeg & 1) ? P9_PATCH1 : 0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { a[ks][0] = ldA(0, 0, toff, ks, pb0); a[ks][1] = ldA(0, 1, toff, ks, pb0); b0k[ks] = ldB(0, ks); }
  }
  int tap = 0, chunk = cbeg;
#pragma unroll 1
  for (int k = 0; k < NK; ++k) {
    const int par = k & 1;
    const int b1s = (par * 2 + 1) * P8_SLOT, b0n = ((par ^ 1) * 2 + 0) * P8_SLOT;
    const int toff = toff_of(tap);
    const bool last_tap = tap == 8;
    const int toffn = toff_of(last_tap ? 0 : tap + 1);
    const int pb = dbuf && (chunk & 1) ? P9_PATCH1 : 0, pbn = dbuf && last_tap ? (pb ? 0 : P9_PATCH1) : pb;      // patch buffer of this / of the next K-tile
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int ph = q >> 2, ks = q & 3;
      const int ih = ph >> 1, jh = (ph == 1 || ph == 2) ? 1 : 0;
      if (ph == 0) b1k[ks] = ldB(b1s, ks);
      const uint4 bq = (ph == 0 || ph == 3) ? b0k[ks] : b1k[ks];
      Mma<T>::run(bq, a[ks][0], acc[ih * 2 + 0][jh]);
      Mma<T>::run(bq, a[ks][1], acc[ih * 2 + 1][jh]);
      if (ph == 1) { a[ks][0] = ldA(1, 0, toff, ks, pb); a[ks][1] = ldA(1, 1, toff, ks, pb); }   // image rows 8-15 of the tile
      if (ph == 2 && ks == 3) {
        wait_vmcnt<0>();                                                  // B(k + 1) (and, with two patch buffers, the next chunk's patch once it has been requested)
        __builtin_amdgcn_s_barrier();
        if (dbuf && tap == 0 && chunk + 1 < nchunk) issue_patch(chunk + 1);      // the other buffer: every wave is past the previous chunk
        if (k + 2 < NK) {
          int t2 = tap + 2, c2 = chunk;
          if (t2 >= 9) { t2 -= 9; ++c2; }
          issueB(t2, c2, par);
        }
      }
      if (ph == 3) {
        b0k[ks] = ldB(b0n, ks);                                           // B0 of the next K-tile
        if (!last_tap || dbuf) { a[ks][0] = ldA(0, 0, toffn, ks, pbn); a[ks][1] = ldA(0, 1, toffn, ks, pbn); }
      }
    }
    if (last_tap) {
      tap = 0; ++chunk;
      if (chunk < nchunk && !dbuf) {
        __builtin_amdgcn_s_barrier();                                     // every wave is past its last read of the patch
        issue_patch(chunk);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { a[ks][0] = ldA(0, 0, toffn, ks); a[ks][1] = ldA(0, 1, toffn, ks); }
      }
    } else ++tap;
  }
  if constexpr (SPLITK) { conv_store_partial(p, acc, part, bm, bn, wm, wn, lane); return; }
  lds_barrier();
  conv_epilogue_lds<T, BM, BN, 2, 4, true, GATE>(p, acc, bm, bn, wm, wn, lane, tid, smem);
}

int g_conv_p9 = -1;      // SEGLAND_CONV_P9 / sl_debug_conv_p9
static bool p9_on() {
  if (g_conv_p9 < 0) g_conv_p9 = (getenv("SEGLAND_CONV_P9") && getenv("SEGLAND_CONV_P9")[0] == '0') ? 0 : 1;
  return g_conv_p9 != 0;
}
static bool p9_shape(const ConvGemmParams& p) {
  return p9_on() && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == p.dil && (p.dil == 1 || p.dil == 2 || p.dil == 4) && p.C2 == 0 && p.C1 % 64 == 0 && p.N % 256 == 0 &&
         p.Hs == p.Hd && p.Ws == p.Wd && p.Hs % 16 == 0 && p.Ws % 16 == 0 && ((long long)p.M >= 32768 || p.ksplit > 1) &&
         !(p.out2 || p.row_scale);                                       // every epilogue with the tile16 row map (fast: store / statistics / gated addend; generic: bias, folded BN, ReLU, pre-addend)
}
int launch_p9(ConvGemmParams& p, hipStream_t st) {
  p.gridM = p.M / 256; p.gridN = p.N / 256; p.tile16 = 1;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_gemm_p9_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(P9_LDS1 > P9_LDS ? P9_LDS1 : P9_LDS));
    (void)hipFuncSetAttribute((const void*)conv_gemm_p9_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(P9_LDS1 > P9_LDS ? P9_LDS1 : P9_LDS));
    (void)hipFuncSetAttribute((const void*)conv_gemm_p9_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(P9_LDS1 > P9_LDS ? P9_LDS1 : P9_LDS));
    attr_set = true;
  }
  if (p.ksplit > 1) {
    hipLaunchKernelGGL(conv_gemm_p9_kernel<2>, dim3(p.gridM * p.gridN * p.ksplit), dim3(512), p.dil == 1 ? P9_LDS1 : P9_LDS, st, p);
    SL_LAUNCH_CHECK("conv_gemm_p9_kernel (split-K)");
    return launch_splitk_finish(p, st);
  }
  if (p.gate) hipLaunchKernelGGL(conv_gemm_p9_kernel<1>, dim3(p.gridM * p.gridN), dim3(512), p.dil == 1 ? P9_LDS1 : P9_LDS, st, p);
  else        hipLaunchKernelGGL(conv_gemm_p9_kernel<0>, dim3(p.gridM * p.gridN), dim3(512), p.dil == 1 ? P9_LDS1 : P9_LDS, st, p);
  SL_LAUNCH_CHECK("conv_gemm_p9_kernel");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// 64 -> 64 channels, 3x3, stride 1, dilation 1 (layer1.conv2 at 128 x 128), forward and data gradient.  The tile kernels above fetch the pixel
// operand once per tap: with only 64 output channels per 64 input channels that makes the launch LDS-fill bound at a third of what the MFMAs
// could do.  Here a persistent block owns 16 x 16-pixel tiles: the input patch WITH its halo goes to the LDS once (324 pixels for 256 outputs)
// and the nine taps are nine shifted 16-byte reads per fragment; the whole weight tensor (64 x 576) sits in the LDS in MFMA fragment order; the
// next tile's patch is loaded into registers while the current one is multiplied.  Same result layout, same BN statistic partials (one row per
// TILE: sl_conv2d_stat_rows knows) as the generic path.  mode 1 (data gradient) = the same kernel on the transposed weights with the taps flipped.
constexpr int C64_T = 16, C64_PW = C64_T + 2, C64_PITCH = 144;        // 128 B of channels + 16 B pad: conflict-free 16-byte fragment reads
constexpr int C64_WFRAG = 2 * 36 * 64 * 16;                           // weights in fragment order, bytes
constexpr int C64_LDS = C64_WFRAG + C64_PW * C64_PW * C64_PITCH + 4 * 2 * 64 * (int)sizeof(float);
__global__ __launch_bounds__(256) void conv_c64k3_kernel(ConvGemmParams p, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* wl = (uint4*)smem;                                           // [nb][ks][lane]
  unsigned char* patch = smem + C64_WFRAG;                            // [18*18][144 B]; the output tile [256][144 B] takes its place after the MFMA loop
  float* red = (float*)(patch + C64_PW * C64_PW * C64_PITCH);         // [4][2][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, fh = lane >> 5;
  const int H = p.Hs, W = p.Ws;
  const int tx = cdiv(W, C64_T), ty = cdiv(H, C64_T);
  const bf16_t* src = (const bf16_t*)p.src1;
  for (int e = tid; e < 2 * 36 * 64; e += 256) {                      // wt [64][9][64] -> fragments: rows nb*32 + l31, tap ks/4, channels (ks%4)*16 + fh*8 .. +8
    const int ln = e & 63, ks = (e >> 6) % 36, nb = (e >> 6) / 36;
    wl[e] = *(const uint4*)((const bf16_t*)p.wt + ((size_t)(nb * 32 + (ln & 31)) * 9 + (ks >> 2)) * 64 + (ks & 3) * 16 + (ln >> 5) * 8);
  }
  constexpr int NCH = C64_PW * C64_PW * 8, CPT = (NCH + 255) / 256;   // 11 chunks of 16 B per thread
  uint4 stage[CPT];
  auto fetch = [&](int tile) {
    int blk = tile;
    const int bx = blk % tx; blk /= tx;
    const int by = blk % ty; const int b = blk / ty;
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      const int e = tid + u * 256;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (e < NCH) {
        const int ch8 = e & 7, pp = e >> 3, py = pp / C64_PW, px = pp - py * C64_PW;
        const int iy = by * C64_T - 1 + py, ix = bx * C64_T - 1 + px;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = *(const uint4*)(src + ((size_t)(b * H + iy) * W + ix) * 64 + ch8 * 8);
      }
      stage[u] = v;
    }
  };
  int pbase[2];                                                       // patch byte offset of this lane's pixel in row block rb (tap (0,0))
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int pidx = wave * 64 + rb * 32 + l31;
    pbase[rb] = ((pidx >> 4) * C64_PW + (pidx & 15)) * C64_PITCH + fh * 16;
  }
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int blk = tile;
    const int bx = blk % tx; blk /= tx;
    const int by = blk % ty; const int b = blk / ty;
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      const int e = tid + u * 256;
      if (e < NCH) *(uint4*)(patch + (e >> 3) * C64_PITCH + (e & 7) * 16) = stage[u];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
    f32x16_t acc[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][nb][r] = 0.f;
#pragma unroll 4
    for (int ks = 0; ks < 36; ++ks) {
      int tap = ks >> 2;
      if (p.mode) tap = 8 - tap;                                      // data gradient: the correlation with the flipped window
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int toff = (ky * C64_PW + kx) * C64_PITCH + (ks & 3) * 32;
      const uint4 w0 = wl[(0 * 36 + ks) * 64 + lane], w1 = wl[(1 * 36 + ks) * 64 + lane];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const uint4 a = *(const uint4*)(patch + pbase[rb] + toff);
        acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w0), __builtin_bit_cast(bf16x8_t, a), acc[rb][0], 0, 0, 0);
        acc[rb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w1), __builtin_bit_cast(bf16x8_t, a), acc[rb][1], 0, 0, 0);
      }
    }
    __syncthreads();                                                  // the patch is dead: stage the tile (lane = pixel, register r = channel (r&3) + 8(r>>2) + 4fh)
    unsigned char* outt = patch;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int pidx = wave * 64 + rb * 32 + l31;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint2 pk;
          pk.x = (unsigned)f2bf(acc[rb][nb][4 * q + 0]) | ((unsigned)f2bf(acc[rb][nb][4 * q + 1]) << 16);
          pk.y = (unsigned)f2bf(acc[rb][nb][4 * q + 2]) | ((unsigned)f2bf(acc[rb][nb][4 * q + 3]) << 16);
          *(uint2*)(outt + pidx * C64_PITCH + (nb * 32 + 8 * q + 4 * fh) * 2) = pk;
        }
    }
    __syncthreads();
    bf16_t* out = (bf16_t*)p.out;
    float sa[8], sq[8];                                               // this thread's 8 channels (tid & 7) over its 8 pixels: column sums of the stored (rounded) tile
#pragma unroll
    for (int c = 0; c < 8; ++c) { sa[c] = 0.f; sq[c] = 0.f; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + u * 256;
      const int pidx = e >> 3, ch8 = e & 7;
      const int oy = by * C64_T + (pidx >> 4), ox = bx * C64_T + (pidx & 15);
      if (oy < H && ox < W) {
        const uint4 v = *(const uint4*)(outt + pidx * C64_PITCH + ch8 * 16);
        st16(out + ((size_t)(b * H + oy) * W + ox) * 64 + ch8 * 8, v);
        if (p.stat_partial) {
          const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float lo = __uint_as_float(wv[c] << 16), hi = __uint_as_float(wv[c] & 0xffff0000u);
            sa[2 * c] += lo; sq[2 * c] += lo * lo; sa[2 * c + 1] += hi; sq[2 * c + 1] += hi * hi;
          }
        }
      }
    }
    if (p.stat_partial) {                                             // lanes 8 apart share the channel octet: three xor steps, then the four waves through the LDS
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { sa[c] += __shfl_xor(sa[c], o); sq[c] += __shfl_xor(sq[c], o); }
      if (lane < 8) {
#pragma unroll
        for (int c = 0; c < 8; ++c) { red[(wave * 2 + 0) * 64 + lane * 8 + c] = sa[c]; red[(wave * 2 + 1) * 64 + lane * 8 + c] = sq[c]; }
      }
      __syncthreads();
      if (tid < 128) {
        const int which = tid >> 6, ch = tid & 63;
        p.stat_partial[((size_t)tile * 2 + which) * 64 + ch] = red[(0 * 2 + which) * 64 + ch] + red[(1 * 2 + which) * 64 + ch] + red[(2 * 2 + which) * 64 + ch] + red[(3 * 2 + which) * 64 + ch];
      }
    }
    __syncthreads();                                                  // the next iteration overwrites the tile with its patch
  }
}

static bool c64k3_shape(int dtype, int KH, int KW, int stride, int pad, int dil, int Cin, int C1, int Cout, long long M) {
  static const bool off = getenv("SEGLAND_CONV_C64K3") && getenv("SEGLAND_CONV_C64K3")[0] == '0';
  return !off && dtype == SL_BF16 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && dil == 1 && Cin == 64 && C1 == 64 && Cout == 64 && M >= 65536;
}

int launch_c64k3(ConvGemmParams& p, hipStream_t st) {
  const int ntiles = p.B * cdiv(p.Hs, C64_T) * cdiv(p.Ws, C64_T);
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv_c64k3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS); attr_set = true; }
  hipLaunchKernelGGL(conv_c64k3_kernel, dim3(ntiles < 256 ? ntiles : 256), dim3(256), C64_LDS, st, p, ntiles);
  SL_LAUNCH_CHECK("conv_c64k3_kernel");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Short-K 1x1 convs (64 / 128 / 256 input channels, stride 1, >= 65 536 pixels): out[M][N] = A[M][K] x W[N][K]^T is bound by the HBM traffic of
// `out` (and of the residual addend in the data gradient), not by the MFMAs: the tile kernels above spend a round per 256 x 256 tile on
// load -> 1..4 K-tiles -> store with nothing overlapping the store.  Here the PIXEL operand is stationary: a block owns 256 rows, each of its
// 8 waves keeps its 32 rows x K in registers as MFMA fragments (read once, straight from global memory) and walks over N in steps of 64
// columns; only the weight rows of a step (64 x K, from the L2) go through a two-slot LDS ring (LDS-DMA, rows XOR-swizzled for the fragment
// reads).  Per step a wave: issues its share of the next step's weight rows and this step's addend loads, runs 2 x K/16 MFMAs, stages its
// 32 x 64 result through its own LDS patch (row-major, rounded), waits for everything it has in flight (the stores of the PREVIOUS step have
// had a whole step to drain), adds / gates / stores 16 bytes per lane in full 128-byte lines, and meets the other waves at a barrier.
// BN statistic partials (one row per 256-row block, like the tile kernels) come from the rounded values in the store loop.
// SKEW: the two waves of a SIMD (w, w + 4) run HALF A STEP APART: while waves 0-3 multiply step s (matrix pipe), waves 4-7 add / gate / store
// step s - 1 (VALU + memory), and vice versa -- two barriers per step; waves 0-3 issue all the LDS-DMA.
template <int KS> struct SkGeom {
  static constexpr int RB = KS * 32;                                   // operand row bytes (K bf16)
  static constexpr int BSTEP = 64 * RB;                                // weight rows of one step
  static constexpr int STG_PITCH = 144, STG_WAVE = 32 * STG_PITCH;     // 32 rows x (128 B + pad) per wave
  static constexpr int OFF_STG = 2 * BSTEP, OFF_RED = OFF_STG + 8 * STG_WAVE;
  static constexpr int LDS = OFF_RED + 2 * 8 * 3 * 64 * (int)sizeof(float);      // red[parity][wave][sum, sq, sq2][64 columns] (sq2: the second BatchNorm of MODE 5's dual form)
};
template <int KS> __device__ __forceinline__ int sk_swz(int row) { return KS == 16 ? (row & 31) : (KS == 8 ? (row & 15) : ((row >> 1) & 7)); }
// sums over the lanes 8, 16 and 32 apart (the lanes of a wave that share lane & 7), without the LDS: one rotation inside the 16-lane rows, then the
// row swaps of gfx950 (permlane16_swap: odd rows of the first operand <-> even rows of the second; permlane32_swap: upper half <-> lower half)
__device__ __forceinline__ float sk_sum_8_16_32(float v) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));       // row_ror:8
  u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r.x) + __uint_as_float(r.y);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

template <int KS, int MODE, bool SKEW>       // MODE 1: store (+ statistics), 2: + (bit-gated) addend, 5: + addend, result gated with the ReLU bits of its own positions + BN-backward column sums (ConvGemmParams::gate)
__global__ __launch_bounds__(512, (KS == 4 && MODE != 5) ? 4 : 2) void conv_gemm_sk_kernel(ConvGemmParams p) {      // MODE 5 holds 81 KiB of LDS at KS = 4: one block per CU whatever the registers allow
  using G = SkGeom<KS>;
  using T = bf16_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                                           // SKEW: 0 leads, 1 is half a step behind
  const int l31 = lane & 31, fh = lane >> 5;
  const int bm = blockIdx.x;
  const int NS = p.N / 64;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  // weight rows of a step: 2 KS wave-instructions of 1 KiB shared by the issuing waves; the swizzle is applied to the SOURCE chunk (the LDS side of LDS-DMA is lane-linear)
  constexpr int NWI = SKEW ? 4 : 8, NI = 2 * KS / NWI, LPR = 2 * KS, RPI = 64 / LPR;
  const int iw = SKEW ? (wave & 3) : wave;
  const bool issuer = !SKEW || grp == 0;
  const unsigned char* bsrc[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = (iw * NI + j) * RPI + lane / LPR, pos = lane % LPR;
    bsrc[j] = (const unsigned char*)p.wt + (size_t)row * G::RB + ((pos ^ sk_swz<KS>(row)) << 4);
  }
  auto issueB = [&](int s, int buf) {
#pragma unroll
    for (int j = 0; j < NI; ++j) glds16_asm(bsrc[j] + (size_t)s * G::BSTEP, lds_base + buf * G::BSTEP + (iw * NI + j) * 1024);
  };
  if (issuer) issueB(0, 0);
  uint4 a[KS];
  {
    const unsigned char* arow = (const unsigned char*)p.src1 + ((size_t)bm * 256 + wave * 32 + l31) * G::RB + fh * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = *(const uint4*)(arow + ks * 32);
  }
  int foff[KS];
  {
    const int x = sk_swz<KS>(l31);                                     // rows l31 and 32 + l31 of the step share the swizzle (all three patterns have period <= 32)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) foff[ks] = l31 * G::RB + (((2 * ks + fh) ^ x) << 4);
  }
  unsigned char* stg = smem + G::OFF_STG + wave * G::STG_WAVE;
  float* red = (float*)(smem + G::OFF_RED);
  const int srow = lane >> 3, sch = lane & 7;                          // store phase: row it * 8 + srow, 16-byte chunk sch of the wave's 32 x 64 patch (a full 128-byte line per row)
  const size_t orow = (size_t)bm * 256 + wave * 32 + srow;
  // addend rows of this lane's four result rows (the same in every step): the result row itself, or -- addend_half -- row (b, y/2, x/2) of the half-resolution
  // tensor at even (y, x) and none (-1) elsewhere: dx of a 1x1 stride-2 conv is zero at the odd positions (resnet.py:109-110 downsample backward)
  long long arow[4];
  if constexpr (MODE == 2 || MODE == 5) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long long m = (long long)orow + it * 8;
      arow[it] = m;
      if (p.addend_half) {
        const int hw = p.Hd * p.Wd;
        const int b = (int)(m / hw), rem = (int)(m - (long long)b * hw), y = rem / p.Wd, x = rem - y * p.Wd;
        arow[it] = ((y | x) & 1) ? -1 : ((long long)b * (p.Hd >> 1) + (y >> 1)) * (p.Wd >> 1) + (x >> 1);
      }
    }
  }
  uint4 addv[4], cxv[4], cxv2[4];                                      // MODE 5: cxv = the BN input c of the result's positions (cxv2: the second BatchNorm's, dual form)
  float bmu[8], bis[8], bmu2[8], bis2[8];
  const bool dual = MODE == 5 && p.bn_x2 != nullptr;                   // wave-uniform
  // the gate bytes of the block's 256 rows (N / 8 per row, contiguous over the rows) are copied to the LDS once: read step by step from global memory, each step would
  // pull 8 useful bytes out of every row's line, and 256 lines per step do not survive in the 32 KiB L1 next to the addend stream (measured: 58 -> 73 us on 1024 -> 256)
  const int mpitch = p.N / 8 + 16;
  unsigned char* msk = smem + (MODE == 5 ? G::LDS : G::OFF_RED);       // MODE 5 keeps the statistic partials too: its gate bytes sit behind them
  if ((MODE == 2 && p.addend_mask) || MODE == 5) {
    const int cpr = p.N / 128;                                         // 16-byte chunks per row
    const unsigned char* src = (MODE == 5 ? p.gate : p.addend_mask) + (size_t)bm * 256 * (p.N / 8);
    for (int e = tid; e < 256 * cpr; e += 512) {
      const int row = e / cpr, c = e - row * cpr;
      *(uint4*)(msk + row * mpitch + c * 16) = *(const uint4*)(src + (size_t)e * 16);
    }
  }

  auto multiply = [&](int s) {
    const int cur = s & 1;
    if (issuer && s + 1 < NS) issueB(s + 1, cur ^ 1);
    if constexpr (MODE == 2 || MODE == 5) {
      const int ncol = s * 64 + sch * 8;
#pragma unroll
      for (int it = 0; it < 4; ++it) addv[it] = arow[it] >= 0 ? *(const uint4*)((const T*)p.addend + arow[it] * p.N + ncol) : make_uint4(0, 0, 0, 0);
      if constexpr (MODE == 5) {
#pragma unroll
        for (int it = 0; it < 4; ++it) cxv[it] = *(const uint4*)((const T*)p.bn_x + (orow + it * 8) * p.N + ncol);
#pragma unroll
        for (int e = 0; e < 8; ++e) { bmu[e] = p.bn_mean[ncol + e]; bis[e] = p.bn_invstd[ncol + e]; }
        if (dual) {
#pragma unroll
          for (int it = 0; it < 4; ++it) cxv2[it] = *(const uint4*)((const T*)p.bn_x2 + (orow + it * 8) * p.N + ncol);
#pragma unroll
          for (int e = 0; e < 8; ++e) { bmu2[e] = p.bn_mean2[ncol + e]; bis2[e] = p.bn_invstd2[ncol + e]; }
        }
      }
    }
    f32x16_t acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned char* bb = smem + cur * G::BSTEP;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint4 b0 = *(const uint4*)(bb + foff[ks]), b1 = *(const uint4*)(bb + 32 * G::RB + foff[ks]);
      Mma<T>::run(b0, a[ks], acc[0]);
      Mma<T>::run(b1, a[ks], acc[1]);
    }
    // D layout: lane = pixel (l31), register r = column (r & 3) + 8 (r >> 2) + 4 fh of column half j
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint2 v;
        v.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[j][4 * q + 0], acc[j][4 * q + 1]}, bf16x2_t));
        v.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[j][4 * q + 2], acc[j][4 * q + 3]}, bf16x2_t));
        *(uint2*)(stg + l31 * G::STG_PITCH + 64 * j + 16 * q + 8 * fh) = v;
      }
  };

  auto store = [&](int s) {
    const int cur = s & 1;
    const int ncol = s * 64 + sch * 8;
    wait_vmcnt<0>();                                                   // the next step's weight rows, this step's addend, the previous step's stores
    float sa[8], sq[8], sq2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sa[e] = 0.f; sq[e] = 0.f; sq2[e] = 0.f; }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const uint4 raw = *(const uint4*)(stg + (it * 8 + srow) * G::STG_PITCH + sch * 16);
      T* o = (T*)p.out + (orow + it * 8) * p.N + ncol;
      if constexpr (MODE == 1) {
        st16(o, raw);
        if (p.stat_partial) {
          const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float lo = __uint_as_float(w[c] << 16), hi = __uint_as_float(w[c] & 0xffff0000u);
            sa[2 * c] += lo; sq[2 * c] += lo * lo; sa[2 * c + 1] += hi; sq[2 * c + 1] += hi * hi;
          }
        }
      } else {
        uint4 ad = addv[it];
        if (MODE == 2 && p.addend_mask) {
          const unsigned b = msk[(wave * 32 + it * 8 + srow) * mpitch + s * 8 + sch];
          ad.x &= ((unsigned)__builtin_amdgcn_sbfe(b, 0, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 1, 1) & 0xffff0000u);
          ad.y &= ((unsigned)__builtin_amdgcn_sbfe(b, 2, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 3, 1) & 0xffff0000u);
          ad.z &= ((unsigned)__builtin_amdgcn_sbfe(b, 4, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 5, 1) & 0xffff0000u);
          ad.w &= ((unsigned)__builtin_amdgcn_sbfe(b, 6, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 7, 1) & 0xffff0000u);
        }
        const unsigned rw[4] = {raw.x, raw.y, raw.z, raw.w}, aw[4] = {ad.x, ad.y, ad.z, ad.w};
        unsigned ow[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x2_t v = (f32x2_t){__uint_as_float(rw[c] << 16) + __uint_as_float(aw[c] << 16), __uint_as_float(rw[c] & 0xffff0000u) + __uint_as_float(aw[c] & 0xffff0000u)};
          ow[c] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
        }
        if constexpr (MODE == 5) {
          // gate the ROUNDED sum with the ReLU bits of its own positions (what the separate passes would see), then the column sums of g and g * xhat
          const unsigned b = msk[(wave * 32 + it * 8 + srow) * mpitch + s * 8 + sch];
          ow[0] &= ((unsigned)__builtin_amdgcn_sbfe(b, 0, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 1, 1) & 0xffff0000u);
          ow[1] &= ((unsigned)__builtin_amdgcn_sbfe(b, 2, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 3, 1) & 0xffff0000u);
          ow[2] &= ((unsigned)__builtin_amdgcn_sbfe(b, 4, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 5, 1) & 0xffff0000u);
          ow[3] &= ((unsigned)__builtin_amdgcn_sbfe(b, 6, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 7, 1) & 0xffff0000u);
          const unsigned xw[4] = {cxv[it].x, cxv[it].y, cxv[it].z, cxv[it].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float glo = __uint_as_float(ow[c] << 16), ghi = __uint_as_float(ow[c] & 0xffff0000u);
            const float xlo = __uint_as_float(xw[c] << 16), xhi = __uint_as_float(xw[c] & 0xffff0000u);
            sa[2 * c] += glo; sq[2 * c] += glo * ((xlo - bmu[2 * c]) * bis[2 * c]);
            sa[2 * c + 1] += ghi; sq[2 * c + 1] += ghi * ((xhi - bmu[2 * c + 1]) * bis[2 * c + 1]);
          }
          if (dual) {
            const unsigned yw[4] = {cxv2[it].x, cxv2[it].y, cxv2[it].z, cxv2[it].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float glo = __uint_as_float(ow[c] << 16), ghi = __uint_as_float(ow[c] & 0xffff0000u);
              const float ylo = __uint_as_float(yw[c] << 16), yhi = __uint_as_float(yw[c] & 0xffff0000u);
              sq2[2 * c] += glo * ((ylo - bmu2[2 * c]) * bis2[2 * c]);
              sq2[2 * c + 1] += ghi * ((yhi - bmu2[2 * c + 1]) * bis2[2 * c + 1]);
            }
          }
        }
        st16(o, make_uint4(ow[0], ow[1], ow[2], ow[3]));
      }
    }
    if ((MODE == 1 || MODE == 5) && p.stat_partial) {                  // lanes 8 apart share the column octet
#pragma unroll
      for (int e = 0; e < 8; ++e) { sa[e] = sk_sum_8_16_32(sa[e]); sq[e] = sk_sum_8_16_32(sq[e]); }
      if (dual) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sq2[e] = sk_sum_8_16_32(sq2[e]);
      }
      if (lane < 8) {
        float* r0 = red + ((cur * 8 + wave) * 3) * 64 + sch * 8;
        *(float4*)(r0) = make_float4(sa[0], sa[1], sa[2], sa[3]); *(float4*)(r0 + 4) = make_float4(sa[4], sa[5], sa[6], sa[7]);
        *(float4*)(r0 + 64) = make_float4(sq[0], sq[1], sq[2], sq[3]); *(float4*)(r0 + 68) = make_float4(sq[4], sq[5], sq[6], sq[7]);
        if (dual) { *(float4*)(r0 + 128) = make_float4(sq2[0], sq2[1], sq2[2], sq2[3]); *(float4*)(r0 + 132) = make_float4(sq2[4], sq2[5], sq2[6], sq2[7]); }
      }
    }
  };
  auto finalize = [&](int s, int t0) {                                 // 128 (dual: 192) threads from t0 on, after the barrier behind the last store phase of step s
    if ((MODE == 1 || MODE == 5) && p.stat_partial && tid >= t0 && tid < t0 + (dual ? 192 : 128)) {
      const int which = (tid - t0) >> 6, col = tid & 63;               // 0: sum g, 1: sum g * xhat, 2: sum g * xhat2
      const float* r0 = red + ((s & 1) * 24 + which) * 64 + col;
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += r0[k * 192];
      if (which < 2) p.stat_partial[((size_t)bm * 2 + which) * p.N + s * 64 + col] = t;
      if (dual && which != 1) p.stat_partial2[((size_t)bm * 2 + (which >> 1)) * p.N + s * 64 + col] = t;      // the second BatchNorm's partials repeat sum g
    }
  };

  // raw barriers: __syncthreads() would also wait for the stores in flight.  LDS writes (statistic partials; the staging patch is wave-private) are drained explicitly.
  auto bar = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  wait_vmcnt<0>();
  bar();
  if constexpr (!SKEW) {
#pragma unroll 1
    for (int s = 0; s < NS; ++s) {
      multiply(s);
      store(s);
      bar();
      finalize(s, 0);
    }
  } else if (grp == 0) {
#pragma unroll 1
    for (int s = 0; s < NS; ++s) {
      multiply(s);
      bar();
      store(s);
      bar();
    }
    bar();
  } else {
    bar();
#pragma unroll 1
    for (int s = 0; s < NS; ++s) {
      multiply(s);
      bar();
      store(s);
      bar();
      finalize(s, 256);
    }
  }
}

static bool sk_shape(int dtype, int KH, int KW, int stride, int pad, int Cin, int C1, int N, long long M) {
  static const bool off = getenv("SEGLAND_CONV_SK") && getenv("SEGLAND_CONV_SK")[0] == '0';
  return !off && dtype == SL_BF16 && KH == 1 && KW == 1 && stride == 1 && pad == 0 && C1 == Cin && (Cin == 64 || Cin == 128 || Cin == 256) &&
         N % 64 == 0 && M >= 65536 && M % 256 == 0;
}

template <int KS, int MODE, bool SKEW>
int launch_sk_t(ConvGemmParams& p, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv_gemm_sk_kernel<KS, MODE, SKEW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
  const int lds = MODE == 5 ? SkGeom<KS>::LDS + 256 * (p.N / 8 + 16)
                            : (MODE == 2 && p.addend_mask ? SkGeom<KS>::OFF_RED + 256 * (p.N / 8 + 16) : SkGeom<KS>::LDS);      // statistic partials and / or the block's gate bytes behind the staging patches
  hipLaunchKernelGGL((conv_gemm_sk_kernel<KS, MODE, SKEW>), dim3(p.M / 256), dim3(512), lds, st, p);
  SL_LAUNCH_CHECK("conv_gemm_sk_kernel");
  return 0;
}
template <int MODE, bool SKEW>
int launch_sk_k(ConvGemmParams& p, hipStream_t st) {
  return p.C1 == 256 ? launch_sk_t<16, MODE, SKEW>(p, st) : (p.C1 == 128 ? launch_sk_t<8, MODE, SKEW>(p, st) : launch_sk_t<4, MODE, SKEW>(p, st));
}
int launch_sk(ConvGemmParams& p, hipStream_t st) {
  // measured (tools/sk_time.sh): half-a-step-apart wave groups pay where the store phase carries the statistics (256 -> 1024 forward: 53 -> 49 us) and cost where it is pure
  // memory traffic (data gradient 1024 -> 256: 41 -> 47 us, with addend 58 -> 67 us).  bit 0: statistics, bit 1: plain store, bit 2: addend
  constexpr int skew = 1;
  p.gridM = p.M / 256; p.gridN = 1;
  if (p.gate) return launch_sk_k<5, false>(p, st);
  if (p.addend) return (skew & 4) ? launch_sk_k<2, true>(p, st) : launch_sk_k<2, false>(p, st);
  return (skew & (p.stat_partial ? 1 : 2)) ? launch_sk_k<1, true>(p, st) : launch_sk_k<1, false>(p, st);
}

// ---------------------------------------------------------------------------------------------------------------
// K = 512 1x1 convs at >= 65 536 pixels (layer4 conv3 forward 512 -> 2048, the data gradient of layer4 conv1 2048 <- 512 with its shortcut addend, 512 -> 1024 / 512 -> 256):
// on the half-tile kernel a 256 x 256 tile of these layers is 8 K-tiles of main loop between a prologue and a store phase that nothing overlaps (tools/p8_trace.py: 15 / 61 / 24 %
// and 15 / 46 / 40 %; 600-700 TFLOP/s), and BOTH operands pass the LDS-DMA path at 32 B per clock and CU.  The pixel-stationary form of conv_gemm_sk_kernel at K = 512:
//   * each wave keeps its 32 pixel rows x 512 channels in 128 registers as MFMA fragments (read once from HBM); only the WEIGHT rows stream (from the L2): 16 B per clock
//     and CU at the MFMA rate; a step's 64 x 32 x 32 result tile is stored (full 128-byte lines, + addend / gate bits / statistics) while the next step multiplies;
//   * the weight rows of a step (64 rows x 1 KiB) arrive as two halves of 32 rows through a ring of THREE 32 KiB slots, each half issued a full step before its first
//     read: half 2s+3 right behind the barrier that ends the reads of half 2s (its slot), half 2s+4 behind the step's second barrier;
//   * counted vmcnt waits (a wave's VMEM order per step: addend (+ gate byte) loads, 4 LDS-DMA, 4 stores, 4 LDS-DMA), vmcnt(0) on the last two steps.
struct Sk5Geom {
  static constexpr int KS = 32, RB = 1024;                             // k-steps, operand row bytes
  static constexpr int HSLOT = 32 * RB;                                // half a step of weight rows
  static constexpr int OFF_STG = 3 * HSLOT;
  static constexpr int STG_PITCH = 144, STG_WAVE = 32 * STG_PITCH;
  static constexpr int OFF_RED = OFF_STG + 8 * STG_WAVE;
  static constexpr int LDS = OFF_RED + 2 * 8 * 2 * 64 * (int)sizeof(float);
};
template <int MODE>       // 1: store (+ statistics), 2: + addend, gated by the bits of addend_mask when given
__global__ __launch_bounds__(512, 2) void conv_gemm_sk512_kernel(ConvGemmParams p) {
  using G = Sk5Geom;
  using T = bf16_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, fh = lane >> 5;
  const int bm = blockIdx.x;
  const int NS = p.N / 64, NH = 2 * NS;
  const unsigned long long tr_entry = p.trace ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  // half-step H = weight rows 32 H .. 32 H + 31, one KiB instruction per row, four rows per wave; source-side swizzle: chunk ^ (row & 31)
  const unsigned char* bsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = wave * 4 + j;
    bsrc[j] = (const unsigned char*)p.wt + (size_t)row * G::RB + ((lane ^ (row & 31)) << 4);
  }
  auto issueH = [&](int H, int slot) {
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16_asm(bsrc[j] + (size_t)H * G::HSLOT, lds_base + slot * G::HSLOT + (wave * 4 + j) * 1024);
  };
  issueH(0, 0);
  issueH(1, 1);
  uint4 a[G::KS];
  {
    const unsigned char* arow = (const unsigned char*)p.src1 + ((size_t)bm * 256 + wave * 32 + l31) * G::RB + fh * 16;
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) a[ks] = *(const uint4*)(arow + ks * 32);
  }
  const int fbase = l31 * G::RB, fx = l31;                              // fragment chunk (2 ks + fh) ^ (row & 31) of weight row l31 of the half
  unsigned char* stg = smem + G::OFF_STG + wave * G::STG_WAVE;
  float* red = (float*)(smem + G::OFF_RED);
  const int srow = lane >> 3, sch = lane & 7;
  const size_t orow = (size_t)bm * 256 + wave * 32 + srow;
  const bool gated = MODE == 2 && p.addend_mask != nullptr;
  uint4 addv[4];
  unsigned gbyte[4];

  // weight fragments are read PF k-steps ahead of their MFMA (the compiler's own schedule keeps ONE ds_read_b128 in flight: a wave alone on its SIMD then issues an
  // MFMA every ~86 clocks instead of every 32 -- tools/sk512_trace.py)
  auto half = [&](const unsigned char* bb, f32x16_t& acc) {
    constexpr int PF = SL_SK512_PF;
    const unsigned char* rowp = bb + fbase;
    uint4 bq[PF];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < PF; ++u) bq[u] = *(const uint4*)(rowp + (((2 * u + fh) ^ fx) << 4));
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const uint4 b = bq[ks % PF];
      if (ks + PF < G::KS) bq[ks % PF] = *(const uint4*)(rowp + (((2 * (ks + PF) + fh) ^ fx) << 4));
      Mma<T>::run(b, a[ks], acc);
    }
    // pin the order the source has (hipcc would sink every read to just in front of its MFMA): PF reads, then MFMA / read pairs, then the last PF MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
    for (int ks = 0; ks < G::KS - PF; ++ks) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, PF, 0);
  };
  auto stage = [&](const f32x16_t& acc, int j) {                      // D layout: lane = pixel (l31), register r = column (r & 3) + 8 (r >> 2) + 4 fh of column half j
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint2 v;
      v.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[4 * q + 0], acc[4 * q + 1]}, bf16x2_t));
      v.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[4 * q + 2], acc[4 * q + 3]}, bf16x2_t));
      *(uint2*)(stg + l31 * G::STG_PITCH + 64 * j + 16 * q + 8 * fh) = v;
    }
  };
  auto store = [&](int s) {
    const int cur = s & 1;
    const int ncol = s * 64 + sch * 8;
    float sa[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sa[e] = 0.f; sq[e] = 0.f; }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const uint4 raw = *(const uint4*)(stg + (it * 8 + srow) * G::STG_PITCH + sch * 16);
      T* o = (T*)p.out + (orow + it * 8) * p.N + ncol;
      if constexpr (MODE == 1) {
        st16(o, raw);
        if (p.stat_partial) {
          const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float lo = __uint_as_float(w[c] << 16), hi = __uint_as_float(w[c] & 0xffff0000u);
            sa[2 * c] += lo; sq[2 * c] += lo * lo; sa[2 * c + 1] += hi; sq[2 * c + 1] += hi * hi;
          }
        }
      } else {
        uint4 ad = addv[it];
        if (gated) {
          const unsigned b = gbyte[it];
          ad.x &= ((unsigned)__builtin_amdgcn_sbfe(b, 0, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 1, 1) & 0xffff0000u);
          ad.y &= ((unsigned)__builtin_amdgcn_sbfe(b, 2, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 3, 1) & 0xffff0000u);
          ad.z &= ((unsigned)__builtin_amdgcn_sbfe(b, 4, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 5, 1) & 0xffff0000u);
          ad.w &= ((unsigned)__builtin_amdgcn_sbfe(b, 6, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 7, 1) & 0xffff0000u);
        }
        const unsigned rw[4] = {raw.x, raw.y, raw.z, raw.w}, aw[4] = {ad.x, ad.y, ad.z, ad.w};
        unsigned ow[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x2_t v = (f32x2_t){__uint_as_float(rw[c] << 16) + __uint_as_float(aw[c] << 16), __uint_as_float(rw[c] & 0xffff0000u) + __uint_as_float(aw[c] & 0xffff0000u)};
          ow[c] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
        }
        st16(o, make_uint4(ow[0], ow[1], ow[2], ow[3]));
      }
    }
    if (MODE == 1 && p.stat_partial) {                                 // lanes 8 apart share the column octet
#pragma unroll
      for (int e = 0; e < 8; ++e) { sa[e] = sk_sum_8_16_32(sa[e]); sq[e] = sk_sum_8_16_32(sq[e]); }
      if (lane < 8) {
        float* r0 = red + ((cur * 8 + wave) * 2) * 64 + sch * 8;
        *(float4*)(r0) = make_float4(sa[0], sa[1], sa[2], sa[3]); *(float4*)(r0 + 4) = make_float4(sa[4], sa[5], sa[6], sa[7]);
        *(float4*)(r0 + 64) = make_float4(sq[0], sq[1], sq[2], sq[3]); *(float4*)(r0 + 68) = make_float4(sq[4], sq[5], sq[6], sq[7]);
      }
    }
  };
  auto finalize = [&](int s) {                                         // 128 threads, behind the barrier that follows the store phase of step s
    if (MODE == 1 && p.stat_partial && tid < 128) {
      const int which = (tid >> 6) & 1, col = tid & 63;
      const float* r0 = red + ((s & 1) * 16 + which) * 64 + col;
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += r0[k * 128];
      p.stat_partial[((size_t)bm * 2 + which) * p.N + s * 64 + col] = t;
    }
  };
  auto bar = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  wait_vmcnt<0>();
  bar();
  unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = 0;        // debug (tools/sk512_trace.py): [0] entry -> ring primed, then per-phase sums over the steps
  const bool tron = p.trace != nullptr;
  if (tron) { tl = __builtin_amdgcn_s_memtime(); tr[0] = tl - tr_entry; }
  auto lap = [&](int k) { if (tron) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tr[k] += t - tl; tl = t; } };
  // SKEW BY ONE HALF: the two waves of a SIMD (w in group 0, w + 4 in group 1) must not store at the same time, and the one that does not store must have MFMAs to
  // issue meanwhile.  In interval k (one barrier per interval) group g multiplies half H = k - g; a group stores step s at the START of the interval that follows its
  // second half (group 0: interval 2s+2, group 1: 2s+3) -- while the other group's wave of the SIMD runs its half at the full rate of the matrix pipe.  Half H is read
  // in intervals H and H+1, its slot is refilled with half H+3 behind the barrier that ends interval H+1, one interval (~2 000 clocks; the weights are L2-resident)
  // before its first read.  Every interval ends with vmcnt(0) (the stores were issued at its start, the LDS-DMA behind them) + the barrier.
  // First version (both groups in lockstep, store phase behind the second half): 7 000 ticks per step against 4 096 MFMA-issue cycles (tools/sk512_trace.py).
  const int grp = wave >> 2;
  auto load_addend = [&](int s) {
    if constexpr (MODE == 2) {
      const int ncol = s * 64 + sch * 8;
#pragma unroll
      for (int it = 0; it < 4; ++it) addv[it] = *(const uint4*)((const T*)p.addend + (orow + it * 8) * p.N + ncol);
      if (gated) {
#pragma unroll
        for (int it = 0; it < 4; ++it) gbyte[it] = p.addend_mask[(orow + it * 8) * (size_t)(p.N / 8) + s * 8 + sch];
      }
    }
  };
  int hslot = 0;                                                       // slot of this group's half H = k - grp (advanced from H = 0 on)
  int islot = 2;                                                       // slot of the half issued in interval k: H = k + 1
#pragma unroll 1
  for (int k = 0; k <= NH + 1; ++k) {
    const int H = k - grp;
    if (H >= 2 && !(H & 1)) { store((H >> 1) - 1); lap(5); }
    if (k >= 1 && k + 1 < NH) issueH(k + 1, islot);
    if (H >= 0 && H < NH) {
      if (H & 1) load_addend(H >> 1);
      f32x16_t acc;
      half(smem + hslot * G::HSLOT, acc);
      stage(acc, H & 1);
      lap(1);
    }
    wait_vmcnt<0>();
    lap(2);
    bar();
    lap(3);
    if (k >= 3 && (k & 1)) finalize((k - 3) >> 1);
    if (H >= 0) { if (++hslot == 3) hslot = 0; }
    if (k >= 1) { if (++islot == 3) islot = 0; }
  }
  if (tron && lane == 0) {
    unsigned long long* t = p.trace + ((size_t)blockIdx.x * 8 + wave) * 8;          // per WAVE (tools/sk512_trace.py)
    tr[7] = __builtin_amdgcn_s_memtime() - tr_entry;
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = tr[k];
  }
}

int g_conv_sk512 = -1;      // SEGLAND_CONV_SK512=1 / sl_debug_conv_sk512(1): default OFF -- faster in isolation, equal inside the step (profiles/r5_ab_sk512.txt)
static bool sk512_shape(const ConvGemmParams& p) {
  static const bool on = getenv("SEGLAND_CONV_SK512") && getenv("SEGLAND_CONV_SK512")[0] == '1';
  if (g_conv_sk512 < 0) g_conv_sk512 = on ? 1 : 0;
  return g_conv_sk512 && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.C1 == 512 && p.C2 == 0 && p.N % 64 == 0 && p.M >= 65536 && p.M % 256 == 0 &&
         p.Hs == p.Hd && p.Ws == p.Wd && !(p.bias || p.scale || p.relu || p.mask_src || p.pre_addend || p.row_scale || p.out2 || p.gate || p.ksplit > 1) &&
         !(p.addend && p.stat_partial) && (p.addend || !p.addend_mask);
}
int launch_sk512(ConvGemmParams& p, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_gemm_sk512_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_gemm_sk512_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  p.gridM = p.M / 256; p.gridN = 1;
  p.trace = g_p8_trace;                 // debug hook shared with the half-tile kernel (sl_debug_p8_trace)
  if (p.addend) hipLaunchKernelGGL(conv_gemm_sk512_kernel<2>, dim3(p.M / 256), dim3(512), Sk5Geom::LDS, st, p);
  else hipLaunchKernelGGL(conv_gemm_sk512_kernel<1>, dim3(p.M / 256), dim3(512), Sk5Geom::LDS, st, p);
  SL_LAUNCH_CHECK("conv_gemm_sk512_kernel");
  return 0;
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_glds(ConvGemmParams& p, hipStream_t st) {
  p.gridM = cdiv(p.M, BM);
  p.gridN = p.N / BN;
  const size_t lds = EpiGeom<T, BM, BN, WM, WN>::LDS_BYTES;
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_gemm_glds_kernel<T, BM, BN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_gemm_glds_kernel<T, BM, BN, WM, WN>), dim3(p.gridM * p.gridN), dim3(64 * WM * WN), lds, st, p);
  SL_LAUNCH_CHECK("conv_gemm_glds_kernel");
  return 0;
}

// rows per block of the kernel that will run for an M-row problem (also the granularity of the BN partial statistics): 256-row tiles need enough row blocks
// to fill 256 CUs; the 4-stage ring replaces the 2-stage kernel from 16 blocks of 128 rows (Swin-T stage 3 / 4 GEMMs of 8 192 / 2 048 tokens: +4 %)
constexpr int MIN_TILES256 = 96, RING128_MIN = 16;
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

long long g_ring64_max_tiles = 160;      // 64 x 128 ring tiles when the 128 x 128 grid would have at most this many blocks; tuning hook sl_debug_ring64_max_tiles (0: never)
// Which kernel a launch runs on: 1000000 * family + 1000 * BM + BN (family 9 = pixel-stationary K = 512, 8 = 3x3 patch (+ 10000000: split-K), 7 = 64 -> 64 patch,
// 6 = pixel-stationary K <= 256, 5 = half-tile, 4 = ring, 2 = two-stage glds).  The ONE predicate chain: launch_gemm switches on it, sl_conv2d_tile_config(_ex) and
// sl_conv2d_stat_rows answer from it (round-4 advisor: the query had drifted from the dispatch).
static int choose_kernel(const ConvGemmParams& p, int dtype) {
  const bool n128 = (p.N % 128 == 0), n256 = (p.N % 256 == 0);
  const bool big = block_rows(p.M) == 256;           // tiny problems (PPM stages, prototype rows) stay on 128-row tiles
  if (dtype == SL_BF16) {
    if (p.ksplit > 1) return 18256256;                                                // planned by splitk_parts: the shape is served by the patch kernel
    if (c64k3_shape(SL_BF16, p.KH, p.KW, p.stride, p.pad, p.dil, p.C1 + p.C2, p.C1, p.N, p.M) && p.Hs == p.Hd && p.Ws == p.Wd &&
        !(p.bias || p.scale || p.relu || p.addend || p.mask_src || p.pre_addend || p.row_scale || p.out2 || p.gate))
      return 7016016;
    if (sk_shape(SL_BF16, p.KH, p.KW, p.stride, p.pad, p.C1 + p.C2, p.C1, p.N, p.M) && p.Hs == p.Hd && p.Ws == p.Wd &&
        !(p.bias || p.scale || p.relu || p.mask_src || p.pre_addend || p.row_scale || p.out2) && (p.gate || !(p.addend && p.stat_partial)) && (p.addend || !p.addend_mask) &&
        (!p.gate || (p.addend && !p.addend_mask && p.stat_partial)) && (!(p.addend_mask || p.gate) || (p.N % 128 == 0 && p.N <= 1024)))
      return 6256064;
    if (sk512_shape(p)) return 9256064;
    if (p9_shape(p)) return 8256256;
    // half-tile kernel: needs the affine row -> pixel map (forward, or data gradient of a stride-1 conv) and <= 32 taps in the mask
    if (big && n256 && (p.mode == 0 || p.stride == 1)) return 5256256;
  }
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
    if (n128 && !p.stat_partial && !p.gate && p.M >= 128LL * RING128_MIN && (long long)cdiv(p.M, 128) * (p.N / 128) <= g_ring64_max_tiles) return 4064128;
  }
  if (n128 && p.M >= 128LL * RING128_MIN) return 4128128;
  if (n128) return 2128128;
  return 2128064;
}

template <typename T>
int launch_gemm(ConvGemmParams& p, hipStream_t st) {
  const int cfg = choose_kernel(p, sizeof(T) == 2 ? SL_BF16 : SL_F32);
  if constexpr (sizeof(T) == 2) {
    switch (cfg) {
      case 18256256: case 8256256: return launch_p9(p, st);
      case 7016016: return launch_c64k3(p, st);
      case 6256064: return launch_sk(p, st);
      case 9256064: return launch_sk512(p, st);
      case 5256256: return launch_p8(p, st);
      case 4064128: return launch_ring<T, 64, 128, 2, 2, 64, 4>(p, st);
      default: break;
    }
  }
  switch (cfg) {
    case 4256256: return launch_ring<T, 256, 256, 2, 4, 64, 4>(p, st);
    case 4256128: return launch_ring<T, 256, 128, 4, 2, 64, 4>(p, st);
    case 2256064: return launch_glds<T, 256, 64, 8, 1>(p, st);
    case 4128128: return launch_ring<T, 128, 128, 2, 2, 64, 4>(p, st);
    case 2128128: return launch_glds<T, 128, 128, 2, 2>(p, st);
    case 2128064: return launch_glds<T, 128, 64, 2, 2>(p, st);
    default: break;
  }
  sl_set_error("conv: no kernel for configuration %d", cfg);
  return SL_EINVAL;
}

int g_conv_affine = -1;    // 1 (default): branch-free affine store phase for biased / folded-BN epilogues; 0: the generic one everywhere
int run_gemm(int dtype, ConvGemmParams& p, hipStream_t st) {
  if (g_conv_affine < 0) g_conv_affine = 1;
  if (!g_conv_affine) p.flags |= 2;
  const int bke = dtype == SL_BF16 ? 64 : 32;
  SL_REQUIRE(dtype == SL_BF16 || dtype == SL_F32, "conv: bad dtype %d", dtype);
  SL_REQUIRE(p.C1 > 0 && p.C1 % bke == 0 && p.C2 % bke == 0, "conv: source channels (%d,%d) must be multiples of %d", p.C1, p.C2, bke);
  SL_REQUIRE(p.N > 0 && p.N % 64 == 0, "conv: output channels %d must be a multiple of 64", p.N);
  SL_REQUIRE(p.M > 0, "conv: empty output");
  if (dtype == SL_BF16) return launch_gemm<bf16_t>(p, st);
  return launch_gemm<float>(p, st);
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

// test hook (not part of the public ABI)
extern "C" void sl_debug_conv_affine(int v) { g_conv_affine = v ? 1 : 0; }      // test hook: affine store phase on / off
extern "C" void sl_debug_ring64_max_tiles(int v) { g_ring64_max_tiles = v; }      // tuning hook: see launch_gemm
extern "C" void sl_debug_conv_sk512(int v) { g_conv_sk512 = v ? 1 : 0; }      // test / A-B hook: K = 512 pixel-stationary kernel on / off
extern "C" void sl_debug_conv_p9(int v) { g_conv_p9 = (v & 1) ? 1 : 0; }      // test hook: 3x3 patch kernel on / off
extern "C" void sl_debug_p8_trace(void* buf) { g_p8_trace = (unsigned long long*)buf; }      // test hook: [blocks][8] u64, see ConvGemmParams::trace

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

// Data gradient whose result is gated with the ReLU bits of its own positions and reduced for the BatchNorm backward of the layer below (see ConvGemmParams::gate).
// rows of stat_partial: sl_conv2d_bwd_data_bnstat_rows(d), 0 = this shape is not served (the caller runs sl_conv2d_bwd_data + sl_bn_bwd_reduce instead): served are the
// shapes that the tile kernels with the LDS-staged store phase take (half-tile, 3x3 patch, ring, two-stage) when every row block is full.
extern "C" int sl_conv2d_bwd_data_bnstat_rows(const SlConvDesc* d) {
  if (!d) return 0;
  static const bool off = getenv("SEGLAND_BN_FUSE") && getenv("SEGLAND_BN_FUSE")[0] == '0';
  if (off) return 0;
  const int cfg = sl_conv2d_tile_config_ex(d, 1, SL_EPI_GATE);
  const int fam = cfg / 1000000, bm = (cfg / 1000) % 1000;
  const long long M = (long long)d->B * d->H * d->W;
  if (!(fam == 5 || fam == 8 || fam == 4 || fam == 2) || bm <= 0 || M % bm != 0) return 0;
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

// The same across a block boundary (resnet.py:71-78 backward): the data gradient of conv1 plus the shortcut gradient `addend` IS the gradient wrt the previous block's
// output relu(bn3(c3) + res); gated with that ReLU's bits and reduced against c3 it hands the previous block its bn3 backward column sums -- its reduce pass over
// (dout, c3) disappears, and dout arrives gated.  Served: the shapes of the pixel-stationary kernel (1x1, K = 64 / 128 / 256, N % 128 == 0, N <= 1024, M % 256 == 0).
extern "C" int sl_conv2d_bwd_data_addend_bnstat_rows(const SlConvDesc* d) {
  if (!d) return 0;
  static const bool off = getenv("SEGLAND_BN_FUSE_CROSS") && getenv("SEGLAND_BN_FUSE_CROSS")[0] == '0';
  const long long M = (long long)d->B * d->H * d->W;
  if (off || d->dtype != SL_BF16 || d->H != d->Ho || d->W != d->Wo) return 0;
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

// ------------------------------------------------------------------------------------------------ weight prep
namespace {
template <typename T>
__global__ void weight_prep_kernel(const float* __restrict__ w, int Cout, int Cin, int KHW, T* wf, T* wb) {
  // one thread per (o, i, t) element of the OIHW tensor
  const long long n = (long long)Cout * Cin * KHW;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(e % KHW);
    const long long oi = e / KHW;
    const int i = (int)(oi % Cin), o = (int)(oi / Cin);
    const T v = from_f<T>(w[e]);
    if (wf) wf[((size_t)o * KHW + t) * Cin + i] = v;
    if (wb) wb[((size_t)i * KHW + t) * Cout + o] = v;
  }
}
}  // namespace

namespace {
struct PrepEntry { const float* src; void* wf; void* wb; int O, I, KHW, dtype; long long start; };

// all conv weights of a model in ONE launch: entry table in device memory.  Every conv is cut into tiles of 64 output x 32 input
// channels (all taps, <= 9).  A block stages one tile in LDS (converted), then writes
// wf[o][t][i0..i0+31] (64-byte runs) and wb[i][t][o0..o0+63] (128-byte runs): both layouts leave the block coalesced, where the
// element-wise version scattered 2-byte stores.
constexpr int WP_O = 64, WP_I = 32;
template <typename T>
__device__ __forceinline__ void weight_prep_tile(const PrepEntry& t, long long tile, unsigned char* smem, int src_I = 0, int src_off = 0,
                                                 long long f_so = 0, long long f_sk = 0, long long b_si = 0, long long b_sk = 0) {
  if (src_I == 0) src_I = t.I;                             // source input-channel count / first channel (a channel slice of a wider weight)
  // element strides of the two outputs: wf[o*f_so + k*f_sk + i], wb[i*b_si + k*b_sk + o]  (defaults: [O][KHW][I] and [I][KHW][O])
  if (f_so == 0) { f_so = (long long)t.KHW * t.I; f_sk = t.I; b_si = (long long)t.KHW * t.O; b_sk = t.O; }
  constexpr int V = 16 / (int)sizeof(T);                   // elements per 16-byte store
  const int it_n = t.I / WP_I;
  const int o0 = (int)(tile / it_n) * WP_O, i0 = (int)(tile % it_n) * WP_I;
  const int KHW = t.KHW, row = WP_I * KHW;                 // contiguous source floats per output channel of the tile
  T* lds = (T*)smem;                                       // [WP_O][row (+pad)]
  const int pitch = row + 2;
  // source rows as float4 (row = 32 * KHW is a multiple of 4), converted on the way into the LDS
  const int row4 = row >> 2;
  for (int e = threadIdx.x; e < WP_O * row4; e += blockDim.x) {
    const int o = e / row4, r = (e - o * row4) << 2;
    const float4 v = *(const float4*)(t.src + ((size_t)(o0 + o) * src_I + src_off + i0) * KHW + r);      // r = i*KHW + k
    T* d = lds + o * pitch + r;
    d[0] = from_f<T>(v.x); d[1] = from_f<T>(v.y); d[2] = from_f<T>(v.z); d[3] = from_f<T>(v.w);
  }
  __syncthreads();
  if (t.wf) {                                              // wf[o][k][i0 .. i0+31]: 16-byte stores of V consecutive input channels
    T* wf = (T*)t.wf;
    constexpr int CH = WP_I / V;
    for (int e = threadIdx.x; e < WP_O * KHW * CH; e += blockDim.x) {
      const int c = e % CH, ok = e / CH, k = ok % KHW, o = ok / KHW;
      T tmp[V];
#pragma unroll
      for (int j = 0; j < V; ++j) tmp[j] = lds[o * pitch + (c * V + j) * KHW + k];
      *(uint4*)(wf + (size_t)(o0 + o) * f_so + (size_t)k * f_sk + i0 + c * V) = *(const uint4*)tmp;
    }
  }
  if (t.wb) {                                              // wb[i][k][o0 .. o0+63]: 16-byte stores of V consecutive output channels
    T* wb = (T*)t.wb;
    constexpr int CH = WP_O / V;
    for (int e = threadIdx.x; e < WP_I * KHW * CH; e += blockDim.x) {
      const int c = e % CH, ik = e / CH, k = ik % KHW, i = ik / KHW;
      T tmp[V];
#pragma unroll
      for (int j = 0; j < V; ++j) tmp[j] = lds[(c * V + j) * pitch + i * KHW + k];
      *(uint4*)(wb + (size_t)(i0 + i) * b_si + (size_t)k * b_sk + o0 + c * V) = *(const uint4*)tmp;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void weight_prep_batched_kernel(const PrepEntry* __restrict__ tab, int n, long long total_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wp_smem[];
  for (long long tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    int lo = 0, hi = n - 1;                               // entry owning this tile: last one with start <= tile (block-uniform)
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab[mid].start <= tile) lo = mid; else hi = mid - 1; }
    const PrepEntry t = tab[lo];
    if (t.dtype == SL_BF16) weight_prep_tile<bf16_t>(t, tile - t.start, wp_smem);
    else                    weight_prep_tile<float>(t, tile - t.start, wp_smem);
  }
}
}  // namespace

// table: device array of n entries {src, w_fwd, w_bwd (void*), O, I, KH*KW, dtype (int), start (int64)} = 48 bytes each;
// start = running count of 64 x 32 channel tiles (O*I/2048) of the preceding entries, total_tiles their grand total.
extern "C" int sl_weight_prep_batched(const void* table_dev, int n, long long total_tiles, sl_stream_t stream) {
  SL_REQUIRE(table_dev && n > 0 && total_tiles > 0, "weight_prep_batched: bad args");
  static_assert(sizeof(PrepEntry) == 48, "table layout is part of the ABI");
  const size_t lds = (size_t)WP_O * (WP_I * 9 + 2) * sizeof(float);      // taps <= 9 (1x1 and 3x3 convs), fp32 worst case
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)weight_prep_batched_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  const int blocks = (int)(total_tiles < 4096 ? total_tiles : 4096);
  hipLaunchKernelGGL(weight_prep_batched_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, (const PrepEntry*)table_dev, n, total_tiles);
  SL_LAUNCH_CHECK("weight_prep_batched_kernel");
  return 0;
}

namespace {
template <typename T>
__global__ void weight_prep_slice_kernel(const float* __restrict__ w, int Cout, int CinTot, int ci_off, int ci_cnt, int KHW, T* wf, T* wb) {
  const long long n = (long long)Cout * ci_cnt * KHW;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(e % KHW);
    const long long oi = e / KHW;
    const int i = (int)(oi % ci_cnt), o = (int)(oi / ci_cnt);
    const T v = from_f<T>(w[((size_t)o * CinTot + ci_off + i) * KHW + t]);
    if (wf) wf[((size_t)o * KHW + t) * ci_cnt + i] = v;
    if (wb) wb[((size_t)i * KHW + t) * Cout + o] = v;
  }
}
}  // namespace

// Factorised PPM priors (ppm.hip): per pyramid level l the slice [l*Cs, (l+1)*Cs) of W_oihw [N][Ctot][3][3] as wq_f [l][tap*N + n][Cs] and
// wq_b [l][c][tap*N + n] (fp32): the same 64 x 32 tiles with the tap-major output strides.
namespace {
__global__ __launch_bounds__(256) void ppm_wq_prep_tiles_kernel(const float* __restrict__ w, int N, int Ctot, int Cs, float* __restrict__ wq_f, float* __restrict__ wq_b, long long tiles_per_level) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wp_smem[];
  const int l = blockIdx.y;
  PrepEntry t{w, wq_f + (size_t)l * 9 * N * Cs, wq_b + (size_t)l * Cs * 9 * N, N, Cs, 9, SL_F32, 0};
  for (long long tile = blockIdx.x; tile < tiles_per_level; tile += gridDim.x)
    weight_prep_tile<float>(t, tile, wp_smem, Ctot, l * Cs, Cs, (long long)N * Cs, 9LL * N, N);
}
}  // namespace

extern "C" int sl_ppm_wq_prep(const float* w_oihw, int N, int Ctot, int Cs, int nlevels, float* wq_f, float* wq_b, sl_stream_t stream) {
  SL_REQUIRE(w_oihw && wq_f && wq_b && N > 0 && Cs > 0 && nlevels >= 1 && nlevels * Cs <= Ctot, "ppm_wq_prep: bad args");
  SL_REQUIRE(N % WP_O == 0 && Cs % WP_I == 0, "ppm_wq_prep: N %% 64 == 0 and Cs %% 32 == 0");
  const long long tiles = (long long)(N / WP_O) * (Cs / WP_I);
  const size_t lds = (size_t)WP_O * (WP_I * 9 + 2) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)ppm_wq_prep_tiles_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  hipLaunchKernelGGL(ppm_wq_prep_tiles_kernel, dim3((unsigned)(tiles < 1024 ? tiles : 1024), nlevels), dim3(256), lds, (hipStream_t)stream, w_oihw, N, Ctot, Cs, wq_f, wq_b, tiles);
  SL_LAUNCH_CHECK("ppm_wq_prep_tiles_kernel");
  return 0;
}

// GEMM layouts of the input-channel slice [ci_off, ci_off + ci_cnt) of an OIHW weight with CinTot input channels
namespace {
__global__ __launch_bounds__(256) void weight_prep_slice_tiles_kernel(PrepEntry t, int CinTot, int ci_off, long long total_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wp_smem[];
  for (long long tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    if (t.dtype == SL_BF16) weight_prep_tile<bf16_t>(t, tile, wp_smem, CinTot, ci_off);
    else                    weight_prep_tile<float>(t, tile, wp_smem, CinTot, ci_off);
  }
}
}  // namespace

extern "C" int sl_weight_prep_slice(int dtype, const float* w_oihw, int Cout, int CinTot, int ci_off, int ci_cnt, int KH, int KW,
                                    void* w_fwd, void* w_bwd, sl_stream_t stream) {
  SL_REQUIRE(w_oihw && (w_fwd || w_bwd) && ci_off >= 0 && ci_cnt > 0 && ci_off + ci_cnt <= CinTot, "weight_prep_slice: bad args");
  if ((dtype == SL_BF16 || dtype == SL_F32) && Cout % WP_O == 0 && ci_cnt % WP_I == 0 && ci_off % 4 == 0 && KH * KW <= 9) {      // tiled, 16-byte stores
    PrepEntry t{w_oihw, w_fwd, w_bwd, Cout, ci_cnt, KH * KW, dtype, 0};
    const long long tiles = (long long)(Cout / WP_O) * (ci_cnt / WP_I);
    const size_t lds = (size_t)WP_O * (WP_I * 9 + 2) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)weight_prep_slice_tiles_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    hipLaunchKernelGGL(weight_prep_slice_tiles_kernel, dim3((unsigned)(tiles < 4096 ? tiles : 4096)), dim3(256), lds, (hipStream_t)stream, t, CinTot, ci_off, tiles);
    SL_LAUNCH_CHECK("weight_prep_slice_tiles_kernel");
    return 0;
  }
  const long long n = (long long)Cout * ci_cnt * KH * KW;
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  if (dtype == SL_BF16) hipLaunchKernelGGL(weight_prep_slice_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, CinTot, ci_off, ci_cnt, KH * KW, (bf16_t*)w_fwd, (bf16_t*)w_bwd);
  else if (dtype == SL_F32) hipLaunchKernelGGL(weight_prep_slice_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, CinTot, ci_off, ci_cnt, KH * KW, (float*)w_fwd, (float*)w_bwd);
  else SL_REQUIRE(false, "weight_prep_slice: bad dtype");
  SL_LAUNCH_CHECK("weight_prep_slice_kernel");
  return 0;
}

extern "C" int sl_weight_prep(int dtype, const float* w_oihw, int Cout, int Cin, int KH, int KW, void* w_fwd,
                              void* w_bwd, sl_stream_t stream) {
  SL_REQUIRE(w_oihw && (w_fwd || w_bwd), "weight_prep: null buffer");
  const long long n = (long long)Cout * Cin * KH * KW;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  if (dtype == SL_BF16)
    hipLaunchKernelGGL(weight_prep_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, KH * KW, (bf16_t*)w_fwd, (bf16_t*)w_bwd);
  else if (dtype == SL_F32)
    hipLaunchKernelGGL(weight_prep_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, KH * KW, (float*)w_fwd, (float*)w_bwd);
  else SL_REQUIRE(false, "weight_prep: bad dtype");
  SL_LAUNCH_CHECK("weight_prep_kernel");
  return 0;
}
