// Shared device/host helpers for libsegland_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/segland_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 storage

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  // hardware conversion (v_cvt_pk_bf16_f32 on gfx950): round to nearest even, NaN preserved
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

// 16-byte vector of T <-> floats
template <typename T> struct Vec16 { static constexpr int N = 16 / sizeof(T); };

template <typename T> __device__ __forceinline__ void unpack16(const uint4& v, float* out);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& v, float* out) {
  out[0] = __uint_as_float(v.x); out[1] = __uint_as_float(v.y); out[2] = __uint_as_float(v.z); out[3] = __uint_as_float(v.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& v, float* out) {
  out[0] = __uint_as_float(v.x << 16); out[1] = __uint_as_float(v.x & 0xffff0000u);
  out[2] = __uint_as_float(v.y << 16); out[3] = __uint_as_float(v.y & 0xffff0000u);
  out[4] = __uint_as_float(v.z << 16); out[5] = __uint_as_float(v.z & 0xffff0000u);
  out[6] = __uint_as_float(v.w << 16); out[7] = __uint_as_float(v.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* in);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* in) {
  return make_uint4(__float_as_uint(in[0]), __float_as_uint(in[1]), __float_as_uint(in[2]), __float_as_uint(in[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* in) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  uint4 r;     // one v_cvt_pk_bf16_f32 per pair
  r.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){in[0], in[1]}, bf16x2_t));
  r.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){in[2], in[3]}, bf16x2_t));
  r.z = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){in[4], in[5]}, bf16x2_t));
  r.w = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){in[6], in[7]}, bf16x2_t));
  return r;
}

// GELU(x) = x/2 (1 + erf(x / sqrt 2)) (torch.nn.GELU, exact form).  fp32 tensors: erff.  bf16 tensors: Abramowitz-Stegun 7.1.26 (|error of erf| < 6.1e-7 in fp32 arithmetic,
// |error of GELU| < 3.7e-7: three orders below the bf16 rounding of the stored value) -- one v_rcp, one v_exp and six FMAs instead of erff's ~40 instructions, which were half of
// the fc1 epilogue's time (113.7 -> 59 us is bias-only at 131 072 x 384; tools/gemm_time.py).
__device__ __forceinline__ float sl_erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
  const float r = fmaf(-p * t, __expf(-ax * ax), 1.f);
  return copysignf(r, x);
}
template <typename T> __device__ __forceinline__ float sl_gelu(float x) {
  if constexpr (sizeof(T) == 4) return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
  else return 0.5f * x * (1.f + sl_erf_as(x * 0.70710678118654752440f));
}

// d GELU(x) / dx = Phi(x) + x phi(x), the same erf per element type as sl_gelu (gelu_bwd_kernel and the data-gradient epilogue with a GELU pre-activation operand)
template <typename T> __device__ __forceinline__ float sl_gelu_grad(float x) {
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  float cdf;
  if constexpr (sizeof(T) == 4) cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
  else cdf = 0.5f * (1.f + sl_erf_as(x * 0.70710678118654752440f));
  return cdf + x * pdf;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- host side
void sl_set_error(const char* fmt, ...);

// Test / tuning hooks (include/segland_hip_debug.h; NOT part of the product ABI): ONE record of process-wide dispatch overrides and trace buffers, every field at its
// default in production.  Only the sl_debug_* functions (api.cpp) write it; sl_debug_reset() restores the defaults (tests/conftest.py calls it after every GPU test).
struct SlDebugState {
  int conv_affine = 1;             // branch-free affine store phase for biased / folded-BN epilogues (0: the generic one everywhere -- the bit-identity test)
  int conv_p9 = 1;                 // 3x3 patch kernel (0: the half-tile / ring kernels take the 3x3 layers)
  int conv_ring192 = 1;            // 128 x 192 ring tile for 192-multiple output widths
  int conv_ringn64 = 1;            // 128 x 64 ring tile for 64-column inference layers
  int conv_rows_small = 1;         // <= 32-row launches on conv_rows_small_kernel
  int ppm_fact_walk = 1;           // factorised PPM prior path: sliding-window scatter / gather kernels (0: the general two-stage kernels they replace)
  int conv_parity = 1;             // stride-2 3x3 data gradients as four parity-plane launches (0: one launch over all nine taps per pixel)
  int ring_small_k = 128;             // > 0: big-M layers with N % 128 == 0 and a reduction of at most this many elements run on 128 x 128 ring tiles (two blocks per CU) instead of 256 x 128
  long long ring64_max_tiles = 256;   // 64 x 128 ring tiles when the 128 x 128 grid would have at most this many blocks (0: never)
  int wgrad3 = 1;                  // nine-tap 3x3 weight gradient (0: per-tap kernels)
  int wgrad_bias = 1;              // bias-gradient column sums out of the weight-gradient kernel's dy fragments (0: blocks of the slab-reduce launch)
  int wgrad_tr = 1;                // bf16 fragments by ds_read_b64_tr_b16 (0: scalar LDS reads; the two must agree bit for bit)
  long long wgrad_pair_min_rows = 0;  // > 0: 1x1 layers with a 64- (not 128-) multiple channel count run as pixel pairs from this many rows
  int attn_valu = -1;              // 1: window attention on the VALU kernels (the MFMA kernels' reference)
  unsigned long long* p8_trace = nullptr;      // [blocks][8] u64 phase stamps (tools/p8_trace.py)
  unsigned long long* wgrad_trace = nullptr;   // [blocks][8] (tools/wgrad_trace.py)
  unsigned long long* wgrad3_trace = nullptr;  // [blocks][8]
  unsigned long long* attn_trace = nullptr;    // [blocks][16] (tools/attn_trace.py)
};
extern SlDebugState g_sl_debug;
#define SL_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      sl_set_error(__VA_ARGS__);         \
      return SL_EINVAL;                  \
    }                                    \
  } while (0)
#define SL_LAUNCH_CHECK(name)                                                \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      sl_set_error("%s: %s", name, hipGetErrorString(e__));                  \
      return (int)e__;                                                       \
    }                                                                        \
  } while (0)

__host__ __device__ static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- column sums over the rows of a [rows][C] tensor (bias gradients): block `block` sums its contiguous row chunk, four rows in flight per thread, into part[block][C]
// (fixed order).  Shared by colsum_rows_partial_kernel (pop_head.hip) and the weight-gradient slab reduce that carries the bias gradient in the same launch (conv_wgrad.hip).
template <typename T>
__device__ __forceinline__ void sl_colsum_rows_block(const T* __restrict__ x, long long rows, int C, long long rows_per_block, float* __restrict__ part, int block, float* red) {
  constexpr int V = Vec16<T>::N;
  const int nvec = C / V;
  const int tpr = nvec < 256 ? nvec : 256, rpb = 256 / tpr;
  const int tr = threadIdx.x / tpr, tc = threadIdx.x % tpr;
  const long long r0 = (long long)block * rows_per_block, r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int vc = tc; vc < nvec; vc += tpr) {
    float s[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s[k] = 0.f;
    if (tr < rpb) {
      long long r = r0 + tr;
      for (; r + 3 * rpb < r1; r += 4 * rpb) {
        uint4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = ((const uint4*)x)[(size_t)(r + u * rpb) * nvec + vc];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float v[V];
          unpack16<T>(q[u], v);
#pragma unroll
          for (int k = 0; k < V; ++k) s[k] += v[k];
        }
      }
      for (; r < r1; r += rpb) {
        float v[V];
        unpack16<T>(((const uint4*)x)[(size_t)r * nvec + vc], v);
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] += v[k];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < V; ++k) red[threadIdx.x * V + k] = s[k];
    __syncthreads();
    if (tr == 0) {
#pragma unroll
      for (int k = 0; k < V; ++k) {
        float a = 0.f;
        for (int j = 0; j < rpb; ++j) a += red[(j * tpr + tc) * V + k];
        part[(size_t)block * C + vc * V + k] = a;
      }
    }
  }
}


// rows per block of the column-sum partials (the same rule for every caller: the partial buffer has ceil(rows / chunk) rows)
inline long long sl_colsum_rows_chunk(long long rows, int C, int esize) {
  const int nvec = C * esize / 16, tpr = nvec < 256 ? nvec : 256, rpb = 256 / tpr;
  long long rpblk = (long long)rpb * 16;                         // >= four 4-deep iterations per thread
  const long long cap = 2048;                                    // partial rows the finalize sums
  if ((rows + rpblk - 1) / rpblk > cap) rpblk = ((rows + cap - 1) / cap + rpb - 1) / rpb * rpb;
  return rpblk;
}
