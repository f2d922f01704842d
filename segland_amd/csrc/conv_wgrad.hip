// Weight gradient of the implicit-GEMM convolution on MFMA (gfx950).
//
//   dw[n][tap][c] = sum_m dy[m][n] * x[pix(m, tap)][c]        (m = output pixels: the reduction dimension)
//
// Both operands are stored NHWC, i.e. the reduction index m is the STRIDED one.  Tiles are staged row-major
// ([m][n] and [m][c]) exactly as they stream from HBM; the MFMA operands (k-packed per lane) are produced
//   - bf16: by ds_read_b64_tr_b16 (LDS transpose read, gfx950), two per 32x32x16 fragment,
//   - f32 : by plain ds_read_b32 (v_mfma_f32_32x32x2_f32 takes one k per lane), conflict-free.
// Split over m with per-split fp32 slabs in the caller's workspace and a fixed-order reduce that also converts
// [Cout][tap][Cin] -> OIHW, so results are bit-stable run to run.
#include <stdlib.h>
#include "common.h"

namespace {

struct WgradParams {
  const void* src1; const void* src2; int C1, C2;   // x (virtual concat)
  const void* dy;
  float* ws;                                        // [splits][Cout][taps][Cin]
  int B, H, W, Ho, Wo, Cout;
  int KH, KW, stride, pad, dil;
  int M;             // B*Ho*Wo
  int rows_per_split;  // multiple of 32
  int gridN, gridC, taps, splits;
};

constexpr int KM = 32;  // reduction rows per LDS stage

template <typename T> struct WPad;                       // row padding (bytes) of the staged tiles
template <> struct WPad<bf16_t> { static constexpr int v = 64; };
template <> struct WPad<float> { static constexpr int v = 16; };

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ uint2 lds_tr16_b64(const unsigned char* p) {
  // compiler-tracked LDS transpose read (ds_read_b64_tr_b16); p is an LDS address, 8-byte aligned
  const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  return __builtin_bit_cast(uint2, v);
}

template <typename T, int BNN, int BCC, bool USE_TR>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
  constexpr int ES = sizeof(T);
  constexpr int EPC = 16 / ES;
  constexpr int PITCH_A = BNN * ES + WPad<T>::v, PITCH_B = BCC * ES + WPad<T>::v;
  constexpr int CPR_A = BNN * ES / 16, CPR_B = BCC * ES / 16;       // 16-byte chunks per row
  constexpr int NA = KM * CPR_A / 256, NB = KM * CPR_B / 256;       // chunks per thread
  constexpr int TM = BNN / 64, TN = BCC / 64;                        // 2x2 waves, 32x32 MFMA tiles per wave
  static_assert(NA >= 1 && NB >= 1, "tile too small");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* lds_a = smem;                         // [2][KM][PITCH_A]
  unsigned char* lds_b = smem + 2 * KM * PITCH_A;      // [2][KM][PITCH_B]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  int bid = blockIdx.x;
  const int bc = bid % p.gridC; bid /= p.gridC;
  const int bn = bid % p.gridN; bid /= p.gridN;
  const int tap = bid % p.taps; const int split = bid / p.taps;
  const int ky = tap / p.KW, kx = tap - ky * p.KW;

  const int CT = p.C1 + p.C2;
  const int c0 = bc * BCC;
  const T* xbase; int xpitch, xoff;
  if (c0 < p.C1) { xbase = (const T*)p.src1; xpitch = p.C1; xoff = c0; }
  else           { xbase = (const T*)p.src2; xpitch = p.C2; xoff = c0 - p.C1; }
  const T* dyb = (const T*)p.dy + bn * BNN;
  const bool ident = (p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0);

  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);
  const int nit = (m_end - m_begin + KM - 1) / KM;

  uint4 areg[NA], breg[NB];
  auto load_global = [&](int it) {
    const int m0 = m_begin + it * KM;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int id = tid + 256 * i, row = id / CPR_A, cc = id % CPR_A;
      const int m = m0 + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < m_end) v = *(const uint4*)(dyb + (size_t)m * p.Cout + cc * EPC);
      areg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int id = tid + 256 * i, row = id / CPR_B, cc = id % CPR_B;
      const int m = m0 + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < m_end) {
        if (ident) v = *(const uint4*)(xbase + (size_t)m * xpitch + xoff + cc * EPC);
        else {
          const int hw = p.Ho * p.Wo;
          const int b = m / hw, rem = m - b * hw, yo = rem / p.Wo, xo = rem - yo * p.Wo;
          const int ys = yo * p.stride - p.pad + ky * p.dil, xs = xo * p.stride - p.pad + kx * p.dil;
          if ((unsigned)ys < (unsigned)p.H && (unsigned)xs < (unsigned)p.W)
            v = *(const uint4*)(xbase + ((size_t)(b * p.H + ys) * p.W + xs) * xpitch + xoff + cc * EPC);
        }
      }
      breg[i] = v;
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int id = tid + 256 * i, row = id / CPR_A, cc = id % CPR_A;
      *(uint4*)(lds_a + (buf * KM + row) * PITCH_A + cc * 16) = areg[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int id = tid + 256 * i, row = id / CPR_B, cc = id % CPR_B;
      *(uint4*)(lds_b + (buf * KM + row) * PITCH_B + cc * 16) = breg[i];
    }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nit > 0) {
    load_global(0);
    store_lds(0);
  }
  __syncthreads();

  const int frow = lane & 31, fhalf = lane >> 5;
  for (int it = 0; it < nit; ++it) {
    const int buf = it & 1;
    if (it + 1 < nit) load_global(it + 1);
    const unsigned char* la = lds_a + buf * KM * PITCH_A;
    const unsigned char* lb = lds_b + buf * KM * PITCH_B;
    if constexpr (sizeof(T) == 4) {
      // f32: one k per lane: A[i = frow][k = fhalf]
#pragma unroll
      for (int ks = 0; ks < KM / 2; ++ks) {
        float af[TM], bf[TN];
        const int row = 2 * ks + fhalf;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *(const float*)(la + row * PITCH_A + (wm * (BNN / 2) + i * 32 + frow) * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *(const float*)(lb + row * PITCH_B + (wn * (BCC / 2) + j * 32 + frow) * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KM / 16; ++ks) {
        uint4 af[TM], bf[TN];
        if constexpr (USE_TR) {
          // 16-lane group g = lane>>4 supplies rows 16ks + 8(g>>1) + 4h + (l>>2), cols base + 16(g&1) + 4(l&3);
          // lane receives k = 8(g>>1) + 4h + 0..3 of column base + (lane&31).
          const int g = lane >> 4, l = lane & 15;
          const int rbase = 16 * ks + 8 * (g >> 1) + (l >> 2);
          const int cofs = 16 * (g & 1) + 4 * (l & 3);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const unsigned char* a0 = la + rbase * PITCH_A + (wm * (BNN / 2) + i * 32 + cofs) * 2;
            const uint2 lo = lds_tr16_b64(a0), hi = lds_tr16_b64(a0 + 4 * PITCH_A);
            af[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const unsigned char* b0 = lb + rbase * PITCH_B + (wn * (BCC / 2) + j * 32 + cofs) * 2;
            const uint2 lo = lds_tr16_b64(b0), hi = lds_tr16_b64(b0 + 4 * PITCH_B);
            bf[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
        } else {
          const int rb = 16 * ks + 8 * fhalf;
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const unsigned char* q = la + rb * PITCH_A + (wm * (BNN / 2) + i * 32 + frow) * 2;
            unsigned e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = *(const unsigned short*)(q + k * PITCH_A);
            af[i] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const unsigned char* q = lb + rb * PITCH_B + (wn * (BCC / 2) + j * 32 + frow) * 2;
            unsigned e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = *(const unsigned short*)(q + k * PITCH_B);
            bf[j] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bf[j]), acc[i][j], 0, 0, 0);
      }
    }
    if (it + 1 < nit) store_lds(buf ^ 1);
    __syncthreads();
  }

  // slab store: ws[split][n][tap][c]
  float* ws = p.ws + (size_t)split * p.Cout * p.taps * CT;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c = c0 + wn * (BCC / 2) + j * 32 + frow;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = bn * BNN + wm * (BNN / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        ws[((size_t)n * p.taps + tap) * CT + c] = acc[i][j][r];
      }
    }
}

// dw_oihw[n][c][t] = sum_s ws[s][n][t][c]
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cout, int Cin, int taps, int splits) {
  const long long total = (long long)Cout * taps * Cin;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % Cin);
    const long long nt = e / Cin;
    const int t = (int)(nt % taps), n = (int)(nt / taps);
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[(size_t)k * total + e];
    dw[((size_t)n * Cin + c) * taps + t] = s;
  }
}

int g_use_tr = -1;
int use_tr() {
  if (g_use_tr < 0) { const char* e = getenv("SEGLAND_WGRAD_TR"); g_use_tr = (e && e[0] == '0') ? 0 : 1; }
  return g_use_tr;
}

struct WgradPlan { int bnn, bcc, gridN, gridC, taps, splits, rows_per_split; size_t ws_bytes; };

WgradPlan plan(const SlConvDesc* d) {
  WgradPlan pl;
  const int c2 = d->Cin - d->C1;
  pl.bnn = d->Cout % 128 == 0 ? 128 : 64;
  pl.bcc = (d->C1 % 128 == 0 && c2 % 128 == 0) ? 128 : 64;
  pl.gridN = d->Cout / pl.bnn; pl.gridC = d->Cin / pl.bcc; pl.taps = d->KH * d->KW;
  const long long tiles = (long long)pl.gridN * pl.gridC * pl.taps;
  const long long M = (long long)d->B * d->Ho * d->Wo;
  long long splits = (1536 + tiles - 1) / tiles;
  const long long max_by_rows = (M + 4 * KM - 1) / (4 * KM);          // at least 128 rows per split
  if (splits > max_by_rows) splits = max_by_rows;
  const size_t slab = (size_t)d->Cout * pl.taps * d->Cin * sizeof(float);
  while (splits > 1 && slab * splits > ((size_t)192 << 20)) --splits;
  if (splits < 1) splits = 1;
  long long rps = (M + splits - 1) / splits;
  rps = (rps + KM - 1) / KM * KM;
  splits = (M + rps - 1) / rps;
  pl.splits = (int)splits; pl.rows_per_split = (int)rps; pl.ws_bytes = slab * splits;
  return pl;
}

template <typename T, bool TR>
int launch_wgrad(const WgradPlan& pl, WgradParams& p, hipStream_t st) {
  dim3 grid(pl.gridN * pl.gridC * pl.taps * pl.splits);
#define WG_LAUNCH(BNN, BCC)                                                                                   \
  do {                                                                                                        \
    const size_t lds = 2 * KM * ((BNN + BCC) * sizeof(T) + 2 * WPad<T>::v);                                   \
    hipLaunchKernelGGL((conv_wgrad_kernel<T, BNN, BCC, TR>), grid, dim3(256), lds, st, p);                   \
  } while (0)
  if (pl.bnn == 128 && pl.bcc == 128) WG_LAUNCH(128, 128);
  else if (pl.bnn == 128) WG_LAUNCH(128, 64);
  else if (pl.bcc == 128) WG_LAUNCH(64, 128);
  else WG_LAUNCH(64, 64);
#undef WG_LAUNCH
  SL_LAUNCH_CHECK("conv_wgrad_kernel");
  return 0;
}

}  // namespace

// test hook (not part of the public ABI): select the bf16 fragment path, 1 = ds_read_b64_tr_b16, 0 = scalar LDS reads
extern "C" void sl_debug_wgrad_tr(int v) { g_use_tr = v ? 1 : 0; }

extern "C" size_t sl_conv2d_bwd_weight_workspace(const SlConvDesc* d) {
  if (!d || d->Cout % 64 || d->Cin % 32) return 0;
  return plan(d).ws_bytes;
}

extern "C" int sl_conv2d_bwd_weight(const SlConvDesc* d, const void* x, const void* x2, const void* dy, float* dw,
                                    void* workspace, size_t workspace_bytes, sl_stream_t stream) {
  SL_REQUIRE(d && x && dy && dw && workspace, "conv bwd_weight: null buffer");
  const int bke = d->dtype == SL_BF16 ? 64 : 32;
  const int c2 = d->Cin - d->C1;
  SL_REQUIRE(d->dtype == SL_BF16 || d->dtype == SL_F32, "conv bwd_weight: bad dtype");
  SL_REQUIRE(d->Cout % 64 == 0 && d->C1 % 64 == 0 && c2 % 64 == 0, "conv bwd_weight: channels must be multiples of 64 (Cout %d, C1 %d, C2 %d)", d->Cout, d->C1, c2);
  (void)bke;
  SL_REQUIRE(c2 == 0 || x2, "conv bwd_weight: x2 missing");
  const WgradPlan pl = plan(d);
  if (workspace_bytes < pl.ws_bytes) { sl_set_error("conv bwd_weight: workspace %zu < %zu", workspace_bytes, pl.ws_bytes); return SL_EWORKSPACE; }
  WgradParams p{};
  p.src1 = x; p.src2 = x2; p.C1 = d->C1; p.C2 = c2; p.dy = dy; p.ws = (float*)workspace;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.M = d->B * d->Ho * d->Wo; p.rows_per_split = pl.rows_per_split;
  p.gridN = pl.gridN; p.gridC = pl.gridC; p.taps = pl.taps; p.splits = pl.splits;
  hipStream_t st = (hipStream_t)stream;
  int e;
  if (d->dtype == SL_F32) e = launch_wgrad<float, false>(pl, p, st);
  else if (use_tr()) e = launch_wgrad<bf16_t, true>(pl, p, st);
  else e = launch_wgrad<bf16_t, false>(pl, p, st);
  if (e) return e;
  const long long total = (long long)d->Cout * pl.taps * d->Cin;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, d->Cout, d->Cin, pl.taps, pl.splits);
  SL_LAUNCH_CHECK("wgrad_reduce_kernel");
  return 0;
}
