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
#include <math.h>
#include <stdlib.h>
#include "common.h"

// conv_wgrad3.hip
bool sl_wgrad3_eligible(const SlConvDesc* d, size_t* ws_bytes);
int sl_wgrad3_run(const SlConvDesc* d, const void* x, const void* x2, const void* dy, float* dw, int dw_cin_total, int dw_ci_off, void* workspace, size_t workspace_bytes,
                  hipStream_t st);

namespace {

struct WgradParams {
  const void* src1; const void* src2; int C1, C2;   // x (virtual concat)
  const void* dy;
  const void* dy2; int Cout1;                       // conv_wgrad_glds_kernel only: output rows [Cout1, Cout) come from a SECOND gradient tensor dy2 of pitch Cout - Cout1 (sl_conv2d_bwd_weight_dy2); null: one tensor
  float* ws;                                        // [splits][Cout][taps][Cin]
  int B, H, W, Ho, Wo, Cout;
  int KH, KW, stride, pad, dil;
  int M;             // B*Ho*Wo
  int rows_per_split;  // multiple of 32
  int gridN, gridC, taps, splits;
  float* colsum;     // BIAS instantiations: [splits * (pair ? 2 : 1)][colsum_cout] column sums of dy per split (the bias gradient partials of an nn.Linear), or null
  int colsum_cout;   // the layer's own Cout (Cout / 2 when rows are pixel pairs)
  int pair;          // 1: a row is a PAIR of consecutive pixels (64-channel layers, see plan()); Cout, C1, M are the paired sizes
  unsigned long long* trace;   // debug (tools/wgrad_trace.py, conv_wgrad_glds_kernel only): per block {s_memtime at entry, ring primed, main loop done, slab stored, HW_ID, XCC_ID, stages}; null in production
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

// ---------------------------------------------------------------------------------------------------------------
// v2: up to 256 x 256 output tiles (8 waves) and HBM -> LDS by global_load_lds.  A stage holds KM = 32 pixel rows of dy
// ([m][BNN]) and of x ([m][BCC]) unpadded; the 16-byte chunk c of stage row r is stored at position c ^ ((r&3)<<2)
// (source-side swizzle, the LDS image of each 1 KiB wave-instruction stays lane-linear).  For the bf16 transpose reads a
// 32-lane service group touches 4 rows x 2 column groups x 32 B = eight distinct 32-byte segments of one 256-byte bank
// window: conflict-free.  The f32 ds_read_b32 pattern (32 consecutive columns of one row) is conflict-free under any
// per-row chunk permutation.
template <typename T> struct WgKM { static constexpr int v = sizeof(T) == 2 ? 64 : 32; };
__device__ __attribute__((aligned(256))) unsigned char g_wzero_page[256];

typedef __attribute__((address_space(3))) void wlds_void_t;
typedef const __attribute__((address_space(1))) void wgbl_void_t;
__device__ __forceinline__ void wglds16_asm(const void* g, unsigned lds_addr) {      // see conv_gemm_common.h: glds16_asm
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned wlds_addr_of(const unsigned char* p) {
  return (unsigned)(size_t)((const __attribute__((address_space(3))) unsigned char*)p);
}
#ifndef SL_WG_NST_256
#define SL_WG_NST_256 4      // ring depth of the 256 x 256 bf16 tile (32 KiB stages; 5 = the whole 160 KiB) and of the 128 x 256 / 256 x 128 tiles (24 KiB stages; up to 6): A/B builds
#endif
#ifndef SL_WG_NST_384
#define SL_WG_NST_384 4
#endif
template <typename T, int BNN, int BCC> __host__ __device__ constexpr int wg_ring_depth() {
  return (BNN + BCC == 512 && sizeof(T) == 2) ? SL_WG_NST_256 : ((BNN + BCC == 384 && sizeof(T) == 2) ? SL_WG_NST_384 : 4);
}
template <int N> __device__ __forceinline__ void wg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// BIAS: the waves of the first c tile's first wave column also multiply their dy fragments with an all-ones x fragment -- every column of that product is the column
// sum of dy over the block's rows, i.e. this split's share of the bias gradient (p.colsum): the separate pass over dy (colsum blocks in the reduce launch: a second read
// of the layer's largest tensor) goes away.  TM extra MFMAs per TM x TN in a quarter / half of the waves of kernels that wait for HBM.
template <typename T, int BNN, int BCC, int WNN, int WCC, bool USE_TR, bool BIAS = false>
__global__ __launch_bounds__(64 * WNN * WCC) void conv_wgrad_glds_kernel(WgradParams p) {
  constexpr int ES = sizeof(T), EPC = 16 / ES, NW = WNN * WCC;
  constexpr int KM = 32;                                           // rows per stage
  constexpr int NST = wg_ring_depth<T, BNN, BCC>();               // stages in the LDS ring
  constexpr int D = NST - 1;                                       // stages issued ahead of the one being multiplied
  constexpr int RB_A = BNN * ES, RB_B = BCC * ES;                  // stage row bytes
  constexpr int RPI_A = 1024 / RB_A, RPI_B = 1024 / RB_B;          // rows per 1 KiB wave-instruction
  constexpr int IA = KM / RPI_A / NW, IB = KM / RPI_B / NW;        // LDS-DMA instructions per wave per stage
  constexpr int L = IA + IB;
  constexpr int TM = BNN / WNN / 32, TN = BCC / WCC / 32;
  constexpr int STAGE = KM * (RB_A + RB_B);
  constexpr int KS = sizeof(T) == 2 ? KM / 16 : KM / 2;            // MFMA k-steps per stage (even)
  static_assert(RB_A >= 256 && RB_B >= 256 && RB_A <= 1024 && RB_B <= 1024, "row bytes 256..1024");
  static_assert(IA >= 1 && IB >= 1 && TM >= 1 && TN >= 1 && KS % 2 == 0, "tile/wave shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WCC, wn = wave % WCC;
  unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
  if (p.trace) tr0 = __builtin_amdgcn_s_memtime();

  int bid = blockIdx.x;
  if (p.gridC * p.gridN * p.taps <= 64) {   // XCD-aware remap: consecutive logical blocks (same dy tile, neighbouring c tiles / taps) share
    // one XCD's L2.  Measured: +10% on the 512->512 / 256->256 3x3 layers, -15% on the 4096-channel PPM conv (16 c-tiles
    // per dy tile thrash one L2), hence the gate on the tile count.
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bc = bid % p.gridC; bid /= p.gridC;
  const int bn = bid % p.gridN; bid /= p.gridN;
  const int tap = bid % p.taps; const int split = bid / p.taps;
  const int ky = tap / p.KW, kx = tap - ky * p.KW;

  const int CT = p.C1 + p.C2;
  const int c0 = bc * BCC;
  const T* xbase; int xpitch, xoff;
  if (c0 < p.C1) { xbase = (const T*)p.src1; xpitch = p.C1; xoff = c0; }
  else           { xbase = (const T*)p.src2; xpitch = p.C2; xoff = c0 - p.C1; }
  // two gradient tensors side by side (block-uniform: a tile never straddles Cout1, the entry point checks)
  const bool dsec = p.dy2 && bn * BNN >= p.Cout1;
  const T* dyb = dsec ? (const T*)p.dy2 + (bn * BNN - p.Cout1) : (const T*)p.dy + bn * BNN;
  const int dpitch = p.dy2 ? (dsec ? p.Cout - p.Cout1 : p.Cout1) : p.Cout;
  const bool ident = (p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0);

  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);
  const int nit = (m_end - m_begin + KM - 1) / KM;

  const int a_rin = (lane * 16) / RB_A, a_pos = ((lane * 16) % RB_A) / 16;
  const int b_rin = (lane * 16) / RB_B, b_pos = ((lane * 16) % RB_B) / 16;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(wlds_addr_of(smem));
  const unsigned char* zsrc = g_wzero_page + (lane & 7) * 16;

  // Output-pixel coordinates of this lane's x rows for the NEXT stage to be issued, advanced incrementally (stages are
  // issued in order, KM rows apart): no integer divisions inside the loop.
  const int pxr = p.pair ? 2 : 1;             // pixels per stage row
  int qb[IB], qy[IB], qx[IB];
#pragma unroll
  for (int j = 0; j < IB; ++j) {
    const int m = (m_begin + (wave * IB + j) * RPI_B + b_rin) * pxr;
    const int hw = p.Ho * p.Wo;
    qb[j] = m / hw; const int rem = m - qb[j] * hw; qy[j] = rem / p.Wo; qx[j] = rem - qy[j] * p.Wo;
  }
  auto issue = [&](int it, int slot) {
    const int m0 = m_begin + it * KM;
    // (BIAS: the wave-uniform branch around the extra MFMAs makes the compiler's divergence analysis give up on the ring slot -- say that it is uniform)
    const unsigned la = BIAS ? (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + slot * STAGE)) : lds_base + slot * STAGE, lb = la + KM * RB_A;
#pragma unroll
    for (int j = 0; j < IA; ++j) {
      const int ins = wave * IA + j, r = ins * RPI_A + a_rin, m = m0 + r;
      const void* src = (m < m_end) ? (const void*)(dyb + (size_t)m * dpitch + (a_pos ^ ((r & 3) << 2)) * EPC) : (const void*)zsrc;
      wglds16_asm(src, la + ins * 1024);
    }
#pragma unroll
    for (int j = 0; j < IB; ++j) {
      const int ins = wave * IB + j, r = ins * RPI_B + b_rin, m = m0 + r;
      const void* src = (const void*)zsrc;
      if (m < m_end) {
        const int sc = b_pos ^ ((r & 3) << 2);                 // source chunk of this lane inside the stage row
        if (ident) src = (const void*)(xbase + (size_t)m * xpitch + xoff + sc * EPC);
        else {
          // paired rows: the first half of the row's chunks is pixel 2m, the second half pixel 2m+1 (same image row, Wo is even)
          constexpr int CPP = RB_B / 32;
          const int par = p.pair ? sc / CPP : 0;
          const int ce = p.pair ? (sc % CPP) * EPC : xoff + sc * EPC;
          const int pitch = p.pair ? xpitch / 2 : xpitch;
          const int ys = qy[j] * p.stride - p.pad + ky * p.dil, xs = (qx[j] + par) * p.stride - p.pad + kx * p.dil;
          if ((unsigned)ys < (unsigned)p.H && (unsigned)xs < (unsigned)p.W)
            src = (const void*)(xbase + ((size_t)(qb[j] * p.H + ys) * p.W + xs) * pitch + ce);
        }
      }
      wglds16_asm(src, lb + ins * 1024);
      if (!ident) {                       // advance KM rows
        qx[j] += KM * pxr;
        while (qx[j] >= p.Wo) { qx[j] -= p.Wo; ++qy[j]; }
        while (qy[j] >= p.Ho) { qy[j] -= p.Ho; ++qb[j]; }
      }
    }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool bias_on = BIAS && __builtin_amdgcn_readfirstlane((int)(p.colsum != nullptr && bc == 0 && tap == 0 && wn == 0)) != 0;      // wave-uniform (and said so)
  f32x16_t accb[BIAS ? TM : 1];
  if constexpr (BIAS) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
  }
  const uint4 ones = sizeof(T) == 2 ? make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u) : make_uint4(0x3f800000u, 0, 0, 0);
  const int frow = lane & 31, fhalf = lane >> 5;
  // fragment loaders: bf16 -> one uint4 (8 k) per 32-column tile via two transpose reads; f32 -> one float (1 k)
  auto ldfrag = [&](uint4* af, uint4* bf, int slot, int ks) {
    const unsigned char* la = smem + slot * STAGE;
    const unsigned char* lb = la + KM * RB_A;
    if constexpr (sizeof(T) == 4) {
      const int row = 2 * ks + fhalf, sw = (row & 3) << 2;
#pragma unroll
      for (int i = 0; i < TM; ++i) { const int n = wm * (BNN / WNN) + i * 32 + frow; af[i].x = *(const unsigned*)(la + row * RB_A + (((n >> 2) ^ sw) << 4) + (n & 3) * 4); }
#pragma unroll
      for (int j = 0; j < TN; ++j) { const int c = wn * (BCC / WCC) + j * 32 + frow; bf[j].x = *(const unsigned*)(lb + row * RB_B + (((c >> 2) ^ sw) << 4) + (c & 3) * 4); }
    } else if constexpr (USE_TR) {
      const int g = lane >> 4, l = lane & 15;
      const int r = 16 * ks + 8 * (g >> 1) + (l >> 2);
      const int sw = (l >> 2) << 2;
      const int cofs = 16 * (g & 1) + 4 * (l & 3);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int e = wm * (BNN / WNN) + i * 32 + cofs;
        const unsigned char* a0 = la + r * RB_A + (((e >> 3) ^ sw) << 4) + (e & 7) * 2;
        const uint2 lo = lds_tr16_b64(a0), hi = lds_tr16_b64(a0 + 4 * RB_A);
        af[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int e = wn * (BCC / WCC) + j * 32 + cofs;
        const unsigned char* b0 = lb + r * RB_B + (((e >> 3) ^ sw) << 4) + (e & 7) * 2;
        const uint2 lo = lds_tr16_b64(b0), hi = lds_tr16_b64(b0 + 4 * RB_B);
        bf[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    } else {
      const int rb = 16 * ks + 8 * fhalf;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int n = wm * (BNN / WNN) + i * 32 + frow;
        unsigned e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = *(const unsigned short*)(la + (rb + k) * RB_A + (((n >> 3) ^ (((rb + k) & 3) << 2)) << 4) + (n & 7) * 2);
        af[i] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int c = wn * (BCC / WCC) + j * 32 + frow;
        unsigned e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = *(const unsigned short*)(lb + (rb + k) * RB_B + (((c >> 3) ^ (((rb + k) & 3) << 2)) << 4) + (c & 7) * 2);
        bf[j] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
      }
    }
  };
  auto mma = [&](const uint4* af, const uint4* bf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (sizeof(T) == 4) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(af[i].x), __uint_as_float(bf[j].x), acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bf[j]), acc[i][j], 0, 0, 0);
      }
    if constexpr (BIAS) {
      if (bias_on) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if constexpr (sizeof(T) == 4) accb[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(af[i].x), __uint_as_float(ones.x), accb[i], 0, 0, 0);
          else accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, ones), accb[i], 0, 0, 0);
        }
      }
    }
  };

  // same pipeline as conv_gemm_ring_kernel: stages i, i+1 complete at the top of iteration i, fragments of the next
  // k-step (reaching into stage i+1) are loaded before the MFMAs of the current one, LDS-DMA of i+2 / i+3 in flight.
  uint4 afA[TM], bfA[TN], afB[TM], bfB[TN];
#pragma unroll
  for (int st = 0; st < D; ++st)
    if (st < nit) issue(st, st);
  // stages 0 and 1 complete; later ones may stay in flight.  fl = issued stages that need not have landed yet (block-uniform)
  auto wait_fl = [&](int fl) {
    if constexpr (D >= 4 && 3 * L <= 63) { if (fl >= 3) { wg_wait_vmcnt<3 * L>(); return; } }
    if constexpr (D >= 3 && 2 * L <= 63) { if (fl >= 2) { wg_wait_vmcnt<2 * L>(); return; } }
    if (fl >= 1) wg_wait_vmcnt<L>(); else wg_wait_vmcnt<0>();
  };
#ifndef SL_WG_LATE
#define SL_WG_LATE 0       // 1: a stage is waited for where its first fragments are read (the ring kernel's round-6 schedule, conv_gemm_tiles.hip SL_RING_LATE).  Measured NEGATIVE here: ResNet-50 256 x 256 tiles 1.15 -> 1.23 ms per step, Swin-T neutral (profiles/r6_ab_wg_late.txt) -- these launches are fill-bound, and the barrier in front of the last k-step holds the MFMA queue back
#endif
  wait_fl(min(nit, D) - (SL_WG_LATE ? 1 : 2));
  __builtin_amdgcn_s_barrier();
  if (p.trace) tr1 = __builtin_amdgcn_s_memtime();
  ldfrag(afA, bfA, 0, 0);
  int slot = 0;
  // the two waves of a SIMD (w, w + NW/2 in an 8-wave block) place their address/issue section one MFMA cluster apart, so one
  // wave's MFMAs cover the other's VALU + LDS-DMA issue (measured on the conv kernel: +1-3 %)
  const bool late = NW == 8 && wave >= 4;
#ifndef SL_WG_ABL
#define SL_WG_ABL 0        // ablation builds (tools/wgrad_ablation.sh; never in the product): bit 0 = no LDS-DMA in the loop, bit 1 = no fragment reads in the loop
#endif
  if constexpr ((SL_WG_ABL & 2) != 0) ldfrag(afB, bfB, 0, 1);
  for (int it = 0; it < nit; ++it) {
    int ns3 = slot + D; if (ns3 >= NST) ns3 -= NST;
    if (!(SL_WG_ABL & 1) && !late && it + D < nit) issue(it + D, ns3);
    int nslot = slot + 1; if (nslot == NST) nslot = 0;
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      if constexpr (!(SL_WG_ABL & 2)) ldfrag(afB, bfB, slot, ks + 1);
      mma(afA, bfA);
      if (!(SL_WG_ABL & 1) && ks == 0 && late && it + D < nit) issue(it + D, ns3);
      if constexpr (!(SL_WG_ABL & 2)) {
        if (ks + 2 < KS) ldfrag(afA, bfA, slot, ks + 2);
        else if (SL_WG_LATE) {
          if (it + 1 < nit) {
            wait_fl(min(it + D, nit - 1) - (it + 1));          // stage it + 1 landed; it + 2 .. it + D may stay in flight
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's fragment reads of the slot are done: behind the barrier it may be refilled
            __builtin_amdgcn_s_barrier();
            ldfrag(afA, bfA, nslot, 0);
          }
        } else ldfrag(afA, bfA, nslot, 0);
      }
      mma(afB, bfB);
    }
    if (!SL_WG_LATE) {
      wait_fl(min(it + D, nit - 1) - (it + 2));                // stage it + 2 complete before anyone starts iteration it + 1; it + 3 .. it + D may stay in flight
      __builtin_amdgcn_s_barrier();
    }
    slot = nslot;
  }
  if (p.trace) tr2 = __builtin_amdgcn_s_memtime();

  float* ws = p.ws + (size_t)split * p.Cout * p.taps * CT;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c = c0 + wn * (BCC / WCC) + j * 32 + frow;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = bn * BNN + wm * (BNN / WNN) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        ws[((size_t)n * p.taps + tap) * CT + c] = acc[i][j][r];
      }
    }
  if constexpr (BIAS) {
    if (bias_on && frow == 0) {              // column 0 of the product (lanes 0 and 32): rows (r & 3) + 8 (r >> 2) + 4 fhalf
      const int pr = p.pair ? 2 : 1;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = bn * BNN + wm * (BNN / WNN) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
          const int par = n >= p.colsum_cout ? 1 : 0;          // pixel pairs: the second half of the paired row's channels is the odd pixel
          p.colsum[((size_t)split * pr + par) * p.colsum_cout + (n - par * p.colsum_cout)] = accb[i][r];
        }
    }
  }
  if (p.trace && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the stamp counts the slab stores of this wave as issued AND accepted
    unsigned long long* t = p.trace + (size_t)blockIdx.x * 8;
    t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = __builtin_amdgcn_s_memtime();
    t[4] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); t[5] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
    t[6] = (unsigned long long)nit; t[7] = (unsigned long long)bid;
  }
}

// dw_oihw[n][c][t] = sum_s ws[s][n][t][c].  One block per (n, 64-channel chunk): the taps x 64 slab values are read as
// 256-byte rows (coalesced), transposed through LDS and written as ONE contiguous run of 64*taps floats.
// n_valid / c_valid (all reduce kernels): output / input channels that exist in dw -- a gradient computed at zero-padded channel counts (Swin: 96 -> 128) lands in the
// parameter's own shape [n_valid][dw_cin_total][taps] without a slicing copy behind it.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cout, int Cin, int taps, int splits,
                                                           int dw_cin_total, int dw_ci_off, int n_valid, int c_valid) {
  __shared__ float tile[64 * 49];      // [c][t], taps <= 49
  const int cchunks = Cin / 64;
  const int n = blockIdx.x / cchunks, c0 = (blockIdx.x % cchunks) * 64;
  if (n >= n_valid || c0 >= c_valid) return;
  const size_t slab = (size_t)Cout * taps * Cin;
  for (int e = threadIdx.x; e < taps * 64; e += 256) {
    const int t = e / 64, c = e % 64;
    const size_t off = ((size_t)n * taps + t) * Cin + c0 + c;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[(size_t)k * slab + off];
    tile[c * taps + t] = s;
  }
  __syncthreads();
  float* o = dw + ((size_t)n * dw_cin_total + dw_ci_off + c0) * taps;
  const int lim = taps * (c_valid - c0 < 64 ? c_valid - c0 : 64);
  for (int e = threadIdx.x; e < lim; e += 256) o[e] = tile[e];
}

// taps == 1: the slab layout already equals OIHW
__global__ void wgrad_reduce_flat_kernel(const float* __restrict__ ws, float* __restrict__ dw, long long total, int splits, int Cin, int dw_cin_total, int dw_ci_off, int n_valid, int c_valid) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < splits; k += 4) {
      s0 += ws[(size_t)k * total + e]; s1 += ws[(size_t)(k + 1) * total + e];
      s2 += ws[(size_t)(k + 2) * total + e]; s3 += ws[(size_t)(k + 3) * total + e];
    }
    for (; k < splits; ++k) s0 += ws[(size_t)k * total + e];
    if (e / Cin < n_valid && e % Cin < c_valid) dw[(e / Cin) * dw_cin_total + dw_ci_off + (e % Cin)] = (s0 + s1) + (s2 + s3);
  }
}

// The flat slab reduce and the column sums of dy (the bias gradient of the same nn.Linear) in ONE launch: blocks [0, nred) reduce the slabs, blocks [nred, nred + ncol) each
// sum one row chunk of dy into colsum_part -- two ~6 us launches per linear and step otherwise (96 of the 767 launches of a Swin-T step).
template <typename T>
__global__ __launch_bounds__(256) void wgrad_reduce_flat_colsum_kernel(const float* __restrict__ ws, float* __restrict__ dw, long long total, int splits, int Cin, int dw_cin_total,
                                                                       int dw_ci_off, int nred, const T* __restrict__ dy, long long rows, int Cout, long long rows_per_block,
                                                                       float* __restrict__ colsum_part, int n_valid, int c_valid) {
  __shared__ float red[256 * Vec16<T>::N];
  if ((int)blockIdx.x >= nred) { sl_colsum_rows_block<T>(dy, rows, Cout, rows_per_block, colsum_part, (int)blockIdx.x - nred, red); return; }
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)nred * blockDim.x) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < splits; k += 4) {
      s0 += ws[(size_t)k * total + e]; s1 += ws[(size_t)(k + 1) * total + e];
      s2 += ws[(size_t)(k + 2) * total + e]; s3 += ws[(size_t)(k + 3) * total + e];
    }
    for (; k < splits; ++k) s0 += ws[(size_t)k * total + e];
    if (e / Cin < n_valid && e % Cin < c_valid) dw[(e / Cin) * dw_cin_total + dw_ci_off + (e % Cin)] = (s0 + s1) + (s2 + s3);
  }
}

// The flat slab reduces (+ bias column sums) of up to SL_WGRAD_BATCH_MAX layers in ONE launch (sl_wgrad_reduce_multi): a transformer block has four nn.Linear weight
// gradients whose reduces are ~5 us launches each.  Block ranges per item: [start[i], start[i] + nred_i) reduce, then ncol_i column-sum blocks; per element the sums run
// in the order of the single-layer kernels (same bits).
struct WgradReduceBatch { SlWgradReduce it[SL_WGRAD_BATCH_MAX]; int start[SL_WGRAD_BATCH_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const WgradReduceBatch b) {
  __shared__ float red[256 * 8];
  int i = 0;
  while (i + 1 < b.n && (int)blockIdx.x >= b.start[i + 1]) ++i;
  const SlWgradReduce& t = b.it[i];
  const int blk = (int)blockIdx.x - b.start[i];
  const int nred = (int)((t.total + 255) / 256);
  if (blk >= nred) {
    if (t.dtype == SL_BF16) sl_colsum_rows_block<bf16_t>((const bf16_t*)t.dy, t.rows, t.Cout, t.rows_per_block, t.colsum_part, blk - nred, red);
    else sl_colsum_rows_block<float>((const float*)t.dy, t.rows, t.Cout, t.rows_per_block, t.colsum_part, blk - nred, red);
    return;
  }
  const long long e = blk * 256LL + threadIdx.x;
  if (e >= t.total) return;
  const float* __restrict__ ws = t.ws;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 3 < t.splits; k += 4) {
    s0 += ws[(size_t)k * t.total + e]; s1 += ws[(size_t)(k + 1) * t.total + e];
    s2 += ws[(size_t)(k + 2) * t.total + e]; s3 += ws[(size_t)(k + 3) * t.total + e];
  }
  for (; k < t.splits; ++k) s0 += ws[(size_t)k * t.total + e];
  if (e / t.Cin < t.n_valid && e % t.Cin < t.c_valid) t.dw[(e / t.Cin) * t.dw_cin_total + t.dw_ci_off + (e % t.Cin)] = (s0 + s1) + (s2 + s3);
}

// paired rows (plan(): 64-channel layers): slab = [2*Cout][taps][2*Cin]; the weight gradient is the sum of the two parity-diagonal blocks
__global__ void wgrad_reduce_pair_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cout, int Cin, int taps, int splits, int dw_cin_total, int dw_ci_off, int n_valid, int c_valid) {
  const long long total = (long long)Cout * Cin * taps, slab = 4 * total;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(e % taps), c = (int)((e / taps) % Cin), n = (int)(e / ((long long)taps * Cin));
    const size_t o0 = ((size_t)n * taps + t) * (2 * Cin) + c, o1 = ((size_t)(Cout + n) * taps + t) * (2 * Cin) + Cin + c;
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < splits; ++k) { s0 += ws[(size_t)k * slab + o0]; s1 += ws[(size_t)k * slab + o1]; }
    if (n < n_valid && c < c_valid) dw[((size_t)n * dw_cin_total + dw_ci_off + c) * taps + t] = s0 + s1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 64 -> 64 channels, 3x3, stride 1, dilation 1 (layer1.conv2 at 128 x 128: the layers the review named at 116-157 TFLOP/s).  The generic kernels
// give every tap its own blocks, which re-read dy and x nine times through the LDS (the launch is LDS-fill bound, and the pixel-pair trick that
// makes 64 channels fit the 128-wide tiles discards half of its MFMAs).  Here a block owns 16 x 16-pixel tiles: the dy tile and the x patch WITH
// its one-pixel halo go to the LDS once, and all nine taps read their shifted pixel rows from that patch (ds_read_b64_tr_b16, the reduction index
// is the pixel).  Wave w accumulates taps {w, w+4, w+8}; persistent blocks; one [64][9][64] slab per block, summed by the fixed-order
// column-sum kernel and turned into OIHW by the slab-reduce kernel (a slab written directly in OIHW order was 4-byte stores 36 B apart: 3x slower).  Row pitch 192 B: four pixel rows x 64 B tile the 64 banks (conflict-free transpose reads).
constexpr int C3_T = 16;                       // tile edge
constexpr int C3_PITCH = 192;                  // bytes per pixel row in the LDS (128 B of channels + 64 B pad)
constexpr int C3_PW = C3_T + 2;                // patch edge
__global__ __launch_bounds__(256) void conv_wgrad_c64k3_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ ws,
                                                               int B, int H, int W, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* lx = smem;                                  // [18*18][192 B]
  unsigned char* ld = smem + C3_PW * C3_PW * C3_PITCH;       // [256][192 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx = cdiv(W, C3_T), ty = cdiv(H, C3_T);
  f32x16_t acc[3][2][2];                                     // [tap slot][n block][c block]
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;
  // transpose-read geometry (see conv_wgrad_kernel): 16-lane group g supplies rows 8(g>>1) + (l>>2) (+4 for the second read), columns 16(g&1) + 4(l&3)
  const int g = lane >> 4, l = lane & 15;
  const int rsub = 8 * (g >> 1) + (l >> 2), cofs = (16 * (g & 1) + 4 * (l & 3)) * 2;
  // the tile's 4 640 16-byte chunks (x patch with halo: 324 pixels, dy: 256 pixels, 8 chunks each) travel global -> registers -> LDS; the loads of
  // the NEXT tile are issued before the MFMA loop of the current one and land while it runs
  constexpr int NCH = (C3_PW * C3_PW + 256) * 8, CPT = (NCH + 255) / 256;        // 19 chunks per thread
  uint4 stage[CPT];
  auto fetch = [&](int tile) {
    int blk = tile;
    const int bx = blk % tx; blk /= tx;
    const int by = blk % ty; const int b = blk / ty;
    const int y0 = by * C3_T, x0 = bx * C3_T;
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      const int e = tid + u * 256;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (e < C3_PW * C3_PW * 8) {
        const int ch8 = e & 7, pp = e >> 3, py = pp / C3_PW, px = pp - py * C3_PW;
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = *(const uint4*)(x + ((size_t)(b * H + iy) * W + ix) * 64 + ch8 * 8);
      } else if (e < NCH) {
        const int e2 = e - C3_PW * C3_PW * 8, ch8 = e2 & 7, pp = e2 >> 3;
        const int iy = y0 + (pp >> 4), ix = x0 + (pp & 15);
        if (iy < H && ix < W) v = *(const uint4*)(dy + ((size_t)(b * H + iy) * W + ix) * 64 + ch8 * 8);
      }
      stage[u] = v;
    }
  };
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      const int e = tid + u * 256;
      if (e < C3_PW * C3_PW * 8) *(uint4*)(lx + (e >> 3) * C3_PITCH + (e & 7) * 16) = stage[u];
      else if (e < NCH) { const int e2 = e - C3_PW * C3_PW * 8; *(uint4*)(ld + (e2 >> 3) * C3_PITCH + (e2 & 7) * 16) = stage[u]; }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
#pragma unroll 2
    for (int ks = 0; ks < 16; ++ks) {                        // k-step = one tile row of 16 pixels
      uint4 af[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned char* a0 = ld + (ks * 16 + rsub) * C3_PITCH + i * 64 + cofs;
        const uint2 lo = lds_tr16_b64(a0), hi = lds_tr16_b64(a0 + 4 * C3_PITCH);
        af[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int tap = wave + 4 * t;
        if (tap < 9) {                                        // wave-uniform
          const int ky = tap / 3, kx = tap - 3 * ky;
          const unsigned char* rowb = lx + ((ks + ky) * C3_PW + kx + rsub) * C3_PITCH + cofs;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const uint2 lo = lds_tr16_b64(rowb + j * 64), hi = lds_tr16_b64(rowb + j * 64 + 4 * C3_PITCH);
            const uint4 bf = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bf), acc[t][i][j], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }
  // D layout: lane & 31 = input channel of the block (second operand row), register r = output channel (r&3) + 8(r>>2) + 4(lane>>5)
  float* out = ws + (size_t)blockIdx.x * 64 * 64 * 9;
  const int frow = lane & 31, fhalf = lane >> 5;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int tap = wave + 4 * t;
    if (tap < 9) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf, c = j * 32 + frow;
            out[((size_t)n * 9 + tap) * 64 + c] = acc[t][i][j][r];       // slab layout [n][tap][c]: the lanes of a store are 32 consecutive c
          }
    }
  }
}
inline bool c64k3_eligible(const SlConvDesc* d, int dw_cin_total, int dw_ci_off) {
  return d->dtype == SL_BF16 && d->Cin == 64 && d->Cout == 64 && d->C1 == 64 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->dil == 1 && d->pad == 1 &&
         dw_cin_total == 64 && dw_ci_off == 0 && (long long)d->B * d->H * d->W >= 65536;
}
inline int c64k3_blocks(const SlConvDesc* d) { const int t = d->B * cdiv(d->H, C3_T) * cdiv(d->W, C3_T); return t < 256 ? t : 256; }

// ---------------------------------------------------------------------------------------------------------------
// 1x1 layers with a 64-channel side at >= 65 536 pixels (layer1: 64 -> 256, 256 -> 64, 64 -> 64 at 128 x 128): dw [Cout][Cin] is at most 16 K numbers and the
// launch is bound by reading x and dy ONCE (168 MB for 64 <-> 256; 34 us at 5 TB/s).  The pixel-pair path above reaches 2.3 TB/s (half of its MFMAs and of its LDS
// traffic are discarded).  Here a persistent block walks over 128-pixel tiles: both tiles go global -> registers -> LDS (the next tile's loads are in flight during
// the MFMA loop), fragments by ds_read_b64_tr_b16 (the reduction index is the pixel), the whole [Cout][Cin] result stays in the accumulators of the block's four
// waves (64 x 64 each; 64 -> 64: the four waves split the pixels instead), ONE slab per block (per wave for 64 -> 64), summed in a fixed order by
// wgrad_reduce_flat_kernel.  D layout: lane & 31 = input channel, register = output channel: the lanes of a slab store are 32 consecutive input channels.
constexpr int CP_T = 128;                                                // pixels per tile (two tiles in flight per block: 2 x 20 or 2 x 8 register chunks per thread)
template <int CA, int CB>                                                // CA = Cout (dy channels), CB = Cin (x channels)
__global__ __launch_bounds__(256) void conv_wgrad_c64p_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ ws, int ntiles) {
  constexpr int PA = CA * 2 + 64, PB = CB * 2 + 64;                       // LDS row pitches: 64 B of pad keep four pixel rows on distinct 64-byte bank groups (192 and 576 B)
  constexpr int CHA = CA / 8, CHB = CB / 8;                               // 16-byte chunks per pixel row
  constexpr int NCH = CP_T * (CHA + CHB), CPT = NCH / 256;                // chunks per tile / per thread (8, 16 or 20)
  constexpr bool KSPLIT = CA == 64 && CB == 64;
  constexpr int CPT_A = CP_T * CHA / 256;                                 // the first CPT_A chunks of a thread are dy, the rest x
  static_assert(NCH % 256 == 0 && (CP_T * CHA) % 256 == 0, "tile chunks divide over the block, operand by operand");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* la = smem;                                               // dy tile [128][PA]
  unsigned char* lb = smem + CP_T * PA;                                   // x tile  [128][PB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NB64 = CB / 64;
  static_assert((CA / 64) * NB64 == 4 || (CA == 64 && CB == 64), "four 64 x 64 blocks, one per wave");
  const int ia = KSPLIT ? 0 : wave / NB64, jb = KSPLIT ? 0 : wave % NB64;  // this wave's 64 x 64 block of the gradient (256 x 64, 64 x 256 or 128 x 128)
  f32x16_t acc[2][2];                                                     // [output-channel 32-block][input-channel 32-block]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int g = lane >> 4, l = lane & 15;
  const int rsub = 8 * (g >> 1) + (l >> 2), cofs = (16 * (g & 1) + 4 * (l & 3)) * 2;      // transpose-read geometry, see conv_wgrad_c64k3_kernel
  // TWO tiles in flight: a tile's loads are issued two tiles ahead of its use (global -> registers), so a block keeps 2 x (CA + CB) x 256 B outstanding while it
  // multiplies -- with one tile ahead the short MFMA loop left most of the memory latency exposed (76 us against 50 us for the pixel-pair path).
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;      // NOT HIP's uint4: arrays of that struct stayed in scratch memory here (ScratchSize 656 B/lane)
  u32x4_t st0[CPT], st1[CPT];
  // (every fetch is UNCONDITIONAL -- past the end it re-reads the last tile: with loads under a branch the compiler can no longer count them and falls back to
  // vmcnt(0) in front of every LDS write, which also waits for the tile that was just requested.  The two register stages are named, not passed by reference:
  // an array handed to a lambda as a parameter lands in scratch memory.)
  auto fetch0 = [&](int t_) {
    const size_t m0 = (size_t)(t_ < ntiles ? t_ : ntiles - 1) * CP_T;
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      if (u < CPT_A) { const int e = tid + u * 256; st0[u] = *(const u32x4_t*)(dy + (m0 + e / CHA) * CA + (e % CHA) * 8); }          // compile-time split: no per-thread branch
      else { const int e2 = tid + (u - CPT_A) * 256; st0[u] = *(const u32x4_t*)(x + (m0 + e2 / CHB) * CB + (e2 % CHB) * 8); }
    }
  };
  auto fetch1 = [&](int t_) {
    const size_t m0 = (size_t)(t_ < ntiles ? t_ : ntiles - 1) * CP_T;
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      if (u < CPT_A) { const int e = tid + u * 256; st1[u] = *(const u32x4_t*)(dy + (m0 + e / CHA) * CA + (e % CHA) * 8); }          // compile-time split: no per-thread branch
      else { const int e2 = tid + (u - CPT_A) * 256; st1[u] = *(const u32x4_t*)(x + (m0 + e2 / CHB) * CB + (e2 % CHB) * 8); }
    }
  };
  auto multiply = [&]() {
    constexpr int KS0 = CP_T / 16 / (KSPLIT ? 4 : 1);                    // k-steps of 16 pixels per wave
#pragma unroll
    for (int k = 0; k < KS0; ++k) {
      const int ks = KSPLIT ? wave * KS0 + k : k;
      uint4 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned char* a0 = la + (ks * 16 + rsub) * PA + ia * 128 + i * 64 + cofs;
        const uint2 lo = lds_tr16_b64(a0), hi = lds_tr16_b64(a0 + 4 * PA);
        af[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned char* b0 = lb + (ks * 16 + rsub) * PB + jb * 128 + j * 64 + cofs;
        const uint2 lo = lds_tr16_b64(b0), hi = lds_tr16_b64(b0 + 4 * PB);
        bfr[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
    }
  };
  // a step: the stage holds the tile; after the hand-off to the LDS it is refilled with the tile two grid strides ahead
  const int G = gridDim.x;
  int tile = blockIdx.x;
  fetch0(tile);
  fetch1(tile + G);
  for (; tile + G < ntiles; tile += 2 * G) {
    _Pragma("unroll") for (int u = 0; u < CPT; ++u) {
      if (u < CPT_A) { const int e = tid + u * 256; *(u32x4_t*)(la + (e / CHA) * PA + (e % CHA) * 16) = st0[u]; }
      else { const int e2 = tid + (u - CPT_A) * 256; *(u32x4_t*)(lb + (e2 / CHB) * PB + (e2 % CHB) * 16) = st0[u]; }
    }
    __syncthreads(); fetch0(tile + 2 * G); multiply(); __syncthreads();
    _Pragma("unroll") for (int u = 0; u < CPT; ++u) {
      if (u < CPT_A) { const int e = tid + u * 256; *(u32x4_t*)(la + (e / CHA) * PA + (e % CHA) * 16) = st1[u]; }
      else { const int e2 = tid + (u - CPT_A) * 256; *(u32x4_t*)(lb + (e2 / CHB) * PB + (e2 % CHB) * 16) = st1[u]; }
    }
    __syncthreads(); fetch1(tile + 3 * G); multiply(); __syncthreads();
  }
  if (tile < ntiles) {                                                    // odd number of tiles for this block
    _Pragma("unroll") for (int u = 0; u < CPT; ++u) {
      if (u < CPT_A) { const int e = tid + u * 256; *(u32x4_t*)(la + (e / CHA) * PA + (e % CHA) * 16) = st0[u]; }
      else { const int e2 = tid + (u - CPT_A) * 256; *(u32x4_t*)(lb + (e2 / CHB) * PB + (e2 % CHB) * 16) = st0[u]; }
    }
    __syncthreads(); multiply(); __syncthreads();
  }
  float* out = ws + (size_t)(KSPLIT ? blockIdx.x * 4 + wave : blockIdx.x) * CA * CB;
  const int frow = lane & 31, fhalf = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = ia * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf, c = jb * 64 + j * 32 + frow;
        out[(size_t)n * CB + c] = acc[i][j][r];
      }
}
inline bool c64p_eligible(const SlConvDesc* d) {
  const long long M = (long long)d->B * d->H * d->W;
  const bool shape = (d->Cout == 256 && d->Cin == 64) || (d->Cout == 64 && d->Cin == 256) || (d->Cout == 64 && d->Cin == 64) || (d->Cout == 128 && d->Cin == 128);
  return d->dtype == SL_BF16 && shape && d->C1 == d->Cin && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && M >= 65536 && M % CP_T == 0;
}
inline int c64p_blocks(const SlConvDesc* d) { const long long t = (long long)d->B * d->H * d->W / CP_T; return t < 256 ? (int)t : 256; }
inline int c64p_slabs(const SlConvDesc* d) { return c64p_blocks(d) * ((d->Cout == 64 && d->Cin == 64) ? 4 : 1); }

int use_tr() { return g_sl_debug.wgrad_tr != 0; }      // bf16 fragments by ds_read_b64_tr_b16 (default) or scalar LDS reads (test hook sl_debug_wgrad_tr: the two must agree bit for bit)

struct WgradPlan { int bnn, bcc, gridN, gridC, taps, splits, rows_per_split; bool glds, pair; size_t ws_bytes; };

WgradPlan plan_shape(const SlConvDesc* d, long long M);

// 64-channel layers (layer1, the stem GEMM): the 128-wide glds tiles do not fit them and the 64-wide register-staged kernel runs at
// ~75 TFLOP/s.  Reading two consecutive pixels as ONE row doubles both channel counts ([M][64] is bit-identical to [M/2][128]):
// the paired problem runs on the glds kernel and the true gradient is the sum of the two parity-diagonal blocks of its result
// (half of the MFMA work is discarded -- still 2-3x faster).  3x3 layers gather the pair per lane, so they need Cin == 64 (one tile).
// From how many rows a 1x1 layer with a 64-multiple (not 128-multiple) channel count runs as pixel pairs.  64-channel layers: 2^19 (round 1: shorter ones are faster on the
// 64-wide register-staged kernel; ResNet's layer1 has kernels of its own anyway).  Layers with >= 192 channels on both sides -- Swin stage 2: 192 <-> 576 / 768 on 32 768
// tokens -- from 16 384 rows (round 4, tools/gemm_time.py: 192 -> 576 66.2 -> 37.4 us, 192 -> 768 53.1 -> 38.3, 768 -> 192 54.3 -> 39.5, 192 -> 192 27.8 -> 24.8;
// Swin-T POP 717.3 -> 733.7 tiles/s on one box).  The tuning hook sl_debug_wgrad_pair_min (> 0) overrides both.
static long long pair_min_rows(const SlConvDesc* d) {
  if (g_sl_debug.wgrad_pair_min_rows > 0) return g_sl_debug.wgrad_pair_min_rows;
  return (d->Cin >= 192 && d->Cout >= 192) ? 16384 : (1 << 19);
}
WgradPlan plan(const SlConvDesc* d) {
  const int c2 = d->Cin - d->C1;
  const bool all128 = d->Cout % 128 == 0 && d->C1 % 128 == 0 && c2 % 128 == 0;
  const bool ident = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0;
  const long long M = (long long)d->B * d->Ho * d->Wo;
  if (!all128 && c2 == 0 && d->dtype == SL_BF16 && M % 2 == 0 && d->Wo % 2 == 0 && d->Cout % 64 == 0 && d->Cin % 64 == 0 &&
      ((ident && M >= pair_min_rows(d)) || (!ident && d->Cin == 64 && d->stride == 1))) {
    SlConvDesc d2 = *d;
    d2.Cout = 2 * d->Cout; d2.Cin = d2.C1 = 2 * d->Cin;
    WgradPlan pl = plan_shape(&d2, M / 2);
    pl.pair = true;
    return pl;
  }
  WgradPlan pl = plan_shape(d, M);
  pl.pair = false;
  return pl;
}

WgradPlan plan_shape(const SlConvDesc* d, long long M) {
  WgradPlan pl;
  const int c2 = d->Cin - d->C1;
  const bool all128 = d->Cout % 128 == 0 && d->C1 % 128 == 0 && c2 % 128 == 0;
  pl.glds = all128;                               // 128- / 256-wide glds tiles; 64-channel sides run on the register-staged kernel
  if (pl.glds) {
    const bool wide = d->dtype == SL_BF16;          // f32 stages are twice as large: 128-wide tiles keep the ring in 128 KiB
    pl.bnn = (wide && d->Cout % 256 == 0) ? 256 : 128;
    pl.bcc = (wide && d->C1 % 256 == 0 && c2 % 256 == 0) ? 256 : 128;
    // few output tiles (a 256 x 1024 gradient is FOUR 256 x 256 tiles: 64 splits over the pixels, 64 MB of slabs for 1 MB of gradient): with at most four
    // 256 x 256 tiles (x taps) the n side drops to 128 rows, i.e. twice the tiles, half the splits and slab bytes.  Measured (profiles/r3_ab_wgrad_tiles.txt):
    // -0.16 ms per ResNet-50 step (the 1x1 256 <-> 1024 and 512 -> 512 gradients); with the 3x3 256 -> 256 layers included (nine tiles) +0.03, both sides at 128 slower
    // (<= 5 since round 6: the 1280 x 256 product [g | x]^T x of a folded BatchNorm apply pass, sl_conv2d_bwd_weight_dy2, needs the 128-row tile's in-kernel column sums)
    if (pl.bnn == 256 && pl.bcc == 256 && (long long)(d->Cout / 256) * (d->Cin / 256) * d->KH * d->KW <= 5) pl.bnn = 128;
  } else {
    pl.bnn = d->Cout % 128 == 0 ? 128 : 64;
    pl.bcc = (d->C1 % 128 == 0 && c2 % 128 == 0) ? 128 : 64;
  }
  pl.gridN = d->Cout / pl.bnn; pl.gridC = d->Cin / pl.bcc; pl.taps = d->KH * d->KW;
  const long long tiles = (long long)pl.gridN * pl.gridC * pl.taps;
  const double slab = (double)d->Cout * pl.taps * d->Cin * sizeof(float);
  // split-K factor: balance the block count over the 256 CUs (time ~ number of block waves) against the slab
  // traffic (written once, read once by the reduce), at >= 128 reduction rows per split and <= 1 GiB of slabs.
  const double flops = 2.0 * M * d->Cout * (double)d->Cin * pl.taps;
  const double cu_rate = 2.5e12, hbm = 3.0e12;      // ~640 TFLOP/s chip-wide when every CU is busy
  const long long max_s = (M + 127) / 128;
  int best = 1; double best_t = 1e30;
  for (int sp = 1; sp <= 512 && sp <= max_s; ++sp) {
    if (slab * sp > 1024.0 * 1024.0 * 1024.0 && sp > 1) break;
    const double blocks = (double)tiles * sp;
    const double waves = ceil(blocks / 256.0);
    const double t = flops / blocks / cu_rate * waves + 2.0 * slab * sp / hbm + 3e-6 * waves;
    if (t < best_t) { best_t = t; best = sp; }
  }
  long long splits = best;
  long long rps = (M + splits - 1) / splits;
  rps = (rps + 63) / 64 * 64;
  splits = (M + rps - 1) / rps;
  pl.splits = (int)splits; pl.rows_per_split = (int)rps; pl.ws_bytes = (size_t)(slab * splits);
  return pl;
}

template <typename T, int BNN, int BCC, int WNN, int WCC, bool TR>
int launch_wgrad_glds(dim3 grid, WgradParams& p, hipStream_t st) {
  const size_t lds = (size_t)wg_ring_depth<T, BNN, BCC>() * 32 * (BNN + BCC) * sizeof(T);        // ring of 32-row stages
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_glds_kernel<T, BNN, BCC, WNN, WCC, TR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  if (p.colsum) {
    if constexpr (BNN == 256 && BCC == 256) { sl_set_error("conv bwd_weight: no bias instantiation of the 256 x 256 tile"); return SL_EINVAL; }
    else {
      static bool attr_set_b = false;
      if (!attr_set_b && lds > 64 * 1024) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_glds_kernel<T, BNN, BCC, WNN, WCC, TR, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set_b = true;
      }
      hipLaunchKernelGGL((conv_wgrad_glds_kernel<T, BNN, BCC, WNN, WCC, TR, true>), grid, dim3(64 * WNN * WCC), lds, st, p);
      SL_LAUNCH_CHECK("conv_wgrad_glds_kernel (bias)");
      return 0;
    }
  }
  hipLaunchKernelGGL((conv_wgrad_glds_kernel<T, BNN, BCC, WNN, WCC, TR>), grid, dim3(64 * WNN * WCC), lds, st, p);
  SL_LAUNCH_CHECK("conv_wgrad_glds_kernel");
  return 0;
}

template <typename T, bool TR>
int launch_wgrad(const WgradPlan& pl, WgradParams& p, hipStream_t st) {
  dim3 grid(pl.gridN * pl.gridC * pl.taps * pl.splits);
  if (pl.glds) {
    if (pl.bnn == 256 && pl.bcc == 256) return launch_wgrad_glds<T, 256, 256, 2, 4, TR>(grid, p, st);     // (four waves of 128 x 128 measured slower: profiles/r3_ab_wgrad_w4.txt)
    if (pl.bnn == 256) return launch_wgrad_glds<T, 256, 128, 4, 2, TR>(grid, p, st);
    if (pl.bcc == 256) return launch_wgrad_glds<T, 128, 256, 2, 4, TR>(grid, p, st);
    return launch_wgrad_glds<T, 128, 128, 2, 2, TR>(grid, p, st);
  }
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

// Which kernel sl_conv2d_bwd_weight runs for a shape (bench.py attributes HIP-event timings to rocprof kernel names with it):
// 1 conv_wgrad_c64k3_kernel, 2 conv_wgrad_c64p_kernel, 10000000 + 1000*BNN + BCC conv_wgrad_glds_kernel, 20000000 + ... conv_wgrad_kernel;
// + 500000 when the rows are pixel pairs.
extern "C" int sl_conv2d_wgrad_config(const SlConvDesc* d) {
  if (!d || d->Cout % 64 || d->Cin % 64) return SL_EINVAL;
  if (c64k3_eligible(d, d->Cin, 0) && use_tr()) return 1;
  if (c64p_eligible(d) && use_tr()) return 2;
  if (use_tr() && sl_wgrad3_eligible(d, nullptr)) return 3;
  const WgradPlan pl = plan(d);
  return (pl.glds ? 10000000 : 20000000) + (pl.pair ? 500000 : 0) + 1000 * pl.bnn + pl.bcc;
}

extern "C" size_t sl_conv2d_bwd_weight_workspace(const SlConvDesc* d) {
  if (!d || d->Cout % 64 || d->Cin % 32) return 0;
  size_t need = plan(d).ws_bytes;
  { size_t n3 = 0; if (sl_wgrad3_eligible(d, &n3) && n3 > need) need = n3; }
  if (c64k3_eligible(d, 64, 0)) { const size_t n2 = (size_t)(c64k3_blocks(d) + 1) * 64 * 64 * 9 * sizeof(float); if (n2 > need) need = n2; }
  if (c64p_eligible(d)) { const size_t n2 = (size_t)(c64p_slabs(d) + 1) * d->Cout * d->Cin * sizeof(float); if (n2 > need) need = n2; }
  return need;
}

extern "C" int sl_colsum_rows_partial(int dtype, const void* x, long long rows, int C, float* partial, sl_stream_t stream);
extern "C" int sl_conv2d_bwd_weight_ex(const SlConvDesc* d, const void* x, const void* x2, const void* dy, float* dw, int dw_cin_total,
                                       int dw_ci_off, void* workspace, size_t workspace_bytes, sl_stream_t stream);

extern "C" int sl_conv2d_bwd_weight(const SlConvDesc* d, const void* x, const void* x2, const void* dy, float* dw,
                                    void* workspace, size_t workspace_bytes, sl_stream_t stream) {
  return sl_conv2d_bwd_weight_ex(d, x, x2, dy, dw, d ? d->Cin : 0, 0, workspace, workspace_bytes, stream);
}
// dw may be a wider OIHW tensor [Cout][dw_cin_total][KH][KW]; this conv's Cin channels land at input-channel offset dw_ci_off
static int bwd_weight_impl(const SlConvDesc* d, const void* x, const void* x2, const void* dy, float* dw, int dw_cin_total,
                           int dw_ci_off, void* workspace, size_t workspace_bytes, sl_stream_t stream, float* colsum_partial, int n_valid = 0, int c_valid = 0,
                           SlWgradReduce* defer = nullptr, const void* dy2 = nullptr, int cout1 = 0);

extern "C" int sl_conv2d_bwd_weight_ex(const SlConvDesc* d, const void* x, const void* x2, const void* dy, float* dw, int dw_cin_total,
                                       int dw_ci_off, void* workspace, size_t workspace_bytes, sl_stream_t stream) {
  return bwd_weight_impl(d, x, x2, dy, dw, dw_cin_total, dw_ci_off, workspace, workspace_bytes, stream, nullptr);
}

// Weight gradient + bias gradient partials of one nn.Linear / biased conv: colsum_partial [sl_colsum_rows_blocks(B*Ho*Wo, Cout, dtype)][Cout] receives the column sums of dy
// per row chunk (finalize: sl_colsum_finalize / _multi).  1x1 layers on the tile kernels carry them in the slab-reduce launch; other shapes run the separate kernel.
extern "C" int sl_conv2d_bwd_weight_bias(const SlConvDesc* d, const void* x, const void* x2, const void* dy, float* dw, void* workspace, size_t workspace_bytes,
                                         float* colsum_partial, sl_stream_t stream) {
  SL_REQUIRE(colsum_partial, "conv bwd_weight_bias: null partial buffer");
  return bwd_weight_impl(d, x, x2, dy, dw, d ? d->Cin : 0, 0, workspace, workspace_bytes, stream, colsum_partial);
}

// The same at zero-padded channel counts: the problem is d->Cout x d->Cin (multiples of 64), dw is the parameter's own [n_valid][c_valid][KH][KW]; colsum_partial may be null.
extern "C" int sl_conv2d_bwd_weight_clip(const SlConvDesc* d, const void* x, const void* x2, const void* dy, float* dw, int n_valid, int c_valid, void* workspace,
                                         size_t workspace_bytes, float* colsum_partial, sl_stream_t stream) {
  SL_REQUIRE(d && n_valid > 0 && c_valid > 0 && n_valid <= d->Cout && c_valid <= d->Cin, "conv bwd_weight_clip: bad valid channel counts");
  return bwd_weight_impl(d, x, x2, dy, dw, c_valid, 0, workspace, workspace_bytes, stream, colsum_partial, n_valid, c_valid);
}

// sl_conv2d_bwd_weight_clip with the slab reduce DEFERRED: item receives what is left to do (item->splits == 0: nothing -- this shape's path reduced by itself); the workspace
// must stay untouched until sl_wgrad_reduce_multi has run (one workspace per deferred layer).
extern "C" int sl_conv2d_bwd_weight_defer(const SlConvDesc* d, const void* x, const void* x2, const void* dy, float* dw, int n_valid, int c_valid, void* workspace,
                                          size_t workspace_bytes, float* colsum_partial, SlWgradReduce* item, sl_stream_t stream) {
  SL_REQUIRE(d && item && n_valid >= 0 && c_valid >= 0 && n_valid <= d->Cout && c_valid <= d->Cin, "conv bwd_weight_defer: bad arguments");
  return bwd_weight_impl(d, x, x2, dy, dw, c_valid > 0 ? c_valid : d->Cin, 0, workspace, workspace_bytes, stream, colsum_partial, n_valid, c_valid, item);
}

extern "C" int sl_wgrad_reduce_multi(const SlWgradReduce* items, int n, sl_stream_t stream) {
  SL_REQUIRE(items && n > 0 && n <= SL_WGRAD_BATCH_MAX, "wgrad_reduce_multi: 1..%d items", SL_WGRAD_BATCH_MAX);
  WgradReduceBatch b{};
  int blocks = 0, m = 0;
  for (int i = 0; i < n; ++i) {
    const SlWgradReduce& t = items[i];
    if (t.splits == 0) continue;
    SL_REQUIRE(t.ws && t.dw && t.total > 0 && t.splits > 0 && t.Cin > 0 && t.total % t.Cin == 0, "wgrad_reduce_multi: item %d is not a deferred reduce", i);
    SL_REQUIRE(t.ncol == 0 || (t.dy && t.colsum_part && t.rows > 0 && t.rows_per_block > 0 && (t.dtype == SL_BF16 || t.dtype == SL_F32)), "wgrad_reduce_multi: item %d: bad column-sum part", i);
    b.it[m] = t; b.start[m] = blocks;
    blocks += (int)((t.total + 255) / 256) + t.ncol;
    ++m;
  }
  if (m == 0) return 0;
  b.start[m] = blocks; b.n = m;
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
  SL_LAUNCH_CHECK("wgrad_reduce_multi_kernel");
  return 0;
}

// Two gradient tensors side by side: dw [d->Cout][d->Cin] = [dy1 | dy2]^T x with dy1 [rows][Cout1], dy2 [rows][d->Cout - Cout1] (1x1 layers on the LDS-DMA tile kernel) --
// g^T x and x^T x of a folded BatchNorm apply pass (sl_bn_fold_wgrad) from ONE pass over x.  colsum_partial as in sl_conv2d_bwd_weight_bias (may be NULL).
extern "C" int sl_conv2d_bwd_weight_dy2(const SlConvDesc* d, const void* x, const void* dy1, const void* dy2, int Cout1, float* dw, void* workspace, size_t workspace_bytes,
                                        float* colsum_partial, sl_stream_t stream) {
  SL_REQUIRE(d && dy2 && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0, "conv bwd_weight_dy2: 1x1 stride-1 layers");
  return bwd_weight_impl(d, x, nullptr, dy1, dw, d->Cin, 0, workspace, workspace_bytes, stream, colsum_partial, 0, 0, nullptr, dy2, Cout1);
}

// Does the weight-gradient kernel of this shape carry the bias-gradient partials itself (BIAS instantiation: one row per split, two for pixel pairs)?
static bool wgrad_bias_fused(const SlConvDesc* d, const WgradPlan& pl) {      // test hook sl_debug_wgrad_bias(0): the column sums of dy back in the reduce launch
  return g_sl_debug.wgrad_bias && pl.glds && pl.taps == 1 && !(pl.bnn == 256 && pl.bcc == 256) && use_tr();
}
extern "C" int sl_colsum_rows_blocks(long long rows, int C, int dtype);
// rows of the colsum_partial buffer sl_conv2d_bwd_weight_bias / _clip (n_valid, c_valid: 0 = all channels) fill: the same path selection as bwd_weight_impl
extern "C" int sl_conv2d_bwd_weight_bias_rows(const SlConvDesc* d, int n_valid, int c_valid) {
  if (!d || d->Cout % 64 || d->Cin % 64) { sl_set_error("conv bwd_weight_bias_rows: null descriptor or channel counts that are not multiples of 64"); return SL_EINVAL; }
  const int generic = sl_colsum_rows_blocks((long long)d->B * d->Ho * d->Wo, d->Cout, d->dtype);
  const bool full = (n_valid <= 0 || n_valid == d->Cout) && (c_valid <= 0 || c_valid == d->Cin);
  if (use_tr() && ((full && c64k3_eligible(d, d->Cin, 0)) || c64p_eligible(d) || (full && sl_wgrad3_eligible(d, nullptr)))) return generic;
  const WgradPlan pl = plan(d);
  return wgrad_bias_fused(d, pl) ? pl.splits * (pl.pair ? 2 : 1) : generic;
}

static int bwd_weight_impl(const SlConvDesc* d, const void* x, const void* x2, const void* dy, float* dw, int dw_cin_total,
                           int dw_ci_off, void* workspace, size_t workspace_bytes, sl_stream_t stream, float* colsum_partial, int n_valid, int c_valid, SlWgradReduce* defer,
                           const void* dy2, int cout1) {
  SL_REQUIRE(d && x && dy && dw && workspace, "conv bwd_weight: null buffer");
  if (defer) defer->splits = 0;           // 0 = nothing left to do (this path ran its own reduce)
  const int nv = n_valid > 0 ? n_valid : d->Cout, cv = c_valid > 0 ? c_valid : d->Cin;
  SL_REQUIRE(nv <= d->Cout && cv <= d->Cin, "conv bwd_weight: valid channel counts exceed the (padded) problem");
  // column sums of dy by the stand-alone kernel: every path below that does not carry them in its reduce launch
  // the caller sized colsum_partial by sl_conv2d_bwd_weight_bias_rows, a second statement of the path selection below: every path checks that the rows it is about to
  // write are the rows that query promised (round-5 advisor: the two must not drift apart silently)
  const int promised_rows = colsum_partial ? sl_conv2d_bwd_weight_bias_rows(d, n_valid, c_valid) : 0;
  auto colsum_separately = [&]() -> int {
    if (!colsum_partial) return 0;
    SL_REQUIRE(sl_colsum_rows_blocks((long long)d->B * d->Ho * d->Wo, d->Cout, d->dtype) == promised_rows, "conv bwd_weight: bias partial rows (stand-alone pass) != sl_conv2d_bwd_weight_bias_rows (%d)", promised_rows);
    return sl_colsum_rows_partial(d->dtype, dy, (long long)d->B * d->Ho * d->Wo, d->Cout, colsum_partial, stream);
  };
  // the fixed-order slab reduces follow their MFMA kernel on the same stream (on a second stream they measured slower inside the captured step:
  // profiles/r3_ab_switches.txt, 27.10 vs 26.53 ms -- every fork / join is a pair of cross-branch dependencies in the graph)
  SL_REQUIRE(dw_ci_off >= 0 && dw_ci_off + cv <= dw_cin_total, "conv bwd_weight: bad dw channel window");
  const int bke = d->dtype == SL_BF16 ? 64 : 32;
  const int c2 = d->Cin - d->C1;
  SL_REQUIRE(d->dtype == SL_BF16 || d->dtype == SL_F32, "conv bwd_weight: bad dtype");
  SL_REQUIRE(d->Cout % 64 == 0 && d->C1 % 64 == 0 && c2 % 64 == 0, "conv bwd_weight: channels must be multiples of 64 (Cout %d, C1 %d, C2 %d)", d->Cout, d->C1, c2);
  (void)bke;
  SL_REQUIRE(c2 == 0 || x2, "conv bwd_weight: x2 missing");
  if (c64k3_eligible(d, dw_cin_total, dw_ci_off) && use_tr() && nv == d->Cout && cv == d->Cin) {
    const int nblk = c64k3_blocks(d), ntiles = d->B * cdiv(d->H, C3_T) * cdiv(d->W, C3_T);
    const size_t need = (size_t)(nblk + 1) * 64 * 64 * 9 * sizeof(float);         // nblk slabs + their sum
    if (workspace_bytes < need) { sl_set_error("conv bwd_weight: workspace %zu < %zu", workspace_bytes, need); return SL_EWORKSPACE; }
    const size_t lds = (size_t)(C3_PW * C3_PW + 256) * C3_PITCH;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv_wgrad_c64k3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    hipLaunchKernelGGL(conv_wgrad_c64k3_kernel, dim3(nblk), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)dy, (float*)workspace, d->B, d->H, d->W, ntiles);
    SL_LAUNCH_CHECK("conv_wgrad_c64k3_kernel");
    float* sum = (float*)workspace + (size_t)nblk * 64 * 64 * 9;
    hipStream_t rs = (hipStream_t)stream;
    if (int e = sl_colsum_finalize((const float*)workspace, nblk, 64 * 64 * 9, sum, (sl_stream_t)rs)) return e;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(64), dim3(256), 0, rs, (const float*)sum, dw, 64, 64, 9, 1, 64, 0, 64, 64);      // [n][tap][c] -> OIHW
    SL_LAUNCH_CHECK("wgrad_reduce_kernel");
    return colsum_separately();
  }
  if (c64p_eligible(d) && use_tr() && !dy2) {
    const int nblk = c64p_blocks(d), nslab = c64p_slabs(d), ntiles = (int)((long long)d->B * d->H * d->W / CP_T);
    const size_t need = (size_t)(nslab + 1) * d->Cout * d->Cin * sizeof(float);       // the slabs + their sum
    if (workspace_bytes < need) { sl_set_error("conv bwd_weight: workspace %zu < %zu", workspace_bytes, need); return SL_EWORKSPACE; }
    const size_t lds = (size_t)CP_T * (2 * (d->Cout + d->Cin) + 128);
    hipStream_t st = (hipStream_t)stream;
#define SL_C64P(CA, CB) do { static bool attr_set = false; \
      if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv_wgrad_c64p_kernel<CA, CB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; } \
      hipLaunchKernelGGL((conv_wgrad_c64p_kernel<CA, CB>), dim3(nblk), dim3(256), lds, st, (const bf16_t*)x, (const bf16_t*)dy, (float*)workspace, ntiles); } while (0)
    if (d->Cout == 256) SL_C64P(256, 64); else if (d->Cin == 256) SL_C64P(64, 256); else if (d->Cout == 128) SL_C64P(128, 128); else SL_C64P(64, 64);
#undef SL_C64P
    SL_LAUNCH_CHECK("conv_wgrad_c64p_kernel");
    const long long total = (long long)d->Cout * d->Cin;
    float* sum = (float*)workspace + (size_t)nslab * total;
    hipStream_t rs = (hipStream_t)stream;
    if (int e = sl_colsum_finalize((const float*)workspace, nslab, (int)total, sum, (sl_stream_t)rs)) return e;          // fixed-order column sums over the slabs (one block per 64 numbers)
    hipLaunchKernelGGL(wgrad_reduce_flat_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, rs, (const float*)sum, dw, total, 1, d->Cin, dw_cin_total, dw_ci_off, nv, cv);
    SL_LAUNCH_CHECK("wgrad_reduce_flat_kernel");
    return colsum_separately();
  }
  // 3x3 stride-1 layers with >= 128 x 64 channels: all nine taps from one pass over dy and x (conv_wgrad3.hip)
  if (use_tr() && nv == d->Cout && cv == d->Cin && sl_wgrad3_eligible(d, nullptr)) {
    if (int e = sl_wgrad3_run(d, x, x2, dy, dw, dw_cin_total, dw_ci_off, workspace, workspace_bytes, (hipStream_t)stream)) return e;
    return colsum_separately();
  }
  const WgradPlan pl = plan(d);
  if (workspace_bytes < pl.ws_bytes) { sl_set_error("conv bwd_weight: workspace %zu < %zu", workspace_bytes, pl.ws_bytes); return SL_EWORKSPACE; }
  WgradParams p{};
  p.src1 = x; p.src2 = x2; p.C1 = d->C1; p.C2 = c2; p.dy = dy; p.ws = (float*)workspace;
  if (dy2) {
    SL_REQUIRE(pl.glds && !pl.pair && pl.taps == 1 && use_tr() && d->dtype == SL_BF16 && cout1 > 0 && cout1 < d->Cout && cout1 % pl.bnn == 0 && (d->Cout - cout1) % 8 == 0,
               "conv bwd_weight_dy2: served by the LDS-DMA tile kernel only, Cout1 (%d) a multiple of its %d-row tile", cout1, pl.bnn);
    SL_REQUIRE(!colsum_partial || wgrad_bias_fused(d, pl), "conv bwd_weight_dy2: column sums only where the kernel forms them itself (two gradient tensors have no common pitch for a separate pass)");
    p.dy2 = dy2; p.Cout1 = cout1;
  }
  p.B = d->B; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.M = d->B * d->Ho * d->Wo; p.rows_per_split = pl.rows_per_split;
  p.gridN = pl.gridN; p.gridC = pl.gridC; p.taps = pl.taps; p.splits = pl.splits;
  p.pair = pl.pair ? 1 : 0;
  p.trace = g_sl_debug.wgrad_trace;
  if (pl.pair) { p.Cout = 2 * d->Cout; p.C1 = 2 * d->Cin; p.M /= 2; }
  const bool bias_fused = colsum_partial && wgrad_bias_fused(d, pl);
  if (bias_fused) SL_REQUIRE(pl.splits * (pl.pair ? 2 : 1) == promised_rows, "conv bwd_weight: bias partial rows (in-kernel) != sl_conv2d_bwd_weight_bias_rows (%d)", promised_rows);
  if (bias_fused) { p.colsum = colsum_partial; p.colsum_cout = d->Cout; colsum_partial = nullptr; }      // the kernel writes the partials; nothing left for the reduce launch
  hipStream_t st = (hipStream_t)stream;
  int e;
  if (d->dtype == SL_F32) e = launch_wgrad<float, false>(pl, p, st);
  else if (use_tr()) e = launch_wgrad<bf16_t, true>(pl, p, st);
  else e = launch_wgrad<bf16_t, false>(pl, p, st);
  if (e) return e;
  SL_REQUIRE(pl.taps <= 49, "conv bwd_weight: kernel window larger than 7x7");
  if (pl.pair) {
    const long long total = (long long)d->Cout * d->Cin * pl.taps;
    hipLaunchKernelGGL(wgrad_reduce_pair_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)workspace, dw, d->Cout, d->Cin, pl.taps, pl.splits, dw_cin_total, dw_ci_off, nv, cv);
  } else if (pl.taps == 1) {
    const long long total = (long long)d->Cout * d->Cin;
    const int nred = (int)((total + 255) / 256);
    if (defer) {
      // the flat slab reduce (and the column sums of dy that would ride in it) joins the caller's batch: sl_wgrad_reduce_multi runs it with the other layers' reduces
      defer->ws = (const float*)workspace; defer->dw = dw; defer->total = total; defer->splits = pl.splits; defer->Cin = d->Cin; defer->dw_cin_total = dw_cin_total;
      defer->dw_ci_off = dw_ci_off; defer->n_valid = nv; defer->c_valid = cv; defer->dtype = d->dtype; defer->dy = nullptr; defer->colsum_part = nullptr; defer->ncol = 0;
      defer->rows = 0; defer->Cout = d->Cout; defer->rows_per_block = 0;
      if (colsum_partial) {
        const long long rows = (long long)d->B * d->Ho * d->Wo, ch = sl_colsum_rows_chunk(rows, d->Cout, d->dtype == SL_BF16 ? 2 : 4);
        const int ncol = (int)((rows + ch - 1) / ch);
        SL_REQUIRE(ncol == promised_rows, "conv bwd_weight: bias partial rows (deferred reduce) != sl_conv2d_bwd_weight_bias_rows (%d)", promised_rows);
        defer->dy = dy; defer->rows = rows; defer->rows_per_block = ch; defer->colsum_part = colsum_partial; defer->ncol = ncol;
      }
      return 0;
    }
    if (colsum_partial) {
      const long long rows = (long long)d->B * d->Ho * d->Wo, ch = sl_colsum_rows_chunk(rows, d->Cout, d->dtype == SL_BF16 ? 2 : 4);
      const int ncol = (int)((rows + ch - 1) / ch);
      SL_REQUIRE(ncol == promised_rows, "conv bwd_weight: bias partial rows (reduce launch) != sl_conv2d_bwd_weight_bias_rows (%d)", promised_rows);
      if (d->dtype == SL_BF16)
        hipLaunchKernelGGL(wgrad_reduce_flat_colsum_kernel<bf16_t>, dim3(nred + ncol), dim3(256), 0, st, (const float*)workspace, dw, total, pl.splits, d->Cin, dw_cin_total, dw_ci_off,
                           nred, (const bf16_t*)dy, rows, d->Cout, ch, colsum_partial, nv, cv);
      else
        hipLaunchKernelGGL(wgrad_reduce_flat_colsum_kernel<float>, dim3(nred + ncol), dim3(256), 0, st, (const float*)workspace, dw, total, pl.splits, d->Cin, dw_cin_total, dw_ci_off,
                           nred, (const float*)dy, rows, d->Cout, ch, colsum_partial, nv, cv);
      SL_LAUNCH_CHECK("wgrad_reduce_flat_colsum_kernel");
        return 0;
    }
    hipLaunchKernelGGL(wgrad_reduce_flat_kernel, dim3((unsigned)nred), dim3(256), 0, st, (const float*)workspace, dw, total, pl.splits, d->Cin, dw_cin_total, dw_ci_off, nv, cv);
  } else {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(d->Cout * (d->Cin / 64)), dim3(256), 0, st, (const float*)workspace, dw, d->Cout, d->Cin, pl.taps, pl.splits, dw_cin_total, dw_ci_off, nv, cv);
  }
  SL_LAUNCH_CHECK("wgrad_reduce_kernel");
  return colsum_separately();
}
