// Weight layouts of the implicit-GEMM convolution: OIHW fp32 master weights -> [N][taps][C] forward / data-gradient operands (one launch per tensor or per table).
#include "common.h"

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

