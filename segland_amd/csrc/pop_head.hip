// POP head kernels (networks/pspnet_pop.py:95-121 orthogonal_decompose, :178-182 / :210-219 classifier on components)
// in the collapsed form of SURVEY.md 0.7: the [B,K,C,N] component tensors are never built.
//   proj[r][k] = S_k . q_r            bg_r = q_r - sum_k proj[r][k] S_k          (fp32, as the reference forces)
//   pred[r][0] = MLP(bg_r)            pred[r][1+k] = a_k max(p,0) + b_k max(-p,0),  a_k = MLP(S_k), b_k = MLP(-S_k)
// A pixel row is owned by LPR lanes of a wavefront (C = LPR x 16-byte vectors x NV: 64 lanes for the 512-channel PSPNet head, 16 / 32
// lanes -- 4 / 2 rows per wavefront -- for the 128-channel Swin head), shuffle reductions inside the LPR-lane group for the K projections.
#include "common.h"

namespace {

constexpr int KMAXP = 16;  // max prototypes (base + novel)

template <int LPR> __device__ __forceinline__ float group_sum(float v) {      // sum over the LPR lanes that share a row
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int LPR> __device__ __forceinline__ float across_groups(float v) {  // sum over the 64/LPR row groups (same channel, different rows)
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// KM per-lane partial sums (one per prototype) -> the KM row totals, every lane of the LPR-lane row group holding all of them.  A plain tree is KM x log2(LPR) shuffles (48 at
// KM = 8, LPR = 64 -- these kernels were bound by the LDS pipe that serves ds_bpermute).  Here each step halves the values a lane carries (the lane keeps the half chosen by its
// lane bit and sends the other): KM / 2 + KM / 4 + ... + 1 shuffles, then plain steps on the single value left, then KM broadcasts: 10 + 8 at KM = 8, LPR = 64.
template <int LPR, int KM>
__device__ __forceinline__ void group_sum_multi(float (&d)[KM], int lane) {
  static_assert((KM & (KM - 1)) == 0 && KM <= LPR, "power-of-two prototype slots, not more than lanes per row");
  float cur[KM];
#pragma unroll
  for (int k = 0; k < KM; ++k) cur[k] = d[k];
  int ofs = LPR / 2;
#pragma unroll
  for (int n = KM; n > 1; n >>= 1, ofs >>= 1) {          // n values -> n / 2: bit `ofs` of the lane picks the upper half
    const bool up = (lane & ofs) != 0;
#pragma unroll
    for (int i = 0; i < n / 2; ++i) {
      const float mine = up ? cur[n / 2 + i] : cur[i], send = up ? cur[i] : cur[n / 2 + i];
      cur[i] = mine + __shfl_xor(send, ofs, 64);
    }
  }
  float r = cur[0];
#pragma unroll
  for (; ofs > 0; ofs >>= 1) r += __shfl_xor(r, ofs, 64);
  // lanes whose upper log2(KM) bits (inside the row group) spell k hold total k: lane index inside the group = k * (LPR / KM) + anything
  const int base = lane & ~(LPR - 1);
#pragma unroll
  for (int k = 0; k < KM; ++k) d[k] = __shfl(r, base + k * (LPR / KM), 64);
}

// prototype rows into the LDS, zero beyond the n valid floats: N floats (a multiple of 1024) as float4 loads that are all in flight before the first LDS store
// (a scalar load -> store loop was 16 dependent memory round trips at the top of every block)
template <int N>
__device__ __forceinline__ void pop_fill_protos(const float* __restrict__ S, int n, float* Sl) {
  static_assert(N % 1024 == 0, "prototype tile: whole float4 sweeps of the block");
  constexpr int IT = N / 1024;
  float4 v[IT];
#pragma unroll
  for (int u = 0; u < IT; ++u) {
    const int e = (u * 256 + threadIdx.x) * 4;
    v[u] = e + 3 < n ? *(const float4*)(S + e) : make_float4(e < n ? S[e] : 0.f, e + 1 < n ? S[e + 1] : 0.f, e + 2 < n ? S[e + 2] : 0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < IT; ++u) *(float4*)(Sl + (u * 256 + threadIdx.x) * 4) = v[u];
}

template <typename T, int NV, int LPR, int KM>   // NV 16-byte vectors per lane, LPR lanes per row, KM >= Kt prototypes (unrolled)
__global__ __launch_bounds__(256) void pop_decompose_fwd_kernel(const T* __restrict__ feats, const float* __restrict__ S, int Kt,
                                                                float* __restrict__ proj, T* __restrict__ bg, long long R, int C) {
  constexpr int V = Vec16<T>::N, RPW = 64 / LPR;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Sl = sm;                       // [KM][C], rows >= Kt are zero
  pop_fill_protos<KM * LPR * NV * V>(S, Kt * C, Sl);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPR, grp = lane / LPR;
  constexpr bool SREG = KM * NV * V <= 64;               // 64 registers: 8 prototypes x 8 channels per lane (the 512-channel head), 8 x 8 at 128 channels / 16 lanes
  float sreg[SREG ? KM : 1][NV * V];
  if constexpr (SREG) {
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
      for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < V; ++e) sreg[k][j * V + e] = Sl[k * C + (j * LPR + sub) * V + e];
  }
  // the next row's vectors are loaded while the current row is reduced: one row per wave and iteration was a chain load -> dot -> shuffle tree -> store per row
  const long long rstep = gridDim.x * 4LL * RPW;
  uint4 nxt[NV];
  {
    const long long r = (blockIdx.x * 4LL + wave) * RPW + grp;
#pragma unroll
    for (int j = 0; j < NV; ++j) nxt[j] = r < R ? *(const uint4*)(feats + (size_t)r * C + (j * LPR + sub) * V) : make_uint4(0, 0, 0, 0);
  }
  for (long long rb = (blockIdx.x * 4LL + wave) * RPW; rb < R; rb += rstep) {
    const long long r = rb + grp;
    const bool live = r < R;
    float q[NV * V], o[NV * V];
#pragma unroll
    for (int j = 0; j < NV; ++j) unpack16<T>(nxt[j], &q[j * V]);             // rows past R were loaded as zeros
    {
      const long long rn = r + rstep;
#pragma unroll
      for (int j = 0; j < NV; ++j) nxt[j] = (rb + rstep < R && rn < R) ? *(const uint4*)(feats + (size_t)rn * C + (j * LPR + sub) * V) : make_uint4(0, 0, 0, 0);
    }
    // all projections first (they use q, not the running residual); the lane's slice of the prototypes lives in registers when it fits (SREG)
    float d[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < V; ++e) a = fmaf(q[j * V + e], SREG ? sreg[k][j * V + e] : Sl[k * C + (j * LPR + sub) * V + e], a);
      d[k] = a;
    }
    group_sum_multi<LPR, KM>(d, lane);
#pragma unroll
    for (int e = 0; e < NV * V; ++e) o[e] = q[e];
#pragma unroll
    for (int k = 0; k < KM; ++k) {                        // same order of the subtractions as before (k ascending)
      if (k < Kt) {
        if (sub == 0 && live) proj[(size_t)r * Kt + k] = d[k];
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
          for (int e = 0; e < V; ++e) o[j * V + e] -= d[k] * (SREG ? sreg[k][j * V + e] : Sl[k * C + (j * LPR + sub) * V + e]);
      }
    }
    if (live) {
#pragma unroll
      for (int j = 0; j < NV; ++j) *(uint4*)(bg + (size_t)r * C + (j * LPR + sub) * V) = pack16<T>(&o[j * V]);
    }
  }
}

template <typename T>
__global__ void pop_proto_rows_kernel(const float* __restrict__ S, int Kt, int C, T* __restrict__ dst) {
  const int n = Kt * C;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    dst[e] = from_f<T>(S[e]);
    dst[n + e] = from_f<T>(-S[e]);
  }
}

template <typename T, int NV, int LPR>
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const T* __restrict__ h, const float* __restrict__ w, float* __restrict__ z,
                                                         long long R, int C) {
  constexpr int V = Vec16<T>::N, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPR, grp = lane / LPR;
  float wv[NV * V];
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int e = 0; e < V; ++e) wv[j * V + e] = w[(j * LPR + sub) * V + e];
  for (long long rb = (blockIdx.x * 4LL + wave) * RPW; rb < R; rb += gridDim.x * 4LL * RPW) {
    const long long r = rb + grp;
    float d = 0.f;
    if (r < R) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float t[V];
        unpack16<T>(*(const uint4*)(h + (size_t)r * C + (j * LPR + sub) * V), t);
#pragma unroll
        for (int e = 0; e < V; ++e) d = fmaf(t[e], wv[j * V + e], d);
      }
    }
    d = group_sum<LPR>(d);
    if (sub == 0 && r < R) z[r] = d;
  }
}

// dh = dz * w * (h > 0); partial[blk][c] = sum_r dz[r] * h[r][c]
template <typename T, int NV, int LPR>
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const T* __restrict__ h, const float* __restrict__ w, const float* __restrict__ dz,
                                                         T* __restrict__ dh, float* __restrict__ part, long long R, int C,
                                                         long long rows_per_blk) {
  constexpr int V = Vec16<T>::N, RPW = 64 / LPR;
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPR, grp = lane / LPR;
  float wv[NV * V], acc[NV * V];
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int e = 0; e < V; ++e) { wv[j * V + e] = w[(j * LPR + sub) * V + e]; acc[j * V + e] = 0.f; }
  const long long r0 = blockIdx.x * rows_per_blk;
  long long r1 = r0 + rows_per_blk; if (r1 > R) r1 = R;
  for (long long r = r0 + wave * RPW + grp; r < r1; r += 4 * RPW) {
    const float g = dz[r];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      float t[V], o[V];
      unpack16<T>(*(const uint4*)(h + (size_t)r * C + (j * LPR + sub) * V), t);
#pragma unroll
      for (int e = 0; e < V; ++e) { acc[j * V + e] = fmaf(g, t[e], acc[j * V + e]); o[e] = t[e] > 0.f ? g * wv[j * V + e] : 0.f; }
      *(uint4*)(dh + (size_t)r * C + (j * LPR + sub) * V) = pack16<T>(o);
    }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float t = across_groups<LPR>(acc[j * V + e]);
      if (grp == 0) sm[wave * C + (j * LPR + sub) * V + e] = t;
    }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) part[(size_t)blockIdx.x * C + c] = sm[c] + sm[C + c] + sm[2 * C + c] + sm[3 * C + c];
}

// out[c] = sum_blk part[blk][c]: 64 columns x 16 row-lanes per block, fixed-order LDS tree (bit-stable)
__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ out) {
  __shared__ double red[16][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s = 0.0;
  if (c < C) {
    int r = rl;
    // up to 2 048 partial rows = 128 loads per lane: sixteen in flight (round 5: four made this launch 14 us of pure load latency, 12 launches per Swin-T step); the additions
    // keep the row order, so the sums did not change
    for (; r + 15 * 16 < nblk; r += 16 * 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = part[(size_t)(r + u * 16) * C + c];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += (double)v[u];
    }
    for (; r + 3 * 16 < nblk; r += 4 * 16) {
      const float v0 = part[(size_t)r * C + c], v1 = part[(size_t)(r + 16) * C + c], v2 = part[(size_t)(r + 32) * C + c], v3 = part[(size_t)(r + 48) * C + c];
      s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
    }
    for (; r < nblk; r += 16) s += (double)part[(size_t)r * C + c];
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j][cl];
    out[c] = (float)t;
  }
}

// Column sums of a [rows][C] activation tensor (bias gradients of the transformer blocks' linear layers, conv bias gradients): partial[blk][c] over the block's
// contiguous row chunk, four rows in flight per thread.  (Round 2 ran these through the BatchNorm backward reduce with a zero "mean": 120 launches of ~18 us per Swin-T
// step, 15 % of it -- a kernel shaped for 65 536-row tensors with one row in flight per thread and rows / 64 blocks, i.e. 128 blocks for the 8 192-token stage.)
template <typename T>
__global__ __launch_bounds__(256) void colsum_rows_partial_kernel(const T* __restrict__ x, long long rows, int C, long long rows_per_block, float* __restrict__ part) {
  __shared__ float red[256 * Vec16<T>::N];
  sl_colsum_rows_block<T>(x, rows, C, rows_per_block, part, blockIdx.x, red);
}

// The same for up to SL_COLSUM_MAX partial buffers in ONE launch (the bias / LayerNorm / attention-bias gradients of a transformer block
// backward: eight 5 us launches otherwise).  The batch descriptor travels by value in the kernel arguments; entry i owns the blocks
// [first[i], first[i+1]).
struct ColsumBatchDev { int n; int first[SL_COLSUM_MAX + 1]; const float* part[SL_COLSUM_MAX]; float* out[SL_COLSUM_MAX]; int nblk[SL_COLSUM_MAX]; int C[SL_COLSUM_MAX]; };
__global__ __launch_bounds__(1024) void colsum_finalize_multi_kernel(ColsumBatchDev b) {
  __shared__ double red[16][64];
  int e = 0;
  while (e + 1 < b.n && (int)blockIdx.x >= b.first[e + 1]) ++e;
  const float* __restrict__ part = b.part[e];
  const int nblk = b.nblk[e], C = b.C[e];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = ((int)blockIdx.x - b.first[e]) * 64 + cl;
  double s = 0.0;
  if (c < C) {
    int r = rl;
    // up to 2 048 partial rows = 128 loads per lane: sixteen in flight (round 5: four made this launch 14 us of pure load latency, 12 launches per Swin-T step); the additions
    // keep the row order, so the sums did not change
    for (; r + 15 * 16 < nblk; r += 16 * 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = part[(size_t)(r + u * 16) * C + c];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += (double)v[u];
    }
    for (; r + 3 * 16 < nblk; r += 4 * 16) {
      const float v0 = part[(size_t)r * C + c], v1 = part[(size_t)(r + 16) * C + c], v2 = part[(size_t)(r + 32) * C + c], v3 = part[(size_t)(r + 48) * C + c];
      s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
    }
    for (; r < nblk; r += 16) s += (double)part[(size_t)r * C + c];
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j][cl];
    b.out[e][c] = (float)t;
  }
}

// SyncBatchNorm totals: the same column sum kept in double (all-reduced in double by the caller) ...
__global__ __launch_bounds__(1024) void colsum_f64_kernel(const float* __restrict__ part, int nblk, int C, double* __restrict__ out) {
  __shared__ double red[16][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s = 0.0;
  if (c < C) for (int r = rl; r < nblk; r += 16) s += (double)part[(size_t)r * C + c];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j][cl];
    out[c] = t;
  }
}
// ... and handed back to the float partial-sum interface of the finalize kernels as two rows (hi, lo) whose double sum is the total
__global__ void f64_split_kernel(const double* __restrict__ tot, int n, float* __restrict__ hi_lo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double t = tot[i];
  const float hi = (float)t;
  hi_lo[i] = hi;
  hi_lo[n + i] = (float)(t - (double)hi);
}

__global__ void pop_combine_fwd_kernel(const float* __restrict__ proj, const float* __restrict__ zbg, const float* __restrict__ a,
                                       const float* __restrict__ b, int Kt, float* __restrict__ preds, int B, int N) {
  const long long R = (long long)B * N;
  for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < R; r += (long long)gridDim.x * blockDim.x) {
    const int bb = (int)(r / N), n = (int)(r % N);
    float* o = preds + (size_t)bb * (1 + Kt) * N + n;
    o[0] = zbg[r];
    for (int k = 0; k < Kt; ++k) {
      const float p = proj[(size_t)r * Kt + k];
      o[(size_t)(1 + k) * N] = a[k] * fmaxf(p, 0.f) + b[k] * fmaxf(-p, 0.f);
    }
  }
}

__global__ __launch_bounds__(256) void pop_combine_bwd_kernel(const float* __restrict__ dpreds, const float* __restrict__ proj,
                                                              const float* __restrict__ a, const float* __restrict__ b, int Kt,
                                                              float* __restrict__ dzbg, float* __restrict__ dproj,
                                                              float* __restrict__ dab, int B, int N, long long rows_per_blk) {
  __shared__ float red[4][2 * KMAXP];
  const long long R = (long long)B * N;
  float da[KMAXP], db[KMAXP];
#pragma unroll
  for (int k = 0; k < KMAXP; ++k) { da[k] = 0.f; db[k] = 0.f; }
  const long long r0 = blockIdx.x * rows_per_blk;
  long long r1 = r0 + rows_per_blk; if (r1 > R) r1 = R;
  for (long long r = r0 + threadIdx.x; r < r1; r += 256) {
    const int bb = (int)(r / N), n = (int)(r % N);
    const float* g = dpreds + (size_t)bb * (1 + Kt) * N + n;
    dzbg[r] = g[0];
#pragma unroll
    for (int k = 0; k < KMAXP; ++k) {
      if (k < Kt) {
        const float p = proj[(size_t)r * Kt + k], d = g[(size_t)(1 + k) * N];
        dproj[(size_t)r * Kt + k] = p > 0.f ? d * a[k] : (p < 0.f ? -d * b[k] : 0.f);
        da[k] += d * fmaxf(p, 0.f);
        db[k] += d * fmaxf(-p, 0.f);
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < KMAXP; ++k) {
    const float x = wave_sum(da[k]), y = wave_sum(db[k]);
    if (lane == 0) { red[wave][k] = x; red[wave][KMAXP + k] = y; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * Kt) {
    const int k = threadIdx.x % Kt, which = threadIdx.x / Kt;
    const int s = which * KMAXP + k;
    dab[(size_t)blockIdx.x * 2 * Kt + which * Kt + k] = red[0][s] + red[1][s] + red[2][s] + red[3][s];
  }
}

// t_k = dproj[r][k] - dg_r . S_k ;  dq_r = dg_r + sum_k t_k S_k ;  dS_k += t_k q_r - proj[r][k] dg_r
template <typename T, int NV, int KM, int LPR>
__global__ __launch_bounds__(256) void pop_decompose_bwd_kernel(const T* __restrict__ dg, const T* __restrict__ feats, const float* __restrict__ S,
                                                                const float* __restrict__ proj, const float* __restrict__ dproj, int Kt,
                                                                T* __restrict__ dq, float* __restrict__ dSpart, long long R, int C,
                                                                long long rows_per_blk) {
  constexpr int V = Vec16<T>::N, CL = NV * V, RPW = 64 / LPR;     // CL channels per lane
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Sl = sm;                                 // [KM][C]; reused for the cross-wave reduction
  pop_fill_protos<KM * LPR * NV * V>(S, Kt * C, Sl);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPR, grp = lane / LPR;
  float acc[KM][CL];
#pragma unroll
  for (int k = 0; k < KM; ++k)
#pragma unroll
    for (int e = 0; e < CL; ++e) acc[k][e] = 0.f;
  constexpr bool SREG = KM * CL <= 64;                    // the lane's slice of the prototypes in registers (8 prototypes x 8 channels)
  float sreg[SREG ? KM : 1][CL];
  if constexpr (SREG) {
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
      for (int e = 0; e < CL; ++e) sreg[k][e] = Sl[k * C + ((e / V) * LPR + sub) * V + (e % V)];
  }
  const long long r0 = blockIdx.x * rows_per_blk;
  long long r1 = r0 + rows_per_blk; if (r1 > R) r1 = R;
  for (long long rb = r0 + wave * RPW; rb < r1; rb += 4 * RPW) {
    const long long r = rb + grp;
    const bool live = r < r1;
    float g[CL], q[CL], o[CL];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      if (live) {
        unpack16<T>(*(const uint4*)(dg + (size_t)r * C + (j * LPR + sub) * V), &g[j * V]);
        unpack16<T>(*(const uint4*)(feats + (size_t)r * C + (j * LPR + sub) * V), &q[j * V]);
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e) { g[j * V + e] = 0.f; q[j * V + e] = 0.f; }
      }
    }
#pragma unroll
    for (int e = 0; e < CL; ++e) o[e] = g[e];
    // all KM dot products first, one transposed reduction for them (group_sum_multi), then the updates: the chain dot -> shuffle tree -> update per prototype
    // was KM dependent trees per row
    float d[KM], tp[KM], pp[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < CL; ++e) a = fmaf(g[e], SREG ? sreg[k][e] : Sl[k * C + ((e / V) * LPR + sub) * V + (e % V)], a);
      d[k] = a;
      tp[k] = (live && k < Kt) ? dproj[(size_t)r * Kt + k] : 0.f;
      pp[k] = (live && k < Kt) ? proj[(size_t)r * Kt + k] : 0.f;
    }
    group_sum_multi<LPR, KM>(d, lane);
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      if (k < Kt) {
        const float t = live ? tp[k] - d[k] : 0.f, p = pp[k];
#pragma unroll
        for (int e = 0; e < CL; ++e) {
          o[e] = fmaf(t, SREG ? sreg[k][e] : Sl[k * C + ((e / V) * LPR + sub) * V + (e % V)], o[e]);
          acc[k][e] += t * q[e] - p * g[e];
        }
      }
    }
    if (live) {
#pragma unroll
      for (int j = 0; j < NV; ++j) *(uint4*)(dq + (size_t)r * C + (j * LPR + sub) * V) = pack16<T>(&o[j * V]);
    }
  }
  // cross-wave reduction, one prototype at a time through LDS
  __syncthreads();
  float* red = sm;   // [4][C]
#pragma unroll
  for (int k = 0; k < KM; ++k) {
    if (k < Kt) {     // Kt is block-uniform
#pragma unroll
      for (int e = 0; e < CL; ++e) {
        const float t = across_groups<LPR>(acc[k][e]);
        if (grp == 0) red[wave * C + ((e / V) * LPR + sub) * V + (e % V)] = t;
      }
      __syncthreads();
      for (int c = threadIdx.x; c < C; c += 256)
        dSpart[((size_t)blockIdx.x * Kt + k) * C + c] = red[c] + red[C + c] + red[2 * C + c] + red[3 * C + c];
      __syncthreads();
    }
  }
}

inline int row_blocks(long long R, int cap) { long long b = (R + 63) / 64; return (int)(b < 1 ? 1 : (b > cap ? cap : b)); }

// C = LPR * NV * (16 / sizeof(T)).  512 channels: the PSPNet-POP head (d_model, pspnet_pop.py:43); 128 / 256: the Swin-POP head
// (d_model = backbone.get_filters()[0] = 96 / 128 / 192 zero-padded to a multiple of 128, swin_pop.py:182).
template <int N> struct IC { static constexpr int value = N; };
template <typename F>
int pop_dispatch(int dtype, int C, F&& f) {
  if (dtype == SL_BF16 && C == 512) return f(bf16_t{}, IC<1>{}, IC<64>{});
  if (dtype == SL_BF16 && C == 256) return f(bf16_t{}, IC<1>{}, IC<32>{});
  if (dtype == SL_BF16 && C == 128) return f(bf16_t{}, IC<1>{}, IC<16>{});
  if (dtype == SL_F32 && C == 512) return f(float{}, IC<2>{}, IC<64>{});
  if (dtype == SL_F32 && C == 256) return f(float{}, IC<1>{}, IC<64>{});
  if (dtype == SL_F32 && C == 128) return f(float{}, IC<1>{}, IC<32>{});
  SL_REQUIRE(false, "pop head: unsupported dtype/C (%d, %d); C must be 128, 256 or 512", dtype, C);
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------
// Prototype preparation in ONE launch (forward) / ONE launch (backward): S = F.normalize(E, p=2, dim=-1) (pspnet_pop.py:96-99,170-171), the prototype
// similarity G = S_a [S_a ; S_b]^T (pspnet_pop.py:185-186 base: S_b empty, G = S S^T;  :236-239 ft: a = novel, b = base) and the orthogonality term
// mean |G[i][j]|, j > i (criterion.py:37-43, also for the rectangular ft matrix).  In torch this is ~50 launches of 5 us each per step (normalize, matmul,
// gather, abs, mean and their autograd mirrors) -- 1 % of the ResNet-50 step.  K <= 16 rows of C <= 1024 numbers: one block, everything in the LDS.
constexpr int PP_KMAX = 32;
__global__ __launch_bounds__(256) void pop_proto_fwd_kernel(const float* __restrict__ Ea, int Ka, const float* __restrict__ Eb, int Kb, int C,
                                                            float* __restrict__ Sa, float* __restrict__ Sb, float* __restrict__ inv_norm, float* __restrict__ G,
                                                            float* __restrict__ orth) {
  extern __shared__ float sm[];                     // [Ka + Kb][C] normalised rows, then [Ka][Ka + Kb] similarities
  __shared__ float nrm[PP_KMAX];
  const int Kt = Ka + Kb, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = wave; k < Kt; k += 4) {              // one wavefront per row: sum of squares in a fixed order
    const float* e = k < Ka ? Ea + (size_t)k * C : Eb + (size_t)(k - Ka) * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += e[c] * e[c];
    s = wave_sum(s);
    if (lane == 0) nrm[k] = 1.f / fmaxf(sqrtf(s), 1e-12f);
  }
  __syncthreads();
  for (int i = tid; i < Kt * C; i += 256) {
    const int k = i / C, c = i - k * C;
    const float v = (k < Ka ? Ea[(size_t)k * C + c] : Eb[(size_t)(k - Ka) * C + c]) * nrm[k];
    sm[i] = v;
    if (k < Ka) Sa[i] = v; else Sb[(size_t)(k - Ka) * C + c] = v;
  }
  if (tid < Kt) inv_norm[tid] = nrm[tid];
  __syncthreads();
  float* gs = sm + Kt * C;
  for (int pr = wave; pr < Ka * Kt; pr += 4) {
    const int i = pr / Kt, j = pr - i * Kt;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += sm[i * C + c] * sm[j * C + c];
    s = wave_sum(s);
    if (lane == 0) { gs[pr] = s; G[pr] = s; }
  }
  __syncthreads();
  if (tid == 0) {                                   // row-major over the strict upper triangle, like the reference's boolean-mask selection
    float s = 0.f; int n = 0;
    for (int i = 0; i < Ka; ++i)
      for (int j = i + 1; j < Kt; ++j) { s += fabsf(gs[i * Kt + j]); ++n; }
    orth[0] = n ? s / (float)n : 0.f;
  }
}

// dE = normalize' (dS + d orth): dS_total[k] = dS[k] + sum over the similarity entries the row takes part in; dE[k] = (dS_total[k] - S[k] (S[k] . dS_total[k])) * inv_norm[k]
__global__ __launch_bounds__(256) void pop_proto_bwd_kernel(const float* __restrict__ Sa, int Ka, const float* __restrict__ Sb, int Kb, int C,
                                                            const float* __restrict__ inv_norm, const float* __restrict__ G, const float* __restrict__ dSa,
                                                            const float* __restrict__ dSb, const float* __restrict__ dorth, float* __restrict__ dEa, float* __restrict__ dEb) {
  extern __shared__ float sm[];                     // [Kt][C] S, [Kt][C] dS_total
  __shared__ float coef[PP_KMAX * PP_KMAX];         // d orth / d G[i][j]
  __shared__ float dots[PP_KMAX];
  const int Kt = Ka + Kb, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int n = 0;
  for (int i = 0; i < Ka; ++i) n += Kt - i - 1 > 0 ? Kt - i - 1 : 0;
  const float go = (dorth && n) ? dorth[0] / (float)n : 0.f;
  for (int pr = tid; pr < Ka * Kt; pr += 256) {
    const int i = pr / Kt, j = pr - i * Kt;
    const float g = G[pr];
    coef[pr] = j > i ? go * (g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f)) : 0.f;     // torch.sign(0) = 0
  }
  for (int i = tid; i < Kt * C; i += 256) {
    const int k = i / C, c = i - k * C;
    sm[i] = k < Ka ? Sa[i] : Sb[(size_t)(k - Ka) * C + c];
  }
  __syncthreads();
  float* dt = sm + Kt * C;
  for (int i = tid; i < Kt * C; i += 256) {
    const int k = i / C, c = i - k * C;
    float v = k < Ka ? (dSa ? dSa[i] : 0.f) : (dSb ? dSb[(size_t)(k - Ka) * C + c] : 0.f);
    if (k < Ka) for (int j = 0; j < Kt; ++j) v += coef[k * Kt + j] * sm[j * C + c];      // G[k][j] = S_k . S_j as row k
    for (int i2 = 0; i2 < Ka; ++i2) v += coef[i2 * Kt + k] * sm[i2 * C + c];               // G[i2][k] as column k
    dt[i] = v;
  }
  __syncthreads();
  for (int k = wave; k < Kt; k += 4) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += sm[k * C + c] * dt[k * C + c];
    s = wave_sum(s);
    if (lane == 0) dots[k] = s;
  }
  __syncthreads();
  for (int i = tid; i < Kt * C; i += 256) {
    const int k = i / C, c = i - k * C;
    const float v = (dt[i] - sm[i] * dots[k]) * inv_norm[k];
    if (k < Ka) { if (dEa) dEa[i] = v; } else if (dEb) dEb[(size_t)(k - Ka) * C + c] = v;
  }
}
}  // namespace

extern "C" int sl_pop_decompose_fwd(int dtype, const void* feats, const float* S, int Kt, float* proj, void* bg, long long R,
                                    int C, sl_stream_t stream) {
  SL_REQUIRE(feats && S && proj && bg && R > 0 && Kt >= 1 && Kt <= KMAXP, "pop_decompose_fwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  // 16 rows per block (four per wave) until 2048 blocks: every block first brings the prototypes (up to 32 KB) into the LDS and its registers, which four rows per block
  // (2048 blocks for the fine-tune pair's 8 192 rows: 60 us) do not repay; from 32 768 rows on the cap binds and nothing changes
  const int blocks = row_blocks(R * 4, 2048);
  const size_t lds = (size_t)KMAXP * C * sizeof(float);
  if (int e = pop_dispatch(dtype, C, [&](auto t, auto nv, auto lpr) {
        using T = decltype(t);
        if (Kt <= 8) hipLaunchKernelGGL((pop_decompose_fwd_kernel<T, decltype(nv)::value, decltype(lpr)::value, 8>), dim3(blocks), dim3(256), lds, st, (const T*)feats, S, Kt, proj, (T*)bg, R, C);
        else hipLaunchKernelGGL((pop_decompose_fwd_kernel<T, decltype(nv)::value, decltype(lpr)::value, 16>), dim3(blocks), dim3(256), lds, st, (const T*)feats, S, Kt, proj, (T*)bg, R, C);
        return 0; })) return e;
  SL_LAUNCH_CHECK("pop_decompose_fwd_kernel");
  return 0;
}


// One block holds everything in the LDS: the forward [Kt][C] rows + the [Ka][Kt] similarities, the backward TWO [Kt][C] matrices next to 4.2 KB of static
// arrays (coef, dots).  Without the opt-in attribute a block gets 64 KiB, static + dynamic: the forward must refuse what the backward of the same step could not
// run (round-3 advisor: C = 512 with 16..29 prototypes passed the forward and raised mid-step in the backward).
static bool pp_fits(int Ka, int Kb, int C) {
  const size_t Kt = (size_t)Ka + Kb;
  const size_t fwd = (Kt * C + (size_t)Ka * Kt) * sizeof(float) + PP_KMAX * sizeof(float);
  const size_t bwd = 2 * Kt * C * sizeof(float) + (PP_KMAX * PP_KMAX + PP_KMAX) * sizeof(float);
  return Ka >= 1 && Kb >= 0 && Kt <= (size_t)PP_KMAX && C > 0 && fwd <= 64 * 1024 && bwd <= 64 * 1024;
}

extern "C" int sl_pop_proto_ok(int Ka, int Kb, int C) { return pp_fits(Ka, Kb, C) ? 1 : 0; }

extern "C" int sl_pop_proto_fwd(const float* Ea, int Ka, const float* Eb, int Kb, int C, float* Sa, float* Sb, float* inv_norm, float* G, float* orth,
                                sl_stream_t stream) {
  SL_REQUIRE(Ea && Sa && inv_norm && G && orth && Ka >= 1 && Kb >= 0 && Ka + Kb <= PP_KMAX && C > 0 && (Kb == 0 || (Eb && Sb)), "pop_proto_fwd: bad args");
  const size_t lds = ((size_t)(Ka + Kb) * C + (size_t)Ka * (Ka + Kb)) * sizeof(float);
  SL_REQUIRE(pp_fits(Ka, Kb, C), "pop_proto_fwd: %d prototypes of %d channels do not fit one block (forward + backward; sl_pop_proto_ok)", Ka + Kb, C);
  hipLaunchKernelGGL(pop_proto_fwd_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, Ea, Ka, Eb, Kb, C, Sa, Sb, inv_norm, G, orth);
  SL_LAUNCH_CHECK("pop_proto_fwd_kernel");
  return 0;
}

extern "C" int sl_pop_proto_bwd(const float* Sa, int Ka, const float* Sb, int Kb, int C, const float* inv_norm, const float* G, const float* dSa,
                                const float* dSb, const float* dorth, float* dEa, float* dEb, sl_stream_t stream) {
  SL_REQUIRE(Sa && inv_norm && G && Ka >= 1 && Kb >= 0 && Ka + Kb <= PP_KMAX && C > 0 && (Kb == 0 || Sb), "pop_proto_bwd: bad args");
  const size_t lds = 2 * (size_t)(Ka + Kb) * C * sizeof(float);
  SL_REQUIRE(pp_fits(Ka, Kb, C), "pop_proto_bwd: %d prototypes of %d channels do not fit one block (sl_pop_proto_ok)", Ka + Kb, C);
  hipLaunchKernelGGL(pop_proto_bwd_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, Sa, Ka, Sb, Kb, C, inv_norm, G, dSa, dSb, dorth, dEa, dEb);
  SL_LAUNCH_CHECK("pop_proto_bwd_kernel");
  return 0;
}

extern "C" int sl_pop_proto_rows(int dtype, const float* S, int Kt, int C, void* dst, sl_stream_t stream) {
  SL_REQUIRE(S && dst && Kt >= 1 && C > 0, "pop_proto_rows: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SL_BF16) hipLaunchKernelGGL(pop_proto_rows_kernel<bf16_t>, dim3(cdiv(Kt * C, 256)), dim3(256), 0, st, S, Kt, C, (bf16_t*)dst);
  else if (dtype == SL_F32) hipLaunchKernelGGL(pop_proto_rows_kernel<float>, dim3(cdiv(Kt * C, 256)), dim3(256), 0, st, S, Kt, C, (float*)dst);
  else SL_REQUIRE(false, "pop_proto_rows: bad dtype");
  SL_LAUNCH_CHECK("pop_proto_rows_kernel");
  return 0;
}

extern "C" int sl_rowdot_fwd(int dtype, const void* h, const float* w, float* z, long long R, int C, sl_stream_t stream) {
  SL_REQUIRE(h && w && z && R > 0, "rowdot_fwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = row_blocks(R * 16, 2048);
  if (int e = pop_dispatch(dtype, C, [&](auto t, auto nv, auto lpr) {
        using T = decltype(t);
        hipLaunchKernelGGL((rowdot_fwd_kernel<T, decltype(nv)::value, decltype(lpr)::value>), dim3(blocks), dim3(256), 0, st, (const T*)h, w, z, R, C);
        return 0; })) return e;
  SL_LAUNCH_CHECK("rowdot_fwd_kernel");
  return 0;
}

extern "C" int sl_rowdot_bwd_rows(long long R, int C) { (void)C; return row_blocks(R, 512); }

extern "C" int sl_rowdot_bwd(int dtype, const void* h, const float* w, const float* dz, void* dh, float* partial, long long R,
                             int C, sl_stream_t stream) {
  SL_REQUIRE(h && w && dz && dh && partial && R > 0, "rowdot_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = row_blocks(R, 512);
  const long long rpb = (R + nblk - 1) / nblk;
  const size_t lds = 4 * (size_t)C * sizeof(float);
  if (int e = pop_dispatch(dtype, C, [&](auto t, auto nv, auto lpr) {
        using T = decltype(t);
        hipLaunchKernelGGL((rowdot_bwd_kernel<T, decltype(nv)::value, decltype(lpr)::value>), dim3(nblk), dim3(256), lds, st, (const T*)h, w, dz, (T*)dh, partial, R, C, rpb);
        return 0; })) return e;
  SL_LAUNCH_CHECK("rowdot_bwd_kernel");
  return 0;
}


extern "C" int sl_colsum_rows_blocks(long long rows, int C, int dtype) {
  if (rows <= 0 || C <= 0) return 0;
  const long long ch = sl_colsum_rows_chunk(rows, C, dtype == SL_BF16 ? 2 : 4);
  return (int)((rows + ch - 1) / ch);
}
extern "C" int sl_colsum_rows_partial(int dtype, const void* x, long long rows, int C, float* partial, sl_stream_t stream) {
  SL_REQUIRE(x && partial && rows > 0 && C > 0 && (dtype == SL_BF16 ? C % 8 == 0 : C % 4 == 0), "colsum_rows_partial: bad args");
  const long long ch = sl_colsum_rows_chunk(rows, C, dtype == SL_BF16 ? 2 : 4);
  const int nblk = (int)((rows + ch - 1) / ch);
  if (dtype == SL_BF16) hipLaunchKernelGGL(colsum_rows_partial_kernel<bf16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, rows, C, ch, partial);
  else if (dtype == SL_F32) hipLaunchKernelGGL(colsum_rows_partial_kernel<float>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const float*)x, rows, C, ch, partial);
  else SL_REQUIRE(false, "colsum_rows_partial: bad dtype");
  SL_LAUNCH_CHECK("colsum_rows_partial_kernel");
  return 0;
}

extern "C" int sl_colsum_finalize(const float* partial, int nblk, int C, float* out, sl_stream_t stream) {
  SL_REQUIRE(partial && out && nblk > 0 && C > 0, "colsum_finalize: bad args");
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, (hipStream_t)stream, partial, nblk, C, out);
  SL_LAUNCH_CHECK("colsum_finalize_kernel");
  return 0;
}

extern "C" int sl_colsum_finalize_multi(const SlColsumBatch* batch, sl_stream_t stream) {
  SL_REQUIRE(batch && batch->n > 0 && batch->n <= SL_COLSUM_MAX, "colsum_finalize_multi: 1..%d entries", SL_COLSUM_MAX);
  ColsumBatchDev b{};
  b.n = batch->n;
  int blocks = 0;
  for (int i = 0; i < batch->n; ++i) {
    SL_REQUIRE(batch->part[i] && batch->out[i] && batch->nblk[i] > 0 && batch->C[i] > 0, "colsum_finalize_multi: bad entry %d", i);
    b.first[i] = blocks; b.part[i] = batch->part[i]; b.out[i] = batch->out[i]; b.nblk[i] = batch->nblk[i]; b.C[i] = batch->C[i];
    blocks += cdiv(batch->C[i], 64);
  }
  b.first[batch->n] = blocks;
  hipLaunchKernelGGL(colsum_finalize_multi_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, b);
  SL_LAUNCH_CHECK("colsum_finalize_multi_kernel");
  return 0;
}

extern "C" int sl_colsum_f64(const float* partial, int nblk, int C, double* out, sl_stream_t stream) {
  SL_REQUIRE(partial && out && nblk > 0 && C > 0, "colsum_f64: bad args");
  hipLaunchKernelGGL(colsum_f64_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, (hipStream_t)stream, partial, nblk, C, out);
  SL_LAUNCH_CHECK("colsum_f64_kernel");
  return 0;
}

extern "C" int sl_f64_split(const double* totals, int n, float* hi_lo, sl_stream_t stream) {
  SL_REQUIRE(totals && hi_lo && n > 0, "f64_split: bad args");
  hipLaunchKernelGGL(f64_split_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, totals, n, hi_lo);
  SL_LAUNCH_CHECK("f64_split_kernel");
  return 0;
}

extern "C" int sl_pop_combine_fwd(const float* proj, const float* z_bg, const float* a, const float* b, int Kt, float* preds,
                                  int B, int N, sl_stream_t stream) {
  SL_REQUIRE(proj && z_bg && a && b && preds && Kt >= 1 && Kt <= KMAXP && B > 0 && N > 0, "pop_combine_fwd: bad args");
  const long long R = (long long)B * N;
  const int blocks = (int)((R + 255) / 256 < 4096 ? (R + 255) / 256 : 4096);
  hipLaunchKernelGGL(pop_combine_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, proj, z_bg, a, b, Kt, preds, B, N);
  SL_LAUNCH_CHECK("pop_combine_fwd_kernel");
  return 0;
}

extern "C" int sl_pop_combine_bwd_rows(int B, int N) { return row_blocks((long long)B * N / 4, 256); }

extern "C" int sl_pop_combine_bwd(const float* dpreds, const float* proj, const float* a, const float* b, int Kt, float* dz_bg,
                                  float* dproj, float* dab_partial, int B, int N, sl_stream_t stream) {
  SL_REQUIRE(dpreds && proj && a && b && dz_bg && dproj && dab_partial && Kt >= 1 && Kt <= KMAXP, "pop_combine_bwd: bad args");
  const long long R = (long long)B * N;
  const int nblk = sl_pop_combine_bwd_rows(B, N);
  const long long rpb = (R + nblk - 1) / nblk;
  hipLaunchKernelGGL(pop_combine_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dpreds, proj, a, b, Kt, dz_bg, dproj, dab_partial, B, N, rpb);
  SL_LAUNCH_CHECK("pop_combine_bwd_kernel");
  return 0;
}

static int dec_bwd_cap() { return 1024; }     // measured 180 / 160 / 161 us for 512 / 1024 / 2048 blocks at 65 536 rows
// 16 rows per block until the cap binds (from 16 384 rows on: 64 rows per block and more, as before).  A wave walks its rows one after the other and every row is a chain of
// dependent loads and shuffles, so the launch time is (rows per wave) x (row latency) whenever there are too few blocks to hide it: the fine-tune pair's 8 192 rows took
// 120 us on 128 blocks of 64 rows -- as long as 65 536 rows on 1 024 blocks.
static int dec_bwd_blocks(long long R) { return row_blocks(R * 4, dec_bwd_cap()); }
extern "C" int sl_pop_decompose_bwd_rows(long long R) { return dec_bwd_blocks(R); }

extern "C" int sl_pop_decompose_bwd(int dtype, const void* dg, const void* feats, const float* S, const float* proj,
                                    const float* dproj, int Kt, void* dq, float* dS_partial, long long R, int C,
                                    sl_stream_t stream) {
  SL_REQUIRE(dg && feats && S && proj && dproj && dq && dS_partial && R > 0 && Kt >= 1 && Kt <= KMAXP, "pop_decompose_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = dec_bwd_blocks(R);
  const long long rpb = (R + nblk - 1) / nblk;
  if (Kt <= 8) {
    const size_t lds = 8 * (size_t)C * sizeof(float);
    if (int e = pop_dispatch(dtype, C, [&](auto t, auto nv, auto lpr) {
          using T = decltype(t);
          hipLaunchKernelGGL((pop_decompose_bwd_kernel<T, decltype(nv)::value, 8, decltype(lpr)::value>), dim3(nblk), dim3(256), lds, st, (const T*)dg, (const T*)feats, S, proj, dproj, Kt, (T*)dq, dS_partial, R, C, rpb);
          return 0; })) return e;
  } else {
    const size_t lds = 16 * (size_t)C * sizeof(float);
    if (int e = pop_dispatch(dtype, C, [&](auto t, auto nv, auto lpr) {
          using T = decltype(t);
          hipLaunchKernelGGL((pop_decompose_bwd_kernel<T, decltype(nv)::value, 16, decltype(lpr)::value>), dim3(nblk), dim3(256), lds, st, (const T*)dg, (const T*)feats, S, proj, dproj, Kt, (T*)dq, dS_partial, R, C, rpb);
          return 0; })) return e;
  }
  SL_LAUNCH_CHECK("pop_decompose_bwd_kernel");
  return 0;
}
