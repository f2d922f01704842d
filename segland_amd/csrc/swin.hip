// Swin-POP path on gfx950: the token-side kernels of networks/backbones/swintransformer.py and the
// resize / gather helpers of networks/swin_pop.py's UperNet_Decoder_Plus.  The dense contractions (qkv / proj / fc1 / fc2 / patch-merge
// reduction Linear layers, 3x3 decoder convs) run on the MFMA implicit-GEMM kernels of conv_gemm*.hip / conv_wgrad.hip: a token map
// [B,H,W,Cp] IS an NHWC image and nn.Linear a 1x1 conv.  Channel counts that are not multiples of 64 (Swin-T/S: C = 96) are carried with
// a zero-filled channel pad (pitch Cp = 128); every kernel here takes (C, pitch) and keeps the pad at exactly zero.
//
//   patch_embed_*      swintransformer.py:395-433   conv 4x4 s4 (3 -> C) straight from the NCHW float image (K = 48: VALU, LDS-tiled)
//   layernorm_*        nn.LayerNorm (eps 1e-5) over C: rows owned by 16/32/64-lane groups, two-pass statistics, shuffle reductions
//   window_attention_* swintransformer.py:118-149 + the pad / cyclic shift / window partition / mask of :208-238,363-379 as index arithmetic:
//                      one workgroup per (window, head); q, k, v, scores in LDS; softmax(q k^T * scale + bias + mask) v; the backward
//                      recomputes the probabilities (nothing but q, k, v is kept from the forward)
//   gelu_*             exact (erf) GELU of Mlp (:36)
//   merge_gather/scatter  PatchMerging's 2x2 space-to-depth (:280-284)
//   bilinear_*         F.interpolate(mode='bilinear') NHWC, both align_corners modes, optional accumulation into the destination and
//                      channel windows (swin_pop.py:33,150-153,167 and the nn.Upsample of :133-137); backward in gather form (bit-stable)
//   scale_add          x + s[b] * branch (DropPath, timm) and y * m[b][c] (nn.Dropout2d, swin_pop.py:21)
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int WS = 7, WN = 49, HD = 32;       // window edge, tokens per window, head dimension (C / heads = 32 for every Swin variant)
constexpr int QP = 36;                        // LDS row pitch of a [49][32] tile (floats): 16-byte aligned, conflict-free ds_read_b128

inline int ew_grid(long long n) { long long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }

template <int LPR> __device__ __forceinline__ float grp_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------ patch embedding
// out[b][ty][tx][c] = bias[c] + sum_{ci,ky,kx} w[c][ci][ky][kx] * img[b][ci][4ty+ky][4tx+kx]   (zero outside the image, :417-421)
template <typename T>
__global__ __launch_bounds__(256) void patch_embed_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
                                                              T* __restrict__ out, int B, int H, int W, int Ho, int Wo, int C, int Cp) {
  constexpr int V = Vec16<T>::N;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wl = sm;                 // [48][C]
  float* pl = sm + 48 * C;        // [64][49] (odd pitch: the 4 threads of a token broadcast, tokens spread over banks)
  const int tid = threadIdx.x;
  for (int e = tid; e < 48 * C; e += 256) { const int c = e / 48, k = e % 48; wl[k * C + c] = w[e]; }
  const long long ntok = (long long)B * Ho * Wo;
  const long long t0 = (long long)blockIdx.x * 64;
  for (int e = tid; e < 64 * 48; e += 256) {
    const int t = e / 48, k = e % 48;
    const long long tok = t0 + t;
    float v = 0.f;
    if (tok < ntok) {
      const int tx = (int)(tok % Wo), ty = (int)((tok / Wo) % Ho), b = (int)(tok / ((long long)Wo * Ho));
      const int ci = k / 16, ky = (k % 16) / 4, kx = k % 4;
      const int y = 4 * ty + ky, x = 4 * tx + kx;
      if (y < H && x < W) v = img[((size_t)(b * 3 + ci) * H + y) * W + x];
    }
    pl[t * 49 + k] = v;
  }
  __syncthreads();
  const int t = tid >> 2, cg = tid & 3, cpt = C / 4;       // 64 tokens x 4 channel groups
  const long long tok = t0 + t;
  if (tok >= ntok) return;
  T* o = out + (size_t)tok * Cp;
  for (int c0 = cg * cpt; c0 < (cg + 1) * cpt; c0 += 8) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bias[c0 + j];
    for (int k = 0; k < 48; ++k) {
      const float a = pl[t * 49 + k];
      const float4 w0 = *(const float4*)(wl + k * C + c0), w1 = *(const float4*)(wl + k * C + c0 + 4);
      acc[0] = fmaf(a, w0.x, acc[0]); acc[1] = fmaf(a, w0.y, acc[1]); acc[2] = fmaf(a, w0.z, acc[2]); acc[3] = fmaf(a, w0.w, acc[3]);
      acc[4] = fmaf(a, w1.x, acc[4]); acc[5] = fmaf(a, w1.y, acc[5]); acc[6] = fmaf(a, w1.z, acc[6]); acc[7] = fmaf(a, w1.w, acc[7]);
    }
#pragma unroll
    for (int j = 0; j < 8; j += V) *(uint4*)(o + c0 + j) = pack16<T>(&acc[j]);
  }
  if (cg == 3) {
    float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = C; c < Cp; c += V) *(uint4*)(o + c) = pack16<T>(z);
  }
}

// col[t][k] = img patch value k = ci*16 + ky*4 + kx of token t (k < 48), zero for k in [48, 64): the weight gradient of the patch embedding is
// then the MFMA weight-gradient GEMM dW[c][k] = sum_t dy[t][c] * col[t][k] (sl_conv2d_bwd_weight on [1,1,T,64] x [1,1,T,P])
template <typename T>
__global__ void patch_im2col_kernel(const float* __restrict__ img, T* __restrict__ col, int B, int H, int W, int Ho, int Wo) {
  const long long total = (long long)B * Ho * Wo * 16;          // 16 groups of 4 consecutive k (= one kernel row of one input channel)
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int gq = (int)(i % 16); const long long tok = i / 16;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (gq < 12) {
      const int tx = (int)(tok % Wo), ty = (int)((tok / Wo) % Ho), b = (int)(tok / ((long long)Wo * Ho));
      const int ci = gq / 4, ky = gq % 4, y = 4 * ty + ky;
      if (y < H) {
        const float* src = img + ((size_t)(b * 3 + ci) * H + y) * W + 4 * tx;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) if (4 * tx + kx < W) v[kx] = src[kx];
      }
    }
    T* o = col + (size_t)tok * 64 + gq * 4;
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) o[kx] = from_f<T>(v[kx]);
  }
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// rows of C channels (pitch px / py); a row belongs to LPR lanes; stats[row] = {mean, rstd}
// NVL > 0: a lane keeps its (at most NVL) channel vectors of the row in registers -- x is read once (round 5: the three passes re-read it from the L1 / L2, three dependent
// load latencies per row; the 8 192- and 2 048-token maps of stages 3 / 4 ran at 0.6-1.5 TB/s); same arithmetic in the same order as the streaming form (NVL = 0).
template <typename T, int LPR, int NVL>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            T* __restrict__ y, float* __restrict__ stats, long long rows, int C, int px, int py, float eps) {
  constexpr int V = Vec16<T>::N, RPW = 64 / LPR, NR = NVL > 0 ? NVL : 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPR, grp = lane / LPR;
  const int nvec = C / V;
  for (long long rb = ((long long)blockIdx.x * 4 + wave) * RPW; rb < rows; rb += (long long)gridDim.x * 4 * RPW) {
    const long long r = rb + grp;
    const bool live = r < rows;
    const T* xr = x + (size_t)(live ? r : 0) * px;
    float s = 0.f;
    uint4 xv[NR];
    if constexpr (NVL > 0) {
#pragma unroll
      for (int j = 0; j < NVL; ++j) { const int v = sub + j * LPR; if (v < nvec) xv[j] = *(const uint4*)(xr + v * V); }
#pragma unroll
      for (int j = 0; j < NVL; ++j) {
        if (sub + j * LPR < nvec) {
          float t[V];
          unpack16<T>(xv[j], t);
#pragma unroll
          for (int e = 0; e < V; ++e) s += t[e];
        }
      }
    } else {
      for (int v = sub; v < nvec; v += LPR) {
        float t[V];
        unpack16<T>(*(const uint4*)(xr + v * V), t);
#pragma unroll
        for (int e = 0; e < V; ++e) s += t[e];
      }
    }
    const float mean = grp_sum<LPR>(s) / (float)C;
    float q = 0.f;
    if constexpr (NVL > 0) {
#pragma unroll
      for (int j = 0; j < NVL; ++j) {
        if (sub + j * LPR < nvec) {
          float t[V];
          unpack16<T>(xv[j], t);
#pragma unroll
          for (int e = 0; e < V; ++e) { const float d = t[e] - mean; q = fmaf(d, d, q); }
        }
      }
    } else {
      for (int v = sub; v < nvec; v += LPR) {
        float t[V];
        unpack16<T>(*(const uint4*)(xr + v * V), t);
#pragma unroll
        for (int e = 0; e < V; ++e) { const float d = t[e] - mean; q = fmaf(d, d, q); }
      }
    }
    const float rstd = rsqrtf(grp_sum<LPR>(q) / (float)C + eps);
    if (!live) continue;
    if (sub == 0 && stats) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
    T* yr = y + (size_t)r * py;
    auto emit = [&](int v, const uint4* held) {
      float o[V];
      if (v < nvec) {
        float t[V];
        unpack16<T>(held ? *held : *(const uint4*)(xr + v * V), t);
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = (t[e] - mean) * rstd * gamma[v * V + e] + beta[v * V + e];
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = 0.f;
      }
      *(uint4*)(yr + v * V) = pack16<T>(o);
    };
    if constexpr (NVL > 0) {
#pragma unroll
      for (int j = 0; j < NVL; ++j) { const int v = sub + j * LPR; if (v < py / V) emit(v, &xv[j]); }
    } else {
      for (int v = sub; v < py / V; v += LPR) emit(v, nullptr);
    }
  }
}

// dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)) (+ addend),  g = dy * gamma,  xhat = (x - mean) * rstd.
// NV > 0: the lane's (at most NV) channel vectors also accumulate the column sums part[blk][0][c] = sum_rows dy * xhat (dgamma) and
// part[blk][1][c] = sum_rows dy (dbeta) in registers -- no second pass over dy and x; NV = 0: rows wider than NV_MAX * LPR vectors
// leave the column sums to layernorm_bwd_cols_kernel.
template <int LPR> __device__ __forceinline__ float grp_across(float v) {      // same channel, the 64/LPR rows of a wavefront
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int LPR, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ stats, const T* __restrict__ addend, T* __restrict__ dx,
                                                            float* __restrict__ part, long long rows, int C, int pdy, int px, int pdx,
                                                            T* __restrict__ dx2, const float* __restrict__ rscale, long long rps) {
  // dx2 (optional): the ROUNDED dx times rscale[row / rps] -- DropPath's per-sample factor for the branch this gradient enters next (what a scale_add launch over dx
  // would write, bit for bit)
  constexpr int V = Vec16<T>::N, RPW = 64 / LPR, NA = NV > 0 ? NV : 1;
  extern __shared__ __attribute__((aligned(16))) float red[];           // [4][2][C] when NV > 0
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPR, grp = lane / LPR;
  const int nvec = C / V;
  float ag[NA][V], ab[NA][V];
#pragma unroll
  for (int j = 0; j < NA; ++j)
#pragma unroll
    for (int e = 0; e < V; ++e) { ag[j][e] = 0.f; ab[j][e] = 0.f; }
  for (long long rb = ((long long)blockIdx.x * 4 + wave) * RPW; rb < rows; rb += (long long)gridDim.x * 4 * RPW) {
    const long long r = rb + grp;
    const bool live = r < rows;
    const long long rr = live ? r : 0;
    const T* dyr = dy + (size_t)rr * pdy;
    const T* xr = x + (size_t)rr * px;
    const float mean = stats[2 * rr], rstd = stats[2 * rr + 1];
    float s1 = 0.f, s2 = 0.f;
    uint4 gv[NA], tv[NA];               // NV > 0: the lane's vectors of dy and x stay in registers between the two passes (round 5; same arithmetic, same order)
    if constexpr (NV > 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) { const int v = sub + j * LPR; if (v < nvec) { gv[j] = *(const uint4*)(dyr + v * V); tv[j] = *(const uint4*)(xr + v * V); } }
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int v = sub + j * LPR;
        if (v < nvec) {
          float g[V], t[V];
          unpack16<T>(gv[j], g);
          unpack16<T>(tv[j], t);
#pragma unroll
          for (int e = 0; e < V; ++e) { const float ge = g[e] * gamma[v * V + e]; s1 += ge; s2 = fmaf(ge, (t[e] - mean) * rstd, s2); }
        }
      }
    } else {
      for (int v = sub; v < nvec; v += LPR) {
        float g[V], t[V];
        unpack16<T>(*(const uint4*)(dyr + v * V), g);
        unpack16<T>(*(const uint4*)(xr + v * V), t);
#pragma unroll
        for (int e = 0; e < V; ++e) { const float ge = g[e] * gamma[v * V + e]; s1 += ge; s2 = fmaf(ge, (t[e] - mean) * rstd, s2); }
      }
    }
    s1 = grp_sum<LPR>(s1) / (float)C;
    s2 = grp_sum<LPR>(s2) / (float)C;
    if (!live) continue;
    T* dxr = dx + (size_t)r * pdx;
    const float rs = dx2 ? rscale[r / rps] : 0.f;
    auto emit = [&](int j) {
      const int v = sub + j * LPR;
      if (v >= pdx / V) return;
      float o[V];
      if (v < nvec) {
        float g[V], t[V];
        if constexpr (NV > 0) { unpack16<T>(gv[j < NA ? j : 0], g); unpack16<T>(tv[j < NA ? j : 0], t); }
        else { unpack16<T>(*(const uint4*)(dyr + v * V), g); unpack16<T>(*(const uint4*)(xr + v * V), t); }
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float xh = (t[e] - mean) * rstd;
          o[e] = rstd * (g[e] * gamma[v * V + e] - s1 - xh * s2);
          if (NV > 0) { ag[j < NA ? j : 0][e] = fmaf(g[e], xh, ag[j < NA ? j : 0][e]); ab[j < NA ? j : 0][e] += g[e]; }
        }
        if (addend) {
          float a[V];
          unpack16<T>(*(const uint4*)(addend + (size_t)r * pdx + v * V), a);
#pragma unroll
          for (int e = 0; e < V; ++e) o[e] += a[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = 0.f;
      }
      const uint4 pk = pack16<T>(o);
      *(uint4*)(dxr + v * V) = pk;
      if (dx2) {
        unpack16<T>(pk, o);
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] *= rs;
        *(uint4*)(dx2 + (size_t)r * pdx + v * V) = pack16<T>(o);
      }
    };
    if constexpr (NV > 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) emit(j);
    } else {
      for (int j = 0; sub + j * LPR < pdx / V; ++j) emit(j);          // (a 64-fold unrolled loop here cost the dx-only form 300 registers)
    }
  }
  if (NV > 0 && part) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int v = sub + j * LPR;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float a = grp_across<LPR>(ag[j][e]), b = grp_across<LPR>(ab[j][e]);
        if (grp == 0 && v < nvec) { red[(wave * 2 + 0) * C + v * V + e] = a; red[(wave * 2 + 1) * C + v * V + e] = b; }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
      const int which = i / C, c = i % C;
      part[((size_t)blockIdx.x * 2 + which) * C + c] = red[(0 * 2 + which) * C + c] + red[(1 * 2 + which) * C + c] + red[(2 * 2 + which) * C + c] + red[(3 * 2 + which) * C + c];
    }
  }
}

// part[blk][0][c] = sum_rows dy * xhat (dgamma), part[blk][1][c] = sum_rows dy (dbeta): thread per channel, rows of the block's chunk
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_cols_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ stats,
                                                                 float* __restrict__ part, long long rows, int C, int pdy, int px, long long rows_per_blk) {
  const long long r0 = blockIdx.x * rows_per_blk;
  long long r1 = r0 + rows_per_blk; if (r1 > rows) r1 = rows;
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (long long r = r0; r < r1; ++r) {
      const float g = to_f<T>(dy[(size_t)r * pdy + c]);
      a = fmaf(g, (to_f<T>(x[(size_t)r * px + c]) - stats[2 * r]) * stats[2 * r + 1], a);
      b += g;
    }
    part[((size_t)blockIdx.x * 2 + 0) * C + c] = a;
    part[((size_t)blockIdx.x * 2 + 1) * C + c] = b;
  }
}

// ------------------------------------------------------------------------------------------------ GELU (exact)
template <typename T>
__global__ void gelu_fwd_kernel(const T* __restrict__ h, T* __restrict__ y, long long nvec) {
  constexpr int V = Vec16<T>::N;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    float t[V];
    unpack16<T>(((const uint4*)h)[i], t);
#pragma unroll
    for (int e = 0; e < V; ++e) t[e] = sl_gelu<T>(t[e]);
    ((uint4*)y)[i] = pack16<T>(t);
  }
}
template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ h, const T* __restrict__ dy, T* __restrict__ dh, long long nvec) {
  constexpr int V = Vec16<T>::N;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    float t[V], g[V];
    unpack16<T>(((const uint4*)h)[i], t);
    unpack16<T>(((const uint4*)dy)[i], g);
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] *= sl_gelu_grad<T>(t[e]);
    ((uint4*)dh)[i] = pack16<T>(g);
  }
}

// ------------------------------------------------------------------------------------------------ patch merging gather / scatter
// xm[b][i][j][q*C + c] = x[b][2i + (q&1)][2j + (q>>1)][c]  (zero beyond H, W: the odd-size padding of :277-278)
template <typename T>
__global__ void merge_gather_kernel(const T* __restrict__ x, T* __restrict__ xm, int B, int H, int W, int C, int Cp) {
  constexpr int V = Vec16<T>::N;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2, nv = C / V;
  const long long total = (long long)B * H2 * W2 * 4 * nv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % nv); long long r = i / nv;
    const int q = (int)(r % 4); r /= 4;
    const int j = (int)(r % W2); r /= W2;
    const int ii = (int)(r % H2); const int b = (int)(r / H2);
    const int y = 2 * ii + (q & 1), xx = 2 * j + (q >> 1);
    uint4 val = make_uint4(0, 0, 0, 0);
    if (y < H && xx < W) val = *(const uint4*)(x + ((size_t)(b * H + y) * W + xx) * Cp + v * V);
    *(uint4*)(xm + (((size_t)(b * H2 + ii) * W2 + j) * 4 + q) * C + v * V) = val;
  }
}
template <typename T>
__global__ void merge_scatter_kernel(const T* __restrict__ dxm, T* __restrict__ dx, int B, int H, int W, int C, int Cp) {
  constexpr int V = Vec16<T>::N;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2, nv = Cp / V, nvc = C / V;
  const long long total = (long long)B * H * W * nv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % nv); long long r = i / nv;
    const int xx = (int)(r % W); r /= W;
    const int y = (int)(r % H); const int b = (int)(r / H);
    uint4 val = make_uint4(0, 0, 0, 0);
    if (v < nvc) {
      const int q = (y & 1) + 2 * (xx & 1);
      val = *(const uint4*)(dxm + (((size_t)(b * H2 + y / 2) * W2 + xx / 2) * 4 + q) * C + v * V);
    }
    *(uint4*)(dx + ((size_t)(b * H + y) * W + xx) * Cp + v * V) = val;
  }
}

// ------------------------------------------------------------------------------------------------ bilinear resize (NHWC)
// ATen's upsample_bilinear2d source index (UpSample.h area_pixel_compute_source_index), float arithmetic
__device__ __forceinline__ float src_scale(int in, int out, int align) {
  if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return (float)in / (float)out;
}
__device__ __forceinline__ void src_taps(int dst, float scale, int in, int align, int& i0, int& i1, float& l0, float& l1) {
  float s = align ? scale * (float)dst : fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.f);
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0; l0 = 1.f - l1;
}

// V consecutive channels of a TS tensor as floats (TS = T: one 16-byte vector; TS = float beside T = bf16: two)
template <typename TS, int V> __device__ __forceinline__ void ldv(const TS* p, float* o) {
  if constexpr (sizeof(TS) == 4) {
#pragma unroll
    for (int e = 0; e < V; e += 4) { const float4 t = *(const float4*)(p + e); o[e] = t.x; o[e + 1] = t.y; o[e + 2] = t.z; o[e + 3] = t.w; }
  } else unpack16<TS>(*(const uint4*)p, o);
}
template <typename TS, int V> __device__ __forceinline__ void stv(TS* p, const float* o) {
  if constexpr (sizeof(TS) == 4) {
#pragma unroll
    for (int e = 0; e < V; e += 4) *(float4*)(p + e) = make_float4(o[e], o[e + 1], o[e + 2], o[e + 3]);
  } else *(uint4*)p = pack16<TS>(o);
}

// y[b][Y][X][yoff + c] = bilinear(x[b][..][..][xoff + c]) (+ add[b][Y][X][yoff + c]), c < C; add may be y itself (accumulate in place)
template <typename T, typename TS>
__global__ void bilinear_fwd_kernel(const TS* __restrict__ x, T* y, int B, int h, int w, int H, int W, int C, int px, int xoff,
                                    int py, int yoff, int align, const T* add) {
  constexpr int V = Vec16<T>::N;
  const int nv = C / V;
  const float sy = src_scale(h, H, align), sx = src_scale(w, W, align);
  const long long total = (long long)B * H * W * nv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % nv); long long r = i / nv;
    const int X = (int)(r % W); r /= W;
    const int Y = (int)(r % H); const int b = (int)(r / H);
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    src_taps(Y, sy, h, align, y0, y1, ly0, ly1);
    src_taps(X, sx, w, align, x0, x1, lx0, lx1);
    const TS* base = x + (size_t)b * h * w * px + xoff + v * V;
    float a[V], bb[V], c[V], d[V], o[V];
    ldv<TS, V>(base + ((size_t)y0 * w + x0) * px, a);
    ldv<TS, V>(base + ((size_t)y0 * w + x1) * px, bb);
    ldv<TS, V>(base + ((size_t)y1 * w + x0) * px, c);
    ldv<TS, V>(base + ((size_t)y1 * w + x1) * px, d);
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] = ly0 * (lx0 * a[e] + lx1 * bb[e]) + ly1 * (lx0 * c[e] + lx1 * d[e]);      // ATen's association
    const size_t doff = ((size_t)(b * H + Y) * W + X) * py + yoff + v * V;
    T* dst = y + doff;
    if (add) {
      float p[V];
      unpack16<T>(*(const uint4*)(add + doff), p);
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] += p[e];
    }
    *(uint4*)dst = pack16<T>(o);
  }
}

// dx[b][y][x][xoff + c] (+)= sum over destination pixels that read source (y, x) of weight * dy[b][Y][X][yoff + c]  (gather form, fixed order)
template <typename T, typename TS>
__global__ void bilinear_bwd_kernel(const T* __restrict__ dy, TS* __restrict__ dx, int B, int h, int w, int H, int W, int C, int pdx, int xoff,
                                    int pdy, int yoff, int align, int accumulate) {
  constexpr int V = Vec16<T>::N;
  const int nv = C / V;
  const float sy = src_scale(h, H, align), sx = src_scale(w, W, align);
  const long long total = (long long)B * h * w * nv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % nv); long long r = i / nv;
    const int xs = (int)(r % w); r /= w;
    const int ys = (int)(r % h); const int b = (int)(r / h);
    // candidate destination rows / columns: source coordinate in (ys-1, ys+1)
    int Ya, Yb, Xa, Xb;
    // align_corners: src = s * dst;  otherwise src = max(s * (dst + 0.5) - 0.5, 0)
    const float lo_y = align ? (float)ys - 1.f : (float)ys - 0.5f, hi_y = align ? (float)ys + 1.f : (float)ys + 1.5f, sh = align ? 0.f : 0.5f;
    const float lo_x = align ? (float)xs - 1.f : (float)xs - 0.5f, hi_x = align ? (float)xs + 1.f : (float)xs + 1.5f;
    if (sy > 0.f) { Ya = (int)floorf(lo_y / sy - sh) - 1; Yb = (int)ceilf(hi_y / sy - sh) + 1; } else { Ya = 0; Yb = H - 1; }
    if (sx > 0.f) { Xa = (int)floorf(lo_x / sx - sh) - 1; Xb = (int)ceilf(hi_x / sx - sh) + 1; } else { Xa = 0; Xb = W - 1; }
    if (!align && ys == 0) Ya = 0;              // the clamp at 0 maps every early destination row onto source row 0
    if (!align && xs == 0) Xa = 0;
    Ya = Ya < 0 ? 0 : Ya; Xa = Xa < 0 ? 0 : Xa; Yb = Yb > H - 1 ? H - 1 : Yb; Xb = Xb > W - 1 ? W - 1 : Xb;
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    const T* base = dy + (size_t)b * H * W * pdy + yoff + v * V;
    for (int Y = Ya; Y <= Yb; ++Y) {
      int y0, y1; float ly0, ly1;
      src_taps(Y, sy, h, align, y0, y1, ly0, ly1);
      const float wy = (y0 == ys ? ly0 : 0.f) + (y1 == ys ? ly1 : 0.f);
      if (wy == 0.f) continue;
      for (int X = Xa; X <= Xb; ++X) {
        int x0, x1; float lx0, lx1;
        src_taps(X, sx, w, align, x0, x1, lx0, lx1);
        const float wx = (x0 == xs ? lx0 : 0.f) + (x1 == xs ? lx1 : 0.f);
        if (wx == 0.f) continue;
        float g[V];
        unpack16<T>(*(const uint4*)(base + ((size_t)Y * W + X) * pdy), g);
        const float wgt = wy * wx;
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = fmaf(wgt, g[e], acc[e]);
      }
    }
    TS* dst = dx + ((size_t)(b * h + ys) * w + xs) * pdx + xoff + v * V;
    if (accumulate) {
      float p[V];
      ldv<TS, V>(dst, p);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += p[e];
    }
    stv<TS, V>(dst, acc);
  }
}

// Few source pixels (the pyramid priors of the decoder: 1 x 1 .. 6 x 6 grids under a 16 x 16 map): the gather form above has one thread per (source pixel, 8 channels)
// walk every destination pixel -- 128 threads x 256 dependent loads for the 1 x 1 level, 145 us.  Here a block owns one source pixel: thread = (channel vector, candidate
// group), the groups walk the candidate destinations interleaved (fixed order) and are summed in group order through the LDS.
template <typename T, typename TS>
__global__ __launch_bounds__(256) void bilinear_bwd_small_kernel(const T* __restrict__ dy, TS* __restrict__ dx, int B, int h, int w, int H, int W, int C, int pdx, int xoff,
                                                                 int pdy, int yoff, int align, int accumulate) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[256 * V];
  const int nv = C / V, G = 256 / nv;                       // nv divides 256 (checked by the launcher)
  const int v = threadIdx.x % nv, grp = threadIdx.x / nv;
  const int xs = blockIdx.x % w, ys = (blockIdx.x / w) % h, b = blockIdx.x / (w * h);
  const float sy = src_scale(h, H, align), sx = src_scale(w, W, align);
  int Ya, Yb, Xa, Xb;
  const float lo_y = align ? (float)ys - 1.f : (float)ys - 0.5f, hi_y = align ? (float)ys + 1.f : (float)ys + 1.5f, sh = align ? 0.f : 0.5f;
  const float lo_x = align ? (float)xs - 1.f : (float)xs - 0.5f, hi_x = align ? (float)xs + 1.f : (float)xs + 1.5f;
  if (sy > 0.f) { Ya = (int)floorf(lo_y / sy - sh) - 1; Yb = (int)ceilf(hi_y / sy - sh) + 1; } else { Ya = 0; Yb = H - 1; }
  if (sx > 0.f) { Xa = (int)floorf(lo_x / sx - sh) - 1; Xb = (int)ceilf(hi_x / sx - sh) + 1; } else { Xa = 0; Xb = W - 1; }
  if (!align && ys == 0) Ya = 0;
  if (!align && xs == 0) Xa = 0;
  Ya = Ya < 0 ? 0 : Ya; Xa = Xa < 0 ? 0 : Xa; Yb = Yb > H - 1 ? H - 1 : Yb; Xb = Xb > W - 1 ? W - 1 : Xb;
  const int nx = Xb - Xa + 1, ncand = (Yb - Ya + 1) * nx;
  float acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = 0.f;
  const T* base = dy + (size_t)b * H * W * pdy + yoff + v * V;
  for (int c = grp; c < ncand; c += G) {
    const int Y = Ya + c / nx, X = Xa + c % nx;
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    src_taps(Y, sy, h, align, y0, y1, ly0, ly1);
    src_taps(X, sx, w, align, x0, x1, lx0, lx1);
    const float wgt = ((y0 == ys ? ly0 : 0.f) + (y1 == ys ? ly1 : 0.f)) * ((x0 == xs ? lx0 : 0.f) + (x1 == xs ? lx1 : 0.f));
    if (wgt == 0.f) continue;
    float g[V];
    unpack16<T>(*(const uint4*)(base + ((size_t)Y * W + X) * pdy), g);
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = fmaf(wgt, g[e], acc[e]);
  }
#pragma unroll
  for (int e = 0; e < V; ++e) red[(grp * nv + v) * V + e] = acc[e];
  __syncthreads();
  if (grp == 0) {
    for (int k = 1; k < G; ++k)
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += red[(k * nv + v) * V + e];
    TS* dst = dx + ((size_t)(b * h + ys) * w + xs) * pdx + xoff + v * V;
    if (accumulate) {
      float p[V];
      ldv<TS, V>(dst, p);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += p[e];
    }
    stv<TS, V>(dst, acc);
  }
}

// ------------------------------------------------------------------------------------------------ per-sample / per-(sample, channel) scaling
// out = (addend ? addend : 0) + x * s,  s = scale[b] (mode 0: DropPath) or scale[b][c] (mode 1: Dropout2d);  rows_per_b rows of pitch P per sample
template <typename T>
__global__ void scale_add_kernel(const T* __restrict__ x, const float* __restrict__ scale, const T* __restrict__ addend, T* __restrict__ out,
                                 long long rows_per_b, int C, int P, int mode, long long total_vec) {
  constexpr int V = Vec16<T>::N;
  const int nv = P / V;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total_vec; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % nv);
    const long long row = i / nv;
    const int b = (int)(row / rows_per_b);
    float t[V];
    unpack16<T>(((const uint4*)x)[i], t);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int c = v * V + e;
      t[e] *= mode == 0 ? scale[b] : (c < C ? scale[(size_t)b * C + c] : 0.f);
    }
    if (addend) {
      float a[V];
      unpack16<T>(((const uint4*)addend)[i], a);
#pragma unroll
      for (int e = 0; e < V; ++e) t[e] += a[e];
    }
    ((uint4*)out)[i] = pack16<T>(t);
  }
}

// ------------------------------------------------------------------------------------------------ window attention
struct WinGeom {
  int B, H, W, Hp, Wp, nWy, nWx, C, heads, P3, Cp, shift;
};

// token n of window (wy, wx) -> pixel of the un-shifted map, or -1 for a pad token; region id of the shift mask (:363-379)
__device__ __forceinline__ void win_token(const WinGeom& g, int wy, int wx, int n, int& pix, int& region) {
  const int Y = wy * WS + n / WS, X = wx * WS + n % WS;              // position in the cyclically shifted, padded map
  int y = Y + g.shift, x = X + g.shift;                             // shifted[Y] = x[(Y + shift) mod Hp]   (torch.roll by -shift)
  if (y >= g.Hp) y -= g.Hp;
  if (x >= g.Wp) x -= g.Wp;
  pix = (y < g.H && x < g.W) ? y * g.W + x : -1;
  const int by = Y < g.Hp - WS ? 0 : (Y < g.Hp - g.shift ? 1 : 2), bx = X < g.Wp - WS ? 0 : (X < g.Wp - g.shift ? 1 : 2);
  region = 3 * by + bx;
}

// loads q (scaled), k, v of one (window, head) into LDS [49][QP] each; pad tokens carry the qkv bias (their x is 0 AFTER norm1, :208-213)
template <typename T>
__device__ __forceinline__ void win_load_qkv(const WinGeom& g, const T* __restrict__ qkv, const float* __restrict__ qkv_bias, int b, int wy, int wx, int head,
                                             float* q, float* k, float* v, int* pixs, int* regs, float scale) {
  constexpr int V = Vec16<T>::N, CPR = HD / V;      // 16-byte chunks per 32-wide head slice
  const int tid = threadIdx.x;
  if (tid < WN) { int p, r; win_token(g, wy, wx, tid, p, r); pixs[tid] = p; regs[tid] = r; }
  __syncthreads();
  for (int e = tid; e < WN * 3 * CPR; e += 256) {
    const int ch = e % CPR, which = (e / CPR) % 3, n = e / (3 * CPR);
    const int col = which * g.C + head * HD + ch * V;
    float t[V];
    const int p = pixs[n];
    if (p >= 0) unpack16<T>(*(const uint4*)(qkv + ((size_t)b * g.H * g.W + p) * g.P3 + col), t);
    else {
#pragma unroll
      for (int j = 0; j < V; ++j) t[j] = to_f<T>(from_f<T>(qkv_bias[col + j]));      // as the GEMM would have stored it
    }
    float* dst = (which == 0 ? q : (which == 1 ? k : v)) + n * QP + ch * V;
    const float m = which == 0 ? scale : 1.f;
#pragma unroll
    for (int j = 0; j < V; ++j) dst[j] = t[j] * m;
  }
  __syncthreads();
}

// S = q k^T + bias + mask, row softmax in place: P[49][50]
__device__ __forceinline__ void win_probs(const WinGeom& g, const float* q, const float* k, const float* __restrict__ rel_bias, int head, const int* regs, float* P) {
  const int tid = threadIdx.x;
  for (int e = tid; e < WN * WN; e += 256) {
    const int i = e / WN, j = e % WN;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const float4 a = *(const float4*)(q + i * QP + d), b = *(const float4*)(k + j * QP + d);
      s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
    }
    s += rel_bias[(size_t)head * WN * WN + e];
    if (g.shift > 0 && regs[i] != regs[j]) s += -100.f;
    P[i * 50 + j] = s;
  }
  __syncthreads();
  // 4 lanes per row
  const int row = tid >> 2, part = tid & 3;
  if (row < WN) {
    float m = -INFINITY;
    for (int j = part; j < WN; j += 4) m = fmaxf(m, P[row * 50 + j]);
    m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64));
    float z = 0.f;
    for (int j = part; j < WN; j += 4) { const float e = __expf(P[row * 50 + j] - m); P[row * 50 + j] = e; z += e; }
    z += __shfl_xor(z, 1, 64); z += __shfl_xor(z, 2, 64);
    const float inv = 1.f / z;
    for (int j = part; j < WN; j += 4) P[row * 50 + j] *= inv;
  }
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void window_attention_fwd_kernel(WinGeom g, const T* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                                   const float* __restrict__ rel_bias, T* __restrict__ out) {
  constexpr int V = Vec16<T>::N;
  __shared__ __attribute__((aligned(16))) float q[WN * QP], k[WN * QP], v[WN * QP], P[WN * 50];
  __shared__ int pixs[WN], regs[WN];
  int blk = blockIdx.x;
  const int head = blk % g.heads; blk /= g.heads;
  const int wx = blk % g.nWx; blk /= g.nWx;
  const int wy = blk % g.nWy; const int b = blk / g.nWy;
  win_load_qkv<T>(g, qkv, qkv_bias, b, wy, wx, head, q, k, v, pixs, regs, rsqrtf((float)HD));
  win_probs(g, q, k, rel_bias, head, regs, P);
  const int tid = threadIdx.x;
  for (int e = tid; e < WN * (HD / V); e += 256) {
    const int i = e / (HD / V), d0 = (e % (HD / V)) * V;
    const int p = pixs[i];
    if (p < 0) continue;
    float o[V];
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = 0.f;
    for (int jj = 0; jj < WN; ++jj) {
      const float pr = P[i * 50 + jj];
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = fmaf(pr, v[jj * QP + d0 + j], o[j]);
    }
    *(uint4*)(out + ((size_t)b * g.H * g.W + p) * g.Cp + head * HD + d0) = pack16<T>(o);
  }
  if (head == 0 && g.Cp > g.C) {            // keep the channel pad at zero
    for (int e = tid; e < WN * ((g.Cp - g.C) / V); e += 256) {
      const int i = e / ((g.Cp - g.C) / V), c = g.C + (e % ((g.Cp - g.C) / V)) * V;
      if (pixs[i] >= 0) *(uint4*)(out + ((size_t)b * g.H * g.W + pixs[i]) * g.Cp + c) = make_uint4(0, 0, 0, 0);
    }
  }
}

// ---- MFMA forward (bf16): one WAVEFRONT per (window, head); the 4 waves of a block take 4 consecutive windows of one head (shared bias tile).
//   S^T[key][query] = K Q^T     v_mfma_f32_32x32x16_bf16, operands straight from global memory (a lane's 16 bytes = 8 consecutive d of its token)
//   softmax over the keys       keys live in the lane's REGISTERS (D layout: lane = column = query), so max / sum are in-lane + one xor-32 shuffle
//   O^T[d][query]   = V^T P^T   the probability registers ARE the second MFMA operand (the MFMA k index may be any permutation as long as both
//                               operands use it: element e of k-step s <-> key 32*ib + 16*s + 8*(e>>2) + 4*(lane>>5) + (e&3) is exactly the D layout's
//                               row of register 8*s + e); only V needs a transposed copy ([d][key] in LDS)
constexpr int VTP = 68;                       // bf16 pitch of the V^T rows: 136 bytes -> conflict-free ds_read_b64 over 32 rows
constexpr int BLP = 52;                       // float pitch of the bias rows (16-byte aligned float4 reads)

__device__ __forceinline__ uint4 win_frag(const WinGeom& g, const bf16_t* __restrict__ qkv, const float* __restrict__ qkv_bias, int b, int wy, int wx,
                                          int head, int which, int n, int dofs) {
  if (n >= WN) return make_uint4(0, 0, 0, 0);
  int pix, reg;
  win_token(g, wy, wx, n, pix, reg);
  const int col = which * g.C + head * HD + dofs;
  if (pix >= 0) return *(const uint4*)(qkv + ((size_t)b * g.H * g.W + pix) * g.P3 + col);
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = qkv_bias[col + e];
  return pack16<bf16_t>(f);
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8v;
__device__ __forceinline__ f32x16_t mfma16(const uint4& first_rows_to_regs, const uint4& second_rows_to_lanes, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, first_rows_to_regs), __builtin_bit_cast(bf16x8v, second_rows_to_lanes), c, 0, 0, 0);
}

// region ids of the 64 (49 real) window tokens packed 4 bits each: word w holds tokens 8w .. 8w+7 (wave-uniform after the reduction)
__device__ __forceinline__ void win_pack_regions(const WinGeom& g, int wy, int wx, int lane, unsigned (&pk)[8]) {
  int pix, reg = 0;
  if (lane < WN) win_token(g, wy, wx, lane, pix, reg);
  if (g.shift == 0) reg = 0;                  // no shift: no mask; one region, so the comparisons below need no branch on the shift
  unsigned v = (unsigned)reg << (4 * (lane & 7));
  v |= __shfl_xor(v, 1, 64); v |= __shfl_xor(v, 2, 64); v |= __shfl_xor(v, 4, 64);
#pragma unroll
  for (int w = 0; w < 8; ++w) pk[w] = __shfl(v, 8 * w, 64);
}

// scores -> probabilities in place (acc[ib][jb][r]: key 32*ib + (r&3) + 8*(r>>2) + 4*hf, query 32*jb + (lane&31)); returns nothing, rows of invalid
// queries hold finite garbage
__device__ __forceinline__ void win_softmax_regs(const WinGeom& g, f32x16_t (&acc)[2][2], const float* biasl, const unsigned (&pk)[8], int lane, float scale) {
  const int l31 = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    const int j = 32 * jb + l31, jc = j < WN ? j : WN - 1;
    const unsigned regq = (pk[jc >> 3] >> (4 * (jc & 7))) & 15u;
    float mx = -INFINITY;
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int i0 = 32 * ib + 8 * m + 4 * hf;
        const float4 bv = *(const float4*)(biasl + jc * BLP + i0);
        const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = i0 + e;
          float sc = acc[ib][jb][4 * m + e] * scale + bb[e];
          { const unsigned rk = (pk[4 * ib + m] >> (16 * hf + 4 * e)) & 15u; sc += rk != regq ? -100.f : 0.f; }
          sc = i < WN ? sc : -INFINITY;
          acc[ib][jb][4 * m + e] = sc;
          mx = fmaxf(mx, sc);
        }
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float z = 0.f;
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float e = __expf(acc[ib][jb][r] - mx); acc[ib][jb][r] = e; z += e; }
    z += __shfl_xor(z, 32, 64);
    const float inv = 1.f / z;
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ib][jb][r] *= inv;
  }
}

typedef __attribute__((ext_vector_type(4))) short win_s16x4_t;
__device__ __forceinline__ uint2 win_lds_tr(const unsigned char* p) {
  const win_s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) win_s16x4_t*)(p));
  return __builtin_bit_cast(uint2, v);
}
// byte offset of 16-byte chunk c (8 d) of token n in a [64][32 d] tile: tokens of a 16-block in the order {0-3, 8-11, 4-7, 12-15} (see above), chunks XOR-swizzled by the
// 4-row group so that 16-byte accesses down a column of lanes and the transpose reads are both conflict-free
__device__ __forceinline__ int win_nat_off(int n, int c) {
  const int row = (n & ~12) | ((n & 4) << 1) | ((n & 8) >> 1);
  return row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
}
__device__ __forceinline__ uint4 win_trfrag(const unsigned char* tile, int lane, int blk, int s2) {     // rows = d (lane & 31), k = tokens 32*blk + 16*s2 .. +15 in D-layout order
  const int g4 = lane >> 4, l = lane & 15;
  const int row = 32 * blk + 16 * s2 + 8 * (g4 >> 1) + (l >> 2), c = 2 * (g4 & 1) + ((l & 3) >> 1);     // LDS row (already in stored order), logical chunk
  const unsigned char* a = tile + row * 64 + ((c ^ ((row >> 2) & 3)) << 4) + (l & 1) * 8;
  const uint2 lo = win_lds_tr(a), hi = win_lds_tr(tile + (row + 4) * 64 + ((c ^ (((row + 4) >> 2) & 3)) << 4) + (l & 1) * 8);
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}
constexpr int WB2_TILE = 64 * 64;             // bytes of one [64 tokens][32 d] bf16 tile
constexpr int WB2_AP = 68;                    // float pitch of the dS accumulation tile: 272-byte rows -> conflict-free 16-byte accesses down a column of lanes

__global__ __launch_bounds__(256) void window_attention_fwd_mfma_kernel(WinGeom g, const bf16_t* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                                        const float* __restrict__ rel_bias, bf16_t* __restrict__ out, int nwin) {
  __shared__ __attribute__((aligned(16))) float biasl[WN * BLP + 16];
  __shared__ __attribute__((aligned(16))) unsigned char vt[4][WB2_TILE];       // V of the wave's window as a [token][32 d] tile (win_nat_off); V^T fragments are transpose reads of it
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
  const int head = blockIdx.x % g.heads, wi = (blockIdx.x / g.heads) * 4 + wave;
  {
    float bv[10];                                           // 2401 = 9.4 x 256: all loads in flight before the first LDS store
#pragma unroll
    for (int u = 0; u < 10; ++u) { const int e = tid + 256 * u; bv[u] = e < WN * WN ? rel_bias[(size_t)head * WN * WN + e] : 0.f; }
#pragma unroll
    for (int u = 0; u < 10; ++u) { const int e = tid + 256 * u; if (e < WN * WN) biasl[(e / WN) * BLP + e % WN] = bv[u]; }
  }
  __syncthreads();
  if (wi >= nwin) return;
  const int wx = wi % g.nWx, wy = (wi / g.nWx) % g.nWy, b = wi / (g.nWx * g.nWy);
  // the 12 fragments of the window (q, k, v x two 32-token blocks x two 16-d steps: a lane's 16 bytes of its token row) straight from global memory, all loads in flight together
  int pixr[2];                                              // pixel of token 32*rb + l31; -1 pad token, -2 no such token
  uint4 fq[2][2], fk[2][2], fv[2][2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int n = 32 * rb + l31;
    int pix = -2, reg;
    if (n < WN) win_token(g, wy, wx, n, pix, reg);
    pixr[rb] = pix;
    const bf16_t* src = qkv + ((size_t)b * g.H * g.W + (pix >= 0 ? pix : 0)) * g.P3 + head * HD + 8 * hf;     // pad / absent tokens read pixel 0 and drop it
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      fq[rb][s2] = *(const uint4*)(src + 16 * s2);
      fk[rb][s2] = *(const uint4*)(src + g.C + 16 * s2);
      fv[rb][s2] = *(const uint4*)(src + 2 * g.C + 16 * s2);
    }
  }
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
      if (pixr[rb] < 0) { const uint4 z = make_uint4(0, 0, 0, 0); fq[rb][s2] = z; fk[rb][s2] = z; fv[rb][s2] = z; }
  if (__any(pixr[0] == -1 || pixr[1] == -1)) {              // edge windows: pad tokens carry the qkv bias as the GEMM would have stored it
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int col = head * HD + 16 * s2 + 8 * hf;
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = qkv_bias[col + e];
        const uint4 bq = pack16<bf16_t>(f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = qkv_bias[g.C + col + e];
        const uint4 bk = pack16<bf16_t>(f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = qkv_bias[2 * g.C + col + e];
        const uint4 bvv = pack16<bf16_t>(f);
        if (pixr[rb] == -1) { fq[rb][s2] = bq; fk[rb][s2] = bk; fv[rb][s2] = bvv; }
      }
  }
  unsigned char* tv = vt[wave];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) *(uint4*)(tv + win_nat_off(32 * rb + l31, 2 * s2 + hf)) = fv[rb][s2];      // rows of tokens 49..63 are zero (their probabilities are 0, garbage could be NaN)
  unsigned pk[8];
  win_pack_regions(g, wy, wx, lane, pk);
  // S^T = K Q^T
  f32x16_t acc[2][2];
#pragma unroll
  for (int ib = 0; ib < 2; ++ib)
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ib][jb][r] = 0.f;
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) acc[ib][jb] = mfma16(fk[ib][s2], fq[jb][s2], acc[ib][jb]);
  win_softmax_regs(g, acc, biasl, pk, lane, rsqrtf((float)HD));
  __builtin_amdgcn_wave_barrier();            // the wave's own LDS writes of V are complete before its reads (same wave: program order + lgkmcnt)
  // O^T = V^T P^T
  f32x16_t o[2];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[jb][r] = 0.f;
#pragma unroll
  for (int ib = 0; ib < 2; ++ib)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const uint4 vf = win_trfrag(tv, lane, ib, s2);
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = acc[ib][jb][8 * s2 + e];
        o[jb] = mfma16(vf, pack16<bf16_t>(pv), o[jb]);
      }
    }
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    const int pix = pixr[jb];
    if (pix < 0) continue;
    bf16_t* dst = out + ((size_t)b * g.H * g.W + pix) * g.Cp + head * HD + 4 * hf;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
      typedef __attribute__((ext_vector_type(2))) float f32x2_t;
      uint2 v;
      v.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){o[jb][4 * m + 0], o[jb][4 * m + 1]}, bf16x2_t));
      v.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){o[jb][4 * m + 2], o[jb][4 * m + 3]}, bf16x2_t));
      *(uint2*)(dst + 8 * m) = v;
    }
    if (head == 0 && g.Cp > g.C && hf == 0)
      for (int c = g.C; c < g.Cp; c += 8) *(uint4*)(out + ((size_t)b * g.H * g.W + pix) * g.Cp + c) = make_uint4(0, 0, 0, 0);
  }
}

// ---- MFMA backward (bf16): one wavefront per (window, head), probabilities recomputed.  Two register layouts of the 64 x 64 score matrix:
//   T-layout (lane = query, registers = keys):  S^T = K Q^T, dP^T = V dO^T  ->  softmax, row sums, dS^T;  dQ^T[d][query] = K^T dS  (dS^T registers
//            are the second MFMA operand, K^T [d][key] comes from LDS);  the sum of dS over the wave's windows = gradient of the position bias
//   N-layout (lane = key, registers = queries): S = Q K^T, dP = dO V^T with the row max / sum / dot of the T-layout handed over through LDS ->
//            P, dS;  dK^T[d][key] = Q^T dS^T,  dV^T[d][key] = dO^T P^T  (Q^T, dO^T [d][query] from LDS)
// Pad tokens' k / v gradients (they ARE the qkv bias) are summed over the pad keys with shuffles and go to pad_part.
__device__ __forceinline__ void lds_transposed(const WinGeom& g, const bf16_t* __restrict__ src, int pitch, int col0, const float* __restrict__ qkv_bias, int bias_col0,
                                               int b, int wy, int wx, int lane, bf16_t* dst) {
  // dst[d][token] (pitch VTP) = src[token][col0 + d]; pad tokens: qkv bias (bias_col0 >= 0) or zero; token columns 49..63 zero
  for (int c = lane; c < WN * 4; c += 64) {
    const int n = c >> 2, d0 = (c & 3) * 8;
    int pix, reg;
    win_token(g, wy, wx, n, pix, reg);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (pix >= 0) v = *(const uint4*)(src + ((size_t)b * g.H * g.W + pix) * pitch + col0 + d0);
    else if (bias_col0 >= 0) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = qkv_bias[bias_col0 + d0 + e];
      v = pack16<bf16_t>(f);
    }
    const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[(d0 + e) * VTP + n] = (bf16_t)((w4[e >> 1] >> (16 * (e & 1))) & 0xffffu);
  }
  for (int e = lane; e < HD * (64 - WN); e += 64) dst[(e / (64 - WN)) * VTP + WN + e % (64 - WN)] = 0;
}

constexpr int NTP = 40;                       // bf16 pitch of the natural [token][d] tiles: 80 bytes -> conflict-free ds_read_b128 over 32 rows
// One pass over a (window, head) slice of src: rows < 49 of nat[token][d] (pitch NTP) and, when tr != nullptr, tr[d][token] (pitch VTP), both from the
// same 16-byte global loads.  Pad tokens: the qkv bias (bias_col0 >= 0) or zero.  Rows / columns 49..63 are zeroed once per kernel by the caller.
__device__ __forceinline__ void lds_fill(const WinGeom& g, const bf16_t* __restrict__ src, int pitch, int col0, const float* __restrict__ qkv_bias, int bias_col0,
                                         int b, int wy, int wx, int lane, bf16_t* nat, bf16_t* tr) {
  for (int c = lane; c < WN * 4; c += 64) {
    const int n = c >> 2, d0 = (c & 3) * 8;
    int pix, reg;
    win_token(g, wy, wx, n, pix, reg);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (pix >= 0) v = *(const uint4*)(src + ((size_t)b * g.H * g.W + pix) * pitch + col0 + d0);
    else if (bias_col0 >= 0) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = qkv_bias[bias_col0 + d0 + e];
      v = pack16<bf16_t>(f);
    }
    *(uint4*)(nat + n * NTP + d0) = v;
    if (tr) {
      const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) tr[(d0 + e) * VTP + n] = (bf16_t)((w4[e >> 1] >> (16 * (e & 1))) & 0xffffu);
    }
  }
}

__device__ __forceinline__ uint4 lds_tfrag(const bf16_t* t, int l31, int hf, int blk, int s2) {     // first-operand fragment: row l31, k in D-layout order
  const bf16_t* r = t + l31 * VTP + 32 * blk + 16 * s2 + 4 * hf;
  const uint2 lo = *(const uint2*)r, hi = *(const uint2*)(r + 8);
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

__device__ __forceinline__ uint4 regs_frag(const f32x16_t& a, int s2) {
  float pv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) pv[e] = a[8 * s2 + e];
  return pack16<bf16_t>(pv);
}

__device__ __forceinline__ void store_dT(bf16_t* base, const f32x16_t& o, int hf, float mul) {        // registers = d (quads of 4), one token per lane
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    uint2 v;
    v.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){o[4 * m + 0] * mul, o[4 * m + 1] * mul}, bf16x2_t));
    v.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){o[4 * m + 2] * mul, o[4 * m + 3] * mul}, bf16x2_t));
    *(uint2*)(base + 8 * m + 4 * hf) = v;
  }
}

// ---- MFMA backward (round 3; the first form -- operands staged in the LDS by 2-byte scatter, 156 KiB, one block per CU -- ran 130 us per launch against 65.6 and was removed in round 4).
// * the 16 natural fragments of a window (q, k, v, dO x two 32-token blocks x two 16-d steps: a lane's 16 bytes of its token row) go from global memory straight to
//   registers, all loads in flight together, and serve both layouts; nothing natural is staged in the LDS;
// * k, q, dO are also written to the LDS as [token][32 d] tiles of 64-byte rows with 16-byte stores from those registers, and the three transposed operands
//   (K^T, Q^T, dO^T: rows = d, k index = tokens) are ds_read_b64_tr_b16 reads of them -- no 2-byte scatter; the tokens of a 16-block sit in the order
//   {0-3, 8-11, 4-7, 12-15} so that the hardware's k order (rows 8*half + e) is the D-layout order of the probability registers they multiply;
// * 72 KiB of LDS per block instead of 156: two blocks per CU, i.e. a second wave on every SIMD under the first one's load / exp / LDS latencies.
template <bool TRACE>
__global__ __launch_bounds__(256, 2) void window_attention_bwd_mfma2_kernel(WinGeom g, const bf16_t* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                                            const float* __restrict__ rel_bias, const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqkv,
                                                                            float* __restrict__ drel_part, float* __restrict__ pad_part, int wpw, int nwin,
                                                                            unsigned long long* __restrict__ trace) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // debug (tools/attn_trace.py, sl_debug_attn_trace): s_memtime of thread 0 at the phase boundaries of the block's first window, [blocks][16]; null in production
  auto stamp = [&](int i) { if constexpr (TRACE) { if (trace && threadIdx.x == 0) trace[(size_t)blockIdx.x * 16 + i] = __builtin_amdgcn_s_memtime(); } };
  stamp(0);
  float* biasl = (float*)smem_raw;                          // [49][52] (+16): bias[query][key]
  unsigned char* tbase = (unsigned char*)(biasl + WN * BLP + 16);      // per wave: K, Q, dO tiles
  float* mzbase = (float*)(tbase + 4 * 3 * WB2_TILE);       // per wave: [64][4] = {max, 1/sum, rowdot, -}
  float* dacc = mzbase + 4 * 256;                           // [49 queries][WB2_AP]: sum of dS over the block's windows (gradient of the position bias)
  const int tid = threadIdx.x, wave = tid >> 6;
  const int head = blockIdx.x % g.heads, chunk = blockIdx.x / g.heads;
  {
    float bv[10];                                           // 2401 = 9.4 x 256: all loads in flight before the first LDS store
#pragma unroll
    for (int u = 0; u < 10; ++u) { const int e = tid + 256 * u; bv[u] = e < WN * WN ? rel_bias[(size_t)head * WN * WN + e] : 0.f; }
#pragma unroll
    for (int u = 0; u < 10; ++u) { const int e = tid + 256 * u; if (e < WN * WN) biasl[(e / WN) * BLP + e % WN] = bv[u]; }
  }
  for (int e = tid; e < WN * WB2_AP; e += 256) dacc[e] = 0.f;
  __syncthreads();
  stamp(1);
  unsigned char* tk = tbase + wave * 3 * WB2_TILE;
  unsigned char* tq = tk + WB2_TILE;
  unsigned char* tg = tq + WB2_TILE;
  float* mz = mzbase + wave * 256;
  const float scale = rsqrtf((float)HD);

  for (int t = 0; t < wpw; ++t) {
    const int wi = (chunk * wpw + t) * 4 + wave;
    if ((chunk * wpw + t) * 4 >= nwin) break;               // block-uniform
    // a wave past the last window runs on all-zero operands (dS = 0 exactly, nothing stored): the block barriers below stay uniform
    const bool active = wi < nwin;
    int lane = tid & 63;
    asm volatile("" : "+v"(lane));                          // per-window addresses are computed per window: hoisted out of this (usually one-trip) loop they only spill
    const int l31 = lane & 31, hf = lane >> 5;
    const int wic = active ? wi : nwin - 1;
    const int wx = wic % g.nWx, wy = (wic / g.nWx) % g.nWy, b = wic / (g.nWx * g.nWy);
    int pixr[2];                                            // pixel of token 32*rb + l31; -1 pad token, -2 no such token
    uint4 fq[2][2], fk[2][2], fv[2][2], fg[2][2];           // [token block][d step]
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int n = 32 * rb + l31;
      int pix = -2, reg;
      if (n < WN && active) win_token(g, wy, wx, n, pix, reg);
      pixr[rb] = pix;
      const size_t tok = (size_t)b * g.H * g.W + (pix >= 0 ? pix : 0);       // pad / absent tokens read pixel 0 and drop it: no branch around the loads
      const bf16_t* src = qkv + tok * g.P3 + head * HD + 8 * hf;
      const bf16_t* gsrc = dout + tok * g.Cp + head * HD + 8 * hf;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        fq[rb][s2] = *(const uint4*)(src + 16 * s2);
        fk[rb][s2] = *(const uint4*)(src + g.C + 16 * s2);
        fv[rb][s2] = *(const uint4*)(src + 2 * g.C + 16 * s2);
        fg[rb][s2] = *(const uint4*)(gsrc + 16 * s2);
      }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        if (pixr[rb] < 0) { const uint4 z = make_uint4(0, 0, 0, 0); fq[rb][s2] = z; fk[rb][s2] = z; fv[rb][s2] = z; fg[rb][s2] = z; }
    if (__any(pixr[0] == -1 || pixr[1] == -1)) {            // edge windows: pad tokens carry the qkv bias as the GEMM would have stored it (their dO is zero)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int col = head * HD + 16 * s2 + 8 * hf;
          float f[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = qkv_bias[col + e];
          const uint4 bq = pack16<bf16_t>(f);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = qkv_bias[g.C + col + e];
          const uint4 bk = pack16<bf16_t>(f);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = qkv_bias[2 * g.C + col + e];
          const uint4 bvv = pack16<bf16_t>(f);
          if (pixr[rb] == -1) { fq[rb][s2] = bq; fk[rb][s2] = bk; fv[rb][s2] = bvv; }
        }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int o = win_nat_off(32 * rb + l31, 2 * s2 + hf);
        *(uint4*)(tk + o) = fk[rb][s2];
        *(uint4*)(tq + o) = fq[rb][s2];
        *(uint4*)(tg + o) = fg[rb][s2];
      }
    }
    __builtin_amdgcn_sched_barrier(0);                      // q / dO live in the tiles from here on: only k (T-layout) and v stay in registers
    unsigned pk[8];
    win_pack_regions(g, wy, wx, lane, pk);
    __builtin_amdgcn_wave_barrier();
    if (t == 0) stamp(2);
    // ---------------- T-layout, one 32-query block at a time
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
      f32x16_t st[2], dp[2];
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[ib][r] = 0.f; dp[ib][r] = 0.f; }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const uint4 qv = *(const uint4*)(tq + win_nat_off(32 * jb + l31, 2 * s2 + hf)), gv = *(const uint4*)(tg + win_nat_off(32 * jb + l31, 2 * s2 + hf));
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
          st[ib] = mfma16(fk[ib][s2], qv, st[ib]);
          dp[ib] = mfma16(fv[ib][s2], gv, dp[ib]);
        }
      }
      const int j = 32 * jb + l31, jc = j < WN ? j : WN - 1;
      const unsigned regq = (pk[jc >> 3] >> (4 * (jc & 7))) & 15u;
      float mx = -INFINITY;
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int i0 = 32 * ib + 8 * m + 4 * hf;
          const float4 bv = *(const float4*)(biasl + jc * BLP + i0);
          const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float sc = st[ib][4 * m + e] * scale + bb[e];
            { const unsigned rk = (pk[4 * ib + m] >> (16 * hf + 4 * e)) & 15u; sc += rk != regq ? -100.f : 0.f; }
            sc = (i0 + e) < WN ? sc : -INFINITY;
            st[ib][4 * m + e] = sc;
            mx = fmaxf(mx, sc);
          }
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float z = 0.f;
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float e = __expf(st[ib][r] - mx); st[ib][r] = e; z += e; }
      z += __shfl_xor(z, 32, 64);
      const float inv = 1.f / z;
      float rs = 0.f;
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[ib][r] *= inv; rs = fmaf(st[ib][r], dp[ib][r], rs); }
      rs += __shfl_xor(rs, 32, 64);
      if (hf == 0) { mz[4 * j + 0] = mx; mz[4 * j + 1] = inv; mz[4 * j + 2] = rs; }
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[ib][r] = st[ib][r] * (dp[ib][r] - rs);
      if (t == 0) stamp(3 + 3 * jb);
      // sum of dS over the block's windows: the four waves add their registers to the shared tile one after the other (fixed order -> bit-stable)
      for (int w = 0; w < 4; ++w) {
        if (wave == w && j < WN) {
#pragma unroll
          for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              float4* a = (float4*)(dacc + j * WB2_AP + 32 * ib + 8 * m + 4 * hf);
              float4 v = *a;
              v.x += dp[ib][4 * m + 0]; v.y += dp[ib][4 * m + 1]; v.z += dp[ib][4 * m + 2]; v.w += dp[ib][4 * m + 3];
              *a = v;
            }
        }
        __syncthreads();
      }
      if (t == 0) stamp(4 + 3 * jb);
      // dQ^T[d][query] = sum_key K^T[d][key] dS[query][key]
      f32x16_t oq;
#pragma unroll
      for (int r = 0; r < 16; ++r) oq[r] = 0.f;
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) oq = mfma16(win_trfrag(tk, lane, ib, s2), regs_frag(dp[ib], s2), oq);
      if (pixr[jb] >= 0) {
        bf16_t* dst = dqkv + ((size_t)b * g.H * g.W + pixr[jb]) * g.P3;
        store_dT(dst + head * HD, oq, hf, scale);
        if (head == 0 && g.P3 > 3 * g.C && hf == 0)
          for (int c = 3 * g.C; c < g.P3; c += 8) *(uint4*)(dst + c) = make_uint4(0, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (t == 0) stamp(5 + 3 * jb);
    }
    // ---------------- N-layout (lane = key), one 32-key block at a time
    float padk = 0.f, padv = 0.f;                          // lane (half, l31 < 16): sum over the window's pad keys of d-register l31 of that half
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      f32x16_t st[2], dp[2];                              // index = query block (registers); lanes = keys 32*ib + l31
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[jb][r] = 0.f; dp[jb][r] = 0.f; }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const uint4 kv = *(const uint4*)(tk + win_nat_off(32 * ib + l31, 2 * s2 + hf));
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
          const int o = win_nat_off(32 * jb + l31, 2 * s2 + hf);
          st[jb] = mfma16(*(const uint4*)(tq + o), kv, st[jb]);
          dp[jb] = mfma16(*(const uint4*)(tg + o), fv[ib][s2], dp[jb]);
        }
      }
      const int i = 32 * ib + l31, ic = i < WN ? i : WN - 1;
      const unsigned regk = (pk[ic >> 3] >> (4 * (ic & 7))) & 15u;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int j0 = 32 * jb + 8 * m + 4 * hf;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int j = j0 + e;
            float sc = st[jb][4 * m + e] * scale + biasl[(j < WN ? j : WN - 1) * BLP + ic];
            { const unsigned rq = (pk[4 * jb + m] >> (16 * hf + 4 * e)) & 15u; sc += rq != regk ? -100.f : 0.f; }
            const float4 mzv = *(const float4*)(mz + 4 * j);
            const float pr = (i < WN && j < WN) ? __expf(sc - mzv.x) * mzv.y : 0.f;
            st[jb][4 * m + e] = pr;                                            // P[query][key]
            dp[jb][4 * m + e] = pr * (dp[jb][4 * m + e] - mzv.z);              // dS[query][key]
          }
        }
      // dK^T[d][key] = sum_query Q^T[d][query] dS[query][key];  dV^T[d][key] = sum_query dO^T[d][query] P[query][key]
      f32x16_t ok, ov;
#pragma unroll
      for (int r = 0; r < 16; ++r) { ok[r] = 0.f; ov[r] = 0.f; }
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          ok = mfma16(win_trfrag(tq, lane, jb, s2), regs_frag(dp[jb], s2), ok);
          ov = mfma16(win_trfrag(tg, lane, jb, s2), regs_frag(st[jb], s2), ov);
        }
      if (pixr[ib] >= 0) {
        bf16_t* dst = dqkv + ((size_t)b * g.H * g.W + pixr[ib]) * g.P3 + head * HD;
        store_dT(dst + 1 * g.C, ok, hf, scale);
        store_dT(dst + 2 * g.C, ov, hf, 1.f);
      }
      if (__any(pixr[ib] == -1)) {                          // edge windows only: butterfly sums over the 32 key lanes of each half (fixed order)
        const bool pad = pixr[ib] == -1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float a = pad ? ok[r] * scale : 0.f, c = pad ? ov[r] : 0.f;
#pragma unroll
          for (int o = 1; o < 32; o <<= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
          if (l31 == r) { padk += a; padv += c; }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (t == 0) stamp(9);
    float* padp = pad_part + ((size_t)wic * g.heads + head) * 96;
    if (!active) continue;
    if (l31 < 16) {
      const int d = 8 * (l31 >> 2) + 4 * hf + (l31 & 3);    // d of register l31 in half hf
      padp[d] = 0.f; padp[HD + d] = padk; padp[2 * HD + d] = padv;
    }
  }
  __syncthreads();
  float* dr = drel_part + ((size_t)chunk * g.heads + head) * (WN * WN);
  for (int e = tid; e < WN * WN; e += 256) dr[e] = dacc[(e / WN) * WB2_AP + e % WN];
  stamp(10);
  if constexpr (TRACE) { if (trace && threadIdx.x == 0) trace[(size_t)blockIdx.x * 16 + 11] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); }
}

// backward of one (window, head); block (head, chunk) walks `wpb` windows and keeps the sum of dS (= gradient of the relative position bias)
// in registers.  dqkv of real tokens is written in place; k / v gradients of PAD tokens belong to the qkv bias: pad_part[window][head][3*32].
template <typename T>
__global__ __launch_bounds__(256) void window_attention_bwd_kernel(WinGeom g, const T* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                                   const float* __restrict__ rel_bias, const T* __restrict__ dout, T* __restrict__ dqkv,
                                                                   float* __restrict__ drel_part, float* __restrict__ pad_part, int wpb, int nwin) {
  constexpr int V = Vec16<T>::N;
  __shared__ __attribute__((aligned(16))) float q[WN * QP], k[WN * QP], v[WN * QP], dO[WN * QP], P[WN * 50], dS[WN * 50];
  __shared__ float padst[2][WN][HD];         // k / v gradients of the window's pad tokens, summed in token order below
  __shared__ int pixs[WN], regs[WN];
  const int tid = threadIdx.x;
  const int head = blockIdx.x % g.heads, chunk = blockIdx.x / g.heads;
  const float scale = rsqrtf((float)HD);
  float dsum[10];                           // entries tid + 256*m of the 49 x 49 matrix
#pragma unroll
  for (int m = 0; m < 10; ++m) dsum[m] = 0.f;
  for (int wi = chunk * wpb; wi < (chunk + 1) * wpb && wi < nwin; ++wi) {
    const int wx = wi % g.nWx, wy = (wi / g.nWx) % g.nWy, b = wi / (g.nWx * g.nWy);
    __syncthreads();
    win_load_qkv<T>(g, qkv, qkv_bias, b, wy, wx, head, q, k, v, pixs, regs, scale);
    for (int e = tid; e < WN * (HD / V); e += 256) {
      const int i = e / (HD / V), d0 = (e % (HD / V)) * V;
      float t[V];
      if (pixs[i] >= 0) unpack16<T>(*(const uint4*)(dout + ((size_t)b * g.H * g.W + pixs[i]) * g.Cp + head * HD + d0), t);
      else {
#pragma unroll
        for (int j = 0; j < V; ++j) t[j] = 0.f;                  // outputs of pad queries are cropped (:234-235)
      }
#pragma unroll
      for (int j = 0; j < V; ++j) dO[i * QP + d0 + j] = t[j];
    }
    win_probs(g, q, k, rel_bias, head, regs, P);                  // ends with a barrier: dO is visible too
    // dP[i][j] = dO_i . v_j
    for (int e = tid; e < WN * WN; e += 256) {
      const int i = e / WN, j = e % WN;
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        const float4 a = *(const float4*)(dO + i * QP + d), bb = *(const float4*)(v + j * QP + d);
        s = fmaf(a.x, bb.x, s); s = fmaf(a.y, bb.y, s); s = fmaf(a.z, bb.z, s); s = fmaf(a.w, bb.w, s);
      }
      dS[i * 50 + j] = s;
    }
    __syncthreads();
    {                                        // dS = P * (dP - sum_j P dP), 4 lanes per row
      const int row = tid >> 2, part = tid & 3;
      if (row < WN) {
        float t = 0.f;
        for (int j = part; j < WN; j += 4) t = fmaf(P[row * 50 + j], dS[row * 50 + j], t);
        t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64);
        for (int j = part; j < WN; j += 4) dS[row * 50 + j] = P[row * 50 + j] * (dS[row * 50 + j] - t);
      }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 10; ++m) { const int e = tid + 256 * m; if (e < WN * WN) dsum[m] += dS[(e / WN) * 50 + e % WN]; }
    // dq_i = scale * sum_j dS_ij k_j ; dk_j = sum_i dS_ij q_i (q already scaled) ; dv_j = sum_i P_ij dO_i
    float* padp = pad_part + ((size_t)wi * g.heads + head) * 96;
    for (int e = tid; e < 3 * WN * (HD / V); e += 256) {
      const int which = e / (WN * (HD / V)), rem = e % (WN * (HD / V));
      const int n = rem / (HD / V), d0 = (rem % (HD / V)) * V;
      float o[V];
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = 0.f;
      if (which == 0) {
        for (int jj = 0; jj < WN; ++jj) {
          const float w = dS[n * 50 + jj];
#pragma unroll
          for (int j = 0; j < V; ++j) o[j] = fmaf(w, k[jj * QP + d0 + j], o[j]);
        }
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] *= scale;
      } else if (which == 1) {
        for (int ii = 0; ii < WN; ++ii) {
          const float w = dS[ii * 50 + n];
#pragma unroll
          for (int j = 0; j < V; ++j) o[j] = fmaf(w, q[ii * QP + d0 + j], o[j]);
        }
      } else {
        for (int ii = 0; ii < WN; ++ii) {
          const float w = P[ii * 50 + n];
#pragma unroll
          for (int j = 0; j < V; ++j) o[j] = fmaf(w, dO[ii * QP + d0 + j], o[j]);
        }
      }
      if (pixs[n] >= 0) *(uint4*)(dqkv + ((size_t)b * g.H * g.W + pixs[n]) * g.P3 + which * g.C + head * HD + d0) = pack16<T>(o);
      else if (which > 0) {                   // pad token: its k / v are the bias vector
#pragma unroll
        for (int j = 0; j < V; ++j) padst[which - 1][n][d0 + j] = o[j];
      }
    }
    __syncthreads();
    if (tid < 96) {                           // [q | k | v] x 32: fixed-order sum over the pad tokens (q of a pad token has no consumer)
      float t = 0.f;
      if (tid >= HD)
        for (int n = 0; n < WN; ++n)
          if (pixs[n] < 0) t += padst[tid / HD - 1][n][tid % HD];
      padp[tid] = t;
    }
    if (head == 0 && g.P3 > 3 * g.C) {
      for (int e = tid; e < WN * ((g.P3 - 3 * g.C) / V); e += 256) {
        const int i = e / ((g.P3 - 3 * g.C) / V), c = 3 * g.C + (e % ((g.P3 - 3 * g.C) / V)) * V;
        if (pixs[i] >= 0) *(uint4*)(dqkv + ((size_t)b * g.H * g.W + pixs[i]) * g.P3 + c) = make_uint4(0, 0, 0, 0);
      }
    }
  }
  float* dr = drel_part + ((size_t)chunk * g.heads + head) * (WN * WN);
#pragma unroll
  for (int m = 0; m < 10; ++m) { const int e = tid + 256 * m; if (e < WN * WN) dr[e] = dsum[m]; }
}


int check_geom(const SlWinDesc* d, WinGeom& g) {
  SL_REQUIRE(d && d->B > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->heads > 0, "window_attention: bad sizes");
  SL_REQUIRE(d->C == d->heads * HD, "window_attention: C (%d) must be heads (%d) x 32", d->C, d->heads);
  SL_REQUIRE(d->shift == 0 || d->shift == WS / 2, "window_attention: shift must be 0 or 3");
  SL_REQUIRE(d->qkv_pitch >= 3 * d->C && d->out_pitch >= d->C, "window_attention: pitches too small");
  const int vb = d->dtype == SL_BF16 ? 8 : 4;
  SL_REQUIRE(d->dtype == SL_BF16 || d->dtype == SL_F32, "window_attention: bad dtype");
  SL_REQUIRE(d->qkv_pitch % vb == 0 && d->out_pitch % vb == 0, "window_attention: pitches must be multiples of a 16-byte vector");
  g.B = d->B; g.H = d->H; g.W = d->W; g.C = d->C; g.heads = d->heads; g.P3 = d->qkv_pitch; g.Cp = d->out_pitch; g.shift = d->shift;
  g.Hp = cdiv(d->H, WS) * WS; g.Wp = cdiv(d->W, WS) * WS; g.nWy = g.Hp / WS; g.nWx = g.Wp / WS;
  return 0;
}

}  // namespace

#define BY_DTYPE(dtype, CALL_BF, CALL_F32, what)             \
  do {                                                       \
    if ((dtype) == SL_BF16) { CALL_BF; }                     \
    else if ((dtype) == SL_F32) { CALL_F32; }                \
    else SL_REQUIRE(false, what ": bad dtype");              \
  } while (0)

extern "C" int sl_patch_embed_fwd(int dtype, const float* img, const float* w, const float* bias, void* out, int B, int H, int W, int C,
                                  int out_pitch, sl_stream_t stream) {
  SL_REQUIRE(img && w && bias && out && B > 0 && H > 0 && W > 0, "patch_embed_fwd: bad args");
  SL_REQUIRE(C % 32 == 0 && C <= 192 && out_pitch >= C && out_pitch % 8 == 0, "patch_embed_fwd: C must be a multiple of 32, <= 192");
  const int Ho = cdiv(H, 4), Wo = cdiv(W, 4);
  const long long ntok = (long long)B * Ho * Wo;
  const size_t lds = (size_t)(48 * C + 64 * 49) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  BY_DTYPE(dtype,
           hipLaunchKernelGGL(patch_embed_fwd_kernel<bf16_t>, dim3(cdiv(ntok, 64)), dim3(256), lds, st, img, w, bias, (bf16_t*)out, B, H, W, Ho, Wo, C, out_pitch),
           hipLaunchKernelGGL(patch_embed_fwd_kernel<float>, dim3(cdiv(ntok, 64)), dim3(256), lds, st, img, w, bias, (float*)out, B, H, W, Ho, Wo, C, out_pitch),
           "patch_embed_fwd");
  SL_LAUNCH_CHECK("patch_embed_fwd_kernel");
  return 0;
}

extern "C" int sl_patch_im2col(int dtype, const float* img, void* col, int B, int H, int W, sl_stream_t stream) {
  SL_REQUIRE(img && col && B > 0 && H > 0 && W > 0, "patch_im2col: bad args");
  const int Ho = cdiv(H, 4), Wo = cdiv(W, 4);
  const long long n = (long long)B * Ho * Wo * 16;
  hipStream_t st = (hipStream_t)stream;
  BY_DTYPE(dtype, hipLaunchKernelGGL(patch_im2col_kernel<bf16_t>, dim3(ew_grid(n)), dim3(256), 0, st, img, (bf16_t*)col, B, H, W, Ho, Wo),
           hipLaunchKernelGGL(patch_im2col_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, st, img, (float*)col, B, H, W, Ho, Wo), "patch_im2col");
  SL_LAUNCH_CHECK("patch_im2col_kernel");
  return 0;
}

template <typename T>
static int launch_ln_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, long long rows, int C, int px, int py, float eps, hipStream_t st) {
  constexpr int V = Vec16<T>::N;
  const int nvec = C / V, nvp = py / V > nvec ? py / V : nvec;      // vectors of a row incl. the zero pad of the output pitch
  // one row group per wave where the map is small (a grid of rows / 16 blocks left 2 048 rows x 768 channels to 129 blocks: four dependent rows per wave)
  auto grid_of = [&](int lpr) { const long long g = (rows + 4 * (64 / lpr) - 1) / (4 * (64 / lpr)); return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g)); };
#define LN_FWD(LPR, NVL) hipLaunchKernelGGL((layernorm_fwd_kernel<T, LPR, NVL>), dim3(grid_of(LPR)), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, stats, rows, C, px, py, eps)
  if (nvp <= 16) LN_FWD(16, 1); else if (nvp <= 32) LN_FWD(32, 1); else if (nvp <= 64) LN_FWD(64, 1); else if (nvp <= 128) LN_FWD(64, 2); else if (nvp <= 192) LN_FWD(64, 3); else LN_FWD(64, 0);
#undef LN_FWD
  return 0;
}

extern "C" int sl_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean_rstd, long long rows,
                                int C, int x_pitch, int y_pitch, float eps, sl_stream_t stream) {
  SL_REQUIRE(x && gamma && beta && y && rows > 0 && C > 0, "layernorm_fwd: bad args");
  const int vb = dtype == SL_BF16 ? 8 : 4;
  SL_REQUIRE(C % vb == 0 && x_pitch >= C && y_pitch >= C && x_pitch % vb == 0 && y_pitch % vb == 0, "layernorm_fwd: C / pitches must be multiples of a 16-byte vector");
  hipStream_t st = (hipStream_t)stream;
  BY_DTYPE(dtype, launch_ln_fwd<bf16_t>(x, gamma, beta, y, mean_rstd, rows, C, x_pitch, y_pitch, eps, st),
           launch_ln_fwd<float>(x, gamma, beta, y, mean_rstd, rows, C, x_pitch, y_pitch, eps, st), "layernorm_fwd");
  SL_LAUNCH_CHECK("layernorm_fwd_kernel");
  return 0;
}

constexpr int LN_BWD_BLOCKS = 2048;     // partial rows of the fused column sums (grid-stride over the rows)
// blocks (= partial rows) of the fused backward
static int ln_bwd_fused_blocks(long long rows, int nvp) {
  // every block writes a partial row of 2 C floats that the finalize reads again: rows / 16 blocks keep that at a quarter of the tensor's own bytes (one row group per
  // wave -- rows / 4 blocks at 64 lanes per row -- made the 8 192-token backward 13.5 -> 19.8 us); maps of few rows get up to 512 blocks (2 048 tokens: 13.2 -> 10.6 us)
  const int lpr = nvp <= 16 ? 16 : (nvp <= 32 ? 32 : 64), rpb = 4 * (64 / lpr);
  const long long one_group = (rows + rpb - 1) / rpb;
  long long want = rows / 16 + 1;
  if (want < 512) want = one_group < 512 ? one_group : 512;
  return (int)(want > LN_BWD_BLOCKS ? LN_BWD_BLOCKS : (want < 1 ? 1 : want));
}

// fused column sums need every lane's channel vectors in registers: up to 3 per lane (C <= 3 * 64 * 8 = 1536 in bf16, 768 in fp32)

template <typename T>
static int launch_ln_bwd(const void* dy, const void* x, const float* gamma, const float* stats, const void* addend, void* dx, float* part, long long rows, int C,
                         int pdy, int px, int pdx, hipStream_t st, void* dx2 = nullptr, const float* rscale = nullptr, long long rps = 1) {
  constexpr int V = Vec16<T>::N;
  const int nvec = C / V, nvp = pdx / V;
  long long want = rows / 16 + 1;
  const bool fused = part != nullptr && nvp <= 192;
  const int grid = (int)(fused ? ln_bwd_fused_blocks(rows, nvp) : (want > 8192 ? 8192 : want));
  const size_t lds = fused ? (size_t)8 * C * sizeof(float) : 0;
#define LN_BWD(LPR, NV) hipLaunchKernelGGL((layernorm_bwd_kernel<T, LPR, NV>), dim3(grid), dim3(256), lds, st, (const T*)dy, (const T*)x, gamma, stats, (const T*)addend, (T*)dx, part, rows, C, pdy, px, pdx, (T*)dx2, rscale, rps)
  if (fused) {
    if (nvp <= 16) LN_BWD(16, 1); else if (nvp <= 32) LN_BWD(32, 1); else if (nvp <= 64) LN_BWD(64, 1); else if (nvp <= 128) LN_BWD(64, 2); else LN_BWD(64, 3);
  } else {
    if (nvec <= 16) LN_BWD(16, 0); else if (nvec <= 32) LN_BWD(32, 0); else LN_BWD(64, 0);
  }
#undef LN_BWD
  return fused ? grid : 0;
}

extern "C" int sl_layernorm_bwd_rows(int dtype, long long rows, int C, int dx_pitch) {
  const int vb = dtype == SL_BF16 ? 8 : 4;
  if (dx_pitch / vb <= 192) return ln_bwd_fused_blocks(rows, dx_pitch / vb);
  const long long b = (rows + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

static int layernorm_bwd_run(int dtype, const void* dy, const void* x, const float* gamma, const float* mean_rstd, const void* addend, void* dx,
                             float* dgamma_dbeta_partial, long long rows, int C, int dy_pitch, int x_pitch, int dx_pitch, sl_stream_t stream,
                             void* dx2, const float* rscale, long long rps) {
  SL_REQUIRE(dy && x && gamma && mean_rstd && dx && rows > 0 && C > 0, "layernorm_bwd: bad args");
  const int vb = dtype == SL_BF16 ? 8 : 4;
  SL_REQUIRE(dtype == SL_BF16 || dtype == SL_F32, "layernorm_bwd: bad dtype");
  SL_REQUIRE(C % vb == 0 && dy_pitch >= C && x_pitch >= C && dx_pitch >= C && dy_pitch % vb == 0 && x_pitch % vb == 0 && dx_pitch % vb == 0, "layernorm_bwd: bad pitches");
  hipStream_t st = (hipStream_t)stream;
  const bool fused = dgamma_dbeta_partial && dx_pitch / vb <= 192;
  if (dtype == SL_BF16) launch_ln_bwd<bf16_t>(dy, x, gamma, mean_rstd, addend, dx, fused ? dgamma_dbeta_partial : nullptr, rows, C, dy_pitch, x_pitch, dx_pitch, st, dx2, rscale, rps);
  else launch_ln_bwd<float>(dy, x, gamma, mean_rstd, addend, dx, fused ? dgamma_dbeta_partial : nullptr, rows, C, dy_pitch, x_pitch, dx_pitch, st, dx2, rscale, rps);
  SL_LAUNCH_CHECK("layernorm_bwd_kernel");
  if (dgamma_dbeta_partial && !fused) {
    const int nblk = sl_layernorm_bwd_rows(dtype, rows, C, dx_pitch);
    const long long rpb = (rows + nblk - 1) / nblk;
    BY_DTYPE(dtype,
             hipLaunchKernelGGL(layernorm_bwd_cols_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, mean_rstd, dgamma_dbeta_partial, rows, C, dy_pitch, x_pitch, rpb),
             hipLaunchKernelGGL(layernorm_bwd_cols_kernel<float>, dim3(nblk), dim3(256), 0, st, (const float*)dy, (const float*)x, mean_rstd, dgamma_dbeta_partial, rows, C, dy_pitch, x_pitch, rpb),
             "layernorm_bwd");
    SL_LAUNCH_CHECK("layernorm_bwd_cols_kernel");
  }
  return 0;
}

extern "C" int sl_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const float* mean_rstd, const void* addend, void* dx,
                                float* dgamma_dbeta_partial, long long rows, int C, int dy_pitch, int x_pitch, int dx_pitch, sl_stream_t stream) {
  return layernorm_bwd_run(dtype, dy, x, gamma, mean_rstd, addend, dx, dgamma_dbeta_partial, rows, C, dy_pitch, x_pitch, dx_pitch, stream, nullptr, nullptr, 1);
}

extern "C" int sl_layernorm_bwd_scaled(int dtype, const void* dy, const void* x, const float* gamma, const float* mean_rstd, const void* addend, void* dx,
                                       void* dx_scaled, const float* row_scale, long long rows_per_sample,
                                       float* dgamma_dbeta_partial, long long rows, int C, int dy_pitch, int x_pitch, int dx_pitch, sl_stream_t stream) {
  SL_REQUIRE(dx_scaled && row_scale && rows_per_sample > 0 && rows % rows_per_sample == 0, "layernorm_bwd_scaled: dx_scaled, row_scale and rows_per_sample | rows are required");
  return layernorm_bwd_run(dtype, dy, x, gamma, mean_rstd, addend, dx, dgamma_dbeta_partial, rows, C, dy_pitch, x_pitch, dx_pitch, stream, dx_scaled, row_scale, rows_per_sample);
}

extern "C" int sl_gelu_fwd(int dtype, const void* h, void* y, long long n, sl_stream_t stream) {
  SL_REQUIRE(h && y && n > 0 && n % 8 == 0, "gelu_fwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  BY_DTYPE(dtype, hipLaunchKernelGGL(gelu_fwd_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)h, (bf16_t*)y, n / 8),
           hipLaunchKernelGGL(gelu_fwd_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, st, (const float*)h, (float*)y, n / 4), "gelu_fwd");
  SL_LAUNCH_CHECK("gelu_fwd_kernel");
  return 0;
}

extern "C" int sl_gelu_bwd(int dtype, const void* h, const void* dy, void* dh, long long n, sl_stream_t stream) {
  SL_REQUIRE(h && dy && dh && n > 0 && n % 8 == 0, "gelu_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  BY_DTYPE(dtype, hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)h, (const bf16_t*)dy, (bf16_t*)dh, n / 8),
           hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, st, (const float*)h, (const float*)dy, (float*)dh, n / 4), "gelu_bwd");
  SL_LAUNCH_CHECK("gelu_bwd_kernel");
  return 0;
}

extern "C" int sl_patch_merge_gather(int dtype, const void* x, void* xm, int B, int H, int W, int C, int x_pitch, sl_stream_t stream) {
  SL_REQUIRE(x && xm && B > 0 && H > 0 && W > 0 && C % 8 == 0 && x_pitch >= C && x_pitch % 8 == 0, "patch_merge_gather: bad args");
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)B * ((H + 1) / 2) * ((W + 1) / 2) * 4 * C;
  BY_DTYPE(dtype, hipLaunchKernelGGL(merge_gather_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)xm, B, H, W, C, x_pitch),
           hipLaunchKernelGGL(merge_gather_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, st, (const float*)x, (float*)xm, B, H, W, C, x_pitch), "patch_merge_gather");
  SL_LAUNCH_CHECK("merge_gather_kernel");
  return 0;
}

extern "C" int sl_patch_merge_scatter(int dtype, const void* dxm, void* dx, int B, int H, int W, int C, int dx_pitch, sl_stream_t stream) {
  SL_REQUIRE(dxm && dx && B > 0 && H > 0 && W > 0 && C % 8 == 0 && dx_pitch >= C && dx_pitch % 8 == 0, "patch_merge_scatter: bad args");
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)B * H * W * dx_pitch;
  BY_DTYPE(dtype, hipLaunchKernelGGL(merge_scatter_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)dxm, (bf16_t*)dx, B, H, W, C, dx_pitch),
           hipLaunchKernelGGL(merge_scatter_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, st, (const float*)dxm, (float*)dx, B, H, W, C, dx_pitch), "patch_merge_scatter");
  SL_LAUNCH_CHECK("merge_scatter_kernel");
  return 0;
}

static int check_resize(const SlResizeDesc* d) {
  SL_REQUIRE(d && d->B > 0 && d->h > 0 && d->w > 0 && d->H > 0 && d->W > 0 && d->C > 0, "bilinear: bad sizes");
  const int vb = d->dtype == SL_BF16 ? 8 : 4;
  SL_REQUIRE(d->dtype == SL_BF16 || d->dtype == SL_F32, "bilinear: bad dtype");
  SL_REQUIRE(d->C % vb == 0 && d->src_pitch % vb == 0 && d->dst_pitch % vb == 0 && d->src_off % vb == 0 && d->dst_off % vb == 0, "bilinear: channel window must be 16-byte aligned");
  SL_REQUIRE(d->src_off + d->C <= d->src_pitch && d->dst_off + d->C <= d->dst_pitch, "bilinear: channel window outside the tensor");
  return 0;
}

static int bilinear_fwd_launch(const SlResizeDesc* d, const void* src, const void* add, void* dst, hipStream_t st) {
  const long long n = (long long)d->B * d->H * d->W * d->C;
#define BL_ARGS d->B, d->h, d->w, d->H, d->W, d->C, d->src_pitch, d->src_off, d->dst_pitch, d->dst_off, d->align_corners
  if (d->dtype == SL_BF16 && d->src_f32) hipLaunchKernelGGL((bilinear_fwd_kernel<bf16_t, float>), dim3(ew_grid(n / 8)), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, BL_ARGS, (const bf16_t*)add);
  else if (d->dtype == SL_BF16) hipLaunchKernelGGL((bilinear_fwd_kernel<bf16_t, bf16_t>), dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, BL_ARGS, (const bf16_t*)add);
  else hipLaunchKernelGGL((bilinear_fwd_kernel<float, float>), dim3(ew_grid(n / 4)), dim3(256), 0, st, (const float*)src, (float*)dst, BL_ARGS, (const float*)add);
  SL_LAUNCH_CHECK("bilinear_fwd_kernel");
  return 0;
}

extern "C" int sl_bilinear_fwd(const SlResizeDesc* d, const void* src, void* dst, sl_stream_t stream) {
  if (int e = check_resize(d)) return e;
  SL_REQUIRE(src && dst, "bilinear_fwd: null buffer");
  return bilinear_fwd_launch(d, src, d->accumulate ? dst : nullptr, dst, (hipStream_t)stream);
}

extern "C" int sl_bilinear_fwd_add(const SlResizeDesc* d, const void* src, const void* base, void* dst, sl_stream_t stream) {
  if (int e = check_resize(d)) return e;
  SL_REQUIRE(src && base && dst, "bilinear_fwd_add: null buffer");
  SL_REQUIRE(!d->accumulate, "bilinear_fwd_add: accumulate is the in-place form of sl_bilinear_fwd");
  return bilinear_fwd_launch(d, src, base, dst, (hipStream_t)stream);
}

extern "C" int sl_bilinear_bwd(const SlResizeDesc* d, const void* ddst, void* dsrc, sl_stream_t stream) {
  if (int e = check_resize(d)) return e;
  SL_REQUIRE(ddst && dsrc, "bilinear_bwd: null buffer");
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)d->B * d->h * d->w * d->C;
  const int nsrc = d->B * d->h * d->w, nv = d->C / (d->dtype == SL_BF16 ? 8 : 4);
  if (nsrc <= 2048 && (long long)d->H * d->W >= 4LL * d->h * d->w && nv <= 128 && 256 % nv == 0) {       // few source pixels under a large map: a block per source pixel
    if (d->dtype == SL_BF16 && d->src_f32) hipLaunchKernelGGL((bilinear_bwd_small_kernel<bf16_t, float>), dim3(nsrc), dim3(256), 0, st, (const bf16_t*)ddst, (float*)dsrc, BL_ARGS, d->accumulate);
    else if (d->dtype == SL_BF16) hipLaunchKernelGGL((bilinear_bwd_small_kernel<bf16_t, bf16_t>), dim3(nsrc), dim3(256), 0, st, (const bf16_t*)ddst, (bf16_t*)dsrc, BL_ARGS, d->accumulate);
    else hipLaunchKernelGGL((bilinear_bwd_small_kernel<float, float>), dim3(nsrc), dim3(256), 0, st, (const float*)ddst, (float*)dsrc, BL_ARGS, d->accumulate);
    SL_LAUNCH_CHECK("bilinear_bwd_small_kernel");
    return 0;
  }
  if (d->dtype == SL_BF16 && d->src_f32) hipLaunchKernelGGL((bilinear_bwd_kernel<bf16_t, float>), dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)ddst, (float*)dsrc, BL_ARGS, d->accumulate);
  else if (d->dtype == SL_BF16) hipLaunchKernelGGL((bilinear_bwd_kernel<bf16_t, bf16_t>), dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)ddst, (bf16_t*)dsrc, BL_ARGS, d->accumulate);
  else hipLaunchKernelGGL((bilinear_bwd_kernel<float, float>), dim3(ew_grid(n / 4)), dim3(256), 0, st, (const float*)ddst, (float*)dsrc, BL_ARGS, d->accumulate);
  SL_LAUNCH_CHECK("bilinear_bwd_kernel");
  return 0;
}

extern "C" int sl_scale_add(int dtype, const void* x, const float* scale, const void* addend, void* out, int B, long long rows_per_sample, int C,
                            int pitch, int per_channel, sl_stream_t stream) {
  SL_REQUIRE(x && scale && out && B > 0 && rows_per_sample > 0 && C > 0 && pitch >= C && pitch % 8 == 0, "scale_add: bad args");
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)B * rows_per_sample * pitch;
  BY_DTYPE(dtype,
           hipLaunchKernelGGL(scale_add_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)x, scale, (const bf16_t*)addend, (bf16_t*)out, rows_per_sample, C, pitch, per_channel, n / 8),
           hipLaunchKernelGGL(scale_add_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, st, (const float*)x, scale, (const float*)addend, (float*)out, rows_per_sample, C, pitch, per_channel, n / 4),
           "scale_add");
  SL_LAUNCH_CHECK("scale_add_kernel");
  return 0;
}

// test hook (not part of the public ABI): 1 = the fp32-arithmetic VALU kernels (what float32 tensors always run on) also for bf16, 0 / -1 = MFMA kernels for bf16
static bool use_attn_mfma(int dtype) { return dtype == SL_BF16 && g_sl_debug.attn_valu != 1; }      // test hook sl_debug_attn_valu(1): the VALU reference kernels
static bool use_attn_bwd_mfma(int dtype) { return use_attn_mfma(dtype); }

extern "C" int sl_window_attention_fwd(const SlWinDesc* d, const void* qkv, const float* qkv_bias, const float* rel_bias, void* out, sl_stream_t stream) {
  WinGeom g;
  if (int e = check_geom(d, g)) return e;
  SL_REQUIRE(qkv && qkv_bias && rel_bias && out, "window_attention_fwd: null buffer");
  hipStream_t st = (hipStream_t)stream;
  const int nwin = g.B * g.nWy * g.nWx;
  if (use_attn_mfma(d->dtype)) {
    hipLaunchKernelGGL(window_attention_fwd_mfma_kernel, dim3(cdiv(nwin, 4) * g.heads), dim3(256), 0, st, g, (const bf16_t*)qkv, qkv_bias, rel_bias, (bf16_t*)out, nwin);
    SL_LAUNCH_CHECK("window_attention_fwd_mfma_kernel");
    return 0;
  }
  const int grid = nwin * g.heads;
  BY_DTYPE(d->dtype, hipLaunchKernelGGL(window_attention_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, g, (const bf16_t*)qkv, qkv_bias, rel_bias, (bf16_t*)out),
           hipLaunchKernelGGL(window_attention_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, g, (const float*)qkv, qkv_bias, rel_bias, (float*)out), "window_attention_fwd");
  SL_LAUNCH_CHECK("window_attention_fwd_kernel");
  return 0;
}

// windows per block of the backward (more when there are many windows: fewer partial rows of the bias gradient)
static int win_wpb(const WinGeom& g) {
  const long long tasks = (long long)g.B * g.nWy * g.nWx * g.heads;
  long long w = tasks / 2048;
  return (int)(w < 1 ? 1 : (w > 16 ? 16 : w));
}
static int win_wpw(const WinGeom& g) { return (win_wpb(g) + 3) / 4; }         // MFMA path: 4 waves per block, each walks wpw windows

// d relative_position_bias_table[t][h] = sum over the (query, key) pairs with relative offset t of d bias[h][pair] (swintransformer.py:128-131 backward):
// fixed-order gather through the constant pair lists (pairs[t][j], -1 padded) -- the index_add_ of an autograd gather is atomic, this is bit-stable.
__global__ void relpos_table_grad_kernel(const float* __restrict__ dbias, const int* __restrict__ pairs, int rows, int m, int heads, int npair, float* __restrict__ dtable,
                                         int nrel, const float* __restrict__ dbq_in, const float* __restrict__ dpad, float* __restrict__ dbq_out) {
  if ((int)blockIdx.x >= nrel) {
    // the blocks behind the table's: d qkv.bias [3][heads][32] = the column sums of dqkv + what reached the bias through the pad tokens, dpad [heads][3][32]
    // (swintransformer.py:208-213 backward) -- the add that followed this launch as a torch kernel
    const int i = ((int)blockIdx.x - nrel) * blockDim.x + threadIdx.x;
    if (i < 3 * heads * 32) {
      const int j = i & 31, h = (i >> 5) % heads, sct = (i >> 5) / heads;
      dbq_out[i] = dbq_in[i] + dpad[(h * 3 + sct) * 32 + j];
    }
    return;
  }
  // eight lanes per table entry: lane `sub` adds pairs sub, sub + 8, ... (up to 49 per offset: one thread walked them as a chain of dependent loads, 21 us whatever the stage),
  // then a fixed xor tree over the eight partial sums
  const int gi = blockIdx.x * blockDim.x + threadIdx.x, i = gi >> 3, sub = gi & 7;
  const bool live = i < rows * heads;
  const int ic = live ? i : 0;
  const int t = ic / heads, h = ic - t * heads;
  float s = 0.f;
  for (int j = sub; j < m; j += 8) {
    const int pidx = pairs[t * m + j];
    if (pidx >= 0) s += dbias[(size_t)h * npair + pidx];
  }
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
  if (live && sub == 0) dtable[i] = s;
}

extern "C" int sl_relpos_table_grad(const float* dbias, const int* pairs, int rows, int m, int heads, int npair, float* dtable, sl_stream_t stream) {
  SL_REQUIRE(dbias && pairs && dtable && rows > 0 && m > 0 && heads > 0 && npair > 0, "relpos_table_grad: bad args");
  const int nrel = cdiv(rows * heads * 8, 256);
  hipLaunchKernelGGL(relpos_table_grad_kernel, dim3(nrel), dim3(256), 0, (hipStream_t)stream, dbias, pairs, rows, m, heads, npair, dtable, nrel, (const float*)nullptr,
                     (const float*)nullptr, (float*)nullptr);
  SL_LAUNCH_CHECK("relpos_table_grad_kernel");
  return 0;
}

extern "C" int sl_relpos_table_grad_bias(const float* dbias, const int* pairs, int rows, int m, int heads, int npair, float* dtable, const float* dbq_colsum,
                                         const float* dpad, float* dbq, sl_stream_t stream) {
  SL_REQUIRE(dbias && pairs && dtable && rows > 0 && m > 0 && heads > 0 && npair > 0 && dbq_colsum && dpad && dbq, "relpos_table_grad_bias: bad args");
  const int nrel = cdiv(rows * heads * 8, 256);
  hipLaunchKernelGGL(relpos_table_grad_kernel, dim3(nrel + cdiv(3 * heads * 32, 256)), dim3(256), 0, (hipStream_t)stream, dbias, pairs, rows, m, heads, npair, dtable, nrel,
                     dbq_colsum, dpad, dbq);
  SL_LAUNCH_CHECK("relpos_table_grad_kernel");
  return 0;
}

extern "C" int sl_window_attention_bwd_chunks(const SlWinDesc* d) {
  WinGeom g;
  if (check_geom(d, g)) return SL_EINVAL;
  const long long nwin = (long long)g.B * g.nWy * g.nWx;
  return use_attn_bwd_mfma(d->dtype) ? cdiv(nwin, 4 * win_wpw(g)) : cdiv(nwin, win_wpb(g));
}

extern "C" int sl_window_attention_windows(const SlWinDesc* d) {
  WinGeom g;
  if (check_geom(d, g)) return SL_EINVAL;
  return g.B * g.nWy * g.nWx;
}


extern "C" int sl_window_attention_bwd(const SlWinDesc* d, const void* qkv, const float* qkv_bias, const float* rel_bias, const void* dout, void* dqkv,
                                       float* drel_partial, float* pad_partial, sl_stream_t stream) {
  WinGeom g;
  if (int e = check_geom(d, g)) return e;
  SL_REQUIRE(qkv && qkv_bias && rel_bias && dout && dqkv && drel_partial && pad_partial, "window_attention_bwd: null buffer");
  hipStream_t st = (hipStream_t)stream;
  const int nwin = g.B * g.nWy * g.nWx;
  if (use_attn_bwd_mfma(d->dtype)) {
    const int wpw = win_wpw(g), chunks = cdiv(nwin, 4 * wpw);
    const size_t lds2 = (size_t)(WN * BLP + 16) * sizeof(float) + (size_t)4 * 3 * WB2_TILE + (size_t)4 * 256 * sizeof(float) + (size_t)WN * WB2_AP * sizeof(float);
    static bool attr2 = false;
    if (!attr2) {
      (void)hipFuncSetAttribute((const void*)window_attention_bwd_mfma2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      (void)hipFuncSetAttribute((const void*)window_attention_bwd_mfma2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      attr2 = true;
    }
    if (g_sl_debug.attn_trace)      // debug instantiation (the stamps cost 8 registers and 6 spills: not in the production kernel)
      hipLaunchKernelGGL(window_attention_bwd_mfma2_kernel<true>, dim3(chunks * g.heads), dim3(256), lds2, st, g, (const bf16_t*)qkv, qkv_bias, rel_bias, (const bf16_t*)dout,
                         (bf16_t*)dqkv, drel_partial, pad_partial, wpw, nwin, g_sl_debug.attn_trace);
    else
      hipLaunchKernelGGL(window_attention_bwd_mfma2_kernel<false>, dim3(chunks * g.heads), dim3(256), lds2, st, g, (const bf16_t*)qkv, qkv_bias, rel_bias, (const bf16_t*)dout,
                         (bf16_t*)dqkv, drel_partial, pad_partial, wpw, nwin, nullptr);
    SL_LAUNCH_CHECK("window_attention_bwd_mfma2_kernel");
    return 0;
  }
  const int wpb = win_wpb(g);
  const int chunks = cdiv(nwin, wpb);
  BY_DTYPE(d->dtype,
           hipLaunchKernelGGL(window_attention_bwd_kernel<bf16_t>, dim3(chunks * g.heads), dim3(256), 0, st, g, (const bf16_t*)qkv, qkv_bias, rel_bias, (const bf16_t*)dout, (bf16_t*)dqkv, drel_partial, pad_partial, wpb, nwin),
           hipLaunchKernelGGL(window_attention_bwd_kernel<float>, dim3(chunks * g.heads), dim3(256), 0, st, g, (const float*)qkv, qkv_bias, rel_bias, (const float*)dout, (float*)dqkv, drel_partial, pad_partial, wpb, nwin),
           "window_attention_bwd");
  SL_LAUNCH_CHECK("window_attention_bwd_kernel");
  return 0;
}
