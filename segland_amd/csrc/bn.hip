// BatchNorm2d (train + eval) forward/backward around the conv kernels; all HBM-bound, 16-byte vectorised NHWC.
// Reference call sites: networks/backbones/resnet.py:45,48,50,88,111 and networks/pspnet_pop.py:20,28.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ finalize
// Column sums of the [rows][2][C] partial buffers: a block owns 64 channels, 16 row-lanes stride over the rows (8 loads in
// flight each), then a fixed-order 16-way LDS tree -- bit-stable, and ~100x less latency than one serial thread per channel.
template <int FIN_CH, int FIN_RL>
__device__ __forceinline__ void colsum2(const float* __restrict__ part, int rows, int C, int c, int rl, double& s, double& q,
                                        double (*red)[FIN_RL][FIN_CH]) {
  double a0 = 0.0, a1 = 0.0;
  if (c < C) {
    int r = rl;
    for (; r + 7 * FIN_RL < rows; r += 8 * FIN_RL) {
      float v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v0[u] = part[((size_t)(r + u * FIN_RL) * 2 + 0) * C + c];
        v1[u] = part[((size_t)(r + u * FIN_RL) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a0 += (double)v0[u]; a1 += (double)v1[u]; }
    }
    for (; r < rows; r += FIN_RL) { a0 += (double)part[((size_t)r * 2 + 0) * C + c]; a1 += (double)part[((size_t)r * 2 + 1) * C + c]; }
  }
  const int cl = threadIdx.x % FIN_CH;
  red[0][rl][cl] = a0; red[1][rl][cl] = a1;
  __syncthreads();
  s = 0.0; q = 0.0;
  if (rl == 0) {
#pragma unroll
    for (int j = 0; j < FIN_RL; ++j) { s += red[0][j][cl]; q += red[1][j][cl]; }
  }
}

template <int FIN_CH, int FIN_RL>
__global__ __launch_bounds__(FIN_CH * FIN_RL) void bn_finalize_train_kernel(const float* __restrict__ part, int rows, int C, double count,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* rmean, float* rvar, float momentum, float eps,
                                         float* mean, float* invstd, float* scale, float* shift, const float* __restrict__ cbias, int nbias) {
  __shared__ double red[2][FIN_RL][FIN_CH];
  const int c = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH, rl = threadIdx.x / FIN_CH;
  double s, q;
  colsum2<FIN_CH, FIN_RL>(part, rows, C, c, rl, s, q, red);
  if (rl != 0 || c >= C) return;
  const double mu = s / count;
  double var = q / count - mu * mu;
  if (var < 0.0) var = 0.0;
  const double is = 1.0 / sqrt(var + (double)eps);
  mean[c] = (float)mu;
  invstd[c] = (float)is;
  const double g = gamma ? (double)gamma[c] : 1.0, b = beta ? (double)beta[c] : 0.0;
  scale[c] = (float)(g * is);
  shift[c] = (float)(b - mu * g * is);
  if (rmean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    // cbias: the statistics are those of the RAW conv output; the conv's bias shifts the mean only (it cancels in the normalised output) and enters the running mean here
    const double mb = (cbias && c < nbias) ? mu + (double)cbias[c] : mu;
    rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * mb);
    rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * unbiased);
  }
}

__global__ void bn_finalize_eval_kernel(int C, const float* gamma, const float* beta, const float* rmean, const float* rvar,
                                        float eps, float* mean, float* invstd, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double is = 1.0 / sqrt((double)rvar[c] + (double)eps);
  const double g = gamma ? (double)gamma[c] : 1.0, b = beta ? (double)beta[c] : 0.0;
  if (mean) mean[c] = rmean[c];
  if (invstd) invstd[c] = (float)is;
  scale[c] = (float)(g * is);
  shift[c] = (float)(b - (double)rmean[c] * g * is);
}

// ------------------------------------------------------------------------------------------------ forward apply
// FIXEDC: 256 % (C/V) == 0, so a thread's channel vector never changes along its grid-stride walk and the per-channel
// coefficients live in registers; the loop is then a pure 16-byte stream (4 independent loads in flight per operand).
template <typename T, bool FIXEDC>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                  const T* __restrict__ res, int relu, T* __restrict__ y, uint8_t* __restrict__ mask_out, unsigned nvec, unsigned nvc) {
  constexpr int V = Vec16<T>::N;
  float sc[V], sh[V];
  if (FIXEDC) {
    const unsigned c = (threadIdx.x % nvc) * V;
#pragma unroll
    for (int k = 0; k < V; ++k) { sc[k] = scale[c + k]; sh[k] = shift[c + k]; }
  }
  // a block-iteration owns 1024 consecutive vectors (16 KiB per operand).  Walking the tensor from its end (the bytes the producing conv
  // wrote last) was measured for all three BN passes: no difference, the Infinity Cache does not retain these streams
  const unsigned nchunk = (nvec + 1023u) >> 10;
  for (unsigned ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
    const unsigned i0 = (ch << 10) + threadIdx.x;
    uint4 xv[4], rv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned i = i0 + u * 256u;
      if (i < nvec) { xv[u] = ((const uint4*)x)[i]; if (res) rv[u] = ((const uint4*)res)[i]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned i = i0 + u * 256u;
      if (i >= nvec) break;
      if (!FIXEDC) {
        const unsigned c = (i % nvc) * V;
#pragma unroll
        for (int k = 0; k < V; ++k) { sc[k] = scale[c + k]; sh[k] = shift[c + k]; }
      }
      float a[V], r[V], o[V];
      unpack16<T>(xv[u], a);
      if (res) unpack16<T>(rv[u], r);
      unsigned bits = 0;
#pragma unroll
      for (int k = 0; k < V; ++k) {
        float v = a[k] * sc[k] + sh[k];
        if (res) v += r[k];
        if (relu) v = v > 0.f ? v : 0.f;
        bits |= (v > 0.f ? 1u : 0u) << k;
        o[k] = v;
      }
      ((uint4*)y)[i] = pack16<T>(o);
      if (mask_out) mask_out[i] = (uint8_t)bits;       // 1 byte per 16-byte vector: bit k = (y_k > 0)
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward reduce
// partial[blk][0][c] = sum g, partial[blk][1][c] = sum g * (x - mean) * invstd over the block's rows
// DUAL: a second BatchNorm whose output was added to the first one's before the same ReLU (bn3 + downsample BN of a bottleneck, resnet.py:71-76): both backward
// passes gate the SAME gradient, so one sweep over dy and the bits serves both (x2, mean2, invstd2 -> part2)
template <typename T, bool DUAL = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ y, const uint8_t* __restrict__ mask, const T* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            float* __restrict__ part, long long rows, int C,
                                                            const T* __restrict__ x2 = nullptr, const float* __restrict__ mean2 = nullptr, const float* __restrict__ invstd2 = nullptr,
                                                            float* __restrict__ part2 = nullptr) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[(DUAL ? 3 : 2) * 256 * V];
  const unsigned lb = blockIdx.x;
  const int nvec = C / V;
  const int tpr = nvec < 256 ? nvec : 256;      // threads per row
  const int rpb = 256 / tpr;                    // rows handled in parallel
  const int tr = threadIdx.x / tpr, tc = threadIdx.x % tpr;
  // block b owns the row groups b, b + nblk, b + 2 nblk, ...: the chip reads ONE moving window of the tensor, like the streaming passes
  // (measured 4.2 vs 4.0 TB/s for a contiguous row range per block; 2 or 4 rows in flight per thread were SLOWER: 3.9 / 3.2 TB/s)
  const long long r_begin = (long long)lb * rpb, r_end = rows, rstep = (long long)gridDim.x * rpb;
  constexpr int NS = DUAL ? 3 : 2;
  for (int vc = tc; vc < nvec; vc += tpr) {
    float s1[V], s2[V], s3[V], mu[V], is[V], mu2[V], is2[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      s1[k] = 0.f; s2[k] = 0.f; s3[k] = 0.f; mu[k] = mean[vc * V + k]; is[k] = invstd[vc * V + k];
      if (DUAL) { mu2[k] = mean2[vc * V + k]; is2[k] = invstd2[vc * V + k]; }
    }
    if (tr < rpb) {
      auto acc = [&](const uint4& gq, const uint4& xq, const uint4& yq, unsigned mb, const uint4& x2q) {
        float g[V], xv[V], yv[V];
        unpack16<T>(gq, g);
        unpack16<T>(xq, xv);
        if (mask) {
#pragma unroll
          for (int k = 0; k < V; ++k) g[k] = (mb >> k) & 1u ? g[k] : 0.f;
        } else if (y) {
          unpack16<T>(yq, yv);
#pragma unroll
          for (int k = 0; k < V; ++k) g[k] = yv[k] > 0.f ? g[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < V; ++k) { s1[k] += g[k]; s2[k] += g[k] * ((xv[k] - mu[k]) * is[k]); }
        if (DUAL) {
          float x2v[V];
          unpack16<T>(x2q, x2v);
#pragma unroll
          for (int k = 0; k < V; ++k) s3[k] += g[k] * ((x2v[k] - mu2[k]) * is2[k]);
        }
      };
      long long r = r_begin + tr;
      for (; r < r_end; r += rstep) {
        const size_t o = (size_t)r * nvec + vc;
        uint4 yq = make_uint4(0, 0, 0, 0), x2q = make_uint4(0, 0, 0, 0);
        if (!mask && y) yq = ((const uint4*)y)[o];
        if (DUAL) x2q = ((const uint4*)x2)[o];
        acc(((const uint4*)dy)[o], ((const uint4*)x)[o], yq, mask ? mask[o] : 0u, x2q);
      }
    }
    // reduce over the rpb row-lanes that share this channel vector
    __syncthreads();
#pragma unroll
    for (int k = 0; k < V; ++k) {
      red[(threadIdx.x * V + k) * NS + 0] = s1[k]; red[(threadIdx.x * V + k) * NS + 1] = s2[k];
      if (DUAL) red[(threadIdx.x * V + k) * NS + 2] = s3[k];
    }
    __syncthreads();
    if (tr == 0) {
#pragma unroll
      for (int k = 0; k < V; ++k) {
        float a = 0.f, b = 0.f, c3 = 0.f;
        for (int j = 0; j < rpb; ++j) {
          a += red[((j * tpr + tc) * V + k) * NS + 0]; b += red[((j * tpr + tc) * V + k) * NS + 1];
          if (DUAL) c3 += red[((j * tpr + tc) * V + k) * NS + 2];
        }
        part[((size_t)lb * 2 + 0) * C + vc * V + k] = a;
        part[((size_t)lb * 2 + 1) * C + vc * V + k] = b;
        if (DUAL) { part2[((size_t)lb * 2 + 0) * C + vc * V + k] = a; part2[((size_t)lb * 2 + 1) * C + vc * V + k] = c3; }
      }
    }
  }
}

template <int FIN_CH, int FIN_RL>
__global__ __launch_bounds__(FIN_CH * FIN_RL) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int C, double count, const float* gamma,
                                       const float* mean, const float* invstd, int train, float* dgamma, float* dbeta,
                                       float* cA, float* cB, float* cC) {
  __shared__ double red[2][FIN_RL][FIN_CH];
  const int c = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH, rl = threadIdx.x / FIN_CH;
  double s1, s2;
  colsum2<FIN_CH, FIN_RL>(part, nblk, C, c, rl, s1, s2, red);
  if (rl != 0 || c >= C) return;
  if (dgamma) dgamma[c] = (float)s2;
  if (dbeta) dbeta[c] = (float)s1;
  const double g = gamma ? (double)gamma[c] : 1.0, is = (double)invstd[c];
  cA[c] = (float)(g * is);
  if (train) {     // dx = g*is*(dy - S1/M - xhat*S2/M) = cA*dy + cB*(x-mean) + cC
    cB[c] = (float)(-g * is * is * s2 / count);
    cC[c] = (float)(-g * is * s1 / count);
  } else {
    cB[c] = 0.f; cC[c] = 0.f;
  }
}

template <typename T, bool FIXEDC, bool DUAL = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y, const uint8_t* __restrict__ mask, const T* __restrict__ x,
                                    const float* __restrict__ cA, const float* __restrict__ cB, const float* __restrict__ cC,
                                    const float* __restrict__ mean, T* __restrict__ dx, T* __restrict__ dres, unsigned nvec, unsigned nvc,
                                    const T* __restrict__ x2 = nullptr, const float* __restrict__ cA2 = nullptr, const float* __restrict__ cB2 = nullptr,
                                    const float* __restrict__ cC2 = nullptr, const float* __restrict__ mean2 = nullptr, T* __restrict__ dx2 = nullptr) {
  constexpr int V = Vec16<T>::N;
  float a_[V], b_[V], c_[V], cc[V];      // dx = a*g + b*(x - c) + cc
  float a2[V], b2[V], c2[V], cc2[V];     // DUAL: the second BatchNorm fed by the same gated gradient
  auto coeffs = [&](unsigned c) {
#pragma unroll
    for (int k = 0; k < V; ++k) {
      a_[k] = cA[c + k]; b_[k] = cB[c + k]; c_[k] = mean[c + k]; cc[k] = cC[c + k];
      if (DUAL) { a2[k] = cA2[c + k]; b2[k] = cB2[c + k]; c2[k] = mean2[c + k]; cc2[k] = cC2[c + k]; }
    }
  };
  if (FIXEDC) coeffs((threadIdx.x % nvc) * V);
  const unsigned nchunk = (nvec + 511u) >> 9;          // 512 consecutive vectors per block-iteration
  for (unsigned ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
    const unsigned i0 = (ch << 9) + threadIdx.x;
    uint4 gv[2], xv[2], yv[2], x2v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned i = i0 + u * 256u;
      if (i < nvec) {
        gv[u] = ((const uint4*)dy)[i]; xv[u] = ((const uint4*)x)[i];
        if (y && !mask) yv[u] = ((const uint4*)y)[i];
        if (DUAL) x2v[u] = ((const uint4*)x2)[i];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned i = i0 + u * 256u;
      if (i >= nvec) break;
      if (!FIXEDC) coeffs((i % nvc) * V);
      float g[V], xx[V], yy[V], o[V];
      unpack16<T>(gv[u], g);
      unpack16<T>(xv[u], xx);
      if (mask) {
        const unsigned b = mask[i];
#pragma unroll
        for (int k = 0; k < V; ++k) g[k] = (b >> k) & 1u ? g[k] : 0.f;
      } else if (y) {
        unpack16<T>(yv[u], yy);
#pragma unroll
        for (int k = 0; k < V; ++k) g[k] = yy[k] > 0.f ? g[k] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] = a_[k] * g[k] + b_[k] * (xx[k] - c_[k]) + cc[k];
      ((uint4*)dx)[i] = pack16<T>(o);
      if (dres) ((uint4*)dres)[i] = pack16<T>(g);
      if (DUAL) {
        float x2x[V];
        unpack16<T>(x2v[u], x2x);
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = a2[k] * g[k] + b2[k] * (x2x[k] - c2[k]) + cc2[k];
        ((uint4*)dx2)[i] = pack16<T>(o);
      }
    }
  }
}

// Block shape of the two finalize kernels: 32 channels x 32 row lanes (measured against 64 x 16 and 16 x 64 in round 2: -0.26 ms per ResNet-50 step).  Round 4 tried 1024 threads
// per 32 channels with 16-byte loads, every load of a thread in flight at once and a shuffle + 16-wave LDS reduction: 7.7 / 7.2 us per launch against 5.9 / 6.0 for this form
// (gpurun r4c: rocprofv3 --stats of bench.py) -- removed.

inline int ew_blocks(long long nvec) { long long b = (nvec + 255) / 256; return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }
inline int reduce_blocks(long long rows) { long long b = (rows + 63) / 64; return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b)); }

}  // namespace

extern "C" int sl_bn_finalize_train(const float* stat_partial, int stat_rows, int C, long long count, const float* gamma,
                                    const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                    float* mean, float* invstd, float* scale, float* shift, sl_stream_t stream) {
  SL_REQUIRE(stat_partial && mean && invstd && scale && shift && C > 0 && stat_rows > 0 && count > 0, "bn_finalize_train: bad args");
  SL_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize_train: running stats must come in pairs");
  hipLaunchKernelGGL((bn_finalize_train_kernel<32, 32>), dim3(cdiv(C, 32)), dim3(1024), 0, (hipStream_t)stream, stat_partial, stat_rows, C, (double)count, gamma, beta,
                     running_mean, running_var, momentum, eps, mean, invstd, scale, shift, (const float*)nullptr, 0);
  SL_LAUNCH_CHECK("bn_finalize_train_kernel");
  return 0;
}

extern "C" int sl_bn_finalize_train_bias(const float* stat_partial, int stat_rows, int C, long long count, const float* gamma,
                                         const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                         float* mean, float* invstd, float* scale, float* shift, const float* conv_bias, int bias_n, sl_stream_t stream) {
  SL_REQUIRE(stat_partial && mean && invstd && scale && shift && C > 0 && stat_rows > 0 && count > 0, "bn_finalize_train_bias: bad args");
  SL_REQUIRE(running_mean && running_var && conv_bias && bias_n > 0 && bias_n <= C, "bn_finalize_train_bias: running stats and bias_n in 1..C conv bias entries are required");
  hipLaunchKernelGGL((bn_finalize_train_kernel<32, 32>), dim3(cdiv(C, 32)), dim3(1024), 0, (hipStream_t)stream, stat_partial, stat_rows, C, (double)count, gamma, beta,
                     running_mean, running_var, momentum, eps, mean, invstd, scale, shift, conv_bias, bias_n);
  SL_LAUNCH_CHECK("bn_finalize_train_kernel");
  return 0;
}

extern "C" int sl_bn_finalize_eval(int C, const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, float* mean, float* invstd, float* scale,
                                   float* shift, sl_stream_t stream) {
  SL_REQUIRE(running_mean && running_var && scale && shift && C > 0, "bn_finalize_eval: bad args");
  hipLaunchKernelGGL(bn_finalize_eval_kernel, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, C, gamma, beta, running_mean,
                     running_var, eps, mean, invstd, scale, shift);
  SL_LAUNCH_CHECK("bn_finalize_eval_kernel");
  return 0;
}

template <typename T>
static int launch_bn_act(const void* x, const float* scale, const float* shift, const void* residual, int relu, void* y, uint8_t* mask_out, long long rows, int C, hipStream_t st) {
  constexpr int V = Vec16<T>::N;
  const long long nvec = rows * C / V;
  SL_REQUIRE(nvec < (1ll << 31), "bn_act_fwd: tensor too large");
  const unsigned nvc = C / V;
  const bool fixed = nvc <= 256 && 256 % nvc == 0;
  const int blocks = ew_blocks((nvec + 3) / 4);
  if (fixed) hipLaunchKernelGGL((bn_act_fwd_kernel<T, true>), dim3(blocks), dim3(256), 0, st, (const T*)x, scale, shift, (const T*)residual, relu, (T*)y, mask_out, (unsigned)nvec, nvc);
  else hipLaunchKernelGGL((bn_act_fwd_kernel<T, false>), dim3(blocks), dim3(256), 0, st, (const T*)x, scale, shift, (const T*)residual, relu, (T*)y, mask_out, (unsigned)nvec, nvc);
  SL_LAUNCH_CHECK("bn_act_fwd_kernel");
  return 0;
}

extern "C" int sl_bn_act_fwd(int dtype, const void* x, const float* scale, const float* shift, const void* residual,
                             int relu, void* y, uint8_t* relu_mask, long long rows, int C, sl_stream_t stream) {
  SL_REQUIRE(x && scale && shift && y && rows > 0 && C > 0 && C % 8 == 0, "bn_act_fwd: bad args");
  if (dtype == SL_BF16) return launch_bn_act<bf16_t>(x, scale, shift, residual, relu, y, relu_mask, rows, C, (hipStream_t)stream);
  if (dtype == SL_F32) return launch_bn_act<float>(x, scale, shift, residual, relu, y, relu_mask, rows, C, (hipStream_t)stream);
  SL_REQUIRE(false, "bn_act_fwd: bad dtype");
  return 0;
}

extern "C" int sl_bn_bwd_reduce_rows(long long rows, int C) { (void)C; return reduce_blocks(rows); }

extern "C" int sl_bn_bwd_reduce(int dtype, const void* dy, const void* y, const uint8_t* relu_mask, const void* x, const float* mean,
                                const float* invstd, float* partial, long long rows, int C, sl_stream_t stream) {
  SL_REQUIRE(dy && x && mean && invstd && partial && rows > 0 && C > 0 && C % 8 == 0, "bn_bwd_reduce: bad args");
  const int nblk = reduce_blocks(rows);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SL_BF16)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, relu_mask, (const bf16_t*)x, mean, invstd, partial, rows, C);
  else if (dtype == SL_F32)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(nblk), dim3(256), 0, st, (const float*)dy, (const float*)y, relu_mask, (const float*)x, mean, invstd, partial, rows, C);
  else SL_REQUIRE(false, "bn_bwd_reduce: bad dtype");
  SL_LAUNCH_CHECK("bn_bwd_reduce_kernel");
  return 0;
}

extern "C" int sl_bn_bwd_finalize(const float* partial, int nblk, int C, long long count, const float* gamma,
                                  const float* mean, const float* invstd, int train, float* dgamma, float* dbeta,
                                  float* cA, float* cB, float* cC, sl_stream_t stream) {
  SL_REQUIRE(partial && invstd && cA && cB && cC && nblk > 0 && C > 0 && count > 0, "bn_bwd_finalize: bad args");
  hipLaunchKernelGGL((bn_bwd_finalize_kernel<32, 32>), dim3(cdiv(C, 32)), dim3(1024), 0, (hipStream_t)stream, partial, nblk, C, (double)count, gamma, mean, invstd, train,
                     dgamma, dbeta, cA, cB, cC);
  SL_LAUNCH_CHECK("bn_bwd_finalize_kernel");
  return 0;
}

template <typename T>
static int launch_bn_apply(const void* dy, const void* y, const uint8_t* mask, const void* x, const float* cA, const float* cB, const float* cC, const float* mean,
                           void* dx, void* dres, long long rows, int C, hipStream_t st) {
  constexpr int V = Vec16<T>::N;
  const long long nvec = rows * C / V;
  SL_REQUIRE(nvec < (1ll << 31), "bn_bwd_apply: tensor too large");
  const unsigned nvc = C / V;
  const bool fixed = nvc <= 256 && 256 % nvc == 0;
  const int blocks = ew_blocks((nvec + 1) / 2);
  if (fixed) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true>), dim3(blocks), dim3(256), 0, st, (const T*)dy, (const T*)y, mask, (const T*)x, cA, cB, cC, mean, (T*)dx, (T*)dres, (unsigned)nvec, nvc);
  else hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false>), dim3(blocks), dim3(256), 0, st, (const T*)dy, (const T*)y, mask, (const T*)x, cA, cB, cC, mean, (T*)dx, (T*)dres, (unsigned)nvec, nvc);
  SL_LAUNCH_CHECK("bn_bwd_apply_kernel");
  return 0;
}

extern "C" int sl_bn_bwd_apply(int dtype, const void* dy, const void* y, const uint8_t* relu_mask, const void* x, const float* cA, const float* cB,
                               const float* cC, const float* mean, void* dx, void* dres, long long rows, int C,
                               sl_stream_t stream) {
  SL_REQUIRE(dy && x && cA && cB && cC && mean && dx && rows > 0 && C > 0 && C % 8 == 0, "bn_bwd_apply: bad args");
  if (dtype == SL_BF16) return launch_bn_apply<bf16_t>(dy, y, relu_mask, x, cA, cB, cC, mean, dx, dres, rows, C, (hipStream_t)stream);
  if (dtype == SL_F32) return launch_bn_apply<float>(dy, y, relu_mask, x, cA, cB, cC, mean, dx, dres, rows, C, (hipStream_t)stream);
  SL_REQUIRE(false, "bn_bwd_apply: bad dtype");
  return 0;
}

// Two BatchNorms behind one ReLU (bn3 + the downsample BN of the first bottleneck of a stage): ONE sweep over dy and the ReLU bits for both reduces / both applies.
extern "C" int sl_bn_bwd_reduce2(int dtype, const void* dy, const uint8_t* relu_mask, const void* x1, const float* mean1, const float* invstd1, float* partial1,
                                 const void* x2, const float* mean2, const float* invstd2, float* partial2, long long rows, int C, sl_stream_t stream) {
  SL_REQUIRE(dy && x1 && x2 && mean1 && invstd1 && mean2 && invstd2 && partial1 && partial2 && rows > 0 && C > 0 && C % 8 == 0, "bn_bwd_reduce2: bad args");
  const int nblk = reduce_blocks(rows);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SL_BF16)
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_t, true>), dim3(nblk), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)nullptr, relu_mask, (const bf16_t*)x1, mean1, invstd1, partial1, rows, C,
                       (const bf16_t*)x2, mean2, invstd2, partial2);
  else if (dtype == SL_F32)
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, true>), dim3(nblk), dim3(256), 0, st, (const float*)dy, (const float*)nullptr, relu_mask, (const float*)x1, mean1, invstd1, partial1, rows, C,
                       (const float*)x2, mean2, invstd2, partial2);
  else SL_REQUIRE(false, "bn_bwd_reduce2: bad dtype");
  SL_LAUNCH_CHECK("bn_bwd_reduce_kernel<dual>");
  return 0;
}

template <typename T>
static int launch_bn_apply2(const void* dy, const uint8_t* mask, const void* x1, const float* cA1, const float* cB1, const float* cC1, const float* mean1, void* dx1,
                            const void* x2, const float* cA2, const float* cB2, const float* cC2, const float* mean2, void* dx2, long long rows, int C, hipStream_t st) {
  constexpr int V = Vec16<T>::N;
  const long long nvec = rows * C / V;
  SL_REQUIRE(nvec < (1ll << 31), "bn_bwd_apply2: tensor too large");
  const unsigned nvc = C / V;
  const bool fixed = nvc <= 256 && 256 % nvc == 0;
  const int blocks = ew_blocks((nvec + 1) / 2);
  if (fixed) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true, true>), dim3(blocks), dim3(256), 0, st, (const T*)dy, (const T*)nullptr, mask, (const T*)x1, cA1, cB1, cC1, mean1, (T*)dx1, (T*)nullptr,
                                (unsigned)nvec, nvc, (const T*)x2, cA2, cB2, cC2, mean2, (T*)dx2);
  else hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false, true>), dim3(blocks), dim3(256), 0, st, (const T*)dy, (const T*)nullptr, mask, (const T*)x1, cA1, cB1, cC1, mean1, (T*)dx1, (T*)nullptr,
                          (unsigned)nvec, nvc, (const T*)x2, cA2, cB2, cC2, mean2, (T*)dx2);
  SL_LAUNCH_CHECK("bn_bwd_apply_kernel<dual>");
  return 0;
}

extern "C" int sl_bn_bwd_apply2(int dtype, const void* dy, const uint8_t* relu_mask, const void* x1, const float* cA1, const float* cB1, const float* cC1,
                                const float* mean1, void* dx1, const void* x2, const float* cA2, const float* cB2, const float* cC2, const float* mean2, void* dx2,
                                long long rows, int C, sl_stream_t stream) {
  SL_REQUIRE(dy && x1 && x2 && cA1 && cB1 && cC1 && mean1 && dx1 && cA2 && cB2 && cC2 && mean2 && dx2 && rows > 0 && C > 0 && C % 8 == 0, "bn_bwd_apply2: bad args");
  if (dtype == SL_BF16) return launch_bn_apply2<bf16_t>(dy, relu_mask, x1, cA1, cB1, cC1, mean1, dx1, x2, cA2, cB2, cC2, mean2, dx2, rows, C, (hipStream_t)stream);
  if (dtype == SL_F32) return launch_bn_apply2<float>(dy, relu_mask, x1, cA1, cB1, cC1, mean1, dx1, x2, cA2, cB2, cC2, mean2, dx2, rows, C, (hipStream_t)stream);
  SL_REQUIRE(false, "bn_bwd_apply2: bad dtype");
  return 0;
}

// ------------------------------------------------------------------------------------------------ bn3 apply pass folded into conv3's gradients (round 6)
// y = bn(conv1x1(x, W)): with dc = cA g + cB (c - mean) + cC (sl_bn_bwd_finalize) and c = x W^T,
//   dx = dc W      = [g | x] [diag(cA) W ; W^T diag(cB) W] + (cC - cB mean) W          (sl_bn_fold_weights -> sl_conv2d_bwd_data_bnstat_folded)
//   dW = dc^T x    = diag(cA) (g^T x) + diag(cB) W (x^T x) + (cC - cB mean) (x) colsum(x)   (sl_bn_fold_wgrad on the raw products g^T x, x^T x)
// W is the bf16 weight the forward multiplied (c was produced with it).  Both kernels are small GEMMs on 16 x 16 output tiles staged through the LDS.
namespace {
constexpr int FT = 16;
constexpr int FKW = 512;          // reduction elements staged per round trip (global -> LDS -> barrier): the kernels are bound by the NUMBER of round trips, not by their FLOPs
constexpr int FKG = 256;

// wt_ext [Cin][Cout + Cin] (bf16).  grid (Cin / 16, Cin / 16 + Cout / 256):
//   y <  Cin / 16 : the tile (n0, j0) of W^T diag(cB) W
//   else          : columns [256 (y - Cin/16), +256) of diag(cA) W, rows n0 .. n0 + 15 (from the transposed weight w_bwd: coalesced both ways)
__global__ __launch_bounds__(256) void bn_fold_weights_kernel(const bf16_t* __restrict__ wf, const bf16_t* __restrict__ wb, const float* __restrict__ cA, const float* __restrict__ cB,
                                                              bf16_t* __restrict__ wext, int Cout, int Cin) {
  extern __shared__ float fsm[];                       // ta [FKW][17], tb [FKW][17]
  float (*ta)[FT + 1] = (float (*)[FT + 1])fsm;
  float (*tb)[FT + 1] = (float (*)[FT + 1])(fsm + FKW * (FT + 1));
  const int n0 = blockIdx.x * FT, ny = Cin / FT;
  const int KE = Cout + Cin;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  if ((int)blockIdx.y < ny) {
    const int j0 = blockIdx.y * FT;
    float acc = 0.f;
    for (int k0 = 0; k0 < Cout; k0 += FKW) {
      // a thread loads 8 consecutive bf16 (16 bytes) of one weight row per tile and step: FKW rows x 2 x 16 bytes per tile, all loads of the chunk in flight together
#pragma unroll
      for (int u = 0; u < FKW * 2 / 256; ++u) {
        const int e = threadIdx.x + 256 * u, kk = e >> 1, h = e & 1;
        const float cb = cB[k0 + kk];
        const uint4 va = *(const uint4*)(wf + (size_t)(k0 + kk) * Cin + n0 + 8 * h), vb = *(const uint4*)(wf + (size_t)(k0 + kk) * Cin + j0 + 8 * h);
        const unsigned wa[4] = {va.x, va.y, va.z, va.w}, wbv[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          ta[kk][8 * h + 2 * q] = __uint_as_float(wa[q] << 16) * cb; ta[kk][8 * h + 2 * q + 1] = __uint_as_float(wa[q] & 0xffff0000u) * cb;
          tb[kk][8 * h + 2 * q] = __uint_as_float(wbv[q] << 16);     tb[kk][8 * h + 2 * q + 1] = __uint_as_float(wbv[q] & 0xffff0000u);
        }
      }
      __syncthreads();
#pragma unroll 16
      for (int kk = 0; kk < FKW; ++kk) acc = fmaf(ta[kk][ty], tb[kk][tx], acc);
      __syncthreads();
    }
    wext[(size_t)(n0 + ty) * KE + Cout + j0 + tx] = f2bf(acc);
  } else {
    const int k0 = ((int)blockIdx.y - ny) * 256;
#pragma unroll
    for (int u = 0; u < FT; ++u) {
      const int k = k0 + threadIdx.x;
      wext[(size_t)(n0 + u) * KE + k] = f2bf(cA[k] * bf2f(wb[(size_t)(n0 + u) * Cout + k]));
    }
  }
}

// bias[n] = - sum_K avg[K] wt_ext[n][K], avg = [colsum(g) / rows (Cout) | colsum(x) / rows (Cin)]: the constant terms of the apply pass are the MEANS of the two virtual-concat
// inputs against the ROUNDED extended weight (cC = -cA mean(g), cB mean = cB mean(x) W^T), so a rounding error of a weight multiplies a centred input, as in the unfolded
// pass -- formed from the unrounded (cC - cB mean) W the bias left the bf16 rounding of W^T diag(cB) W against the uncentred x (bn2's gradients: cosine 0.978 vs the oracle).
// One wave per output channel (8 elements per lane and step, all loads in flight), fixed-order tree.
__global__ __launch_bounds__(64) void bn_fold_bias_kernel(const bf16_t* __restrict__ wext, const float* __restrict__ gsum, const float* __restrict__ xsum, float inv_rows,
                                                          float* __restrict__ bias, int Cout, int Cin) {
  const int n = blockIdx.x, KE = Cout + Cin;
  float s = 0.f;
  for (int k = threadIdx.x * 8; k < KE; k += 512) {
    const uint4 v = *(const uint4*)(wext + (size_t)n * KE + k);
    const float* av = k < Cout ? gsum + k : xsum + (k - Cout);            // Cout % 8 == 0: a vector lies on one side
    const float4 a0 = *(const float4*)av, a1 = *(const float4*)(av + 4);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) { s = fmaf(a[2 * q], __uint_as_float(w[q] << 16), s); s = fmaf(a[2 * q + 1], __uint_as_float(w[q] & 0xffff0000u), s); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) bias[n] = -s * inv_rows;
}

// dw [Cout][Cin] = diag(cA) G1 + diag(cB) W G2 + (cC - cB mean) (x) s, G1 = g^T x (may be dw itself).   grid (Cin / 16, Cout / 16)
__global__ __launch_bounds__(256) void bn_fold_wgrad_kernel(const float* G1, float* dw, const float* __restrict__ G2, const float* __restrict__ s, const bf16_t* __restrict__ wf,
                                                            const float* __restrict__ cA, const float* __restrict__ cB, const float* __restrict__ cC, const float* __restrict__ mean,
                                                            int Cout, int Cin) {
  __shared__ float tw[FT][FKG + 1], tg[FKG][FT + 1];
  const int j0 = blockIdx.x * FT, k0 = blockIdx.y * FT;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  float acc = 0.f;
  for (int i0 = 0; i0 < Cin; i0 += FKG) {
#pragma unroll
    for (int u = 0; u < FT * FKG / 256; ++u) {
      const int e = threadIdx.x + 256 * u, r = e / FKG, c = e % FKG;
      tw[r][c] = bf2f(wf[(size_t)(k0 + r) * Cin + i0 + c]);
    }
#pragma unroll
    for (int u = 0; u < FKG * FT / 256; ++u) {
      const int e = threadIdx.x + 256 * u, r = e / FT, c = e % FT;
      tg[r][c] = G2[(size_t)(i0 + r) * Cin + j0 + c];
    }
    __syncthreads();
#pragma unroll 16
    for (int ii = 0; ii < FKG; ++ii) acc = fmaf(tw[ty][ii], tg[ii][tx], acc);
    __syncthreads();
  }
  const int k = k0 + ty, j = j0 + tx;
  const size_t o = (size_t)k * Cin + j;
  dw[o] = cA[k] * G1[o] + cB[k] * acc + (cC[k] - cB[k] * mean[k]) * s[j];
}
}  // namespace

extern "C" int sl_bn_fold_weights(int Cout, int Cin, const void* w_fwd, const void* w_bwd, const float* cA, const float* cB, const float* g_colsum, const float* x_colsum,
                                  long long rows, void* wt_ext, float* bias, sl_stream_t stream) {
  SL_REQUIRE(w_fwd && w_bwd && cA && cB && g_colsum && x_colsum && wt_ext && bias && rows > 0, "bn_fold_weights: null buffer");
  SL_REQUIRE(Cout > 0 && Cin > 0 && Cout % FKW == 0 && Cin % 16 == 0, "bn_fold_weights: Cout must be a multiple of %d and Cin of 16 (got %d, %d)", FKW, Cout, Cin);
  const size_t lds = (size_t)2 * FKW * (FT + 1) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)bn_fold_weights_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  hipLaunchKernelGGL(bn_fold_weights_kernel, dim3(Cin / FT, Cin / FT + Cout / 256), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)w_fwd, (const bf16_t*)w_bwd, cA, cB,
                     (bf16_t*)wt_ext, Cout, Cin);
  SL_LAUNCH_CHECK("bn_fold_weights_kernel");
  hipLaunchKernelGGL(bn_fold_bias_kernel, dim3(Cin), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)wt_ext, g_colsum, x_colsum, (float)(1.0 / (double)rows), bias, Cout, Cin);
  SL_LAUNCH_CHECK("bn_fold_bias_kernel");
  return 0;
}

extern "C" int sl_bn_fold_wgrad(int Cout, int Cin, const float* gtx, float* dw, const float* xtx, const float* x_colsum, const void* w_fwd, const float* cA, const float* cB,
                                const float* cC, const float* mean, sl_stream_t stream) {
  SL_REQUIRE(gtx && dw && xtx && x_colsum && w_fwd && cA && cB && cC && mean, "bn_fold_wgrad: null buffer");
  SL_REQUIRE(Cout > 0 && Cin > 0 && Cout % 16 == 0 && Cin % FKG == 0, "bn_fold_wgrad: Cout must be a multiple of 16 and Cin of %d (got %d, %d)", FKG, Cout, Cin);
  hipLaunchKernelGGL(bn_fold_wgrad_kernel, dim3(Cin / FT, Cout / FT), dim3(256), 0, (hipStream_t)stream, gtx, dw, xtx, x_colsum, (const bf16_t*)w_fwd, cA, cB, cC, mean, Cout, Cin);
  SL_LAUNCH_CHECK("bn_fold_wgrad_kernel");
  return 0;
}
