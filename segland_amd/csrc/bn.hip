// BatchNorm2d (train + eval) forward/backward around the conv kernels; all HBM-bound, 16-byte vectorised NHWC.
// Reference call sites: networks/backbones/resnet.py:45,48,50,88,111 and networks/pspnet_pop.py:20,28.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ finalize
__global__ void bn_finalize_train_kernel(const float* __restrict__ part, int rows, int C, double count,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* rmean, float* rvar, float momentum, float eps,
                                         float* mean, float* invstd, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int r = 0; r < rows; ++r) {           // fixed order: bit-stable
    s += (double)part[((size_t)r * 2 + 0) * C + c];
    q += (double)part[((size_t)r * 2 + 1) * C + c];
  }
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
    rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * mu);
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
template <typename T>
__global__ void bn_act_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                  const T* __restrict__ res, int relu, T* __restrict__ y, long long nvec, int C) {
  constexpr int V = Vec16<T>::N;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i * V) % C);
    float xv[V], rv[V], o[V];
    unpack16<T>(((const uint4*)x)[i], xv);
    if (res) unpack16<T>(((const uint4*)res)[i], rv);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float v = xv[k] * scale[c + k] + shift[c + k];
      if (res) v += rv[k];
      if (relu) v = v > 0.f ? v : 0.f;
      o[k] = v;
    }
    ((uint4*)y)[i] = pack16<T>(o);
  }
}

// ------------------------------------------------------------------------------------------------ backward reduce
// partial[blk][0][c] = sum g, partial[blk][1][c] = sum g * (x - mean) * invstd over the block's rows
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            float* __restrict__ part, long long rows, int C, long long rows_per_blk) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[2 * 256 * V];
  const int nvec = C / V;
  const int tpr = nvec < 256 ? nvec : 256;      // threads per row
  const int rpb = 256 / tpr;                    // rows handled in parallel
  const int tr = threadIdx.x / tpr, tc = threadIdx.x % tpr;
  const long long r_begin = blockIdx.x * rows_per_blk;
  long long r_end = r_begin + rows_per_blk; if (r_end > rows) r_end = rows;
  for (int vc = tc; vc < nvec; vc += tpr) {
    float s1[V], s2[V], mu[V], is[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { s1[k] = 0.f; s2[k] = 0.f; mu[k] = mean[vc * V + k]; is[k] = invstd[vc * V + k]; }
    if (tr < rpb) {
      for (long long r = r_begin + tr; r < r_end; r += rpb) {
        const size_t o = (size_t)r * nvec + vc;
        float g[V], xv[V], yv[V];
        unpack16<T>(((const uint4*)dy)[o], g);
        unpack16<T>(((const uint4*)x)[o], xv);
        if (y) {
          unpack16<T>(((const uint4*)y)[o], yv);
#pragma unroll
          for (int k = 0; k < V; ++k) g[k] = yv[k] > 0.f ? g[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < V; ++k) { s1[k] += g[k]; s2[k] += g[k] * ((xv[k] - mu[k]) * is[k]); }
      }
    }
    // reduce over the rpb row-lanes that share this channel vector
    __syncthreads();
#pragma unroll
    for (int k = 0; k < V; ++k) { red[(threadIdx.x * V + k) * 2 + 0] = s1[k]; red[(threadIdx.x * V + k) * 2 + 1] = s2[k]; }
    __syncthreads();
    if (tr == 0) {
#pragma unroll
      for (int k = 0; k < V; ++k) {
        float a = 0.f, b = 0.f;
        for (int j = 0; j < rpb; ++j) { a += red[((j * tpr + tc) * V + k) * 2 + 0]; b += red[((j * tpr + tc) * V + k) * 2 + 1]; }
        part[((size_t)blockIdx.x * 2 + 0) * C + vc * V + k] = a;
        part[((size_t)blockIdx.x * 2 + 1) * C + vc * V + k] = b;
      }
    }
  }
}

__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int C, double count, const float* gamma,
                                       const float* mean, const float* invstd, int train, float* dgamma, float* dbeta,
                                       float* cA, float* cB, float* cC) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int r = 0; r < nblk; ++r) {
    s1 += (double)part[((size_t)r * 2 + 0) * C + c];
    s2 += (double)part[((size_t)r * 2 + 1) * C + c];
  }
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

template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x,
                                    const float* __restrict__ cA, const float* __restrict__ cB, const float* __restrict__ cC,
                                    const float* __restrict__ mean, T* __restrict__ dx, T* __restrict__ dres, long long nvec, int C) {
  constexpr int V = Vec16<T>::N;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i * V) % C);
    float g[V], xv[V], yv[V], o[V];
    unpack16<T>(((const uint4*)dy)[i], g);
    unpack16<T>(((const uint4*)x)[i], xv);
    if (y) {
      unpack16<T>(((const uint4*)y)[i], yv);
#pragma unroll
      for (int k = 0; k < V; ++k) g[k] = yv[k] > 0.f ? g[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = cA[c + k] * g[k] + cB[c + k] * (xv[k] - mean[c + k]) + cC[c + k];
    ((uint4*)dx)[i] = pack16<T>(o);
    if (dres) ((uint4*)dres)[i] = pack16<T>(g);
  }
}

inline int ew_blocks(long long nvec) { long long b = (nvec + 255) / 256; return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }
inline int reduce_blocks(long long rows) { long long b = (rows + 63) / 64; return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b)); }

}  // namespace

extern "C" int sl_bn_finalize_train(const float* stat_partial, int stat_rows, int C, long long count, const float* gamma,
                                    const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                    float* mean, float* invstd, float* scale, float* shift, sl_stream_t stream) {
  SL_REQUIRE(stat_partial && mean && invstd && scale && shift && C > 0 && stat_rows > 0 && count > 0, "bn_finalize_train: bad args");
  SL_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize_train: running stats must come in pairs");
  hipLaunchKernelGGL(bn_finalize_train_kernel, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, stat_partial, stat_rows, C,
                     (double)count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift);
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

extern "C" int sl_bn_act_fwd(int dtype, const void* x, const float* scale, const float* shift, const void* residual,
                             int relu, void* y, long long rows, int C, sl_stream_t stream) {
  SL_REQUIRE(x && scale && shift && y && rows > 0 && C > 0 && C % 8 == 0, "bn_act_fwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SL_BF16) {
    const long long nvec = rows * C / 8;
    hipLaunchKernelGGL(bn_act_fwd_kernel<bf16_t>, dim3(ew_blocks(nvec)), dim3(256), 0, st, (const bf16_t*)x, scale, shift, (const bf16_t*)residual, relu, (bf16_t*)y, nvec, C);
  } else if (dtype == SL_F32) {
    const long long nvec = rows * C / 4;
    hipLaunchKernelGGL(bn_act_fwd_kernel<float>, dim3(ew_blocks(nvec)), dim3(256), 0, st, (const float*)x, scale, shift, (const float*)residual, relu, (float*)y, nvec, C);
  } else SL_REQUIRE(false, "bn_act_fwd: bad dtype");
  SL_LAUNCH_CHECK("bn_act_fwd_kernel");
  return 0;
}

extern "C" int sl_bn_bwd_reduce_rows(long long rows, int C) { (void)C; return reduce_blocks(rows); }

extern "C" int sl_bn_bwd_reduce(int dtype, const void* dy, const void* y, const void* x, const float* mean,
                                const float* invstd, float* partial, long long rows, int C, sl_stream_t stream) {
  SL_REQUIRE(dy && x && mean && invstd && partial && rows > 0 && C > 0 && C % 8 == 0, "bn_bwd_reduce: bad args");
  const int nblk = reduce_blocks(rows);
  const long long rpb = (rows + nblk - 1) / nblk;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SL_BF16)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, mean, invstd, partial, rows, C, rpb);
  else if (dtype == SL_F32)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(nblk), dim3(256), 0, st, (const float*)dy, (const float*)y, (const float*)x, mean, invstd, partial, rows, C, rpb);
  else SL_REQUIRE(false, "bn_bwd_reduce: bad dtype");
  SL_LAUNCH_CHECK("bn_bwd_reduce_kernel");
  return 0;
}

extern "C" int sl_bn_bwd_finalize(const float* partial, int nblk, int C, long long count, const float* gamma,
                                  const float* mean, const float* invstd, int train, float* dgamma, float* dbeta,
                                  float* cA, float* cB, float* cC, sl_stream_t stream) {
  SL_REQUIRE(partial && invstd && cA && cB && cC && nblk > 0 && C > 0 && count > 0, "bn_bwd_finalize: bad args");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, partial, nblk, C, (double)count,
                     gamma, mean, invstd, train, dgamma, dbeta, cA, cB, cC);
  SL_LAUNCH_CHECK("bn_bwd_finalize_kernel");
  return 0;
}

extern "C" int sl_bn_bwd_apply(int dtype, const void* dy, const void* y, const void* x, const float* cA, const float* cB,
                               const float* cC, const float* mean, void* dx, void* dres, long long rows, int C,
                               sl_stream_t stream) {
  SL_REQUIRE(dy && x && cA && cB && cC && mean && dx && rows > 0 && C > 0 && C % 8 == 0, "bn_bwd_apply: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SL_BF16) {
    const long long nvec = rows * C / 8;
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(ew_blocks(nvec)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, cA, cB, cC, mean, (bf16_t*)dx, (bf16_t*)dres, nvec, C);
  } else if (dtype == SL_F32) {
    const long long nvec = rows * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(ew_blocks(nvec)), dim3(256), 0, st, (const float*)dy, (const float*)y, (const float*)x, cA, cB, cC, mean, (float*)dx, (float*)dres, nvec, C);
  } else SL_REQUIRE(false, "bn_bwd_apply: bad dtype");
  SL_LAUNCH_CHECK("bn_bwd_apply_kernel");
  return 0;
}
