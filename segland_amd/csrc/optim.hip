// AdamW over ALL parameters of a model in one launch (train_base.py:197-204 builds torch.optim.AdamW; its loop body steps it
// twice per iteration, train_base.py:262-264).  HBM-bound: 16 B read + 12 B written per element and step; `repeat` consecutive steps on
// the same gradient are applied in registers, so the reference's double step costs one pass instead of two.
// Arithmetic = torch/csrc/.../fused_adam_utils.cuh (ADAMW, amsgrad off, maximize off): decoupled decay, lerp first moment,
// bias-corrected step size, denom = sqrt(v)/sqrt(bc2) + eps.
#include "common.h"

namespace {

struct AdamEntry { float* p; const float* g; float* m; float* v; long long numel; float lr, wd; long long start; int group; int pad_; };
static_assert(sizeof(AdamEntry) == 64, "table layout is part of the ABI");
constexpr int AD_CHUNK = 4096;        // elements per block

__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamEntry* __restrict__ tab, int n, long long total_chunks, float b1, float b2, float eps,
                                                          float bc1_0, float rs2_0, float bc1_1, float rs2_1, int repeat, const float* __restrict__ gscale,
                                                          const float* __restrict__ hyp) {
  const float gs = gscale ? gscale[0] : 1.f;
  if (hyp) { bc1_0 = hyp[0]; rs2_0 = hyp[1]; bc1_1 = hyp[2]; rs2_1 = hyp[3]; }      // step-dependent scalars from device memory (graph replay)
  for (long long chunk = blockIdx.x; chunk < total_chunks; chunk += gridDim.x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab[mid].start <= chunk) lo = mid; else hi = mid - 1; }
    AdamEntry t = tab[lo];
    if (hyp) { t.lr = hyp[4 + 2 * t.group]; t.wd = hyp[5 + 2 * t.group]; }
    const long long base = (chunk - t.start) * AD_CHUNK;
    const float decay = 1.f - t.lr * t.wd;
    for (int k = threadIdx.x * 4; k < AD_CHUNK; k += 1024) {
      const long long e = base + k;
      if (e >= t.numel) break;
      const int cnt = (int)((t.numel - e) < 4 ? (t.numel - e) : 4);
      float p[4], g[4], m[4], v[4];
      if (cnt == 4 && ((t.numel & 3) == 0)) {
        const float4 P = *(const float4*)(t.p + e), G = *(const float4*)(t.g + e), M = *(const float4*)(t.m + e), V = *(const float4*)(t.v + e);
        p[0] = P.x; p[1] = P.y; p[2] = P.z; p[3] = P.w; g[0] = G.x; g[1] = G.y; g[2] = G.z; g[3] = G.w;
        m[0] = M.x; m[1] = M.y; m[2] = M.z; m[3] = M.w; v[0] = V.x; v[1] = V.y; v[2] = V.z; v[3] = V.w;
      } else {
        for (int j = 0; j < cnt; ++j) { p[j] = t.p[e + j]; g[j] = t.g[e + j]; m[j] = t.m[e + j]; v[j] = t.v[e + j]; }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gj = g[j] * gs;
        for (int r = 0; r < repeat; ++r) {
          const float bc1 = r == 0 ? bc1_0 : bc1_1, rs2 = r == 0 ? rs2_0 : rs2_1;
          p[j] *= decay;
          m[j] = m[j] + (gj - m[j]) * (1.f - b1);
          v[j] = b2 * v[j] + (1.f - b2) * gj * gj;
          const float denom = sqrtf(v[j]) / rs2 + eps;
          p[j] -= (t.lr / bc1) * (m[j] / denom);
        }
      }
      if (cnt == 4 && ((t.numel & 3) == 0)) {
        *(float4*)(t.p + e) = make_float4(p[0], p[1], p[2], p[3]);
        *(float4*)(t.m + e) = make_float4(m[0], m[1], m[2], m[3]);
        *(float4*)(t.v + e) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        for (int j = 0; j < cnt; ++j) { t.p[e + j] = p[j]; t.m[e + j] = m[j]; t.v[e + j] = v[j]; }
      }
    }
  }
}

}  // namespace

extern "C" int sl_adamw_multi(const void* table_dev, int n, long long total_chunks, float beta1, float beta2, float eps,
                              float bias_correction1, float bias_correction2_sqrt, float bias_correction1_next, float bias_correction2_sqrt_next,
                              int repeat, const float* grad_scale, sl_stream_t stream) {
  SL_REQUIRE(table_dev && n > 0 && total_chunks > 0 && (repeat == 1 || repeat == 2), "adamw_multi: bad args");
  SL_REQUIRE(bias_correction1 > 0.f && bias_correction2_sqrt > 0.f, "adamw_multi: bias corrections must be positive (step >= 1)");
  const int blocks = (int)(total_chunks < 8192 ? total_chunks : 8192);
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const AdamEntry*)table_dev, n, total_chunks, beta1, beta2, eps,
                     bias_correction1, bias_correction2_sqrt, bias_correction1_next, bias_correction2_sqrt_next, repeat, grad_scale, (const float*)nullptr);
  SL_LAUNCH_CHECK("adamw_multi_kernel");
  return 0;
}

// The same step with every step-dependent scalar in DEVICE memory, so that the launch can sit in a captured HIP graph and be replayed:
// hyper_dev = { bias_correction1, bias_correction2_sqrt, the pair for the next step, then (lr, weight_decay) per parameter group };
// record field `group` (the int after `start`) selects the pair, the records' own lr / wd are ignored.
extern "C" int sl_adamw_multi_dev(const void* table_dev, int n, long long total_chunks, float beta1, float beta2, float eps,
                                  const float* hyper_dev, int repeat, const float* grad_scale, sl_stream_t stream) {
  SL_REQUIRE(table_dev && hyper_dev && n > 0 && total_chunks > 0 && (repeat == 1 || repeat == 2), "adamw_multi_dev: bad args");
  const int blocks = (int)(total_chunks < 8192 ? total_chunks : 8192);
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const AdamEntry*)table_dev, n, total_chunks, beta1, beta2, eps,
                     1.f, 1.f, 1.f, 1.f, repeat, grad_scale, hyper_dev);
  SL_LAUNCH_CHECK("adamw_multi_kernel");
  return 0;
}


// ------------------------------------------------------------------------------------------------------------------------------------------
// Momentum SGD over all trainable parameters in one launch (ft_pop.py:205-209 builds torch.optim.SGD(momentum 0.9, weight_decay); its loop body steps it once per
// iteration, ft_pop.py:252).  Arithmetic = torch/optim/sgd.py (dampening 0, nesterov off, maximize off): d = g + wd p;  buf = momentum buf + d  (the first step's
// buf = d is the same with a zero-initialised buffer);  p -= lr buf.  Same table records as AdamW (`m` = momentum buffer, `v` unused).  hyp != NULL: (lr, wd) per
// parameter group from device memory, so that the launch can sit in a captured HIP graph while the driver changes the learning rate every iteration.
namespace {
__global__ __launch_bounds__(256) void sgd_multi_kernel(const AdamEntry* __restrict__ tab, int n, long long total_chunks, float momentum, const float* __restrict__ gscale,
                                                        const float* __restrict__ hyp) {
  const float gs = gscale ? gscale[0] : 1.f;
  for (long long chunk = blockIdx.x; chunk < total_chunks; chunk += gridDim.x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab[mid].start <= chunk) lo = mid; else hi = mid - 1; }
    AdamEntry t = tab[lo];
    if (hyp) { t.lr = hyp[2 * t.group]; t.wd = hyp[2 * t.group + 1]; }
    const long long base = (chunk - t.start) * AD_CHUNK;
    for (int k = threadIdx.x; k < AD_CHUNK; k += 256) {
      const long long e = base + k;
      if (e >= t.numel) break;
      float d = t.g[e] * gs + t.wd * t.p[e];
      if (t.m) { d = momentum * t.m[e] + d; t.m[e] = d; }
      t.p[e] -= t.lr * d;
    }
  }
}

struct FloatPack { float v[16]; };
__global__ void store_floats_kernel(float* __restrict__ dst, int n, FloatPack pk) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = pk.v[threadIdx.x];
}
}  // namespace

extern "C" int sl_sgd_multi(const void* table_dev, int n, long long total_chunks, float momentum, const float* hyper_dev, const float* grad_scale, sl_stream_t stream) {
  SL_REQUIRE(table_dev && n > 0 && total_chunks > 0 && momentum >= 0.f, "sgd_multi: bad args");
  const int blocks = (int)(total_chunks < 4096 ? total_chunks : 4096);
  hipLaunchKernelGGL(sgd_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const AdamEntry*)table_dev, n, total_chunks, momentum, grad_scale, hyper_dev);
  SL_LAUNCH_CHECK("sgd_multi_kernel");
  return 0;
}

// up to 16 floats into device memory as KERNEL ARGUMENTS (no host -> device copy: a small upload in front of a graph launch sat on the critical path of every step,
// DESIGN.md / optim.py graph_prepare): the per-group (lr, weight_decay) of a captured optimizer step whose learning rate the driver sets every iteration
extern "C" int sl_store_floats(float* dst_dev, int n, const float* values_host, sl_stream_t stream) {
  SL_REQUIRE(dst_dev && values_host && n >= 1 && n <= 16, "store_floats: 1..16 values");
  FloatPack pk;
  for (int i = 0; i < 16; ++i) pk.v[i] = i < n ? values_host[i] : 0.f;
  hipLaunchKernelGGL(store_floats_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst_dev, n, pk);
  SL_LAUNCH_CHECK("store_floats_kernel");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// Many small strided fp32 copies in ONE launch: src [rows][cols] contiguous -> dst rows of pitch dst_pitch (the zero-padded staging copies of a model's
// weights / biases / BatchNorm vectors at the channel pitch, re-filled once per optimizer step: ~55 launches of 4.4 us per Swin-T step otherwise).
// table: device array of n entries {dst, src (float*), rows, cols, dst_pitch (int), pad, start (int64)} = 40 bytes; start = running count of 1024-element chunks.
namespace {
struct CopyEntry { float* dst; const float* src; int rows, cols, dst_pitch, pad_; long long start; };
static_assert(sizeof(CopyEntry) == 40, "table layout is part of the ABI");
__global__ __launch_bounds__(256) void copy2d_multi_kernel(const CopyEntry* __restrict__ tab, int n, long long total_chunks) {
  for (long long chunk = blockIdx.x; chunk < total_chunks; chunk += gridDim.x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab[mid].start <= chunk) lo = mid; else hi = mid - 1; }
    const CopyEntry t = tab[lo];
    const long long total = (long long)t.rows * t.cols, base = (chunk - t.start) * 1024;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long e = base + u * 256 + threadIdx.x;
      if (e < total) { const int r = (int)(e / t.cols), c = (int)(e - (long long)r * t.cols); t.dst[(size_t)r * t.dst_pitch + c] = t.src[e]; }
    }
  }
}
}  // namespace

extern "C" int sl_copy2d_multi(const void* table_dev, int n, long long total_chunks, sl_stream_t stream) {
  SL_REQUIRE(table_dev && n > 0 && total_chunks > 0, "copy2d_multi: bad args");
  const int blocks = (int)(total_chunks < 4096 ? total_chunks : 4096);
  hipLaunchKernelGGL(copy2d_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const CopyEntry*)table_dev, n, total_chunks);
  SL_LAUNCH_CHECK("copy2d_multi_kernel");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// The relative-position bias tiles of every Swin block in ONE launch (swintransformer.py:128-131: table[index].view(n, n, heads).permute(2, 0, 1)):
//   out[h][p] = table[index[p]][h],  p < npair = 49 * 49.   One gather + one permute copy per block and optimizer step otherwise (24 launches of ~5 us per Swin-T step).
// table: device array of n entries {out (float*), table (const float*), index (const int64*), heads, npair (int)} = 32 bytes; block b of the grid serves entry b / 4, quarter b % 4.
namespace {
struct RelGatherEntry { float* out; const float* table; const long long* index; int heads, npair; };
static_assert(sizeof(RelGatherEntry) == 32, "table layout is part of the ABI");
__global__ __launch_bounds__(256) void relpos_gather_multi_kernel(const RelGatherEntry* __restrict__ tab) {
  const RelGatherEntry t = tab[blockIdx.x >> 2];
  const int total = t.heads * t.npair;
  for (int e = (blockIdx.x & 3) * 256 + threadIdx.x; e < total; e += 1024) {
    const int h = e / t.npair, pp = e - h * t.npair;
    t.out[e] = t.table[(size_t)t.index[pp] * t.heads + h];
  }
}
}  // namespace

extern "C" int sl_relpos_gather_multi(const void* table_dev, int n, sl_stream_t stream) {
  SL_REQUIRE(table_dev && n > 0, "relpos_gather_multi: bad args");
  hipLaunchKernelGGL(relpos_gather_multi_kernel, dim3(4 * n), dim3(256), 0, (hipStream_t)stream, (const RelGatherEntry*)table_dev);
  SL_LAUNCH_CHECK("relpos_gather_multi_kernel");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// clip_grad_norm_'s total norm over a list of gradient tensors and the clip coefficient, without torch's ~12 launches (fill, four multi-tensor norm kernels, clean-up,
// two concatenations, a reduce, reciprocal, clamp, copy: 130 us of a ResNet-50 step).  The gradient addresses travel as KERNEL ARGUMENTS (up to SL_NORM_MAX per launch), so
// a captured step bakes them in like every other launch; partial sums of squares per 4096-element chunk, summed in a fixed order by the finalize launch (deterministic).
//   norm = sqrt(sum g^2) * inv_div,   coef = min(1, max_norm / (norm + 1e-6)) * inv_div          (inv_div = 1 / world size when the gradients hold the SUM over ranks)
namespace {
__global__ __launch_bounds__(256) void grad_sqnorm_multi_kernel(SlNormBatch b, float* __restrict__ partial) {
  __shared__ float red[4];
  const int chunk = blockIdx.x;
  int lo = 0, hi = b.n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (b.chunk0[mid] <= chunk) lo = mid; else hi = mid - 1; }
  const float* g = (const float*)b.grad[lo];
  const long long base = (long long)(chunk - b.chunk0[lo]) * 4096, n = b.numel[lo];
  float s = 0.f;
  if (base + 4096 <= n && (((size_t)(g + base)) & 15) == 0) {
    const float4* q = (const float4*)(g + base);
#pragma unroll
    for (int u = 0; u < 4; ++u) { const float4 v = q[u * 256 + threadIdx.x]; s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
  } else {
    for (int u = 0; u < 16; ++u) { const long long e = base + u * 256 + threadIdx.x; if (e < n) { const float v = g[e]; s += v * v; } }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[b.chunk_base + chunk] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(1024) void grad_norm_finalize_kernel(const float* __restrict__ partial, int nchunks, float max_norm, float inv_div, float* __restrict__ out) {
  __shared__ double red[1024];
  double s = 0.0;
  for (int i = threadIdx.x; i < nchunks; i += 1024) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]) * inv_div;
    float coef = max_norm / (norm + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
    out[0] = norm; out[1] = coef * inv_div;
  }
}
}  // namespace

extern "C" int sl_grad_sqnorm_multi(const SlNormBatch* batch, float* partial, sl_stream_t stream) {
  SL_REQUIRE(batch && partial && batch->n > 0 && batch->n <= SL_NORM_MAX && batch->chunk0[0] == 0 && batch->chunk0[batch->n] > 0, "grad_sqnorm_multi: bad batch");
  hipLaunchKernelGGL(grad_sqnorm_multi_kernel, dim3(batch->chunk0[batch->n]), dim3(256), 0, (hipStream_t)stream, *batch, partial);
  SL_LAUNCH_CHECK("grad_sqnorm_multi_kernel");
  return 0;
}
extern "C" int sl_grad_norm_finalize(const float* partial, int nchunks, float max_norm, float inv_div, float* out, sl_stream_t stream) {
  SL_REQUIRE(partial && out && nchunks > 0 && max_norm > 0.f && inv_div > 0.f, "grad_norm_finalize: bad args");
  hipLaunchKernelGGL(grad_norm_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, nchunks, max_norm, inv_div, out);
  SL_LAUNCH_CHECK("grad_norm_finalize_kernel");
  return 0;
}
