// The pooled maps and the stage tensors (B*(1+4+9+36) rows) are ALWAYS float: train-mode BatchNorm over 2..576 samples
// is hypersensitive to rounding of its input, and these tensors are tiny.
// Pyramid pooling side kernels (networks/pspnet_pop.py:26,33): AdaptiveAvgPool2d to (1,2,3,6) in ONE pass over the
// feature map, and the bilinear (align_corners=False) upsampling of the four stage outputs, forward and backward.
// All HBM-bound; the feature map is read exactly once in each direction.
#include "common.h"

namespace {

struct PpmGeom {
  int B, H, W, C, nlevels;
  int sizes[4];
  int rowoff[5];         // first pooled row of each level (rows are (b, i, j) within a level)
  int ncy, ncx;          // cells of the common refinement of all bin boundaries
  int yb[40], xb[40];
};

inline int bin_start(int i, int in, int out) { return (i * in) / out; }
inline int bin_end(int i, int in, int out) { return ((i + 1) * in + out - 1) / out; }
__device__ __forceinline__ int d_bin_start(int i, int in, int out) { return (i * in) / out; }
__device__ __forceinline__ int d_bin_end(int i, int in, int out) { return ((i + 1) * in + out - 1) / out; }

int make_bounds(const SlPpmDesc* d, int in, int* out) {
  bool mark[4097] = {false};
  if (in > 4096) return -1;
  for (int l = 0; l < d->nlevels; ++l)
    for (int i = 0; i < d->sizes[l]; ++i) { mark[bin_start(i, in, d->sizes[l])] = true; mark[bin_end(i, in, d->sizes[l])] = true; }
  int n = 0;
  for (int v = 0; v <= in; ++v) if (mark[v]) { if (n >= 40) return -1; out[n++] = v; }
  return n - 1;
}

int make_geom(const SlPpmDesc* d, PpmGeom& g) {
  SL_REQUIRE(d && d->B > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->C % 8 == 0, "ppm: bad sizes");
  SL_REQUIRE(d->nlevels >= 1 && d->nlevels <= 4, "ppm: 1..4 levels");
  g.B = d->B; g.H = d->H; g.W = d->W; g.C = d->C; g.nlevels = d->nlevels;
  int off = 0;
  for (int l = 0; l < 4; ++l) {
    g.sizes[l] = l < d->nlevels ? d->sizes[l] : 0;
    g.rowoff[l] = off;
    if (l < d->nlevels) { SL_REQUIRE(d->sizes[l] >= 1 && d->sizes[l] <= d->H && d->sizes[l] <= d->W, "ppm: bad level size"); off += d->B * d->sizes[l] * d->sizes[l]; }
  }
  g.rowoff[4] = off;
  g.ncy = make_bounds(d, d->H, g.yb);
  g.ncx = make_bounds(d, d->W, g.xb);
  SL_REQUIRE(g.ncy > 0 && g.ncx > 0, "ppm: too many distinct bin boundaries");
  return 0;
}

// cells[b][cy][cx][c] = sum of x over the cell (fp32)
template <typename T>
__global__ void ppm_cells_kernel(PpmGeom g, const T* __restrict__ x, float* __restrict__ cells) {
  constexpr int V = Vec16<T>::N;
  const int nv = g.C / V;
  const long long total = (long long)g.B * g.ncy * g.ncx * nv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % nv); long long r = i / nv;
    const int cx = (int)(r % g.ncx); r /= g.ncx;
    const int cy = (int)(r % g.ncy); const int b = (int)(r / g.ncy);
    float s[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s[k] = 0.f;
    for (int y = g.yb[cy]; y < g.yb[cy + 1]; ++y)
      for (int xx = g.xb[cx]; xx < g.xb[cx + 1]; ++xx) {
        float t[V];
        unpack16<T>(*(const uint4*)(x + ((size_t)(b * g.H + y) * g.W + xx) * g.C + v * V), t);
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] += t[k];
      }
    float* o = cells + (size_t)i * V;
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = s[k];
  }
}

// pooled[level rows][C] = (sum of the cells inside the bin) / bin area
__global__ void ppm_bins_kernel(PpmGeom g, const float* __restrict__ cells, float* __restrict__ pooled) {
  const long long total = (long long)g.rowoff[4] * g.C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % g.C); const int row = (int)(i / g.C);
    int l = 0;
    while (l + 1 < g.nlevels && row >= g.rowoff[l + 1]) ++l;
    const int s = g.sizes[l];
    int r = row - g.rowoff[l];
    const int j = r % s; r /= s;
    const int ii = r % s; const int b = r / s;
    const int ys = d_bin_start(ii, g.H, s), ye = d_bin_end(ii, g.H, s), xs = d_bin_start(j, g.W, s), xe = d_bin_end(j, g.W, s);
    float sum = 0.f;
    for (int cy = 0; cy < g.ncy; ++cy) {
      if (g.yb[cy] < ys || g.yb[cy + 1] > ye) continue;
      for (int cx = 0; cx < g.ncx; ++cx) {
        if (g.xb[cx] < xs || g.xb[cx + 1] > xe) continue;
        sum += cells[(((size_t)b * g.ncy + cy) * g.ncx + cx) * g.C + c];
      }
    }
    pooled[i] = sum / (float)((ye - ys) * (xe - xs));
  }
}

template <typename T>
__global__ void ppm_pool_bwd_kernel(PpmGeom g, const float* __restrict__ dpooled, const T* __restrict__ dcat, int cat_pitch, int cat_off,
                                    T* __restrict__ dx) {
  constexpr int V = Vec16<T>::N;
  const int nv = g.C / V;
  const long long total = (long long)g.B * g.H * g.W * nv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % nv); long long r = i / nv;
    const int xx = (int)(r % g.W); r /= g.W;
    const int y = (int)(r % g.H); const int b = (int)(r / g.H);
    float acc[V];
    if (dcat) unpack16<T>(*(const uint4*)(dcat + ((size_t)(b * g.H + y) * g.W + xx) * cat_pitch + cat_off + v * V), acc);
    else {
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] = 0.f;
    }
    for (int l = 0; l < g.nlevels; ++l) {
      const int s = g.sizes[l];
      const int i0 = (y * s) / g.H, j0 = (xx * s) / g.W;
      for (int ii = max(0, i0 - 1); ii <= min(s - 1, i0 + 1); ++ii) {
        const int ys = d_bin_start(ii, g.H, s), ye = d_bin_end(ii, g.H, s);
        if (y < ys || y >= ye) continue;
        for (int j = max(0, j0 - 1); j <= min(s - 1, j0 + 1); ++j) {
          const int xs = d_bin_start(j, g.W, s), xe = d_bin_end(j, g.W, s);
          if (xx < xs || xx >= xe) continue;
          const float* t = dpooled + ((size_t)g.rowoff[l] + (size_t)(b * s + ii) * s + j) * g.C + v * V;
          const float inv = 1.f / (float)((ye - ys) * (xe - xs));
#pragma unroll
          for (int k = 0; k < V; ++k) acc[k] += t[k] * inv;
        }
      }
    }
    *(uint4*)(dx + (size_t)i * V) = pack16<T>(acc);
  }
}

// ATen upsample_bilinear2d source index, align_corners = False
__device__ __forceinline__ void src_index_ac0(int dst, int in, int out, int& i0, int& i1, float& l1) {
  const float scale = (float)in / (float)out;
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
}

template <typename T>
__global__ void ppm_upsample_fwd_kernel(PpmGeom g, int Cs, const float* __restrict__ stage, T* __restrict__ priors) {
  constexpr int V = Vec16<T>::N;
  const int nv = Cs / V, pitch = g.nlevels * Cs;
  const long long total = (long long)g.B * g.H * g.W * g.nlevels * nv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % nv); long long r = i / nv;
    const int l = (int)(r % g.nlevels); r /= g.nlevels;
    const int xx = (int)(r % g.W); r /= g.W;
    const int y = (int)(r % g.H); const int b = (int)(r / g.H);
    const int s = g.sizes[l];
    int y0, y1, x0, x1; float ly, lx;
    src_index_ac0(y, s, g.H, y0, y1, ly);
    src_index_ac0(xx, s, g.W, x0, x1, lx);
    const float* base = stage + ((size_t)g.rowoff[l] + (size_t)b * s * s) * Cs + v * V;
    const float* v00 = base + (size_t)(y0 * s + x0) * Cs;
    const float* v01 = base + (size_t)(y0 * s + x1) * Cs;
    const float* v10 = base + (size_t)(y1 * s + x0) * Cs;
    const float* v11 = base + (size_t)(y1 * s + x1) * Cs;
    float o[V];
    const float wy0 = 1.f - ly, wx0 = 1.f - lx;
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = wy0 * (wx0 * v00[k] + lx * v01[k]) + ly * (wx0 * v10[k] + lx * v11[k]);
    *(uint4*)(priors + ((size_t)(b * g.H + y) * g.W + xx) * pitch + l * Cs + v * V) = pack16<T>(o);
  }
}

// stage A: tmp[b][y][lj][c] = sum_x wx(x, j) * dcat[b][y][x][l*Cs + c]   (lj enumerates (level, j))
template <typename T>
__global__ void ppm_upsample_bwd_x_kernel(PpmGeom g, int Cs, const T* __restrict__ dcat, int cat_pitch, float* __restrict__ tmp, int nlj) {
  constexpr int V = Vec16<T>::N;
  const int nv = Cs / V;
  const long long total = (long long)g.B * g.H * nlj * nv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % nv); long long r = i / nv;
    int lj = (int)(r % nlj); r /= nlj;
    const int y = (int)(r % g.H); const int b = (int)(r / g.H);
    int l = 0;
    while (lj >= g.sizes[l]) { lj -= g.sizes[l]; ++l; }
    const int j = lj, s = g.sizes[l];
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    for (int xx = 0; xx < g.W; ++xx) {
      int x0, x1; float lx;
      src_index_ac0(xx, s, g.W, x0, x1, lx);
      const float wgt = (x0 == j ? 1.f - lx : 0.f) + (x1 == j ? lx : 0.f);
      if (wgt == 0.f) continue;
      float t[V];
      unpack16<T>(*(const uint4*)(dcat + ((size_t)(b * g.H + y) * g.W + xx) * cat_pitch + l * Cs + v * V), t);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += wgt * t[k];
    }
    float* o = tmp + (size_t)i * V;
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = acc[k];
  }
}

// stage B: dstage[row(l,b,i,j)][c] = sum_y wy(y, i) * tmp[b][y][lj][c]
__global__ void ppm_upsample_bwd_y_kernel(PpmGeom g, int Cs, const float* __restrict__ tmp, float* __restrict__ dstage, int nlj) {
  const long long total = (long long)g.rowoff[4] * Cs;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cs); const int row = (int)(i / Cs);
    int l = 0, ljoff = 0;
    while (l + 1 < g.nlevels && row >= g.rowoff[l + 1]) { ljoff += g.sizes[l]; ++l; }
    const int s = g.sizes[l];
    int r = row - g.rowoff[l];
    const int j = r % s; r /= s;
    const int ii = r % s; const int b = r / s;
    float acc = 0.f;
    for (int y = 0; y < g.H; ++y) {
      int y0, y1; float ly;
      src_index_ac0(y, s, g.H, y0, y1, ly);
      const float wgt = (y0 == ii ? 1.f - ly : 0.f) + (y1 == ii ? ly : 0.f);
      if (wgt == 0.f) continue;
      acc += wgt * tmp[(((size_t)b * g.H + y) * nlj + ljoff + j) * Cs + c];
    }
    dstage[i] = acc;
  }
}

inline int gs_blocks(long long n) { long long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }
inline int sum_sizes(const PpmGeom& g) { int n = 0; for (int l = 0; l < g.nlevels; ++l) n += g.sizes[l]; return n; }

}  // namespace

extern "C" size_t sl_ppm_workspace(const SlPpmDesc* d) {
  PpmGeom g;
  if (make_geom(d, g)) return 0;
  const size_t cells = (size_t)g.B * g.ncy * g.ncx * g.C * sizeof(float);
  const size_t tmp = (size_t)g.B * g.H * sum_sizes(g) * g.C * sizeof(float);   // upper bound (Cs <= C)
  return cells > tmp ? cells : tmp;
}

extern "C" int sl_ppm_pool_fwd(const SlPpmDesc* d, const void* x, float* pooled, void* workspace, size_t workspace_bytes,
                               sl_stream_t stream) {
  PpmGeom g;
  if (int e = make_geom(d, g)) return e;
  SL_REQUIRE(x && pooled && workspace, "ppm_pool_fwd: null buffer");
  const size_t need = (size_t)g.B * g.ncy * g.ncx * g.C * sizeof(float);
  if (workspace_bytes < need) { sl_set_error("ppm_pool_fwd: workspace %zu < %zu", workspace_bytes, need); return SL_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  float* cells = (float*)workspace;
  if (d->dtype == SL_BF16) {
    hipLaunchKernelGGL(ppm_cells_kernel<bf16_t>, dim3(gs_blocks((long long)g.B * g.ncy * g.ncx * g.C / 8)), dim3(256), 0, st, g, (const bf16_t*)x, cells);
    hipLaunchKernelGGL(ppm_bins_kernel, dim3(gs_blocks((long long)g.rowoff[4] * g.C)), dim3(256), 0, st, g, cells, (float*)pooled);
  } else if (d->dtype == SL_F32) {
    hipLaunchKernelGGL(ppm_cells_kernel<float>, dim3(gs_blocks((long long)g.B * g.ncy * g.ncx * g.C / 4)), dim3(256), 0, st, g, (const float*)x, cells);
    hipLaunchKernelGGL(ppm_bins_kernel, dim3(gs_blocks((long long)g.rowoff[4] * g.C)), dim3(256), 0, st, g, cells, (float*)pooled);
  } else SL_REQUIRE(false, "ppm_pool_fwd: bad dtype");
  SL_LAUNCH_CHECK("ppm_pool_fwd");
  return 0;
}

extern "C" int sl_ppm_pool_bwd(const SlPpmDesc* d, const float* dpooled, const void* dcat, int cat_pitch, int cat_off, void* dx,
                               sl_stream_t stream) {
  PpmGeom g;
  if (int e = make_geom(d, g)) return e;
  SL_REQUIRE(dpooled && dx, "ppm_pool_bwd: null buffer");
  SL_REQUIRE(!dcat || (cat_pitch >= cat_off + g.C && cat_pitch % 8 == 0 && cat_off % 8 == 0), "ppm_pool_bwd: bad concat geometry");
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == SL_BF16)
    hipLaunchKernelGGL(ppm_pool_bwd_kernel<bf16_t>, dim3(gs_blocks((long long)g.B * g.H * g.W * g.C / 8)), dim3(256), 0, st, g, (const float*)dpooled, (const bf16_t*)dcat, cat_pitch, cat_off, (bf16_t*)dx);
  else if (d->dtype == SL_F32)
    hipLaunchKernelGGL(ppm_pool_bwd_kernel<float>, dim3(gs_blocks((long long)g.B * g.H * g.W * g.C / 4)), dim3(256), 0, st, g, (const float*)dpooled, (const float*)dcat, cat_pitch, cat_off, (float*)dx);
  else SL_REQUIRE(false, "ppm_pool_bwd: bad dtype");
  SL_LAUNCH_CHECK("ppm_pool_bwd_kernel");
  return 0;
}

extern "C" int sl_ppm_upsample_fwd(const SlPpmDesc* d, int Cs, const float* stage, void* priors, sl_stream_t stream) {
  PpmGeom g;
  if (int e = make_geom(d, g)) return e;
  SL_REQUIRE(stage && priors && Cs > 0 && Cs % 8 == 0, "ppm_upsample_fwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == SL_BF16)
    hipLaunchKernelGGL(ppm_upsample_fwd_kernel<bf16_t>, dim3(gs_blocks((long long)g.B * g.H * g.W * g.nlevels * Cs / 8)), dim3(256), 0, st, g, Cs, (const float*)stage, (bf16_t*)priors);
  else if (d->dtype == SL_F32)
    hipLaunchKernelGGL(ppm_upsample_fwd_kernel<float>, dim3(gs_blocks((long long)g.B * g.H * g.W * g.nlevels * Cs / 4)), dim3(256), 0, st, g, Cs, (const float*)stage, (float*)priors);
  else SL_REQUIRE(false, "ppm_upsample_fwd: bad dtype");
  SL_LAUNCH_CHECK("ppm_upsample_fwd_kernel");
  return 0;
}

extern "C" int sl_ppm_upsample_bwd(const SlPpmDesc* d, int Cs, const void* dcat, int cat_pitch, float* dstage, void* workspace,
                                   size_t workspace_bytes, sl_stream_t stream) {
  PpmGeom g;
  if (int e = make_geom(d, g)) return e;
  SL_REQUIRE(dcat && dstage && workspace && Cs > 0 && Cs % 8 == 0 && cat_pitch >= g.nlevels * Cs, "ppm_upsample_bwd: bad args");
  const int nlj = sum_sizes(g);
  const size_t need = (size_t)g.B * g.H * nlj * Cs * sizeof(float);
  if (workspace_bytes < need) { sl_set_error("ppm_upsample_bwd: workspace %zu < %zu", workspace_bytes, need); return SL_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  float* tmp = (float*)workspace;
  if (d->dtype == SL_BF16) {
    hipLaunchKernelGGL(ppm_upsample_bwd_x_kernel<bf16_t>, dim3(gs_blocks((long long)g.B * g.H * nlj * Cs / 8)), dim3(256), 0, st, g, Cs, (const bf16_t*)dcat, cat_pitch, tmp, nlj);
    hipLaunchKernelGGL(ppm_upsample_bwd_y_kernel, dim3(gs_blocks((long long)g.rowoff[4] * Cs)), dim3(256), 0, st, g, Cs, tmp, (float*)dstage, nlj);
  } else if (d->dtype == SL_F32) {
    hipLaunchKernelGGL(ppm_upsample_bwd_x_kernel<float>, dim3(gs_blocks((long long)g.B * g.H * nlj * Cs / 4)), dim3(256), 0, st, g, Cs, (const float*)dcat, cat_pitch, tmp, nlj);
    hipLaunchKernelGGL(ppm_upsample_bwd_y_kernel, dim3(gs_blocks((long long)g.rowoff[4] * Cs)), dim3(256), 0, st, g, Cs, tmp, (float*)dstage, nlj);
  } else SL_REQUIRE(false, "ppm_upsample_bwd: bad dtype");
  SL_LAUNCH_CHECK("ppm_upsample_bwd");
  return 0;
}
