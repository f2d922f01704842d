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
    // s > H is legal: the ATen bin rule then yields overlapping / repeated bins (os 32 on small tiles: 5x4 maps against the 6x6 level)
    if (l < d->nlevels) { SL_REQUIRE(d->sizes[l] >= 1 && d->sizes[l] <= 64, "ppm: bad level size"); off += d->B * d->sizes[l] * d->sizes[l]; }
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

// The same sums for small batches (a fine-tune pair: 2 x 8 x 8 cells x 256 channel vectors = 128 blocks of threads that each walk ~100 pixels one dependent load after the
// other: 51 us whatever the batch).  A block owns one cell x 32 channel vectors; its 8 row lanes take the cell's rows y0 + lane, + 8, ... and the lane sums are added in
// lane order through the LDS (fixed order: bit-stable; another order than ppm_cells_kernel's row-major walk, so the two kernels agree to fp32 rounding only).
template <typename T>
__global__ __launch_bounds__(256) void ppm_cells_rows_kernel(PpmGeom g, const T* __restrict__ x, float* __restrict__ cells) {
  constexpr int V = Vec16<T>::N;
  __shared__ float red[8][32][V + 1];
  const int nv = g.C / V, vblocks = (nv + 31) / 32;
  int bid = blockIdx.x;
  const int vb = bid % vblocks; bid /= vblocks;
  const int cx = bid % g.ncx; bid /= g.ncx;
  const int cy = bid % g.ncy; const int b = bid / g.ncy;
  const int lv = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int v = vb * 32 + lv;
  float s[V];
#pragma unroll
  for (int k = 0; k < V; ++k) s[k] = 0.f;
  if (v < nv)
    for (int y = g.yb[cy] + rl; y < g.yb[cy + 1]; y += 8)
      for (int xx = g.xb[cx]; xx < g.xb[cx + 1]; ++xx) {
        float t[V];
        unpack16<T>(*(const uint4*)(x + ((size_t)(b * g.H + y) * g.W + xx) * g.C + v * V), t);
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] += t[k];
      }
#pragma unroll
  for (int k = 0; k < V; ++k) red[rl][lv][k] = s[k];
  __syncthreads();
  if (rl == 0 && v < nv) {
    float* o = cells + ((((size_t)b * g.ncy + cy) * g.ncx + cx) * nv + v) * V;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float a = red[0][lv][k];
#pragma unroll
      for (int r = 1; r < 8; ++r) a += red[r][lv][k];
      o[k] = a;
    }
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

// Backward of the four adaptive poolings.  The gradient is constant on every cell of the common refinement of the bin grids, so it
// is first tabulated per cell (a few hundred cells per image) and then added to the direct (concat) gradient in a pure streaming pass.
//   gcell[b][cy][cx][c] = sum over levels of dpooled[bin containing the cell][c] / bin area
__global__ void ppm_pool_bwd_cells_kernel(PpmGeom g, const float* __restrict__ dpooled, float* __restrict__ gcell) {
  const long long total = (long long)g.B * g.ncy * g.ncx * g.C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % g.C); long long r = i / g.C;
    const int cx = (int)(r % g.ncx); r /= g.ncx;
    const int cy = (int)(r % g.ncy); const int b = (int)(r / g.ncy);
    const int y0 = g.yb[cy], y1 = g.yb[cy + 1], x0 = g.xb[cx], x1 = g.xb[cx + 1];
    float acc = 0.f;
    int nbins = 0;
    for (int l = 0; l < g.nlevels; ++l) nbins += g.sizes[l] * g.sizes[l];
    if (g.C % 64 == 0 && nbins <= 64) {
      // a wavefront shares (b, cy, cx): lane q decides once whether bin q (levels in order, (ii, j) row-major inside a level) contains the cell and with which weight;
      // the loop reads the 50 decisions back lane by lane -- every thread walked all bins with their integer divisions before (the launch was bound by that arithmetic)
      const int lane = threadIdx.x & 63;
      float wq = 0.f; int rowq = 0;
      {
        int q = lane, l = 0;
        while (l < g.nlevels && q >= g.sizes[l] * g.sizes[l]) { q -= g.sizes[l] * g.sizes[l]; ++l; }
        if (l < g.nlevels) {
          const int s = g.sizes[l], ii = q / s, j = q - ii * s;
          const int ys = d_bin_start(ii, g.H, s), ye = d_bin_end(ii, g.H, s), xs = d_bin_start(j, g.W, s), xe = d_bin_end(j, g.W, s);
          if (!(y0 < ys || y1 > ye) && !(x0 < xs || x1 > xe)) { wq = 1.f / (float)((ye - ys) * (xe - xs)); rowq = g.rowoff[l] + (b * s + ii) * s + j; }
        }
      }
      for (int q = 0; q < nbins; ++q) {
        const float w = __shfl(wq, q, 64);
        if (w == 0.f) continue;
        acc += dpooled[(size_t)__shfl(rowq, q, 64) * g.C + c] * w;
      }
      gcell[i] = acc;
      continue;
    }
    for (int l = 0; l < g.nlevels; ++l) {
      const int s = g.sizes[l];
      for (int ii = 0; ii < s; ++ii) {
        const int ys = d_bin_start(ii, g.H, s), ye = d_bin_end(ii, g.H, s);
        if (y0 < ys || y1 > ye) continue;
        for (int j = 0; j < s; ++j) {
          const int xs = d_bin_start(j, g.W, s), xe = d_bin_end(j, g.W, s);
          if (x0 < xs || x1 > xe) continue;
          acc += dpooled[((size_t)g.rowoff[l] + (size_t)(b * s + ii) * s + j) * g.C + c] * (1.f / (float)((ye - ys) * (xe - xs)));
        }
      }
    }
    gcell[i] = acc;
  }
}

// dx[b][y][x][:] = dcat[b][y][x][cat_off ..] + gcell[b][cell(y)][cell(x)][:]      one block per image row
template <typename T>
__global__ __launch_bounds__(256) void ppm_pool_bwd_kernel(PpmGeom g, const float* __restrict__ gcell, const T* __restrict__ dcat, int cat_pitch, int cat_off,
                                                           T* __restrict__ dx) {
  constexpr int V = Vec16<T>::N;
  __shared__ int cxs[4096];
  const int by = blockIdx.x, b = by / g.H, y = by - b * g.H;
  int cy = 0;
  while (cy + 1 < g.ncy && y >= g.yb[cy + 1]) ++cy;
  for (int xx = threadIdx.x; xx < g.W; xx += 256) {
    int cx = 0;
    while (cx + 1 < g.ncx && xx >= g.xb[cx + 1]) ++cx;
    cxs[xx] = cx;
  }
  __syncthreads();
  const int nv = g.C / V;
  const float* gc = gcell + ((size_t)b * g.ncy + cy) * g.ncx * g.C;
  const size_t row = (size_t)by * g.W;
  for (int i = threadIdx.x; i < g.W * nv; i += 256) {
    const int xx = i / nv, v = i - xx * nv;
    float acc[V];
    if (dcat) unpack16<T>(*(const uint4*)(dcat + (row + xx) * cat_pitch + cat_off + v * V), acc);
    else {
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] = 0.f;
    }
    const float* t = gc + (size_t)cxs[xx] * g.C + v * V;
#pragma unroll
    for (int k = 0; k < V; k += 4) {
      const float4 q = *(const float4*)(t + k);
      acc[k] += q.x; acc[k + 1] += q.y; acc[k + 2] += q.z; acc[k + 3] += q.w;
    }
    *(uint4*)(dx + ((row + xx) * nv + v) * V) = pack16<T>(acc);
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

// ------------------------------------------------------------------------------------------------------------------
// Factorised prior half of the PPM bottleneck conv (networks/pspnet_pop.py:19,33-34).  For level l with stage map
// P_l [B,s,s,Cs]:   conv3x3(upsample(P_l))[y,x,n] = sum_tap sum_(i,j) u_tap(y,x; i,j) * Q_l[i,j,(tap,n)],
// Q_l[i,j,(tap,n)] = sum_c W[n][l*Cs+c][tap] * P_l[i,j,c]  (a 1x1 conv on the s x s grid, done by the MFMA kernel), and
// u_tap = [pixel shifted by the tap is inside the map] * (bilinear weight of the shifted pixel on cell (i,j)), which is
// separable in y and x.  Exact by linearity; removes 77 of the 154.6 GFLOP/tile of this conv (and of its dgrad / wgrad).

// weight re-layout W_oihw [N][Ctot][3][3] -> per level wq_f [9N][Cs] (1x1 forward layout) and wq_b [Cs][9N] (dgrad layout): sl_ppm_wq_prep in conv_weight_prep.hip

// dwq [l][(tap,n)][c] -> dw_oihw[n][l*Cs + c][tap]
__global__ void ppm_dwq_scatter_kernel(const float* __restrict__ dwq, int N, int Ctot, int Cs, int nl, float* __restrict__ dw) {
  const long long total = (long long)nl * 9 * N * Cs;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % Cs); long long r = e / Cs;
    const int n = (int)(r % N); r /= N;
    const int tap = (int)(r % 9); const int l = (int)(r / 9);
    dw[((size_t)n * Ctot + l * Cs + c) * 9 + tap] = dwq[e];
  }
}

// bilinear weight of destination index d (shifted by the tap, may fall outside) on source cell `cell`
__device__ __forceinline__ float tap_weight(int d, int s, int size, int cell) {
  if ((unsigned)d >= (unsigned)size) return 0.f;
  int i0, i1; float l1;
  src_index_ac0(d, s, size, i0, i1, l1);
  return (i0 == cell ? 1.f - l1 : 0.f) + (i1 == cell ? l1 : 0.f);
}

// stage 1 of the gather: t[b][y][kx][lj][n] = sum_ky sum_i wy_ky(y,i) * q_l[b][i][j][(ky*3+kx)*N + n]
__global__ void ppm_fact_gather1_kernel(PpmGeom g, int N, const float* __restrict__ q, float* __restrict__ t, int nlj) {
  const int nv = N / 4;
  const long long total = (long long)g.B * g.H * 3 * nlj * nv;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(e % nv); long long r = e / nv;
    int lj = (int)(r % nlj); r /= nlj;
    const int kx = (int)(r % 3); r /= 3;
    const int y = (int)(r % g.H); const int b = (int)(r / g.H);
    int l = 0;
    while (lj >= g.sizes[l]) { lj -= g.sizes[l]; ++l; }
    const int s = g.sizes[l], j = lj;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < 3; ++ky) {
      const int d = y + ky - 1;
      if ((unsigned)d >= (unsigned)g.H) continue;
      int i0, i1; float l1;
      src_index_ac0(d, s, g.H, i0, i1, l1);
      const float* base = q + ((size_t)g.rowoff[l] + (size_t)b * s * s) * 9 * N + (ky * 3 + kx) * N + v * 4;
      const float4 a = *(const float4*)(base + (size_t)(i0 * s + j) * 9 * N);
      const float4 c = *(const float4*)(base + (size_t)(i1 * s + j) * 9 * N);
      const float w0 = 1.f - l1;
      acc.x += w0 * a.x + l1 * c.x; acc.y += w0 * a.y + l1 * c.y; acc.z += w0 * a.z + l1 * c.z; acc.w += w0 * a.w + l1 * c.w;
    }
    *(float4*)(t + (size_t)e * 4) = acc;
  }
}

// stage 2: gout[b][y][x][n] = sum_l sum_kx sum_j wx_kx(x,j) * t[b][y][kx][ljoff_l + j][n]   (written in the compute dtype)
template <typename T>
__global__ void ppm_fact_gather2_kernel(PpmGeom g, int N, const float* __restrict__ t, T* __restrict__ gout, int nlj) {
  constexpr int V = Vec16<T>::N;
  const int nv = N / V;
  const long long total = (long long)g.B * g.H * g.W * nv;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(e % nv); long long r = e / nv;
    const int x = (int)(r % g.W); r /= g.W;
    const int y = (int)(r % g.H); const int b = (int)(r / g.H);
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    int ljoff = 0;
    for (int l = 0; l < g.nlevels; ++l) {
      const int s = g.sizes[l];
      for (int kx = 0; kx < 3; ++kx) {
        const int d = x + kx - 1;
        if ((unsigned)d >= (unsigned)g.W) continue;
        int j0, j1; float l1;
        src_index_ac0(d, s, g.W, j0, j1, l1);
        const float* base = t + ((((size_t)b * g.H + y) * 3 + kx) * nlj + ljoff) * N + v * V;
        const float w0 = 1.f - l1;
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] += w0 * base[(size_t)j0 * N + k] + l1 * base[(size_t)j1 * N + k];
      }
      ljoff += s;
    }
    *(uint4*)(gout + (size_t)e * V) = pack16<T>(acc);
  }
}

// transpose of the gather, stage A: sa[b][y][kx][lj][n] = sum_x wx_kx(x,j) * dcb[b][y][x][n]
template <typename T>
__global__ void ppm_fact_scatterA_kernel(PpmGeom g, int N, const T* __restrict__ dcb, float* __restrict__ sa, int nlj) {
  constexpr int V = Vec16<T>::N;
  const int nv = N / V;
  const long long total = (long long)g.B * g.H * 3 * nlj * nv;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(e % nv); long long r = e / nv;
    int lj = (int)(r % nlj); r /= nlj;
    const int kx = (int)(r % 3); r /= 3;
    const int y = (int)(r % g.H); const int b = (int)(r / g.H);
    int l = 0;
    while (lj >= g.sizes[l]) { lj -= g.sizes[l]; ++l; }
    const int s = g.sizes[l], j = lj;
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    if (nv % 64 == 0) {
      // the 64 lanes of a wavefront differ only in v: lane x evaluates the weight of column x0 + x once and the loop reads it back lane by lane -- every thread evaluated
      // all W weights before (~40 instructions each, the kernel was bound by them), and the test for a zero weight is wave-uniform now
      const int lane = threadIdx.x & 63;
      for (int x0 = 0; x0 < g.W; x0 += 64) {
        const float wl = x0 + lane < g.W ? tap_weight(x0 + lane + kx - 1, s, g.W, j) : 0.f;
        const int xe = g.W - x0 < 64 ? g.W - x0 : 64;
        for (int xx = 0; xx < xe; ++xx) {
          const float wgt = __shfl(wl, xx, 64);
          if (wgt == 0.f) continue;
          float d[V];
          unpack16<T>(*(const uint4*)(dcb + (((size_t)b * g.H + y) * g.W + x0 + xx) * N + v * V), d);
#pragma unroll
          for (int k = 0; k < V; ++k) acc[k] += wgt * d[k];
        }
      }
    } else {
      for (int x = 0; x < g.W; ++x) {
        const float wgt = tap_weight(x + kx - 1, s, g.W, j);
        if (wgt == 0.f) continue;
        float d[V];
        unpack16<T>(*(const uint4*)(dcb + (((size_t)b * g.H + y) * g.W + x) * N + v * V), d);
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] += wgt * d[k];
      }
    }
    float* o = sa + (size_t)e * V;
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = acc[k];
  }
}

// stage B: gq[row_l(b,i,j)][(ky*3+kx)*N + n] = sum_y wy_ky(y,i) * sa[b][y][kx][lj][n]
__global__ void ppm_fact_scatterB_kernel(PpmGeom g, int N, const float* __restrict__ sa, float* __restrict__ gq, int nlj) {
  const long long total = (long long)g.rowoff[4] * 9 * N;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(e % N); long long r = e / N;
    const int tap = (int)(r % 9); const int row = (int)(r / 9);
    const int ky = tap / 3, kx = tap - 3 * ky;
    int l = 0, ljoff = 0;
    while (l + 1 < g.nlevels && row >= g.rowoff[l + 1]) { ljoff += g.sizes[l]; ++l; }
    const int s = g.sizes[l];
    int rr = row - g.rowoff[l];
    const int j = rr % s; rr /= s;
    const int i = rr % s; const int b = rr / s;
    float acc = 0.f;
    if (N % 64 == 0) {                                   // the wavefront shares (row, tap): weights once per wave, as in stage A
      const int lane = threadIdx.x & 63;
      for (int y0 = 0; y0 < g.H; y0 += 64) {
        const float wl = y0 + lane < g.H ? tap_weight(y0 + lane + ky - 1, s, g.H, i) : 0.f;
        const int ye = g.H - y0 < 64 ? g.H - y0 : 64;
        for (int yy = 0; yy < ye; ++yy) {
          const float wgt = __shfl(wl, yy, 64);
          if (wgt == 0.f) continue;
          acc += wgt * sa[((((size_t)b * g.H + y0 + yy) * 3 + kx) * nlj + ljoff + j) * N + n];
        }
      }
    } else {
      for (int y = 0; y < g.H; ++y) {
        const float wgt = tap_weight(y + ky - 1, s, g.H, i);
        if (wgt == 0.f) continue;
        acc += wgt * sa[((((size_t)b * g.H + y) * 3 + kx) * nlj + ljoff + j) * N + n];
      }
    }
    gq[e] = acc;
  }
}


// ---- Round 6: the same two stages as sliding windows.  u_tap(pos; cell) = W(pos + k - 1; cell) with W the bilinear weight of a destination index on a source cell: walking
// the SHIFTED index p = pos + k - 1 instead of pos, the cell pair (c(p), c(p) + 1) and its weights are the same for all three taps of the axis and the taps differ only in
// which datum they multiply (pos = p + 1, p, p - 1).  c(p) is non-decreasing and advances by at most one per step (levels no finer than the map), so a thread keeps two
// open accumulators per (level, tap) -- the cell being left and the cell being entered -- and writes a cell out when the walk leaves it: every input element is read ONCE
// (stage A re-read each map row 36 times from the L2, stage B each intermediate row ~6 times: 131 + 100 us per ResNet-50 step), no weight is evaluated per element.
struct FactWalk { int c; float a, b; };                       // position p touches cell c with weight a and cell c + 1 with weight b
__device__ __forceinline__ FactWalk fact_walk(int p, int s, int size) {
  int i0, i1; float l1;
  src_index_ac0(p, s, size, i0, i1, l1);
  FactWalk w; w.c = i0;
  if (i1 == i0) { w.a = (1.f - l1) + l1; w.b = 0.f; } else { w.a = 1.f - l1; w.b = l1; }      // (the sum as tap_weight forms it: bit-identical to the general kernels)
  return w;
}
template <typename T> __device__ __forceinline__ float2 ld_pair(const T* p);
template <> __device__ __forceinline__ float2 ld_pair<float>(const float* p) { return *(const float2*)p; }
template <> __device__ __forceinline__ float2 ld_pair<bf16_t>(const bf16_t* p) { const unsigned v = *(const unsigned*)p; return make_float2(__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u)); }

// stage A: sa[b][y][kx][lj][n] = sum_x W_l(x + kx - 1; j) * dcb[b][y][x][n].  One block per map row (b, y), a thread per channel PAIR.
template <typename T>
__global__ __launch_bounds__(256) void ppm_fact_scatterA2_kernel(PpmGeom g, int N, const T* __restrict__ dcb, float* __restrict__ sa, int nlj) {
  __shared__ FactWalk tab[4][64];
  const int tid = threadIdx.x;
  for (int e = tid; e < g.nlevels * g.W; e += 256) { const int l = e / g.W, p = e - l * g.W; tab[l][p] = fact_walk(p, g.sizes[l], g.W); }
  __syncthreads();
  const int by = blockIdx.x;                                  // b * H + y
  const T* row = dcb + (size_t)by * g.W * N;
  float* out = sa + (size_t)by * 3 * nlj * N;
  for (int pr = tid; pr < N / 2; pr += 256) {
    const int n = 2 * pr;
    float2 acc[4][3][2];
    int cur[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) { cur[l] = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) { acc[l][k][0] = make_float2(0.f, 0.f); acc[l][k][1] = make_float2(0.f, 0.f); } }
    float2 dm = make_float2(0.f, 0.f), d0 = ld_pair<T>(row + n), dp = g.W > 1 ? ld_pair<T>(row + (size_t)N + n) : make_float2(0.f, 0.f);      // pixels p - 1, p, p + 1
    int ljoff_[4]; { int o = 0;
#pragma unroll
      for (int l = 0; l < 4; ++l) { ljoff_[l] = o; o += g.sizes[l]; } }
    for (int p = 0; p < g.W; ++p) {
      const float2 dn = p + 2 < g.W ? ld_pair<T>(row + (size_t)(p + 2) * N + n) : make_float2(0.f, 0.f);      // next step's pixel p + 1
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        if (l < g.nlevels) {
          const FactWalk w = tab[l][p];
          if (w.c != cur[l]) {                                   // the walk left cell cur: it is complete (block-uniform branch)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              *(float2*)(out + ((size_t)k * nlj + ljoff_[l] + cur[l]) * N + n) = acc[l][k][0];
              acc[l][k][0] = acc[l][k][1]; acc[l][k][1] = make_float2(0.f, 0.f);
            }
            cur[l] = w.c;
          }
          // tap kx multiplies pixel p - kx + 1: kx = 0 -> dp, 1 -> d0, 2 -> dm
          acc[l][0][0].x += w.a * dp.x; acc[l][0][0].y += w.a * dp.y; acc[l][0][1].x += w.b * dp.x; acc[l][0][1].y += w.b * dp.y;
          acc[l][1][0].x += w.a * d0.x; acc[l][1][0].y += w.a * d0.y; acc[l][1][1].x += w.b * d0.x; acc[l][1][1].y += w.b * d0.y;
          acc[l][2][0].x += w.a * dm.x; acc[l][2][0].y += w.a * dm.y; acc[l][2][1].x += w.b * dm.x; acc[l][2][1].y += w.b * dm.y;
        }
      }
      dm = d0; d0 = dp; dp = dn;
    }
#pragma unroll
    for (int l = 0; l < 4; ++l)
      if (l < g.nlevels) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          *(float2*)(out + ((size_t)k * nlj + ljoff_[l] + cur[l]) * N + n) = acc[l][k][0];
          for (int c = cur[l] + 1; c < g.sizes[l]; ++c)          // the cell being entered (and, on a map narrower than the level's support, cells never reached: zero)
            *(float2*)(out + ((size_t)k * nlj + ljoff_[l] + c) * N + n) = c == cur[l] + 1 ? acc[l][k][1] : make_float2(0.f, 0.f);
        }
      }
  }
}

// stage B: gq[row_l(b,i,j)][(ky*3+kx)*N + n] = sum_y W_l(y + ky - 1; i) * sa[b][y][kx][lj][n].  A thread per (b, kx, lj, channel pair) column of sa, walking y.
__global__ __launch_bounds__(256) void ppm_fact_scatterB2_kernel(PpmGeom g, int N, const float* __restrict__ sa, float* __restrict__ gq, int nlj) {
  const int np = N / 2;
  const long long total = (long long)g.B * 3 * nlj * np;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int pr = (int)(e % np); long long r = e / np;
    int lj = (int)(r % nlj); r /= nlj;
    const int kx = (int)(r % 3); const int b = (int)(r / 3);
    const int ljall = lj;
    int l = 0;
    while (lj >= g.sizes[l]) { lj -= g.sizes[l]; ++l; }
    const int s = g.sizes[l], j = lj, n = 2 * pr;
    const float* col = sa + (((size_t)b * g.H * 3 + kx) * nlj + ljall) * N + n;      // row y at col + y * 3 * nlj * N
    const size_t ystep = (size_t)3 * nlj * N;
    float2 acc[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k) { acc[k][0] = make_float2(0.f, 0.f); acc[k][1] = make_float2(0.f, 0.f); }
    int cur = 0;
    auto flush = [&](int i, int slot) {
      float* o = gq + ((size_t)g.rowoff[l] + ((size_t)b * s + i) * s + j) * 9 * N + (size_t)kx * N + n;
#pragma unroll
      for (int k = 0; k < 3; ++k) *(float2*)(o + (size_t)k * 3 * N) = acc[k][slot];
    };
    float2 dm = make_float2(0.f, 0.f), d0 = *(const float2*)col, dp = g.H > 1 ? *(const float2*)(col + ystep) : make_float2(0.f, 0.f);
    float2 dq = g.H > 2 ? *(const float2*)(col + 2 * ystep) : make_float2(0.f, 0.f);       // one more row in flight
    for (int p = 0; p < g.H; ++p) {
      const float2 dn = p + 3 < g.H ? *(const float2*)(col + (size_t)(p + 3) * ystep) : make_float2(0.f, 0.f);
      const FactWalk w = fact_walk(p, s, g.H);
      if (w.c != cur) {
        flush(cur, 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) { acc[k][0] = acc[k][1]; acc[k][1] = make_float2(0.f, 0.f); }
        cur = w.c;
      }
      acc[0][0].x += w.a * dp.x; acc[0][0].y += w.a * dp.y; acc[0][1].x += w.b * dp.x; acc[0][1].y += w.b * dp.y;
      acc[1][0].x += w.a * d0.x; acc[1][0].y += w.a * d0.y; acc[1][1].x += w.b * d0.x; acc[1][1].y += w.b * d0.y;
      acc[2][0].x += w.a * dm.x; acc[2][0].y += w.a * dm.y; acc[2][1].x += w.b * dm.x; acc[2][1].y += w.b * dm.y;
      dm = d0; d0 = dp; dp = dq; dq = dn;
    }
    flush(cur, 0);
    for (int c = cur + 1; c < s; ++c) {
      if (c == cur + 1) flush(c, 1);
      else { float* o = gq + ((size_t)g.rowoff[l] + ((size_t)b * s + c) * s + j) * 9 * N + (size_t)kx * N + n;
#pragma unroll
        for (int k = 0; k < 3; ++k) *(float2*)(o + (size_t)k * 3 * N) = make_float2(0.f, 0.f); }
    }
  }
}

// (Round 6, gather stage 2: two forms that serve the operand rows from registers / the LDS instead of the L2 were built and measured against ppm_fact_gather2_kernel's
// 113 us per ResNet-50 step -- a sliding window per channel pair (64 dependent steps per thread on 8 waves per CU): 170 us; the general body reading a row staged in the
// LDS (32-byte lane accesses, bank conflicts): 340 us.  Neither kept; profiles/r6_ab_ppm_fact.txt.)

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
    const long long cell_vecs = (long long)g.B * g.ncy * g.ncx * (g.C / 8);
    if (cell_vecs < 256 * 1024)      // fewer than 1024 blocks of the one-thread-per-cell kernel (batch < 16 at 64 x 64 x 2048): rows of a cell in parallel
      hipLaunchKernelGGL(ppm_cells_rows_kernel<bf16_t>, dim3((unsigned)(g.B * g.ncy * g.ncx * ((g.C / 8 + 31) / 32))), dim3(256), 0, st, g, (const bf16_t*)x, cells);
    else
      hipLaunchKernelGGL(ppm_cells_kernel<bf16_t>, dim3(gs_blocks(cell_vecs)), dim3(256), 0, st, g, (const bf16_t*)x, cells);
    hipLaunchKernelGGL(ppm_bins_kernel, dim3(gs_blocks((long long)g.rowoff[4] * g.C)), dim3(256), 0, st, g, cells, (float*)pooled);
  } else if (d->dtype == SL_F32) {
    hipLaunchKernelGGL(ppm_cells_kernel<float>, dim3(gs_blocks((long long)g.B * g.ncy * g.ncx * g.C / 4)), dim3(256), 0, st, g, (const float*)x, cells);
    hipLaunchKernelGGL(ppm_bins_kernel, dim3(gs_blocks((long long)g.rowoff[4] * g.C)), dim3(256), 0, st, g, cells, (float*)pooled);
  } else SL_REQUIRE(false, "ppm_pool_fwd: bad dtype");
  SL_LAUNCH_CHECK("ppm_pool_fwd");
  return 0;
}

extern "C" int sl_ppm_pool_bwd(const SlPpmDesc* d, const float* dpooled, const void* dcat, int cat_pitch, int cat_off, void* dx,
                               void* workspace, size_t workspace_bytes, sl_stream_t stream) {
  PpmGeom g;
  if (int e = make_geom(d, g)) return e;
  SL_REQUIRE(dpooled && dx && workspace, "ppm_pool_bwd: null buffer");
  SL_REQUIRE(!dcat || (cat_pitch >= cat_off + g.C && cat_pitch % 8 == 0 && cat_off % 8 == 0), "ppm_pool_bwd: bad concat geometry");
  SL_REQUIRE(g.W <= 4096, "ppm_pool_bwd: W > 4096");
  const size_t need = (size_t)g.B * g.ncy * g.ncx * g.C * sizeof(float);
  if (workspace_bytes < need) { sl_set_error("ppm_pool_bwd: workspace %zu < %zu", workspace_bytes, need); return SL_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  float* gcell = (float*)workspace;
  hipLaunchKernelGGL(ppm_pool_bwd_cells_kernel, dim3(gs_blocks((long long)g.B * g.ncy * g.ncx * g.C)), dim3(256), 0, st, g, (const float*)dpooled, gcell);
  if (d->dtype == SL_BF16)
    hipLaunchKernelGGL(ppm_pool_bwd_kernel<bf16_t>, dim3(g.B * g.H), dim3(256), 0, st, g, (const float*)gcell, (const bf16_t*)dcat, cat_pitch, cat_off, (bf16_t*)dx);
  else if (d->dtype == SL_F32)
    hipLaunchKernelGGL(ppm_pool_bwd_kernel<float>, dim3(g.B * g.H), dim3(256), 0, st, g, (const float*)gcell, (const float*)dcat, cat_pitch, cat_off, (float*)dx);
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

// ---- factorised prior path (see the comment block above ppm_wq_prep_kernel)
extern "C" int sl_ppm_dwq_scatter(const float* dwq, int N, int Ctot, int Cs, int nlevels, float* dw_oihw, sl_stream_t stream) {
  SL_REQUIRE(dwq && dw_oihw && N > 0 && Cs > 0 && nlevels >= 1 && nlevels * Cs <= Ctot, "ppm_dwq_scatter: bad args");
  hipLaunchKernelGGL(ppm_dwq_scatter_kernel, dim3(gs_blocks((long long)nlevels * 9 * N * Cs)), dim3(256), 0, (hipStream_t)stream, dwq, N, Ctot, Cs, nlevels, dw_oihw);
  SL_LAUNCH_CHECK("ppm_dwq_scatter_kernel");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Grouped skinny GEMM over the pyramid rows:  y[r][n] = sum_k x[r][k] * w[l(r)][n][k]   (exact fp32, v_mfma_f32_32x32x2_f32)
// The four levels of the pyramid have 16..576 rows (B = 16) against K, N of 512..4608: one launch covers all levels, 64x64
// output tiles, and K is split across blockIdx.z (slabs summed in a fixed order by the finish kernel -> deterministic).
// The MFMA operand roles are swapped (A = weight rows, B = x rows) so that a lane ends up with 4 consecutive n of one row.
struct RowsGemm {
  const float* x; const float* w[SL_PPM_MAX_LEVELS]; float* out;        // w[l]: level l's [N][K] weights
  int K, N, nl, kslice;
  int row_off[5], tile_off[5];
  long long slab_stride;
};

constexpr int RG_KT = 32, RG_LD = 36;

__global__ __launch_bounds__(256) void ppm_rows_gemm_kernel(RowsGemm p) {
  __shared__ __attribute__((aligned(16))) float Xs[2][64][RG_LD];
  __shared__ __attribute__((aligned(16))) float Ws[2][64][RG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int l = 0;
  while (l + 1 < p.nl && (int)blockIdx.y >= p.tile_off[l + 1]) ++l;
  const int m0 = p.row_off[l] + ((int)blockIdx.y - p.tile_off[l]) * 64, mend = p.row_off[l + 1];
  const int n0 = blockIdx.x * 64;
  const int kbeg = blockIdx.z * p.kslice, kend = min(p.K, kbeg + p.kslice);
  const float* wl = p.w[l];
  const int lr = tid >> 3, lc = (tid & 7) * 4;             // this thread loads rows lr, lr+32; floats lc..lc+3 of the k tile
  float4 rx[2], rw[2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = m0 + lr + 32 * h;
      rx[h] = m < mend ? *reinterpret_cast<const float4*>(p.x + (size_t)m * p.K + k0 + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
      rw[h] = *reinterpret_cast<const float4*>(wl + (size_t)(n0 + lr + 32 * h) * p.K + k0 + lc);
    }
  };
  f32x16_t acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int wn = (wave & 1) * 32, wm = (wave >> 1) * 32, li = lane & 31, lh = (lane >> 5) * 4;
  gload(kbeg);
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += RG_KT, buf ^= 1) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<float4*>(&Xs[buf][lr + 32 * h][lc]) = rx[h];
      *reinterpret_cast<float4*>(&Ws[buf][lr + 32 * h][lc]) = rw[h];
    }
    __syncthreads();                                   // one barrier per k tile: the other buffer was last read a full iteration ago
    if (k0 + RG_KT < kend) gload(k0 + RG_KT);
#pragma unroll
    for (int kk = 0; kk < RG_KT; kk += 8) {
      const float4 a = *reinterpret_cast<const float4*>(&Ws[buf][wn + li][kk + lh]);
      const float4 b = *reinterpret_cast<const float4*>(&Xs[buf][wm + li][kk + lh]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
  }
  const int m = m0 + wm + li;
  if (m < mend) {
    float* o = p.out + (size_t)blockIdx.z * p.slab_stride + (size_t)m * p.N + n0 + wn + lh;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(o + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
  }
}

// y = sum of the K slabs (fixed order) and/or the per-(level, 128-row group) column sums  part[g][0][n] = sum y, part[g][1][n] = sum y^2.
__global__ __launch_bounds__(256) void ppm_rows_finish_kernel(const float* slabs, int ks, long long slab_stride, float* y, int N, RowsGemm g,
                                                              float* part) {
  // 16 lanes x float4 cover the block's 64 columns, 16 row phases cover its <= 128 rows: 8 rows x ks 16-byte loads per thread
  __shared__ float4 red[2][16][16];
  int l = 0;                                            // tile_off here = prefix of 128-row groups per level
  while (l + 1 < g.nl && (int)blockIdx.y >= g.tile_off[l + 1]) ++l;
  const int m0 = g.row_off[l] + ((int)blockIdx.y - g.tile_off[l]) * 128, mend = min(g.row_off[l + 1], m0 + 128);
  const int cq = threadIdx.x & 15, rq = threadIdx.x >> 4, n = blockIdx.x * 64 + cq * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s;
  for (int m = m0 + rq; m < mend; m += 16) {
    const float* src = slabs + (size_t)m * N + n;
    float4 v = *reinterpret_cast<const float4*>(src);
    int k = 1;
    for (; k + 4 <= ks; k += 4) {                       // four slabs in flight, summed in slab order
      const float4 a = *reinterpret_cast<const float4*>(src + (size_t)k * slab_stride);
      const float4 b = *reinterpret_cast<const float4*>(src + (size_t)(k + 1) * slab_stride);
      const float4 c = *reinterpret_cast<const float4*>(src + (size_t)(k + 2) * slab_stride);
      const float4 d = *reinterpret_cast<const float4*>(src + (size_t)(k + 3) * slab_stride);
      v.x = (((v.x + a.x) + b.x) + c.x) + d.x; v.y = (((v.y + a.y) + b.y) + c.y) + d.y;
      v.z = (((v.z + a.z) + b.z) + c.z) + d.z; v.w = (((v.w + a.w) + b.w) + c.w) + d.w;
    }
    for (; k < ks; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(src + (size_t)k * slab_stride);
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    if (ks > 1 || y != slabs) *reinterpret_cast<float4*>(y + (size_t)m * N + n) = v;
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
  }
  if (!part) return;
  red[0][rq][cq] = s; red[1][rq][cq] = s2;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int which = threadIdx.x >> 4;
    float4 t = red[which][0][cq];
#pragma unroll
    for (int r = 1; r < 16; ++r) { const float4 u = red[which][r][cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    *reinterpret_cast<float4*>(part + ((size_t)blockIdx.y * 2 + which) * N + n) = t;
  }
}

static int rows_gemm_plan(const SlPpmDesc* d, int K, int N, RowsGemm& mm, RowsGemm& fin, int& ks, int& mtiles, int& groups) {
  SL_REQUIRE(d && d->nlevels >= 1 && d->nlevels <= 4 && d->B > 0, "ppm_rows_gemm: bad descriptor");
  SL_REQUIRE(K > 0 && N > 0 && K % 32 == 0 && N % 64 == 0, "ppm_rows_gemm: K % 32 == 0 and N % 64 == 0 required");
  mm.K = fin.K = K; mm.N = fin.N = N; mm.nl = fin.nl = d->nlevels;
  mm.row_off[0] = fin.row_off[0] = mm.tile_off[0] = fin.tile_off[0] = 0;
  for (int l = 0; l < d->nlevels; ++l) {
    const int rows = d->B * d->sizes[l] * d->sizes[l];
    mm.row_off[l + 1] = fin.row_off[l + 1] = mm.row_off[l] + rows;
    mm.tile_off[l + 1] = mm.tile_off[l] + cdiv(rows, 64);
    fin.tile_off[l + 1] = fin.tile_off[l] + cdiv(rows, 128);
  }
  mtiles = mm.tile_off[d->nlevels]; groups = fin.tile_off[d->nlevels];
  const int tiles = mtiles * (N / 64);
  ks = std::max(1, std::min(cdiv(1024, tiles), K / 128));
  mm.kslice = cdiv(cdiv(K, ks), 32) * 32;
  ks = cdiv(K, mm.kslice);
  mm.slab_stride = fin.slab_stride = (long long)mm.row_off[d->nlevels] * N;
  return 0;
}

extern "C" int sl_ppm_rows_gemm_stat_rows(const SlPpmDesc* d) {
  if (!d || d->nlevels < 1 || d->nlevels > 4) return 0;
  int g = 0;
  for (int l = 0; l < d->nlevels; ++l) g += cdiv(d->B * d->sizes[l] * d->sizes[l], 128);
  return g;
}

extern "C" size_t sl_ppm_rows_gemm_workspace(const SlPpmDesc* d, int K, int N) {
  RowsGemm mm{}, fin{}; int ks, mt, gr;
  if (rows_gemm_plan(d, K, N, mm, fin, ks, mt, gr)) return 0;
  return ks > 1 ? (size_t)ks * mm.slab_stride * sizeof(float) : 0;
}

// w_levels: one [N][K] weight tensor per level (the parameters' own prepared copies: no stacked copy per step)
extern "C" int sl_ppm_rows_gemm_levels(const SlPpmDesc* d, int K, int N, const float* x, const float* const* w_levels, float* y, float* stat_partial,
                                       void* workspace, size_t workspace_bytes, sl_stream_t stream) {
  RowsGemm mm{}, fin{}; int ks, mtiles, groups;
  if (int e = rows_gemm_plan(d, K, N, mm, fin, ks, mtiles, groups)) return e;
  SL_REQUIRE(x && w_levels && y, "ppm_rows_gemm: null buffer");
  for (int l = 0; l < d->nlevels; ++l) { SL_REQUIRE(w_levels[l], "ppm_rows_gemm: null weight of a level"); mm.w[l] = w_levels[l]; }
  SL_REQUIRE(ks == 1 || (workspace && workspace_bytes >= (size_t)ks * mm.slab_stride * sizeof(float)), "ppm_rows_gemm: workspace too small");
  mm.x = x; mm.out = ks > 1 ? (float*)workspace : y;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ppm_rows_gemm_kernel, dim3(N / 64, mtiles, ks), dim3(256), 0, st, mm);
  if (ks > 1 || stat_partial)
    hipLaunchKernelGGL(ppm_rows_finish_kernel, dim3(N / 64, groups), dim3(256), 0, st, (const float*)mm.out, ks, mm.slab_stride, y, N, fin, stat_partial);
  return (int)hipGetLastError();
}

extern "C" int sl_ppm_rows_gemm(const SlPpmDesc* d, int K, int N, const float* x, const float* w, float* y, float* stat_partial,
                                void* workspace, size_t workspace_bytes, sl_stream_t stream) {
  SL_REQUIRE(d && d->nlevels >= 1 && d->nlevels <= SL_PPM_MAX_LEVELS && w, "ppm_rows_gemm: bad descriptor / null weights");
  const float* lv[SL_PPM_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};
  for (int l = 0; l < d->nlevels; ++l) lv[l] = w + (size_t)l * N * K;
  return sl_ppm_rows_gemm_levels(d, K, N, x, lv, y, stat_partial, workspace, workspace_bytes, stream);
}


// ---- weight gradients of the grouped row GEMMs (round 6): dw[l][n][k] = sum over the rows r of level l of a[r][n] * x[r][k].  The reduction has 16 ... 576 rows per level:
// the generic weight-gradient tile kernel ran each level as a launch of its own with split-K slabs and a reduce launch (eight launches, 0.24 ms per ResNet-50 step for
// 5.5 GFLOP).  Here one launch covers all levels: a block owns a 64 x 64 tile of one level's gradient, walks that level's rows in chunks of 32 (the MFMA k index is the row)
// and writes the tile itself -- no slabs, fixed summation order (ascending rows).  fp32 operands, v_mfma_f32_32x32x2_f32 (exact fp32 products).
struct RowsWgrad {
  const float* a; const float* x; float* dw[SL_PPM_MAX_LEVELS];
  int N, K, nl;
  int row_off[5];
};
constexpr int RW_LD = 68;            // LDS row pitch in floats (64 + 4: the float4 stores of the 16 threads of a row do not collide with the next row's)
__global__ __launch_bounds__(256) void ppm_rows_wgrad_kernel(RowsWgrad p) {
  __shared__ __attribute__((aligned(16))) float As[2][32][RW_LD];
  __shared__ __attribute__((aligned(16))) float Xs[2][32][RW_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l = blockIdx.z, n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int r0 = p.row_off[l], r1 = p.row_off[l + 1];
  const int lr = tid >> 4, lc = (tid & 15) * 4;                  // this thread loads rows lr and lr + 16 of a chunk, floats lc .. lc + 3
  float4 ra[2], rx[2];
  auto gload = [&](int r) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = r + lr + 16 * h;
      const bool ok = row < r1;
      ra[h] = ok ? *reinterpret_cast<const float4*>(p.a + (size_t)row * p.N + n0 + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
      rx[h] = ok ? *reinterpret_cast<const float4*>(p.x + (size_t)row * p.K + k0 + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  f32x16_t acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int wn = (wave & 1) * 32, wk = (wave >> 1) * 32, li = lane & 31, lh = lane >> 5;
  gload(r0);
  int buf = 0;
  for (int r = r0; r < r1; r += 32, buf ^= 1) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<float4*>(&As[buf][lr + 16 * h][lc]) = ra[h];
      *reinterpret_cast<float4*>(&Xs[buf][lr + 16 * h][lc]) = rx[h];
    }
    __syncthreads();                                   // one barrier per chunk: the other buffer was last read a full iteration ago
    if (r + 32 < r1) gload(r + 32);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)                    // k = row 2 kk + (lane >> 5) of the chunk
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][2 * kk + lh][wn + li], Xs[buf][2 * kk + lh][wk + li], acc, 0, 0, 0);
  }
  // D layout: register r = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the first operand (n), lane & 31 = column of the second (k)
  float* o = p.dw[l] + (size_t)(n0 + wn) * p.K + k0 + wk + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * p.K] = acc[r];
}

extern "C" int sl_ppm_rows_wgrad(const SlPpmDesc* d, int N, int K, const float* a, const float* x, float* const* dw, sl_stream_t stream) {
  SL_REQUIRE(d && d->nlevels >= 1 && d->nlevels <= SL_PPM_MAX_LEVELS && d->B > 0, "ppm_rows_wgrad: bad descriptor");
  SL_REQUIRE(a && x && dw && N > 0 && K > 0 && N % 64 == 0 && K % 64 == 0, "ppm_rows_wgrad: N % 64 == 0 and K % 64 == 0 required");
  RowsWgrad p{};
  p.a = a; p.x = x; p.N = N; p.K = K; p.nl = d->nlevels;
  for (int l = 0; l < d->nlevels; ++l) {
    SL_REQUIRE(dw[l] && d->sizes[l] > 0, "ppm_rows_wgrad: null output / bad level size");
    p.dw[l] = dw[l];
    p.row_off[l + 1] = p.row_off[l] + d->B * d->sizes[l] * d->sizes[l];
  }
  hipLaunchKernelGGL(ppm_rows_wgrad_kernel, dim3(N / 64, K / 64, d->nlevels), dim3(256), 0, (hipStream_t)stream, p);
  SL_LAUNCH_CHECK("ppm_rows_wgrad_kernel");
  return 0;
}

extern "C" size_t sl_ppm_fact_workspace(const SlPpmDesc* d, int N) {
  PpmGeom g;
  if (make_geom(d, g)) return 0;
  return (size_t)g.B * g.H * 3 * sum_sizes(g) * N * sizeof(float);
}

extern "C" int sl_ppm_fact_gather(const SlPpmDesc* d, int N, const float* q, void* gout, void* workspace, size_t workspace_bytes, sl_stream_t stream) {
  PpmGeom g;
  if (int e = make_geom(d, g)) return e;
  SL_REQUIRE(q && gout && workspace && N > 0 && N % 8 == 0, "ppm_fact_gather: bad args");
  const int nlj = sum_sizes(g);
  const size_t need = (size_t)g.B * g.H * 3 * nlj * N * sizeof(float);
  if (workspace_bytes < need) { sl_set_error("ppm_fact_gather: workspace %zu < %zu", workspace_bytes, need); return SL_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  float* t = (float*)workspace;
  hipLaunchKernelGGL(ppm_fact_gather1_kernel, dim3(gs_blocks((long long)g.B * g.H * 3 * nlj * N / 4)), dim3(256), 0, st, g, N, q, t, nlj);
  if (d->dtype == SL_BF16) hipLaunchKernelGGL(ppm_fact_gather2_kernel<bf16_t>, dim3(gs_blocks((long long)g.B * g.H * g.W * N / 8)), dim3(256), 0, st, g, N, t, (bf16_t*)gout, nlj);
  else if (d->dtype == SL_F32) hipLaunchKernelGGL(ppm_fact_gather2_kernel<float>, dim3(gs_blocks((long long)g.B * g.H * g.W * N / 4)), dim3(256), 0, st, g, N, t, (float*)gout, nlj);
  else SL_REQUIRE(false, "ppm_fact_gather: bad dtype");
  SL_LAUNCH_CHECK("ppm_fact_gather");
  return 0;
}

extern "C" int sl_ppm_fact_scatter(const SlPpmDesc* d, int N, const void* dcb, float* gq, void* workspace, size_t workspace_bytes, sl_stream_t stream) {
  PpmGeom g;
  if (int e = make_geom(d, g)) return e;
  SL_REQUIRE(dcb && gq && workspace && N > 0 && N % 8 == 0, "ppm_fact_scatter: bad args");
  const int nlj = sum_sizes(g);
  const size_t need = (size_t)g.B * g.H * 3 * nlj * N * sizeof(float);
  if (workspace_bytes < need) { sl_set_error("ppm_fact_scatter: workspace %zu < %zu", workspace_bytes, need); return SL_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  float* sa = (float*)workspace;
  // the sliding-window kernels (round 6) need every level no finer than the map (the cell index advances by at most one per step) and rows of at most 64 pixels for the
  // weight table; anything else takes the two general kernels.  Test hook: sl_debug_ppm_fact_walk(0).
  bool walk = g_sl_debug.ppm_fact_walk && g.W <= 64 && g.H >= 1 && N % 2 == 0;
  for (int l = 0; l < g.nlevels; ++l) walk = walk && g.sizes[l] <= g.W && g.sizes[l] <= g.H;
  if (walk) {
    if (d->dtype == SL_BF16) hipLaunchKernelGGL(ppm_fact_scatterA2_kernel<bf16_t>, dim3(g.B * g.H), dim3(256), 0, st, g, N, (const bf16_t*)dcb, sa, nlj);
    else if (d->dtype == SL_F32) hipLaunchKernelGGL(ppm_fact_scatterA2_kernel<float>, dim3(g.B * g.H), dim3(256), 0, st, g, N, (const float*)dcb, sa, nlj);
    else SL_REQUIRE(false, "ppm_fact_scatter: bad dtype");
    hipLaunchKernelGGL(ppm_fact_scatterB2_kernel, dim3(gs_blocks((long long)g.B * 3 * nlj * (N / 2))), dim3(256), 0, st, g, N, sa, gq, nlj);
    SL_LAUNCH_CHECK("ppm_fact_scatter (walk)");
    return 0;
  }
  if (d->dtype == SL_BF16) hipLaunchKernelGGL(ppm_fact_scatterA_kernel<bf16_t>, dim3(gs_blocks((long long)g.B * g.H * 3 * nlj * N / 8)), dim3(256), 0, st, g, N, (const bf16_t*)dcb, sa, nlj);
  else if (d->dtype == SL_F32) hipLaunchKernelGGL(ppm_fact_scatterA_kernel<float>, dim3(gs_blocks((long long)g.B * g.H * 3 * nlj * N / 4)), dim3(256), 0, st, g, N, (const float*)dcb, sa, nlj);
  else SL_REQUIRE(false, "ppm_fact_scatter: bad dtype");
  hipLaunchKernelGGL(ppm_fact_scatterB_kernel, dim3(gs_blocks((long long)g.rowoff[4] * 9 * N)), dim3(256), 0, st, g, N, sa, gq, nlj);
  SL_LAUNCH_CHECK("ppm_fact_scatter");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm + ReLU backward of ALL pyramid stages (pspnet_pop.py:12-16 backward) in one launch.  The stage tensors are [rows = B * sum s^2][C] float (16 .. 576 rows per
// level at the bench shape): the general path ran reduce / finalize / apply per level -- twelve latency-bound launches, 137 us per step (profiles/r4_kernel_sequence).
// Here a block owns (level, 32 channels): eight row lanes sum (g, g * xhat) over the level's rows (g = dy gated by y > 0), the block combines them in fp64 in a fixed
// order, forms the coefficients exactly as bn_bwd_finalize_kernel does and applies dx = cA g + cB (x - mean) + cC in a second sweep over the same (cache-resident) rows.
struct PpmStageBn {
  const float* mean[SL_PPM_MAX_LEVELS]; const float* invstd[SL_PPM_MAX_LEVELS]; const float* gamma[SL_PPM_MAX_LEVELS];
  float* dgamma[SL_PPM_MAX_LEVELS]; float* dbeta[SL_PPM_MAX_LEVELS];
  int row0[SL_PPM_MAX_LEVELS + 1]; int train[SL_PPM_MAX_LEVELS];
};
// (round 6: 32 row lanes per channel instead of 8, four rows in flight per thread -- the 576-row level was a chain of 72 dependent row reads per thread, twice: 83 -> ~20 us)
constexpr int SBN_RL = 32;
__global__ __launch_bounds__(32 * SBN_RL) void ppm_stage_bn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x, float* __restrict__ dx,
                                                                      int C, PpmStageBn q) {
  __shared__ double red[2][SBN_RL][32];
  __shared__ float co[3][32];
  const int cg = C / 32, lvl = blockIdx.x / cg, c = (blockIdx.x % cg) * 32 + (threadIdx.x & 31), rl = threadIdx.x >> 5;
  const int r0 = q.row0[lvl], r1 = q.row0[lvl + 1];
  const float mu = q.mean[lvl][c], is = q.invstd[lvl][c];
  float s1 = 0.f, s2 = 0.f;
  for (int r = r0 + rl; r < r1; r += 4 * SBN_RL) {
    float gy[4], gd[4], gx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r + u * SBN_RL;
      const size_t o = (size_t)(rr < r1 ? rr : r) * C + c;
      gy[u] = y[o]; gd[u] = dy[o]; gx[u] = x[o];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (r + u * SBN_RL < r1) {                       // rows in ascending order per lane, as before
        const float g = gy[u] > 0.f ? gd[u] : 0.f;
        s1 += g; s2 += g * ((gx[u] - mu) * is);
      }
  }
  red[0][rl][threadIdx.x & 31] = (double)s1; red[1][rl][threadIdx.x & 31] = (double)s2;
  __syncthreads();
  if (rl == 0) {
    double a = 0.0, b = 0.0;
    for (int j = 0; j < SBN_RL; ++j) { a += red[0][j][threadIdx.x]; b += red[1][j][threadIdx.x]; }
    if (q.dgamma[lvl]) q.dgamma[lvl][c] = (float)b;
    if (q.dbeta[lvl]) q.dbeta[lvl][c] = (float)a;
    const double gm = q.gamma[lvl] ? (double)q.gamma[lvl][c] : 1.0, isd = (double)is, count = (double)(r1 - r0);
    co[0][threadIdx.x] = (float)(gm * isd);
    co[1][threadIdx.x] = q.train[lvl] ? (float)(-gm * isd * isd * b / count) : 0.f;
    co[2][threadIdx.x] = q.train[lvl] ? (float)(-gm * isd * a / count) : 0.f;
  }
  __syncthreads();
  const float cA = co[0][threadIdx.x & 31], cB = co[1][threadIdx.x & 31], cC = co[2][threadIdx.x & 31];
  for (int r = r0 + rl; r < r1; r += SBN_RL) {
    const size_t o = (size_t)r * C + c;
    const float g = y[o] > 0.f ? dy[o] : 0.f;
    dx[o] = cA * g + cB * (x[o] - mu) + cC;
  }
}

extern "C" int sl_ppm_stage_bn_bwd(const SlPpmDesc* d, int C, const float* dy, const float* y, const float* x, const float* const* mean, const float* const* invstd,
                                   const float* const* gamma, const int* train, float* const* dgamma, float* const* dbeta, float* dx, sl_stream_t stream) {
  SL_REQUIRE(d && dy && y && x && dx && mean && invstd && gamma && train && dgamma && dbeta, "ppm_stage_bn_bwd: null argument");
  SL_REQUIRE(d->nlevels >= 1 && d->nlevels <= SL_PPM_MAX_LEVELS && C > 0 && C % 32 == 0, "ppm_stage_bn_bwd: bad level count / channel count");
  PpmStageBn q{};
  int row = 0;
  for (int l = 0; l < d->nlevels; ++l) {
    SL_REQUIRE(mean[l] && invstd[l], "ppm_stage_bn_bwd: null statistics");
    q.mean[l] = mean[l]; q.invstd[l] = invstd[l]; q.gamma[l] = gamma[l]; q.dgamma[l] = dgamma[l]; q.dbeta[l] = dbeta[l]; q.train[l] = train[l];
    q.row0[l] = row; row += d->B * d->sizes[l] * d->sizes[l];
  }
  q.row0[d->nlevels] = row;
  hipLaunchKernelGGL(ppm_stage_bn_bwd_kernel, dim3(d->nlevels * (C / 32)), dim3(32 * SBN_RL), 0, (hipStream_t)stream, dy, y, x, dx, C, q);
  SL_LAUNCH_CHECK("ppm_stage_bn_bwd_kernel");
  return 0;
}
