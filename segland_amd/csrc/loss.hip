// Fused bilinear upsample (align_corners=True) + cross-entropy (loss/criterion.py:51-52), the ft pseudo-label step
// (networks/pspnet_pop.py:221-231), upsample+argmax (eval_base.py:168-169), IoU histogram (utils/pyt_utils.py:293-305)
// and masked average pooling (networks/pspnet.py:7-15).  The H x W logits are never materialised.
#include "common.h"

namespace {

constexpr int KMAXC = 16;   // max logit channels

struct UpGeom { int B, K, h, w, H, W; float sy, sx; };

__host__ inline float ac1_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

// ATen upsample_bilinear2d source index, align_corners = True
__device__ __forceinline__ void src_index_ac1(int dst, int in, float scale, int& i0, int& i1, float& l1) {
  const float src = scale * (float)dst;
  i0 = (int)src;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
}

// interpolated logits of pixel (Y,X) of image b into v[0..K)
__device__ __forceinline__ void pixel_logits(const UpGeom& g, const float* __restrict__ lg, int b, int Y, int X, float* v) {
  int y0, y1, x0, x1; float ly, lx;
  src_index_ac1(Y, g.h, g.sy, y0, y1, ly);
  src_index_ac1(X, g.w, g.sx, x0, x1, lx);
  const float wy0 = 1.f - ly, wx0 = 1.f - lx;
  const float* p = lg + (size_t)b * g.K * g.h * g.w;
#pragma unroll
  for (int k = 0; k < KMAXC; ++k) {
    if (k < g.K) {
      const float* q = p + (size_t)k * g.h * g.w;
      v[k] = wy0 * (wx0 * q[y0 * g.w + x0] + lx * q[y0 * g.w + x1]) + ly * (wx0 * q[y1 * g.w + x0] + lx * q[y1 * g.w + x1]);
    } else v[k] = -INFINITY;
  }
}

__global__ __launch_bounds__(256) void upsample_ce_fwd_kernel(UpGeom g, const float* __restrict__ lg, const int64_t* __restrict__ tgt,
                                                              int ignore, float* __restrict__ part) {
  __shared__ float red[2][4];
  const long long total = (long long)g.B * g.H * g.W;
  float loss = 0.f, cnt = 0.f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
    const long long t = tgt[i];
    if (t == ignore || t < 0 || t >= g.K) continue;
    const int X = (int)(i % g.W); const long long r = i / g.W;
    const int Y = (int)(r % g.H), b = (int)(r / g.H);
    float v[KMAXC];
    pixel_logits(g, lg, b, Y, X, v);
    float m = v[0], vt = 0.f;
#pragma unroll
    for (int k = 0; k < KMAXC; ++k) { m = fmaxf(m, v[k]); if (k == (int)t) vt = v[k]; }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KMAXC; ++k) if (k < g.K) s += expf(v[k] - m);
    loss += logf(s) + m - vt;
    cnt += 1.f;
  }
  loss = wave_sum(loss); cnt = wave_sum(cnt);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = loss; red[1][wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    part[blockIdx.x * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

__global__ void upsample_ce_finalize_kernel(const float* __restrict__ part, int nblk, float* __restrict__ out) {
  __shared__ double rs[256], rc[256];
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) { s += (double)part[2 * i]; c += (double)part[2 * i + 1]; }
  rs[threadIdx.x] = s; rc[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { rs[threadIdx.x] += rs[threadIdx.x + o]; rc[threadIdx.x] += rc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = (float)(rs[0] / rc[0]); out[1] = (float)rc[0]; }
}

// four lanes per low-resolution cell (each takes every 4th footprint row), gather over the pixels whose bilinear
// footprint touches the cell, then a 4-lane shuffle reduction: deterministic, no atomics.
__global__ __launch_bounds__(256) void upsample_ce_bwd_kernel(UpGeom g, const float* __restrict__ lg, const int64_t* __restrict__ tgt,
                                                              const float* __restrict__ loss_cnt, const float* __restrict__ gscale,
                                                              int ignore, float* __restrict__ dlg) {
  const long long cells = (long long)g.B * g.h * g.w;
  const long long ci = (blockIdx.x * 256LL + threadIdx.x) >> 2;
  const int sub = threadIdx.x & 3;
  const bool live = ci < cells;
  const long long cc = live ? ci : 0;
  const int j = (int)(cc % g.w); const long long r = cc / g.w;
  const int i = (int)(r % g.h), b = (int)(r / g.h);
  float acc[KMAXC];
#pragma unroll
  for (int k = 0; k < KMAXC; ++k) acc[k] = 0.f;
  const int Ylo = g.sy > 0.f ? max(0, (int)floorf((float)(i - 1) / g.sy)) : 0;
  const int Yhi = g.sy > 0.f ? min(g.H - 1, (int)ceilf((float)(i + 1) / g.sy)) : g.H - 1;
  const int Xlo = g.sx > 0.f ? max(0, (int)floorf((float)(j - 1) / g.sx)) : 0;
  const int Xhi = g.sx > 0.f ? min(g.W - 1, (int)ceilf((float)(j + 1) / g.sx)) : g.W - 1;
  if (live) {
    for (int Y = Ylo + sub; Y <= Yhi; Y += 4) {
      int y0, y1; float ly;
      src_index_ac1(Y, g.h, g.sy, y0, y1, ly);
      const float wy = (y0 == i ? 1.f - ly : 0.f) + (y1 == i ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int X = Xlo; X <= Xhi; ++X) {
        int x0, x1; float lx;
        src_index_ac1(X, g.w, g.sx, x0, x1, lx);
        const float wx = (x0 == j ? 1.f - lx : 0.f) + (x1 == j ? lx : 0.f);
        if (wx == 0.f) continue;
        const long long t = tgt[((size_t)b * g.H + Y) * g.W + X];
        if (t == ignore || t < 0 || t >= g.K) continue;
        float v[KMAXC];
        pixel_logits(g, lg, b, Y, X, v);
        float m = v[0];
#pragma unroll
        for (int k = 0; k < KMAXC; ++k) m = fmaxf(m, v[k]);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < KMAXC; ++k) { v[k] = k < g.K ? expf(v[k] - m) : 0.f; s += v[k]; }
        const float wgt = wy * wx, inv = 1.f / s;
#pragma unroll
        for (int k = 0; k < KMAXC; ++k) acc[k] += wgt * (v[k] * inv - (k == (int)t ? 1.f : 0.f));
      }
    }
  }
  const float f = gscale[0] / loss_cnt[1];
#pragma unroll
  for (int k = 0; k < KMAXC; ++k) {
    float a = acc[k];
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    if (live && sub == 0 && k < g.K) dlg[(((size_t)b * g.K + k) * g.h + i) * g.w + j] = a * f;
  }
}

// Tiled form of the same gradient: a block owns CT x CT low-resolution cells and evaluates the softmax of every pixel of their joint
// footprint ONCE (the per-cell gather above evaluates each pixel for each of the ~4 cells it touches, from global memory), then applies the
// separable bilinear weights in two LDS passes:  H1[Y][j][k] = sum_X wx(X, j) G[Y][X][k],  dlg[i][j][k] = sum_Y wy(Y, i) H1[Y][j][k].
// Sums run in a fixed order (deterministic); they are associated differently from the gather kernel (last-bit differences).
constexpr int CT = 4;
struct UpTile { int NYmax, NXmax; };
template <int NT>
__global__ __launch_bounds__(NT) void upsample_ce_bwd_tiled_kernel(UpGeom g, UpTile ut, const float* __restrict__ lg, const int64_t* __restrict__ tgt,
                                                                    const float* __restrict__ loss_cnt, const float* __restrict__ gscale,
                                                                    int ignore, float* __restrict__ dlg) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = g.K;
  float* lgt = sm;                                   // [(CT+2)^2][K] logits of the cells around the tile (clamped at the map border)
  float* wy0 = lgt + (CT + 2) * (CT + 2) * K;        // per footprint row: weight on cell y0, weight on y1 ; y0 as int
  float* wx0 = wy0 + 3 * ut.NYmax;
  float* wxj = wx0 + 3 * ut.NXmax;                   // [NX][CT] weight of footprint column X on cell j0 + jr
  float* wyi = wxj + CT * ut.NXmax;                  // [NY][CT]
  float* H1 = wyi + CT * ut.NYmax;                   // [NY][CT][K]
  float* G = H1 + ut.NYmax * CT * K;                 // [NY][NX][K]
  const int tid = threadIdx.x;
  const int tw = cdiv(g.w, CT), th = cdiv(g.h, CT);
  int blk = blockIdx.x;
  const int tj = blk % tw; blk /= tw;
  const int ti = blk % th; const int b = blk / th;
  const int i0 = ti * CT, j0 = tj * CT;
  const int i1 = min(i0 + CT, g.h) - 1, j1 = min(j0 + CT, g.w) - 1;          // last cell of the tile
  const int Ylo = g.sy > 0.f ? max(0, (int)floorf((float)(i0 - 1) / g.sy)) : 0;
  const int Yhi = g.sy > 0.f ? min(g.H - 1, (int)ceilf((float)(i1 + 1) / g.sy)) : g.H - 1;
  const int Xlo = g.sx > 0.f ? max(0, (int)floorf((float)(j0 - 1) / g.sx)) : 0;
  const int Xhi = g.sx > 0.f ? min(g.W - 1, (int)ceilf((float)(j1 + 1) / g.sx)) : g.W - 1;
  const int NY = Yhi - Ylo + 1, NX = Xhi - Xlo + 1;
  const float* lb = lg + (size_t)b * K * g.h * g.w;
  for (int e = tid; e < (CT + 2) * (CT + 2) * K; e += NT) {
    const int k = e % K, c = e / K, cy = c / (CT + 2), cx = c - cy * (CT + 2);
    const int yy = min(max(i0 - 1 + cy, 0), g.h - 1), xx = min(max(j0 - 1 + cx, 0), g.w - 1);
    lgt[e] = lb[((size_t)k * g.h + yy) * g.w + xx];
  }
  for (int e = tid; e < NY; e += NT) {
    int y0, y1; float ly; src_index_ac1(Ylo + e, g.h, g.sy, y0, y1, ly);
    wy0[3 * e] = __int_as_float(y0); wy0[3 * e + 1] = __int_as_float(y1); wy0[3 * e + 2] = ly;
#pragma unroll
    for (int r = 0; r < CT; ++r) wyi[e * CT + r] = (y0 == i0 + r ? 1.f - ly : 0.f) + (y1 == i0 + r ? ly : 0.f);
  }
  for (int e = tid; e < NX; e += NT) {
    int x0, x1; float lx; src_index_ac1(Xlo + e, g.w, g.sx, x0, x1, lx);
    wx0[3 * e] = __int_as_float(x0); wx0[3 * e + 1] = __int_as_float(x1); wx0[3 * e + 2] = lx;
#pragma unroll
    for (int r = 0; r < CT; ++r) wxj[e * CT + r] = (x0 == j0 + r ? 1.f - lx : 0.f) + (x1 == j0 + r ? lx : 0.f);
  }
  __syncthreads();
  // ---- pass 1: softmax - onehot of every footprint pixel, once
  // the thread's labels first, all loads in flight (a load at the top of every iteration was a chain of ~8 memory round trips per block)
  constexpr int TPF = 2048 / NT;
  long long tpre[TPF];
#pragma unroll
  for (int u = 0; u < TPF; ++u) {
    const int pix = tid + NT * u;
    const int yr = pix / NX, xr = pix - yr * NX;
    tpre[u] = pix < NY * NX ? tgt[((size_t)b * g.H + Ylo + yr) * g.W + Xlo + xr] : (long long)ignore;
  }
#pragma unroll 1
  for (int u = 0; tid + NT * u < NY * NX; ++u) {
    const int pix = tid + NT * u;
    const int yr = pix / NX, xr = pix - yr * NX;
    float* out = G + (size_t)pix * K;
    long long t;
    if (u < TPF) {
      t = tpre[0];
#pragma unroll
      for (int q = 1; q < TPF; ++q) t = u == q ? tpre[q] : t;
    } else t = tgt[((size_t)b * g.H + Ylo + yr) * g.W + Xlo + xr];
    if (t == ignore || t < 0 || t >= K) {
      for (int k = 0; k < K; ++k) out[k] = 0.f;
      continue;
    }
    const int y0 = __float_as_int(wy0[3 * yr]), y1 = __float_as_int(wy0[3 * yr + 1]); const float ly = wy0[3 * yr + 2];
    const int x0 = __float_as_int(wx0[3 * xr]), x1 = __float_as_int(wx0[3 * xr + 1]); const float lx = wx0[3 * xr + 2];
    const float wya = 1.f - ly, wxa = 1.f - lx;
    // a footprint pixel at the rim may interpolate from a cell outside the (CT+2)^2 neighbourhood; it then has zero weight on every cell of
    // this tile, so any finite value will do: clamp the index
    const int ry0 = min(max(y0 - i0 + 1, 0), CT + 1), ry1 = min(max(y1 - i0 + 1, 0), CT + 1);
    const int rx0 = min(max(x0 - j0 + 1, 0), CT + 1), rx1 = min(max(x1 - j0 + 1, 0), CT + 1);
    const float* q00 = lgt + (ry0 * (CT + 2) + rx0) * K;
    const float* q01 = lgt + (ry0 * (CT + 2) + rx1) * K;
    const float* q10 = lgt + (ry1 * (CT + 2) + rx0) * K;
    const float* q11 = lgt + (ry1 * (CT + 2) + rx1) * K;
    float v[KMAXC];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < KMAXC; ++k) {
      if (k < K) { v[k] = wya * (wxa * q00[k] + lx * q01[k]) + ly * (wxa * q10[k] + lx * q11[k]); m = fmaxf(m, v[k]); }
      else v[k] = -INFINITY;
    }
    float ssum = 0.f;
#pragma unroll
    for (int k = 0; k < KMAXC; ++k) { v[k] = k < K ? __expf(v[k] - m) : 0.f; ssum += v[k]; }      // hardware exp2-based: 2 ulp, the gradient's tolerance is 1e-3
    const float inv = 1.f / ssum;
#pragma unroll
    for (int k = 0; k < KMAXC; ++k) if (k < K) out[k] = v[k] * inv - (k == (int)t ? 1.f : 0.f);
  }
  __syncthreads();
  // ---- pass 2a: along X
  for (int e = tid; e < NY * CT * K; e += NT) {
    const int k = e % K, r = e / K, jr = r % CT, yr = r / CT;
    const int j = j0 + jr;
    float acc = 0.f;
    if (j <= j1) {
      const int xa = (g.sx > 0.f ? max(Xlo, (int)floorf((float)(j - 1) / g.sx)) : Xlo) - Xlo, xb = (g.sx > 0.f ? min(Xhi, (int)ceilf((float)(j + 1) / g.sx)) : Xhi) - Xlo;
      const float* grow = G + (size_t)yr * NX * K + k;
      for (int xr = xa; xr <= xb; ++xr) acc += wxj[xr * CT + jr] * grow[xr * K];
    }
    H1[e] = acc;
  }
  __syncthreads();
  // ---- pass 2b: along Y, scale, store
  const float f = gscale[0] / loss_cnt[1];
  for (int e = tid; e < CT * CT * K; e += NT) {
    const int k = e % K, c = e / K, ir = c / CT, jr = c - ir * CT;
    const int i = i0 + ir, j = j0 + jr;
    if (i > i1 || j > j1) continue;
    const int ya = g.sy > 0.f ? max(Ylo, (int)floorf((float)(i - 1) / g.sy)) : Ylo, yb = g.sy > 0.f ? min(Yhi, (int)ceilf((float)(i + 1) / g.sy)) : Yhi;
    float acc = 0.f;
    for (int yr = ya - Ylo; yr <= yb - Ylo; ++yr) acc += wyi[yr * CT + ir] * H1[(yr * CT + jr) * K + k];
    dlg[(((size_t)b * K + k) * g.h + i) * g.w + j] = acc * f;
  }
}

template <bool PSEUDO>
__global__ void upsample_argmax_kernel(UpGeom g, const float* __restrict__ lg, uint8_t* __restrict__ labels, int64_t* __restrict__ mask,
                                       int n_base) {
  const long long total = (long long)g.B * g.H * g.W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    if (PSEUDO && mask[i] != 0) continue;
    const int X = (int)(i % g.W); const long long r = i / g.W;
    const int Y = (int)(r % g.H), b = (int)(r / g.H);
    float v[KMAXC];
    pixel_logits(g, lg, b, Y, X, v);
    int best = 0; float bv = v[0];
#pragma unroll
    for (int k = 1; k < KMAXC; ++k) if (k < g.K && v[k] > bv) { bv = v[k]; best = k; }   // first maximum wins (torch.argmax)
    if (PSEUDO) mask[i] = best > 0 ? best + n_base : 0;
    else labels[i] = (uint8_t)best;
  }
}

// probability-map dump of eval_base.py:168,189-190: the full-resolution logits F.interpolate(align_corners=True) produced, NCHW float
__global__ void upsample_logits_kernel(UpGeom g, const float* __restrict__ lg, float* __restrict__ out) {
  const long long total = (long long)g.B * g.H * g.W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int X = (int)(i % g.W); const long long r = i / g.W;
    const int Y = (int)(r % g.H), b = (int)(r / g.H);
    float v[KMAXC];
    pixel_logits(g, lg, b, Y, X, v);
    const size_t plane = (size_t)g.H * g.W;
#pragma unroll
    for (int k = 0; k < KMAXC; ++k) if (k < g.K) out[((size_t)b * g.K + k) * plane + (size_t)Y * g.W + X] = v[k];
  }
}

// fusemat.py:35-52: mats[idx] += prob in list order, argmax(mat / n, axis=0): float sum in the same order, first maximum wins (np.argmax)
__global__ void fuse_argmax_kernel(const float* const* __restrict__ mats, int n, int K, long long hw, uint8_t* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < hw; i += (long long)gridDim.x * blockDim.x) {
    int best = 0; float bv = 0.f;
    for (int k = 0; k < K; ++k) {
      float s = mats[0][(size_t)k * hw + i];
      for (int m = 1; m < n; ++m) s += mats[m][(size_t)k * hw + i];
      s = s / (float)n;
      if (k == 0 || s > bv) { bv = s; best = k; }
    }
    out[i] = (uint8_t)best;
  }
}

__global__ __launch_bounds__(256) void iou_hist_kernel(const uint8_t* __restrict__ pred, const int64_t* __restrict__ tgt, long long n, int K,
                                                       int ignore, unsigned long long* __restrict__ hist) {
  __shared__ unsigned int h[3 * 256];
  for (int i = threadIdx.x; i < 3 * K; i += 256) h[i] = 0;
  __syncthreads();
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
    const long long t = tgt[i];
    if (t == ignore) continue;                       // output[target == ignore] = ignore: contributes to no bin
    const int p = pred[i];
    if (p < K) atomicAdd(&h[K + p], 1u);
    if (t >= 0 && t < K) { atomicAdd(&h[2 * K + (int)t], 1u); if (p == (int)t) atomicAdd(&h[p], 1u); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * K; i += 256) if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// cm[t][p] += 1 for every pixel whose label t != ignore (utils/pyt_utils.py:182-200 get_confusion_matrix over the kept pixels);
// labels outside [0, K) are skipped (the reference would fold them into a neighbouring bin through t*K + p -- they cannot occur:
// the eval drivers filter ignore first and K covers every class).
__global__ __launch_bounds__(256) void confusion_kernel(const uint8_t* __restrict__ pred, const int64_t* __restrict__ tgt, long long n, int K,
                                                        int ignore, unsigned long long* __restrict__ cm) {
  extern __shared__ unsigned int hc[];
  for (int i = threadIdx.x; i < K * K; i += 256) hc[i] = 0;
  __syncthreads();
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
    const long long t = tgt[i];
    if (t == ignore || t < 0 || t >= K) continue;
    const int p = pred[i];
    if (p < K) atomicAdd(&hc[(int)t * K + p], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * K; i += 256) if (hc[i]) atomicAdd(&cm[i], (unsigned long long)hc[i]);
}

template <typename T>
__global__ void masked_avg_pool_kernel(const T* __restrict__ feat, const float* __restrict__ mask, int B, int h, int w, int C, int H, int W,
                                       float sy, float sx, float* __restrict__ proto) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float total = 0.f;
  for (int b = 0; b < B; ++b) {
    float sf = 0.f, sm = 0.f;
    for (int y = 0; y < h; ++y) {
      int y0, y1; float ly;
      src_index_ac1(y, H, sy, y0, y1, ly);
      for (int x = 0; x < w; ++x) {
        int x0, x1; float lx;
        src_index_ac1(x, W, sx, x0, x1, lx);
        const float* mb = mask + (size_t)b * H * W;
        const float m = (1.f - ly) * ((1.f - lx) * mb[y0 * W + x0] + lx * mb[y0 * W + x1]) + ly * ((1.f - lx) * mb[y1 * W + x0] + lx * mb[y1 * W + x1]);
        sf += to_f<T>(feat[((size_t)(b * h + y) * w + x) * C + c]) * m;
        sm += m;
      }
    }
    total += sf / (sm + 1e-5f);
  }
  proto[c] = total / (float)B;
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int B, int HW, int C) {
  const long long total = (long long)B * HW * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW); const long long r = i / HW;     // i enumerates the NCHW destination
    const int c = (int)(r % C), b = (int)(r / C);
    dst[i] = to_f<T>(src[((size_t)b * HW + p) * C + c]);
  }
}
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int HW, int C) {
  const long long total = (long long)B * HW * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); const long long r = i / C;       // i enumerates the NHWC destination
    const int p = (int)(r % HW), b = (int)(r / HW);
    dst[i] = from_f<T>(src[((size_t)b * C + c) * HW + p]);
  }
}

inline UpGeom make_up(int B, int K, int h, int w, int H, int W) {
  UpGeom g; g.B = B; g.K = K; g.h = h; g.w = w; g.H = H; g.W = W; g.sy = ac1_scale(h, H); g.sx = ac1_scale(w, W);
  return g;
}
inline int ce_blocks(long long total) { long long b = (total + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

}  // namespace

extern "C" int sl_upsample_ce_rows(int B, int H, int W) { return ce_blocks((long long)B * H * W); }

extern "C" int sl_upsample_ce_fwd(const float* logits, const int64_t* target, int B, int K, int h, int w, int H, int W,
                                  int ignore_index, float* partial, sl_stream_t stream) {
  SL_REQUIRE(logits && target && partial && B > 0 && K >= 1 && K <= KMAXC && h > 0 && w > 0 && H > 0 && W > 0, "upsample_ce_fwd: bad args");
  const UpGeom g = make_up(B, K, h, w, H, W);
  hipLaunchKernelGGL(upsample_ce_fwd_kernel, dim3(ce_blocks((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, g, logits, target, ignore_index, partial);
  SL_LAUNCH_CHECK("upsample_ce_fwd_kernel");
  return 0;
}

extern "C" int sl_upsample_ce_finalize(const float* partial, int nblk, float* loss_and_count, sl_stream_t stream) {
  SL_REQUIRE(partial && loss_and_count && nblk > 0, "upsample_ce_finalize: bad args");
  hipLaunchKernelGGL(upsample_ce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblk, loss_and_count);
  SL_LAUNCH_CHECK("upsample_ce_finalize_kernel");
  return 0;
}

extern "C" int sl_upsample_ce_bwd(const float* logits, const int64_t* target, const float* loss_and_count, const float* gscale,
                                  int B, int K, int h, int w, int H, int W, int ignore_index, float* dlogits,
                                  sl_stream_t stream) {
  SL_REQUIRE(logits && target && loss_and_count && gscale && dlogits && K >= 1 && K <= KMAXC, "upsample_ce_bwd: bad args");
  const UpGeom g = make_up(B, K, h, w, H, W);
  const long long cells = (long long)B * h * w;
  // tiled kernel when the joint footprint of a CT x CT cell tile fits the LDS (always for an upsampling factor <= ~12 at K <= 16)
  if (g.sy > 0.f && g.sx > 0.f) {
    UpTile ut;
    ut.NYmax = (int)ceilf((float)(CT + 1) / g.sy) + 3; if (ut.NYmax > H) ut.NYmax = H;
    ut.NXmax = (int)ceilf((float)(CT + 1) / g.sx) + 3; if (ut.NXmax > W) ut.NXmax = W;
    const size_t lds = ((size_t)(CT + 2) * (CT + 2) * K + (3 + CT) * (ut.NYmax + ut.NXmax) + (size_t)ut.NYmax * CT * K + (size_t)ut.NYmax * ut.NXmax * K) * sizeof(float);
    if (lds <= 150 * 1024) {
      static size_t attr = 0;
#ifndef SL_UPCE_NT
#define SL_UPCE_NT 512      // threads per block of the tiled kernel (A/B build: 256)
#endif
      if (lds > attr) { (void)hipFuncSetAttribute((const void*)upsample_ce_bwd_tiled_kernel<SL_UPCE_NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr = 150 * 1024; }
      hipLaunchKernelGGL(upsample_ce_bwd_tiled_kernel<SL_UPCE_NT>, dim3((unsigned)(B * cdiv(h, CT) * cdiv(w, CT))), dim3(SL_UPCE_NT), lds, (hipStream_t)stream, g, ut, logits, target, loss_and_count, gscale, ignore_index, dlogits);
      SL_LAUNCH_CHECK("upsample_ce_bwd_tiled_kernel");
      return 0;
    }
  }
  hipLaunchKernelGGL(upsample_ce_bwd_kernel, dim3((unsigned)((cells * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, logits, target, loss_and_count, gscale, ignore_index, dlogits);
  SL_LAUNCH_CHECK("upsample_ce_bwd_kernel");
  return 0;
}

extern "C" int sl_pseudo_label(const float* logits, int K2, int h, int w, int64_t* mask, int B, int H, int W, int n_base,
                               sl_stream_t stream) {
  SL_REQUIRE(logits && mask && K2 >= 1 && K2 <= KMAXC && B > 0, "pseudo_label: bad args");
  const UpGeom g = make_up(B, K2, h, w, H, W);
  hipLaunchKernelGGL(upsample_argmax_kernel<true>, dim3(ce_blocks((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, g, logits, (uint8_t*)nullptr, mask, n_base);
  SL_LAUNCH_CHECK("pseudo_label_kernel");
  return 0;
}

extern "C" int sl_upsample_argmax(const float* logits, int B, int K, int h, int w, int H, int W, uint8_t* labels,
                                  sl_stream_t stream) {
  SL_REQUIRE(logits && labels && K >= 1 && K <= KMAXC && B > 0, "upsample_argmax: bad args");
  const UpGeom g = make_up(B, K, h, w, H, W);
  hipLaunchKernelGGL(upsample_argmax_kernel<false>, dim3(ce_blocks((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, g, logits, labels, (int64_t*)nullptr, 0);
  SL_LAUNCH_CHECK("upsample_argmax_kernel");
  return 0;
}

extern "C" int sl_upsample_logits(const float* logits, int B, int K, int h, int w, int H, int W, float* out, sl_stream_t stream) {
  SL_REQUIRE(logits && out && K >= 1 && K <= KMAXC && B > 0, "upsample_logits: bad args");
  const UpGeom g = make_up(B, K, h, w, H, W);
  hipLaunchKernelGGL(upsample_logits_kernel, dim3(ce_blocks((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, g, logits, out);
  SL_LAUNCH_CHECK("upsample_logits_kernel");
  return 0;
}

extern "C" int sl_fuse_argmax(const void* mats_dev, int n_models, int K, long long hw, uint8_t* labels, sl_stream_t stream) {
  SL_REQUIRE(mats_dev && labels && n_models >= 1 && K >= 1 && K <= 255 && hw > 0, "fuse_argmax: bad args");
  hipLaunchKernelGGL(fuse_argmax_kernel, dim3(ce_blocks(hw)), dim3(256), 0, (hipStream_t)stream, (const float* const*)mats_dev, n_models, K, hw, labels);
  SL_LAUNCH_CHECK("fuse_argmax_kernel");
  return 0;
}

extern "C" int sl_iou_hist(const uint8_t* pred, const int64_t* target, long long n, int K, int ignore_index, long long* hist,
                           sl_stream_t stream) {
  SL_REQUIRE(pred && target && hist && n > 0 && K >= 1 && K <= 256, "iou_hist: bad args");
  hipLaunchKernelGGL(iou_hist_kernel, dim3(ce_blocks(n)), dim3(256), 0, (hipStream_t)stream, pred, target, n, K, ignore_index, (unsigned long long*)hist);
  SL_LAUNCH_CHECK("iou_hist_kernel");
  return 0;
}

extern "C" int sl_confusion_matrix(const uint8_t* pred, const int64_t* target, long long n, int K, int ignore_index, long long* cm,
                                   sl_stream_t stream) {
  SL_REQUIRE(pred && target && cm && n > 0 && K >= 1 && K <= 64, "confusion_matrix: bad args");
  hipLaunchKernelGGL(confusion_kernel, dim3(ce_blocks(n)), dim3(256), (size_t)K * K * sizeof(unsigned), (hipStream_t)stream, pred, target, n, K, ignore_index,
                     (unsigned long long*)cm);
  SL_LAUNCH_CHECK("confusion_kernel");
  return 0;
}

extern "C" int sl_masked_avg_pool(int dtype, const void* feature, const float* mask, int B, int h, int w, int C, int H, int W,
                                  float* proto, sl_stream_t stream) {
  SL_REQUIRE(feature && mask && proto && B > 0 && C > 0, "masked_avg_pool: bad args");
  const float sy = ac1_scale(H, h), sx = ac1_scale(W, w);
  if (dtype == SL_BF16) hipLaunchKernelGGL(masked_avg_pool_kernel<bf16_t>, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)feature, mask, B, h, w, C, H, W, sy, sx, proto);
  else if (dtype == SL_F32) hipLaunchKernelGGL(masked_avg_pool_kernel<float>, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, (const float*)feature, mask, B, h, w, C, H, W, sy, sx, proto);
  else SL_REQUIRE(false, "masked_avg_pool: bad dtype");
  SL_LAUNCH_CHECK("masked_avg_pool_kernel");
  return 0;
}

extern "C" int sl_nhwc_to_nchw_f32(int dtype, const void* src, float* dst, int B, int H, int W, int C, sl_stream_t stream) {
  SL_REQUIRE(src && dst && B > 0, "nhwc_to_nchw: bad args");
  const long long total = (long long)B * H * W * C;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (dtype == SL_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, B, H * W, C);
  else if (dtype == SL_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)src, dst, B, H * W, C);
  else SL_REQUIRE(false, "nhwc_to_nchw: bad dtype");
  SL_LAUNCH_CHECK("nhwc_to_nchw_kernel");
  return 0;
}

extern "C" int sl_nchw_f32_to_nhwc(int dtype, const float* src, void* dst, int B, int H, int W, int C, sl_stream_t stream) {
  SL_REQUIRE(src && dst && B > 0, "nchw_to_nhwc: bad args");
  const long long total = (long long)B * H * W * C;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (dtype == SL_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, B, H * W, C);
  else if (dtype == SL_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, B, H * W, C);
  else SL_REQUIRE(false, "nchw_to_nhwc: bad dtype");
  SL_LAUNCH_CHECK("nchw_to_nhwc_kernel");
  return 0;
}
