// ResNet stem for gfx950: conv 7x7 s2 p3 (3 -> 64) straight from the NCHW float image, fused BN+ReLU+maxpool 3x3 s2 p1,
// and their backward (networks/backbones/resnet.py:86-90,124-125).  Cin = 3 makes this stage bandwidth/VALU bound,
// so it is a direct LDS-tiled fp32 convolution (input patch + all 64x147 weights resident in LDS), not MFMA.
#include "common.h"

namespace {

constexpr int TS = 16;                 // output tile edge
constexpr int PS = (TS - 1) * 2 + 7;   // input patch edge = 37
constexpr int NTAP = 147;              // 3*7*7

__device__ __forceinline__ void load_patch(float* patch, const float* img, int b, int H, int W, int oy0, int ox0, int tid) {
  // patch[c][PS][PS] of image rows 2*oy0-3 .., zero outside
  for (int e = tid; e < 3 * PS * PS; e += 256) {
    const int c = e / (PS * PS), r = (e / PS) % PS, q = e % PS;
    const int iy = 2 * oy0 - 3 + r, ix = 2 * ox0 - 3 + q;
    float v = 0.f;
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = img[((size_t)(b * 3 + c) * H + iy) * W + ix];
    patch[e] = v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void stem_conv_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                            T* __restrict__ y, float* __restrict__ part, int B, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wl = sm;                  // [147][64]
  float* patch = sm + NTAP * 64;   // [3][37][37]
  const int Ho = H / 2, Wo = W / 2;
  const int tx = cdiv(Wo, TS), ty = cdiv(Ho, TS);
  const int tid = threadIdx.x;
  int blk = blockIdx.x;
  const int bx = blk % tx; blk /= tx;
  const int by = blk % ty; const int b = blk / ty;
  for (int e = tid; e < 64 * NTAP; e += 256) { const int n = e / NTAP, t = e % NTAP; wl[t * 64 + n] = w[e]; }
  load_patch(patch, img, b, H, W, by * TS, bx * TS, tid);
  __syncthreads();

  const int pg = tid & 63, cg = tid >> 6;     // wave = channel group of 16, lane = 4-pixel group
  const int py = pg >> 2, px0 = (pg & 3) * 4;
  float acc[4][16];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;
  for (int c = 0; c < 3; ++c)
    for (int ky = 0; ky < 7; ++ky) {
      const float* prow = patch + (c * PS + 2 * py + ky) * PS + 2 * px0;
      const float* wrow = wl + ((c * 7 + ky) * 7) * 64 + cg * 16;
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        float wv[16];
#pragma unroll
        for (int k = 0; k < 16; k += 4) { const float4 t = *(const float4*)(wrow + kx * 64 + k); wv[k] = t.x; wv[k + 1] = t.y; wv[k + 2] = t.z; wv[k + 3] = t.w; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = prow[2 * j + kx];
#pragma unroll
          for (int k = 0; k < 16; ++k) acc[j][k] = fmaf(a, wv[k], acc[j][k]);
        }
      }
    }
  const int oy = by * TS + py;
  float s1[16], s2[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ox = bx * TS + px0 + j;
    if (oy < Ho && ox < Wo) {
      T* o = y + ((size_t)(b * Ho + oy) * Wo + ox) * 64 + cg * 16;
      constexpr int V = Vec16<T>::N;
#pragma unroll
      for (int k = 0; k < 16; k += V) *(uint4*)(o + k) = pack16<T>(&acc[j][k]);
#pragma unroll
      for (int k = 0; k < 16; ++k) { s1[k] += acc[j][k]; s2[k] += acc[j][k] * acc[j][k]; }
    }
  }
  if (part) {
#pragma unroll
    for (int k = 0; k < 16; ++k) { s1[k] = wave_sum(s1[k]); s2[k] = wave_sum(s2[k]); }
    if (pg == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        part[((size_t)blockIdx.x * 2 + 0) * 64 + cg * 16 + k] = s1[k];
        part[((size_t)blockIdx.x * 2 + 1) * 64 + cg * 16 + k] = s2[k];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 forward on the matrix cores: the 16x16-pixel output tile of a block is a [256 pixels] x [160 = 147 taps, zero padded] x [64 channels]
// GEMM whose pixel operand is gathered from the bf16 image patch in the LDS (im2col is never materialised).  Operand roles as in the conv
// kernels: first MFMA operand = weight rows (-> accumulator registers), second = pixel rows (-> lanes).  A wave owns 64 pixels x 64 channels
// (4 accumulators).  The math is 40 MFMAs per wave; what decides the time is the glue around them, so: the weight fragments come ready-made
// from a 20 KiB table (stem_weight_frag_kernel, one tiny launch per step) straight into registers, the patch is loaded without a division
// in the loop, the im2col offsets of a k-step are compile-time constants selected by the lane half, and the BN statistic partials are
// column sums over the staged (rounded) tile, like the conv kernels' epilogue.
constexpr int MP = 38;                 // patch row pitch (bf16 elements)
constexpr int KSTEPS = 10;             // 160 = 147 taps zero padded

// wfrag[(nb*10 + ks)*64 + lane] = W[nb*32 + (lane & 31)][ks*16 + (lane >> 5)*8 .. +8] as bf16 (taps >= 147 zero)
__global__ void stem_weight_frag_kernel(const float* __restrict__ w, uint4* __restrict__ wfrag) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * KSTEPS * 64) return;
  const int lane = t & 63, ks = (t >> 6) % KSTEPS, nb = (t >> 6) / KSTEPS;
  const int n = nb * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
  unsigned short v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = k0 + e < NTAP ? from_f<bf16_t>(w[n * NTAP + k0 + e]) : (unsigned short)0;
  wfrag[t] = make_uint4(v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16), v[6] | ((unsigned)v[7] << 16));
}

__host__ __device__ constexpr int stem_tap_off(int k) { return k < NTAP ? ((k / 49) * PS + (k % 49) / 7) * MP + (k % 7) : 0; }

template <int KS>
__device__ __forceinline__ void stem_kstep(const bf16_t* __restrict__ patch, const int (&pbase)[2], int fh, const uint4* __restrict__ wfrag, int lane, f32x16_t (&acc)[2][2]) {
  const uint4 w0 = wfrag[(0 * KSTEPS + KS) * 64 + lane], w1 = wfrag[(1 * KSTEPS + KS) * 64 + lane];
  int off[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) off[e] = fh ? stem_tap_off(KS * 16 + 8 + e) : stem_tap_off(KS * 16 + e);      // two immediates and a select
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    unsigned v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = patch[pbase[rb] + off[e]];
    // taps >= 147 read patch[pbase + 0] (a finite bf16) against a ZERO weight: no masking needed
    const uint4 a = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
    acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w0), __builtin_bit_cast(bf16x8_t, a), acc[rb][0], 0, 0, 0);
    acc[rb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w1), __builtin_bit_cast(bf16x8_t, a), acc[rb][1], 0, 0, 0);
  }
}

__global__ __launch_bounds__(256) void stem_conv_fwd_mfma_kernel(const float* __restrict__ img, const uint4* __restrict__ wfrag,
                                                                 bf16_t* __restrict__ y, float* __restrict__ part, int B, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* patch = (bf16_t*)smem;                           // [3][PS][MP]
  bf16_t* outt = (bf16_t*)smem;                            // [256][64 + 8]: takes the patch's place after the main loop
  float* red = (float*)(smem + 256 * 72 * sizeof(bf16_t)); // [4][2][64]
  const int Ho = H / 2, Wo = W / 2;
  const int tx = cdiv(Wo, TS), ty = cdiv(Ho, TS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int blk = blockIdx.x;
  const int bx = blk % tx; blk /= tx;
  const int by = blk % ty; const int b = blk / ty;
  {                                                        // 3 x 37 patch rows, two threads per row (19 + 18 elements), no division in the loop
    const int rr = tid >> 1, half = tid & 1;
    if (rr < 3 * PS) {
      const int c = rr / PS, r = rr - c * PS;
      const int iy = 2 * by * TS - 3 + r, ix0 = 2 * bx * TS - 3;
      const bool rowok = (unsigned)iy < (unsigned)H;
      const float* src = img + ((size_t)(b * 3 + c) * H + (rowok ? iy : 0)) * W;
      bf16_t* dst = patch + rr * MP;
      const int q0 = half * 19, q1 = half ? PS : 19;
      for (int q = q0; q < q1; ++q) {
        const int ix = ix0 + q;
        dst[q] = from_f<bf16_t>(rowok && (unsigned)ix < (unsigned)W ? src[ix] : 0.f);
      }
    }
  }
  __syncthreads();

  const int l31 = lane & 31, fh = lane >> 5;
  f32x16_t acc[2][2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][nb][r] = 0.f;
  int pbase[2];                                            // pixel of this lane in row block rb: p = wave*64 + rb*32 + l31 -> (py, px) of the tile
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int pidx = wave * 64 + rb * 32 + l31;
    pbase[rb] = (2 * (pidx >> 4)) * MP + 2 * (pidx & 15);
  }
  stem_kstep<0>(patch, pbase, fh, wfrag, lane, acc); stem_kstep<1>(patch, pbase, fh, wfrag, lane, acc);
  stem_kstep<2>(patch, pbase, fh, wfrag, lane, acc); stem_kstep<3>(patch, pbase, fh, wfrag, lane, acc);
  stem_kstep<4>(patch, pbase, fh, wfrag, lane, acc); stem_kstep<5>(patch, pbase, fh, wfrag, lane, acc);
  stem_kstep<6>(patch, pbase, fh, wfrag, lane, acc); stem_kstep<7>(patch, pbase, fh, wfrag, lane, acc);
  stem_kstep<8>(patch, pbase, fh, wfrag, lane, acc); stem_kstep<9>(patch, pbase, fh, wfrag, lane, acc);
  __syncthreads();                                         // the patch is dead: the staging tile takes its place
  // D layout: lane = pixel (l31), register r = channel (r&3) + 8*(r>>2) + 4*fh of the 32-channel block
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int pidx = wave * 64 + rb * 32 + l31;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint2 pk;
        pk.x = (unsigned)from_f<bf16_t>(acc[rb][nb][4 * q + 0]) | ((unsigned)from_f<bf16_t>(acc[rb][nb][4 * q + 1]) << 16);
        pk.y = (unsigned)from_f<bf16_t>(acc[rb][nb][4 * q + 2]) | ((unsigned)from_f<bf16_t>(acc[rb][nb][4 * q + 3]) << 16);
        *(uint2*)(outt + pidx * 72 + nb * 32 + 8 * q + 4 * fh) = pk;
      }
  }
  __syncthreads();
  for (int e = tid; e < 256 * 8; e += 256) {               // 8 x 16-byte chunks per pixel row
    const int pidx = e >> 3, ch8 = e & 7;
    const int oy = by * TS + (pidx >> 4), ox = bx * TS + (pidx & 15);
    if (oy < Ho && ox < Wo) *(uint4*)(y + ((size_t)(b * Ho + oy) * Wo + ox) * 64 + ch8 * 8) = *(const uint4*)(outt + pidx * 72 + ch8 * 8);
  }
  if (part) {                                              // column sums of the stored (rounded) tile: thread = (channel, quarter of the pixels)
    const int ch = tid & 63, qtr = tid >> 6;
    float a = 0.f, c = 0.f;
    for (int pp = 0; pp < 64; ++pp) {
      const int pidx = qtr * 64 + pp;
      const int oy = by * TS + (pidx >> 4), ox = bx * TS + (pidx & 15);
      if (oy < Ho && ox < Wo) { const float v = to_f<bf16_t>(outt[pidx * 72 + ch]); a += v; c += v * v; }
    }
    red[(qtr * 2 + 0) * 64 + ch] = a; red[(qtr * 2 + 1) * 64 + ch] = c;
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6;
      part[((size_t)blockIdx.x * 2 + which) * 64 + ch] = red[(0 * 2 + which) * 64 + ch] + red[(1 * 2 + which) * 64 + ch] + red[(2 * 2 + which) * 64 + ch] + red[(3 * 2 + which) * 64 + ch];
    }
  }
}

// pooled = maxpool3x3s2p1(relu(c0*scale+shift)); idx = first-max window position ky*3+kx (ATen scan order)
template <typename T>
__global__ void stem_bn_relu_pool_fwd_kernel(const T* __restrict__ c0, const float* __restrict__ scale, const float* __restrict__ shift,
                                             T* __restrict__ pooled, uint8_t* __restrict__ idx, int B, int Hc, int Wc) {
  constexpr int V = Vec16<T>::N, NV = 64 / V;
  const int Hp = Hc / 2, Wp = Wc / 2;
  const long long total = (long long)B * Hp * Wp * NV;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % NV); long long r = i / NV;
    const int px = (int)(r % Wp); r /= Wp;
    const int py = (int)(r % Hp); const int b = (int)(r / Hp);
    float best[V]; int bi[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { best[k] = -INFINITY; bi[k] = 0; }
    float sc[V], sh[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { sc[k] = scale[v * V + k]; sh[k] = shift[v * V + k]; }
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = 2 * py - 1 + ky;
      if ((unsigned)yy >= (unsigned)Hc) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = 2 * px - 1 + kx;
        if ((unsigned)xx >= (unsigned)Wc) continue;
        float xv[V];
        unpack16<T>(*(const uint4*)(c0 + ((size_t)(b * Hc + yy) * Wc + xx) * 64 + v * V), xv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
          float a = xv[k] * sc[k] + sh[k];
          a = a > 0.f ? a : 0.f;
          if (a > best[k] || a != a) { best[k] = a; bi[k] = ky * 3 + kx; }
        }
      }
    }
    const size_t o = ((size_t)(b * Hp + py) * Wp + px) * 64 + v * V;
    *(uint4*)(pooled + o) = pack16<T>(best);
    if (idx) {
#pragma unroll
      for (int k = 0; k < V; ++k) idx[o + k] = (uint8_t)bi[k];
    }
  }
}

// g0[b][y][x][c] = relu'(a0) * sum over windows whose first max sits at (y,x) of dpooled
// STAT: also the reduce pass of bn1's backward (resnet.py:124 backward): the kernel reads c0 anyway, so the column sums (sum g0, sum g0 * xhat) of each block's
// elements leave with it as part[block][2][64] (fixed order; the same partial format as bn_bwd_reduce_kernel) -- bn_bwd_reduce's sweep over g0 and c0 (268 MB at the
// bench shape, 48 us) disappears.  A thread's channel vector is the same in every iteration (the grid stride is a multiple of the vectors per pixel).
template <typename T, bool STAT>
__global__ __launch_bounds__(256) void stem_pool_relu_bwd_kernel(const T* __restrict__ dp, const uint8_t* __restrict__ idx, const T* __restrict__ c0,
                                          const float* __restrict__ scale, const float* __restrict__ shift, T* __restrict__ g0,
                                          int B, int Hc, int Wc, const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ part) {
  constexpr int V = Vec16<T>::N, NV = 64 / V;
  const int Hp = Hc / 2, Wp = Wc / 2;
  const long long total = (long long)B * Hc * Wc * NV;
  float s1[V], s2[V], mu[V], is[V];
  if (STAT) {
    const int v0 = threadIdx.x % NV;
#pragma unroll
    for (int k = 0; k < V; ++k) { s1[k] = 0.f; s2[k] = 0.f; mu[k] = mean[v0 * V + k]; is[k] = invstd[v0 * V + k]; }
  }
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % NV); long long r = i / NV;
    const int x = (int)(r % Wc); r /= Wc;
    const int y = (int)(r % Hc); const int b = (int)(r / Hc);
    float g[V], xv[V];
#pragma unroll
    for (int k = 0; k < V; ++k) g[k] = 0.f;
    // windows py with 2py-1 <= y <= 2py+1
    const int py_lo = y / 2, py_hi = (y + 1) / 2, px_lo = x / 2, px_hi = (x + 1) / 2;
    for (int py = py_lo; py <= py_hi; ++py) {
      if (py >= Hp) continue;
      const int ky = y - (2 * py - 1);
      for (int px = px_lo; px <= px_hi; ++px) {
        if (px >= Wp) continue;
        const int kx = x - (2 * px - 1);
        const size_t o = ((size_t)(b * Hp + py) * Wp + px) * 64 + v * V;
        float d[V];
        unpack16<T>(*(const uint4*)(dp + o), d);
        const int want = ky * 3 + kx;
#pragma unroll
        for (int k = 0; k < V; ++k) if (idx[o + k] == want) g[k] += d[k];
      }
    }
    const size_t oc = ((size_t)(b * Hc + y) * Wc + x) * 64 + v * V;
    unpack16<T>(*(const uint4*)(c0 + oc), xv);
#pragma unroll
    for (int k = 0; k < V; ++k) g[k] = (xv[k] * scale[v * V + k] + shift[v * V + k]) > 0.f ? g[k] : 0.f;
    const uint4 q = pack16<T>(g);
    *(uint4*)(g0 + oc) = q;
    if (STAT) {
      float gr[V];
      unpack16<T>(q, gr);                       // the sums are over the STORED (rounded) gradient, like the stand-alone reduce pass
#pragma unroll
      for (int k = 0; k < V; ++k) { s1[k] += gr[k]; s2[k] += gr[k] * ((xv[k] - mu[k]) * is[k]); }
    }
  }
  if (STAT) {
    __shared__ float red[256 * 2 * V];
#pragma unroll
    for (int k = 0; k < V; ++k) { red[threadIdx.x * 2 * V + k] = s1[k]; red[threadIdx.x * 2 * V + V + k] = s2[k]; }
    __syncthreads();
    if (threadIdx.x < 128) {
      const int which = threadIdx.x >> 6, c = threadIdx.x & 63, v = c / V, k = c % V;
      float a = 0.f;
      for (int j = 0; j < 256 / NV; ++j) a += red[(j * NV + v) * 2 * V + which * V + k];
      part[((size_t)blockIdx.x * 2 + which) * 64 + c] = a;
    }
  }
}

// dw partial per block: ws[blk][64][147] (OIHW order inside), block loops over `tiles_per_blk` 16x16 tiles
template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ img, const T* __restrict__ dc0, float* __restrict__ ws,
                                                         int B, int H, int W, int tiles_per_blk, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* dyl = sm;                    // [256 px][64]
  float* patch = sm + 256 * 64;       // [3][37][37]
  const int Ho = H / 2, Wo = W / 2;
  const int tx = cdiv(Wo, TS), ty = cdiv(Ho, TS);
  const int tid = threadIdx.x, n = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: (c,ky) combos g, g+4, ...
  float acc[6][7];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[i][k] = 0.f;
  for (int t = 0; t < tiles_per_blk; ++t) {
    int tile = blockIdx.x * tiles_per_blk + t;
    if (tile >= ntiles) break;
    const int bx = tile % tx; tile /= tx;
    const int by = tile % ty; const int b = tile / ty;
    __syncthreads();
    load_patch(patch, img, b, H, W, by * TS, bx * TS, tid);
    for (int e = tid; e < 256 * 64; e += 256) {
      const int p = e >> 6, c = e & 63;
      const int oy = by * TS + (p >> 4), ox = bx * TS + (p & 15);
      float v = 0.f;
      if (oy < Ho && ox < Wo) v = to_f<T>(dc0[((size_t)(b * Ho + oy) * Wo + ox) * 64 + c]);
      dyl[e] = v;
    }
    __syncthreads();
    for (int p = 0; p < 256; ++p) {
      const float a = dyl[p * 64 + n];
      const int py = p >> 4, px = p & 15;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int id = g + 4 * i;
        if (id < 21) {
          const int c = id / 7, ky = id - 7 * c;
          const float* prow = patch + (c * PS + 2 * py + ky) * PS + 2 * px;
#pragma unroll
          for (int k = 0; k < 7; ++k) acc[i][k] = fmaf(a, prow[k], acc[i][k]);
        }
      }
    }
  }
  float* o = ws + (size_t)blockIdx.x * 64 * NTAP + n * NTAP;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int id = g + 4 * i;
    if (id < 21) {
#pragma unroll
      for (int k = 0; k < 7; ++k) o[id * 7 + k] = acc[i][k];
    }
  }
}

// bf16 weight gradient on the matrix cores: dw[n][k] = sum over pixels of dc0[p][n] * patch(p, k).  The reduction index is the pixel, so both
// MFMA operands are gathered K-major from the LDS: first operand = dc0^T (rows = output channels -> registers), second = the im2col view of the
// bf16 image patch (rows = taps -> lanes), 8 consecutive pixels of one tile row per lane.  A wave reduces its 64 pixels of every tile the
// persistent block walks into ten 32x32 accumulators (64 channels x 160 taps); the four waves are summed through the LDS at the end and the
// block writes one [64][147] partial (the existing fixed-order reduce kernel adds the blocks).
__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const float* __restrict__ img, const bf16_t* __restrict__ dc0, float* __restrict__ ws,
                                                              int B, int H, int W, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* patch = (bf16_t*)smem;                           // [3][PS][MP]                       8 436 B
  bf16_t* dct = patch + 3 * PS * MP + 6;                   // [256][64 + 8] (16-byte aligned)  36 864 B
  float* red = (float*)smem;                               // [4][16][64] floats, after the tile loop
  const int Ho = H / 2, Wo = W / 2;
  const int tx = cdiv(Wo, TS), ty = cdiv(Ho, TS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, fh = lane >> 5;
  f32x16_t acc[2][5];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int kb = 0; kb < 5; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][kb][r] = 0.f;
  int toff[5];                                             // patch offset of this lane's tap in tap block kb (taps >= 147: any valid address)
#pragma unroll
  for (int kb = 0; kb < 5; ++kb) {
    const int k = kb * 32 + l31;
    const int c = k / 49, r = k - c * 49, ky = r / 7, kx = r - ky * 7;
    toff[kb] = k < NTAP ? (c * PS + ky) * MP + kx : 0;
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int blk = tile;
    const int bx = blk % tx; blk /= tx;
    const int by = blk % ty; const int b = blk / ty;
    {
      const int rr = tid >> 1, half = tid & 1;
      if (rr < 3 * PS) {
        const int c = rr / PS, r = rr - c * PS;
        const int iy = 2 * by * TS - 3 + r, ix0 = 2 * bx * TS - 3;
        const bool rowok = (unsigned)iy < (unsigned)H;
        const float* src = img + ((size_t)(b * 3 + c) * H + (rowok ? iy : 0)) * W;
        bf16_t* dst = patch + rr * MP;
        const int q0 = half * 19, q1 = half ? PS : 19;
        for (int q = q0; q < q1; ++q) {
          const int ix = ix0 + q;
          dst[q] = from_f<bf16_t>(rowok && (unsigned)ix < (unsigned)W ? src[ix] : 0.f);
        }
      }
    }
    for (int e = tid; e < 256 * 8; e += 256) {             // the dc0 tile, zero outside the map
      const int pidx = e >> 3, ch8 = e & 7;
      const int oy = by * TS + (pidx >> 4), ox = bx * TS + (pidx & 15);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (oy < Ho && ox < Wo) v = *(const uint4*)(dc0 + ((size_t)(b * Ho + oy) * Wo + ox) * 64 + ch8 * 8);
      *(uint4*)(dct + pidx * 72 + ch8 * 8) = v;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      // pixels p = wave*64 + ks*16 + fh*8 + e: tile row py = wave*4 + ks, columns px = fh*8 + e
      const int prow = wave * 64 + ks * 16 + fh * 8;
      uint4 fa[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        unsigned v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = dct[(prow + e) * 72 + nb * 32 + l31];
        fa[nb] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
      }
      const int pb = (2 * (wave * 4 + ks)) * MP + 2 * (fh * 8);
#pragma unroll
      for (int kb = 0; kb < 5; ++kb) {
        unsigned v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = patch[pb + toff[kb] + 2 * e];
        const uint4 fb = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          acc[nb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[nb]), __builtin_bit_cast(bf16x8_t, fb), acc[nb][kb], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // D layout: lane l31 = tap of the block, register r = channel (r&3) + 8*(r>>2) + 4*fh of the block.  Sum the four waves, one accumulator at a time.
  float* out = ws + (size_t)blockIdx.x * 64 * NTAP;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int kb = 0; kb < 5; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[nb][kb][r];
      __syncthreads();
      for (int e = tid; e < 16 * 64; e += 256) {
        const int r = e >> 6, ln = e & 63;
        const float t = red[(0 * 16 + r) * 64 + ln] + red[(1 * 16 + r) * 64 + ln] + red[(2 * 16 + r) * 64 + ln] + red[(3 * 16 + r) * 64 + ln];
        const int n = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), k = kb * 32 + (ln & 31);
        if (k < NTAP) out[n * NTAP + k] = t;
      }
      __syncthreads();
    }
}

// im2col of the stem's receptive fields: col[m][t], t = c*49 + ky*7 + kx (OIHW tap order), padded to 192 columns with
// zeros, so that the 7x7 weight gradient becomes a plain [64 x M] x [M x 192] MFMA reduction (conv_wgrad kernel).
template <typename T>
__global__ void stem_im2col_kernel(const float* __restrict__ img, T* __restrict__ col, int B, int H, int W) {
  constexpr int V = Vec16<T>::N, NV = 192 / V;
  const int Ho = H / 2, Wo = W / 2;
  const long long total = (long long)B * Ho * Wo * NV;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % NV); long long r = i / NV;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho); const int b = (int)(r / Ho);
    float o[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const int t = v * V + k;
      float val = 0.f;
      if (t < NTAP) {
        const int c = t / 49, rem = t - c * 49, ky = rem / 7, kx = rem - ky * 7;
        const int iy = 2 * oy - 3 + ky, ix = 2 * ox - 3 + kx;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) val = img[((size_t)(b * 3 + c) * H + iy) * W + ix];
      }
      o[k] = val;
    }
    *(uint4*)(col + (size_t)i * V) = pack16<T>(o);
  }
}

inline int stem_tiles(int B, int H, int W) { return B * cdiv(H / 2, TS) * cdiv(W / 2, TS); }
inline int stem_wgrad_blocks(int ntiles) { return ntiles < 512 ? ntiles : 512; }

}  // namespace

extern "C" int sl_stem_conv_stat_rows(int B, int H, int W) { return stem_tiles(B, H, W); }

extern "C" size_t sl_stem_conv_fwd_workspace(int dtype) { return dtype == SL_BF16 ? 2 * KSTEPS * 64 * sizeof(uint4) : 0; }

extern "C" int sl_stem_conv_fwd(int dtype, const float* img_nchw, const float* w_oihw, void* y, float* stat_partial, int B,
                                int H, int W, void* workspace, sl_stream_t stream) {
  SL_REQUIRE(img_nchw && w_oihw && y && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "stem_conv_fwd: bad args");
  const size_t lds = (NTAP * 64 + 3 * PS * PS) * sizeof(float);
  dim3 grid(stem_tiles(B, H, W));
  if (dtype == SL_BF16) {
    SL_REQUIRE(workspace && ((size_t)workspace & 15) == 0, "stem_conv_fwd: bf16 needs the 16-byte aligned workspace of sl_stem_conv_fwd_workspace()");
    uint4* wfrag = (uint4*)workspace;                                           // the weights in MFMA fragment order, rebuilt on every call (20 KiB)
    hipLaunchKernelGGL(stem_weight_frag_kernel, dim3(cdiv(2 * KSTEPS * 64, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, wfrag);
    const size_t l2 = 256 * 72 * sizeof(bf16_t) + 4 * 2 * 64 * sizeof(float);   // staging tile + statistic scratch (the patch, 8.4 KB, lives in the tile's place first)
    hipLaunchKernelGGL(stem_conv_fwd_mfma_kernel, grid, dim3(256), l2, (hipStream_t)stream, img_nchw, (const uint4*)wfrag, (bf16_t*)y, stat_partial, B, H, W);
    SL_LAUNCH_CHECK("stem_conv_fwd_mfma_kernel");
    return 0;
  }
  if (dtype == SL_F32) hipLaunchKernelGGL(stem_conv_fwd_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, img_nchw, w_oihw, (float*)y, stat_partial, B, H, W);
  else SL_REQUIRE(false, "stem_conv_fwd: bad dtype");
  SL_LAUNCH_CHECK("stem_conv_fwd_kernel");
  return 0;
}

extern "C" int sl_stem_bn_relu_pool_fwd(int dtype, const void* c0, const float* scale, const float* shift, void* pooled,
                                        uint8_t* argmax, int B, int Hc, int Wc, sl_stream_t stream) {
  SL_REQUIRE(c0 && scale && shift && pooled && B > 0 && Hc % 2 == 0 && Wc % 2 == 0, "stem_bn_relu_pool_fwd: bad args");
  const long long total = (long long)B * (Hc / 2) * (Wc / 2) * (dtype == SL_BF16 ? 8 : 16);
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (dtype == SL_BF16) hipLaunchKernelGGL(stem_bn_relu_pool_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)c0, scale, shift, (bf16_t*)pooled, argmax, B, Hc, Wc);
  else if (dtype == SL_F32) hipLaunchKernelGGL(stem_bn_relu_pool_fwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)c0, scale, shift, (float*)pooled, argmax, B, Hc, Wc);
  else SL_REQUIRE(false, "stem_bn_relu_pool_fwd: bad dtype");
  SL_LAUNCH_CHECK("stem_bn_relu_pool_fwd_kernel");
  return 0;
}

extern "C" int sl_stem_pool_relu_bwd(int dtype, const void* dpooled, const uint8_t* argmax, const void* c0, const float* scale,
                                     const float* shift, void* g0, int B, int Hc, int Wc, sl_stream_t stream) {
  SL_REQUIRE(dpooled && argmax && c0 && scale && shift && g0 && B > 0 && Hc % 2 == 0 && Wc % 2 == 0, "stem_pool_relu_bwd: bad args");
  const long long total = (long long)B * Hc * Wc * (dtype == SL_BF16 ? 8 : 16);
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (dtype == SL_BF16) hipLaunchKernelGGL((stem_pool_relu_bwd_kernel<bf16_t, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dpooled, argmax, (const bf16_t*)c0, scale, shift, (bf16_t*)g0, B, Hc, Wc, nullptr, nullptr, nullptr);
  else if (dtype == SL_F32) hipLaunchKernelGGL((stem_pool_relu_bwd_kernel<float, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)dpooled, argmax, (const float*)c0, scale, shift, (float*)g0, B, Hc, Wc, nullptr, nullptr, nullptr);
  else SL_REQUIRE(false, "stem_pool_relu_bwd: bad dtype");
  SL_LAUNCH_CHECK("stem_pool_relu_bwd_kernel");
  return 0;
}

// The same + the column sums of bn1's backward: stat_partial [sl_stem_pool_relu_bwd_bnstat_rows(B, Hc, Wc)][2][64] (consumed by sl_bn_bwd_finalize like bn_bwd_reduce's)
extern "C" int sl_stem_pool_relu_bwd_bnstat_rows(int B, int Hc, int Wc) {
  const long long px = (long long)B * Hc * Wc;
  if (B <= 0 || Hc <= 0 || Wc <= 0) return 0;
  const long long blocks = (px * 8 + 255) / 256;           // at least one 16-byte vector per thread in either dtype
  return (int)(blocks < 2048 ? blocks : 2048);
}
extern "C" int sl_stem_pool_relu_bwd_bnstat(int dtype, const void* dpooled, const uint8_t* argmax, const void* c0, const float* scale, const float* shift,
                                            const float* mean, const float* invstd, void* g0, float* stat_partial, int B, int Hc, int Wc, sl_stream_t stream) {
  SL_REQUIRE(dpooled && argmax && c0 && scale && shift && mean && invstd && g0 && stat_partial && B > 0 && Hc % 2 == 0 && Wc % 2 == 0, "stem_pool_relu_bwd_bnstat: bad args");
  const int blocks = sl_stem_pool_relu_bwd_bnstat_rows(B, Hc, Wc);
  if (dtype == SL_BF16) hipLaunchKernelGGL((stem_pool_relu_bwd_kernel<bf16_t, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dpooled, argmax, (const bf16_t*)c0, scale, shift, (bf16_t*)g0, B, Hc, Wc, mean, invstd, stat_partial);
  else if (dtype == SL_F32) hipLaunchKernelGGL((stem_pool_relu_bwd_kernel<float, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)dpooled, argmax, (const float*)c0, scale, shift, (float*)g0, B, Hc, Wc, mean, invstd, stat_partial);
  else SL_REQUIRE(false, "stem_pool_relu_bwd_bnstat: bad dtype");
  SL_LAUNCH_CHECK("stem_pool_relu_bwd_kernel");
  return 0;
}

extern "C" size_t sl_stem_conv_bwd_weight_workspace(int B, int H, int W) {
  return (size_t)stem_wgrad_blocks(stem_tiles(B, H, W)) * 64 * NTAP * sizeof(float);
}

extern "C" int sl_stem_conv_bwd_weight(int dtype, const float* img_nchw, const void* dc0, float* dw_oihw, void* workspace,
                                       size_t workspace_bytes, int B, int H, int W, sl_stream_t stream) {
  SL_REQUIRE(img_nchw && dc0 && dw_oihw && workspace && B > 0 && H % 2 == 0 && W % 2 == 0, "stem_conv_bwd_weight: bad args");
  const int ntiles = stem_tiles(B, H, W), nblk = stem_wgrad_blocks(ntiles), tpb = cdiv(ntiles, nblk);
  if (workspace_bytes < (size_t)nblk * 64 * NTAP * sizeof(float)) { sl_set_error("stem_conv_bwd_weight: workspace too small"); return SL_EWORKSPACE; }
  const size_t lds = (256 * 64 + 3 * PS * PS) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SL_BF16) {
    const size_t l2 = (size_t)(3 * PS * MP + 6) * sizeof(bf16_t) + 256 * 72 * sizeof(bf16_t);
    hipLaunchKernelGGL(stem_wgrad_mfma_kernel, dim3(nblk), dim3(256), l2, st, img_nchw, (const bf16_t*)dc0, (float*)workspace, B, H, W, ntiles);
    SL_LAUNCH_CHECK("stem_wgrad_mfma_kernel");
  } else if (dtype == SL_F32) hipLaunchKernelGGL(stem_wgrad_kernel<float>, dim3(nblk), dim3(256), lds, st, img_nchw, (const float*)dc0, (float*)workspace, B, H, W, tpb, ntiles);
  else SL_REQUIRE(false, "stem_conv_bwd_weight: bad dtype");
  SL_LAUNCH_CHECK("stem_wgrad_kernel");
  return sl_colsum_finalize((const float*)workspace, nblk, 64 * NTAP, dw_oihw, stream);       // fixed-order sum of the block partials (64 x 16 lanes per block of columns)
}

extern "C" int sl_stem_im2col(int dtype, const float* img_nchw, void* col, int B, int H, int W, sl_stream_t stream) {
  SL_REQUIRE(img_nchw && col && B > 0 && H % 2 == 0 && W % 2 == 0, "stem_im2col: bad args");
  const long long total = (long long)B * (H / 2) * (W / 2) * (dtype == SL_BF16 ? 24 : 48);
  const int blocks = (int)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
  if (dtype == SL_BF16) hipLaunchKernelGGL(stem_im2col_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img_nchw, (bf16_t*)col, B, H, W);
  else if (dtype == SL_F32) hipLaunchKernelGGL(stem_im2col_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img_nchw, (float*)col, B, H, W);
  else SL_REQUIRE(false, "stem_im2col: bad dtype");
  SL_LAUNCH_CHECK("stem_im2col_kernel");
  return 0;
}
