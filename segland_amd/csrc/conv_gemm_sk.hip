// 64 -> 64 3x3 patch kernel (family 7) and the pixel-stationary 1x1 kernels (family 6: K <= 256) of the implicit-GEMM convolution.  See conv_gemm_common.h.
#include "conv_gemm_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// 64 -> 64 channels, 3x3, stride 1, dilation 1 (layer1.conv2 at 128 x 128), forward and data gradient.  The tile kernels above fetch the pixel
// operand once per tap: with only 64 output channels per 64 input channels that makes the launch LDS-fill bound at a third of what the MFMAs
// could do.  Here a persistent block owns 16 x 16-pixel tiles: the input patch WITH its halo goes to the LDS once (324 pixels for 256 outputs)
// and the nine taps are nine shifted 16-byte reads per fragment; the whole weight tensor (64 x 576) sits in the LDS in MFMA fragment order; the
// next tile's patch is loaded into registers while the current one is multiplied.  Same result layout, same BN statistic partials (one row per
// TILE: sl_conv2d_stat_rows knows) as the generic path.  mode 1 (data gradient) = the same kernel on the transposed weights with the taps flipped.
using slconv::C64_T;
constexpr int C64_PW = C64_T + 2, C64_PITCH = 144;        // 128 B of channels + 16 B pad: conflict-free 16-byte fragment reads
constexpr int C64_WFRAG = 2 * 36 * 64 * 16;                           // weights in fragment order, bytes
constexpr int C64_LDS = C64_WFRAG + C64_PW * C64_PW * C64_PITCH + 4 * 2 * 64 * (int)sizeof(float);
__global__ __launch_bounds__(256) void conv_c64k3_kernel(ConvGemmParams p, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* wl = (uint4*)smem;                                           // [nb][ks][lane]
  unsigned char* patch = smem + C64_WFRAG;                            // [18*18][144 B]; the output tile [256][144 B] takes its place after the MFMA loop
  float* red = (float*)(patch + C64_PW * C64_PW * C64_PITCH);         // [4][2][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, fh = lane >> 5;
  const int H = p.Hs, W = p.Ws;
  const int tx = cdiv(W, C64_T), ty = cdiv(H, C64_T);
  const bf16_t* src = (const bf16_t*)p.src1;
  for (int e = tid; e < 2 * 36 * 64; e += 256) {                      // wt [64][9][64] -> fragments: rows nb*32 + l31, tap ks/4, channels (ks%4)*16 + fh*8 .. +8
    const int ln = e & 63, ks = (e >> 6) % 36, nb = (e >> 6) / 36;
    wl[e] = *(const uint4*)((const bf16_t*)p.wt + ((size_t)(nb * 32 + (ln & 31)) * 9 + (ks >> 2)) * 64 + (ks & 3) * 16 + (ln >> 5) * 8);
  }
  constexpr int NCH = C64_PW * C64_PW * 8, CPT = (NCH + 255) / 256;   // 11 chunks of 16 B per thread
  uint4 stage[CPT];
  auto fetch = [&](int tile) {
    int blk = tile;
    const int bx = blk % tx; blk /= tx;
    const int by = blk % ty; const int b = blk / ty;
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      const int e = tid + u * 256;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (e < NCH) {
        const int ch8 = e & 7, pp = e >> 3, py = pp / C64_PW, px = pp - py * C64_PW;
        const int iy = by * C64_T - 1 + py, ix = bx * C64_T - 1 + px;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = *(const uint4*)(src + ((size_t)(b * H + iy) * W + ix) * 64 + ch8 * 8);
      }
      stage[u] = v;
    }
  };
  int pbase[2];                                                       // patch byte offset of this lane's pixel in row block rb (tap (0,0))
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int pidx = wave * 64 + rb * 32 + l31;
    pbase[rb] = ((pidx >> 4) * C64_PW + (pidx & 15)) * C64_PITCH + fh * 16;
  }
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int blk = tile;
    const int bx = blk % tx; blk /= tx;
    const int by = blk % ty; const int b = blk / ty;
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      const int e = tid + u * 256;
      if (e < NCH) *(uint4*)(patch + (e >> 3) * C64_PITCH + (e & 7) * 16) = stage[u];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
    f32x16_t acc[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][nb][r] = 0.f;
#pragma unroll 4
    for (int ks = 0; ks < 36; ++ks) {
      int tap = ks >> 2;
      if (p.mode) tap = 8 - tap;                                      // data gradient: the correlation with the flipped window
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int toff = (ky * C64_PW + kx) * C64_PITCH + (ks & 3) * 32;
      const uint4 w0 = wl[(0 * 36 + ks) * 64 + lane], w1 = wl[(1 * 36 + ks) * 64 + lane];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const uint4 a = *(const uint4*)(patch + pbase[rb] + toff);
        acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w0), __builtin_bit_cast(bf16x8_t, a), acc[rb][0], 0, 0, 0);
        acc[rb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w1), __builtin_bit_cast(bf16x8_t, a), acc[rb][1], 0, 0, 0);
      }
    }
    __syncthreads();                                                  // the patch is dead: stage the tile (lane = pixel, register r = channel (r&3) + 8(r>>2) + 4fh)
    unsigned char* outt = patch;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int pidx = wave * 64 + rb * 32 + l31;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint2 pk;
          pk.x = (unsigned)f2bf(acc[rb][nb][4 * q + 0]) | ((unsigned)f2bf(acc[rb][nb][4 * q + 1]) << 16);
          pk.y = (unsigned)f2bf(acc[rb][nb][4 * q + 2]) | ((unsigned)f2bf(acc[rb][nb][4 * q + 3]) << 16);
          *(uint2*)(outt + pidx * C64_PITCH + (nb * 32 + 8 * q + 4 * fh) * 2) = pk;
        }
    }
    __syncthreads();
    bf16_t* out = (bf16_t*)p.out;
    float sa[8], sq[8];                                               // this thread's 8 channels (tid & 7) over its 8 pixels: column sums of the stored (rounded) tile
#pragma unroll
    for (int c = 0; c < 8; ++c) { sa[c] = 0.f; sq[c] = 0.f; }
    if (p.gate) {
      // data gradient into relu(bn(c)) (round 6): gate with the ReLU bits of these positions, store, accumulate (sum g, sum g * xhat) against the BatchNorm's input c --
      // the store phase MODE 3 of the tile kernels (conv_gemm_common.h), same arithmetic; layer1.conv2's gated data gradient ran on the two-stage 256 x 64 tile kernel at
      // 79 us where this kernel's forward takes 43
      float mu[8], is[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) { mu[c] = p.bn_mean[(tid & 7) * 8 + c]; is[c] = p.bn_invstd[(tid & 7) * 8 + c]; }
      uint4 xv[8]; unsigned bits[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = tid + u * 256, pidx = e >> 3, ch8 = e & 7;
        const int oy = by * C64_T + (pidx >> 4), ox = bx * C64_T + (pidx & 15);
        const bool in = oy < H && ox < W;
        const size_t pix = in ? (size_t)(b * H + oy) * W + ox : 0;
        xv[u] = *(const uint4*)((const bf16_t*)p.bn_x + pix * 64 + ch8 * 8);
        bits[u] = p.gate[pix * 8 + ch8];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = tid + u * 256, pidx = e >> 3, ch8 = e & 7;
        const int oy = by * C64_T + (pidx >> 4), ox = bx * C64_T + (pidx & 15);
        if (oy < H && ox < W) {
          uint4 v = *(const uint4*)(outt + pidx * C64_PITCH + ch8 * 16);
          const unsigned bb = bits[u];
          v.x &= ((unsigned)__builtin_amdgcn_sbfe(bb, 0, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(bb, 1, 1) & 0xffff0000u);
          v.y &= ((unsigned)__builtin_amdgcn_sbfe(bb, 2, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(bb, 3, 1) & 0xffff0000u);
          v.z &= ((unsigned)__builtin_amdgcn_sbfe(bb, 4, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(bb, 5, 1) & 0xffff0000u);
          v.w &= ((unsigned)__builtin_amdgcn_sbfe(bb, 6, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(bb, 7, 1) & 0xffff0000u);
          st16(out + ((size_t)(b * H + oy) * W + ox) * 64 + ch8 * 8, v);
          const unsigned wv[4] = {v.x, v.y, v.z, v.w}, xw[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float lo = __uint_as_float(wv[c] << 16), hi = __uint_as_float(wv[c] & 0xffff0000u);
            const float xl = __uint_as_float(xw[c] << 16), xh = __uint_as_float(xw[c] & 0xffff0000u);
            sa[2 * c] += lo; sq[2 * c] += lo * ((xl - mu[2 * c]) * is[2 * c]);
            sa[2 * c + 1] += hi; sq[2 * c + 1] += hi * ((xh - mu[2 * c + 1]) * is[2 * c + 1]);
          }
        }
      }
    } else
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + u * 256;
      const int pidx = e >> 3, ch8 = e & 7;
      const int oy = by * C64_T + (pidx >> 4), ox = bx * C64_T + (pidx & 15);
      if (oy < H && ox < W) {
        const uint4 v = *(const uint4*)(outt + pidx * C64_PITCH + ch8 * 16);
        st16(out + ((size_t)(b * H + oy) * W + ox) * 64 + ch8 * 8, v);
        if (p.stat_partial) {
          const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float lo = __uint_as_float(wv[c] << 16), hi = __uint_as_float(wv[c] & 0xffff0000u);
            sa[2 * c] += lo; sq[2 * c] += lo * lo; sa[2 * c + 1] += hi; sq[2 * c + 1] += hi * hi;
          }
        }
      }
    }
    if (p.stat_partial) {                                             // lanes 8 apart share the channel octet: three xor steps, then the four waves through the LDS
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { sa[c] += __shfl_xor(sa[c], o); sq[c] += __shfl_xor(sq[c], o); }
      if (lane < 8) {
#pragma unroll
        for (int c = 0; c < 8; ++c) { red[(wave * 2 + 0) * 64 + lane * 8 + c] = sa[c]; red[(wave * 2 + 1) * 64 + lane * 8 + c] = sq[c]; }
      }
      __syncthreads();
      if (tid < 128) {
        const int which = tid >> 6, ch = tid & 63;
        p.stat_partial[((size_t)tile * 2 + which) * 64 + ch] = red[(0 * 2 + which) * 64 + ch] + red[(1 * 2 + which) * 64 + ch] + red[(2 * 2 + which) * 64 + ch] + red[(3 * 2 + which) * 64 + ch];
      }
    }
    __syncthreads();                                                  // the next iteration overwrites the tile with its patch
  }
}

}  // namespace

bool slconv::c64k3_shape(int dtype, int KH, int KW, int stride, int pad, int dil, int Cin, int C1, int Cout, long long M) {
  return dtype == SL_BF16 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && dil == 1 && Cin == 64 && C1 == 64 && Cout == 64 && M >= 65536;
}

int slconv::launch_c64k3(ConvGemmParams& p, hipStream_t st) {
  const int ntiles = p.B * cdiv(p.Hs, C64_T) * cdiv(p.Ws, C64_T);
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv_c64k3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS); attr_set = true; }
  hipLaunchKernelGGL(conv_c64k3_kernel, dim3(ntiles < 256 ? ntiles : 256), dim3(256), C64_LDS, st, p, ntiles);
  SL_LAUNCH_CHECK("conv_c64k3_kernel");
  return 0;
}

namespace {
// ---------------------------------------------------------------------------------------------------------------
// Short-K 1x1 convs (64 / 128 / 256 input channels, stride 1, >= 65 536 pixels): out[M][N] = A[M][K] x W[N][K]^T is bound by the HBM traffic of
// `out` (and of the residual addend in the data gradient), not by the MFMAs: the tile kernels above spend a round per 256 x 256 tile on
// load -> 1..4 K-tiles -> store with nothing overlapping the store.  Here the PIXEL operand is stationary: a block owns 256 rows, each of its
// 8 waves keeps its 32 rows x K in registers as MFMA fragments (read once, straight from global memory) and walks over N in steps of 64
// columns; only the weight rows of a step (64 x K, from the L2) go through a two-slot LDS ring (LDS-DMA, rows XOR-swizzled for the fragment
// reads).  Per step a wave: issues its share of the next step's weight rows and this step's addend loads, runs 2 x K/16 MFMAs, stages its
// 32 x 64 result through its own LDS patch (row-major, rounded), waits for everything it has in flight (the stores of the PREVIOUS step have
// had a whole step to drain), adds / gates / stores 16 bytes per lane in full 128-byte lines, and meets the other waves at a barrier.
// BN statistic partials (one row per 256-row block, like the tile kernels) come from the rounded values in the store loop.
// SKEW: the two waves of a SIMD (w, w + 4) run HALF A STEP APART: while waves 0-3 multiply step s (matrix pipe), waves 4-7 add / gate / store
// step s - 1 (VALU + memory), and vice versa -- two barriers per step; waves 0-3 issue all the LDS-DMA.
template <int KS> struct SkGeom {
  static constexpr int RB = KS * 32;                                   // operand row bytes (K bf16)
  static constexpr int BSTEP = 64 * RB;                                // weight rows of one step
  static constexpr int STG_PITCH = 144, STG_WAVE = 32 * STG_PITCH;     // 32 rows x (128 B + pad) per wave
  static constexpr int OFF_STG = 2 * BSTEP, OFF_RED = OFF_STG + 8 * STG_WAVE;
  static constexpr int LDS = OFF_RED + 2 * 8 * 3 * 64 * (int)sizeof(float);      // red[parity][wave][sum, sq, sq2][64 columns] (sq2: the second BatchNorm of MODE 5's dual form)
};
template <int KS> __device__ __forceinline__ int sk_swz(int row) { return KS == 16 ? (row & 31) : (KS == 8 ? (row & 15) : ((row >> 1) & 7)); }
// sums over the lanes 8, 16 and 32 apart (the lanes of a wave that share lane & 7), without the LDS: one rotation inside the 16-lane rows, then the
// row swaps of gfx950 (permlane16_swap: odd rows of the first operand <-> even rows of the second; permlane32_swap: upper half <-> lower half)
__device__ __forceinline__ float sk_sum_8_16_32(float v) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));       // row_ror:8
  u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r.x) + __uint_as_float(r.y);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

#ifndef SL_SK_ABL
#define SL_SK_ABL 0      // ablation builds (tools/sk_ablation.sh; never in the product): 1 = no global stores, 2 = no addend / BN-input loads, 4 = no column-sum arithmetic, 8 = no MFMAs
#endif
template <int KS, int MODE, bool SKEW>       // MODE 1: store (+ statistics), 2: + (bit-gated) addend, 5: + addend, result gated with the ReLU bits of its own positions + BN-backward column sums (ConvGemmParams::gate)
__global__ __launch_bounds__(512, (KS == 4 && MODE != 5) ? 4 : 2) void conv_gemm_sk_kernel(ConvGemmParams p) {      // MODE 5 holds 81 KiB of LDS at KS = 4: one block per CU whatever the registers allow
  using G = SkGeom<KS>;
  using T = bf16_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                                           // SKEW: 0 leads, 1 is half a step behind
  const int l31 = lane & 31, fh = lane >> 5;
  const int bm = blockIdx.x;
  const int NS = p.N / 64;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  // weight rows of a step: 2 KS wave-instructions of 1 KiB shared by the issuing waves; the swizzle is applied to the SOURCE chunk (the LDS side of LDS-DMA is lane-linear)
  constexpr int NWI = SKEW ? 4 : 8, NI = 2 * KS / NWI, LPR = 2 * KS, RPI = 64 / LPR;
  const int iw = SKEW ? (wave & 3) : wave;
  const bool issuer = !SKEW || grp == 0;
  const unsigned char* bsrc[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = (iw * NI + j) * RPI + lane / LPR, pos = lane % LPR;
    bsrc[j] = (const unsigned char*)p.wt + (size_t)row * G::RB + ((pos ^ sk_swz<KS>(row)) << 4);
  }
  auto issueB = [&](int s, int buf) {
#pragma unroll
    for (int j = 0; j < NI; ++j) glds16_asm(bsrc[j] + (size_t)s * G::BSTEP, lds_base + buf * G::BSTEP + (iw * NI + j) * 1024);
  };
  // (Round 6, measured and left out: walking the column steps in an order rotated by the block index -- so that the resident blocks do not all touch the same 128-byte
  // column slice of their rows at the same time -- changed nothing: 1.842 vs 1.844 ms per step for the family, profiles/r6_ab_sk_rotation.txt.  No channel camping here.)
  auto colstep = [&](int i) { return i; };
  if (issuer) issueB(colstep(0), 0);
  uint4 a[KS];
  {
    const unsigned char* arow = (const unsigned char*)p.src1 + ((size_t)bm * 256 + wave * 32 + l31) * G::RB + fh * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = *(const uint4*)(arow + ks * 32);
  }
  int foff[KS];
  {
    const int x = sk_swz<KS>(l31);                                     // rows l31 and 32 + l31 of the step share the swizzle (all three patterns have period <= 32)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) foff[ks] = l31 * G::RB + (((2 * ks + fh) ^ x) << 4);
  }
  unsigned char* stg = smem + G::OFF_STG + wave * G::STG_WAVE;
  float* red = (float*)(smem + G::OFF_RED);
  const int srow = lane >> 3, sch = lane & 7;                          // store phase: row it * 8 + srow, 16-byte chunk sch of the wave's 32 x 64 patch (a full 128-byte line per row)
  const size_t orow = (size_t)bm * 256 + wave * 32 + srow;
  // addend rows of this lane's four result rows (the same in every step): the result row itself, or -- addend_half -- row (b, y/2, x/2) of the half-resolution
  // tensor at even (y, x) and none (-1) elsewhere: dx of a 1x1 stride-2 conv is zero at the odd positions (resnet.py:109-110 downsample backward)
  long long arow[4];
  if constexpr (MODE == 2 || MODE == 5) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long long m = (long long)orow + it * 8;
      arow[it] = m;
      if (p.addend_half) {
        const int hw = p.Hd * p.Wd;
        const int b = (int)(m / hw), rem = (int)(m - (long long)b * hw), y = rem / p.Wd, x = rem - y * p.Wd;
        arow[it] = ((y | x) & 1) ? -1 : ((long long)b * (p.Hd >> 1) + (y >> 1)) * (p.Wd >> 1) + (x >> 1);
      }
    }
  }
  uint4 addv[4], cxv[4], cxv2[4];                                      // MODE 5: cxv = the BN input c of the result's positions (cxv2: the second BatchNorm's, dual form)
  float bmu[8], bis[8], bmu2[8], bis2[8];
  const bool dual = MODE == 5 && p.bn_x2 != nullptr;                   // wave-uniform
  // the gate bytes of the block's 256 rows (N / 8 per row, contiguous over the rows) are copied to the LDS once: read step by step from global memory, each step would
  // pull 8 useful bytes out of every row's line, and 256 lines per step do not survive in the 32 KiB L1 next to the addend stream (measured: 58 -> 73 us on 1024 -> 256)
  const int mpitch = p.N / 8 + 16;
  unsigned char* msk = smem + (MODE == 5 ? G::LDS : G::OFF_RED);       // MODE 5 keeps the statistic partials too: its gate bytes sit behind them
  if ((MODE == 2 && p.addend_mask) || MODE == 5) {
    const int cpr = p.N / 128;                                         // 16-byte chunks per row
    const unsigned char* src = (MODE == 5 ? p.gate : p.addend_mask) + (size_t)bm * 256 * (p.N / 8);
    for (int e = tid; e < 256 * cpr; e += 512) {
      const int row = e / cpr, c = e - row * cpr;
      *(uint4*)(msk + row * mpitch + c * 16) = *(const uint4*)(src + (size_t)e * 16);
    }
  }

  auto multiply = [&](int i) {
    const int cur = i & 1, s = colstep(i);
    if (issuer && i + 1 < NS) issueB(colstep(i + 1), cur ^ 1);
    if constexpr (MODE == 2 || MODE == 5) {
      const int ncol = s * 64 + sch * 8;
#pragma unroll
      for (int it = 0; it < 4; ++it) addv[it] = (arow[it] >= 0 && !(SL_SK_ABL & 2)) ? *(const uint4*)((const T*)p.addend + arow[it] * p.N + ncol) : make_uint4(0, 0, 0, 0);
      if constexpr (MODE == 5) {
#pragma unroll
        for (int it = 0; it < 4; ++it) cxv[it] = (SL_SK_ABL & 2) ? make_uint4(0, 0, 0, 0) : *(const uint4*)((const T*)p.bn_x + (orow + it * 8) * p.N + ncol);
#pragma unroll
        for (int e = 0; e < 8; ++e) { bmu[e] = p.bn_mean[ncol + e]; bis[e] = p.bn_invstd[ncol + e]; }
        if (dual) {
#pragma unroll
          for (int it = 0; it < 4; ++it) cxv2[it] = *(const uint4*)((const T*)p.bn_x2 + (orow + it * 8) * p.N + ncol);
#pragma unroll
          for (int e = 0; e < 8; ++e) { bmu2[e] = p.bn_mean2[ncol + e]; bis2[e] = p.bn_invstd2[ncol + e]; }
        }
      }
    }
    f32x16_t acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned char* bb = smem + cur * G::BSTEP;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint4 b0 = *(const uint4*)(bb + foff[ks]), b1 = *(const uint4*)(bb + 32 * G::RB + foff[ks]);
      if constexpr (!(SL_SK_ABL & 8)) {
      Mma<T>::run(b0, a[ks], acc[0]);
      Mma<T>::run(b1, a[ks], acc[1]);
      } else { acc[0][0] += __uint_as_float(b0.x); acc[1][0] += __uint_as_float(b1.x); }
    }
    // D layout: lane = pixel (l31), register r = column (r & 3) + 8 (r >> 2) + 4 fh of column half j
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint2 v;
        v.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[j][4 * q + 0], acc[j][4 * q + 1]}, bf16x2_t));
        v.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[j][4 * q + 2], acc[j][4 * q + 3]}, bf16x2_t));
        *(uint2*)(stg + l31 * G::STG_PITCH + 64 * j + 16 * q + 8 * fh) = v;
      }
  };

  auto store = [&](int i) {
    const int cur = i & 1, s = colstep(i);
    const int ncol = s * 64 + sch * 8;
    wait_vmcnt<0>();                                                   // the next step's weight rows, this step's addend, the previous step's stores
    float sa[8], sq[8], sq2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sa[e] = 0.f; sq[e] = 0.f; sq2[e] = 0.f; }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const uint4 raw = *(const uint4*)(stg + (it * 8 + srow) * G::STG_PITCH + sch * 16);
      T* o = (T*)p.out + (orow + it * 8) * p.N + ncol;
      if constexpr (MODE == 1) {
        st16(o, raw);
        if (p.stat_partial) {
          const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float lo = __uint_as_float(w[c] << 16), hi = __uint_as_float(w[c] & 0xffff0000u);
            sa[2 * c] += lo; sq[2 * c] += lo * lo; sa[2 * c + 1] += hi; sq[2 * c + 1] += hi * hi;
          }
        }
      } else {
        uint4 ad = addv[it];
        if (MODE == 2 && p.addend_mask) {
          const unsigned b = msk[(wave * 32 + it * 8 + srow) * mpitch + s * 8 + sch];
          ad.x &= ((unsigned)__builtin_amdgcn_sbfe(b, 0, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 1, 1) & 0xffff0000u);
          ad.y &= ((unsigned)__builtin_amdgcn_sbfe(b, 2, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 3, 1) & 0xffff0000u);
          ad.z &= ((unsigned)__builtin_amdgcn_sbfe(b, 4, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 5, 1) & 0xffff0000u);
          ad.w &= ((unsigned)__builtin_amdgcn_sbfe(b, 6, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 7, 1) & 0xffff0000u);
        }
        const unsigned rw[4] = {raw.x, raw.y, raw.z, raw.w}, aw[4] = {ad.x, ad.y, ad.z, ad.w};
        unsigned ow[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x2_t v = (f32x2_t){__uint_as_float(rw[c] << 16) + __uint_as_float(aw[c] << 16), __uint_as_float(rw[c] & 0xffff0000u) + __uint_as_float(aw[c] & 0xffff0000u)};
          ow[c] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
        }
        if constexpr (MODE == 5) {
          // gate the ROUNDED sum with the ReLU bits of its own positions (what the separate passes would see), then the column sums of g and g * xhat
          const unsigned b = msk[(wave * 32 + it * 8 + srow) * mpitch + s * 8 + sch];
          ow[0] &= ((unsigned)__builtin_amdgcn_sbfe(b, 0, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 1, 1) & 0xffff0000u);
          ow[1] &= ((unsigned)__builtin_amdgcn_sbfe(b, 2, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 3, 1) & 0xffff0000u);
          ow[2] &= ((unsigned)__builtin_amdgcn_sbfe(b, 4, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 5, 1) & 0xffff0000u);
          ow[3] &= ((unsigned)__builtin_amdgcn_sbfe(b, 6, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(b, 7, 1) & 0xffff0000u);
          const unsigned xw[4] = {cxv[it].x, cxv[it].y, cxv[it].z, cxv[it].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if constexpr (SL_SK_ABL & 4) { sa[2 * c] += __uint_as_float(ow[c] ^ xw[c]); continue; }
            const float glo = __uint_as_float(ow[c] << 16), ghi = __uint_as_float(ow[c] & 0xffff0000u);
            const float xlo = __uint_as_float(xw[c] << 16), xhi = __uint_as_float(xw[c] & 0xffff0000u);
            sa[2 * c] += glo; sq[2 * c] += glo * ((xlo - bmu[2 * c]) * bis[2 * c]);
            sa[2 * c + 1] += ghi; sq[2 * c + 1] += ghi * ((xhi - bmu[2 * c + 1]) * bis[2 * c + 1]);
          }
          if (dual) {
            const unsigned yw[4] = {cxv2[it].x, cxv2[it].y, cxv2[it].z, cxv2[it].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float glo = __uint_as_float(ow[c] << 16), ghi = __uint_as_float(ow[c] & 0xffff0000u);
              const float ylo = __uint_as_float(yw[c] << 16), yhi = __uint_as_float(yw[c] & 0xffff0000u);
              sq2[2 * c] += glo * ((ylo - bmu2[2 * c]) * bis2[2 * c]);
              sq2[2 * c + 1] += ghi * ((yhi - bmu2[2 * c + 1]) * bis2[2 * c + 1]);
            }
          }
        }
        if constexpr (SL_SK_ABL & 1) { if (ow[0] == 0x12345678u && ow[1] == 0x9abcdef0u) st16(o, make_uint4(ow[0], ow[1], ow[2], ow[3])); }
        else st16(o, make_uint4(ow[0], ow[1], ow[2], ow[3]));
      }
    }
    if ((MODE == 1 || MODE == 5) && p.stat_partial) {                  // lanes 8 apart share the column octet
#pragma unroll
      for (int e = 0; e < 8; ++e) { sa[e] = sk_sum_8_16_32(sa[e]); sq[e] = sk_sum_8_16_32(sq[e]); }
      if (dual) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sq2[e] = sk_sum_8_16_32(sq2[e]);
      }
      if (lane < 8) {
        float* r0 = red + ((cur * 8 + wave) * 3) * 64 + sch * 8;
        *(float4*)(r0) = make_float4(sa[0], sa[1], sa[2], sa[3]); *(float4*)(r0 + 4) = make_float4(sa[4], sa[5], sa[6], sa[7]);
        *(float4*)(r0 + 64) = make_float4(sq[0], sq[1], sq[2], sq[3]); *(float4*)(r0 + 68) = make_float4(sq[4], sq[5], sq[6], sq[7]);
        if (dual) { *(float4*)(r0 + 128) = make_float4(sq2[0], sq2[1], sq2[2], sq2[3]); *(float4*)(r0 + 132) = make_float4(sq2[4], sq2[5], sq2[6], sq2[7]); }
      }
    }
  };
  auto finalize = [&](int i, int t0) {                                 // 128 (dual: 192) threads from t0 on, after the barrier behind the last store phase of step i
    const int s = colstep(i);
    if ((MODE == 1 || MODE == 5) && p.stat_partial && tid >= t0 && tid < t0 + (dual ? 192 : 128)) {
      const int which = (tid - t0) >> 6, col = tid & 63;               // 0: sum g, 1: sum g * xhat, 2: sum g * xhat2
      const float* r0 = red + ((i & 1) * 24 + which) * 64 + col;
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += r0[k * 192];
      if (which < 2) p.stat_partial[((size_t)bm * 2 + which) * p.N + s * 64 + col] = t;
      if (dual && which != 1) p.stat_partial2[((size_t)bm * 2 + (which >> 1)) * p.N + s * 64 + col] = t;      // the second BatchNorm's partials repeat sum g
    }
  };

  // raw barriers: __syncthreads() would also wait for the stores in flight.  LDS writes (statistic partials; the staging patch is wave-private) are drained explicitly.
  auto bar = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  wait_vmcnt<0>();
  bar();
  if constexpr (!SKEW) {
#pragma unroll 1
    for (int s = 0; s < NS; ++s) {
      multiply(s);
      store(s);
      bar();
      finalize(s, 0);
    }
  } else if (grp == 0) {
#pragma unroll 1
    for (int s = 0; s < NS; ++s) {
      multiply(s);
      bar();
      store(s);
      bar();
    }
    bar();
  } else {
    bar();
#pragma unroll 1
    for (int s = 0; s < NS; ++s) {
      multiply(s);
      bar();
      store(s);
      bar();
      finalize(s, 256);
    }
  }
}

}  // namespace

bool slconv::sk_shape(int dtype, int KH, int KW, int stride, int pad, int Cin, int C1, int N, long long M) {
  return dtype == SL_BF16 && KH == 1 && KW == 1 && stride == 1 && pad == 0 && C1 == Cin && (Cin == 64 || Cin == 128 || Cin == 256) &&
         N % 64 == 0 && M >= 65536 && M % 256 == 0;
}

namespace {
template <int KS, int MODE, bool SKEW>
int launch_sk_t(ConvGemmParams& p, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv_gemm_sk_kernel<KS, MODE, SKEW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
  const int lds = MODE == 5 ? SkGeom<KS>::LDS + 256 * (p.N / 8 + 16)
                            : (MODE == 2 && p.addend_mask ? SkGeom<KS>::OFF_RED + 256 * (p.N / 8 + 16) : SkGeom<KS>::LDS);      // statistic partials and / or the block's gate bytes behind the staging patches
  hipLaunchKernelGGL((conv_gemm_sk_kernel<KS, MODE, SKEW>), dim3(p.M / 256), dim3(512), lds, st, p);
  SL_LAUNCH_CHECK("conv_gemm_sk_kernel");
  return 0;
}
template <int MODE, bool SKEW>
int launch_sk_k(ConvGemmParams& p, hipStream_t st) {
  return p.C1 == 256 ? launch_sk_t<16, MODE, SKEW>(p, st) : (p.C1 == 128 ? launch_sk_t<8, MODE, SKEW>(p, st) : launch_sk_t<4, MODE, SKEW>(p, st));
}
}  // namespace

int slconv::launch_sk(ConvGemmParams& p, hipStream_t st) {
  // measured (tools/sk_time.sh): half-a-step-apart wave groups pay where the store phase carries the statistics (256 -> 1024 forward: 53 -> 49 us) and cost where it is pure
  // memory traffic (data gradient 1024 -> 256: 41 -> 47 us, with addend 58 -> 67 us).  bit 0: statistics, bit 1: plain store, bit 2: addend
  constexpr int skew = 1;
  p.gridM = p.M / 256; p.gridN = 1;
  // MODE 5 (gate + BatchNorm-backward column sums in the store loop) stays un-skewed: round 6's ablation (profiles/r6_sk_ablation.txt) shows the launch bound by its own
  // instruction streams, not by HBM -- 80 of 115 us remain with NO global loads or stores; MFMAs and column-sum arithmetic are worth 26 us each -- but putting one wave
  // group's MFMAs under the other's store loop (SKEW) measured 126 vs 117 us (profiles/r6_ab_sk_skew5.txt): the second barrier per step costs more than the overlap buys.
  if (p.gate) return launch_sk_k<5, false>(p, st);
  if (p.addend) return (skew & 4) ? launch_sk_k<2, true>(p, st) : launch_sk_k<2, false>(p, st);
  return (skew & (p.stat_partial ? 1 : 2)) ? launch_sk_k<1, true>(p, st) : launch_sk_k<1, false>(p, st);
}

