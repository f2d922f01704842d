// Half-tile kernel (family 5, "p8"), 3x3 patch kernel (family 8, "p9") and the split-K finish of the implicit-GEMM convolution.  See conv_gemm_common.h.
#include "conv_gemm_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// v5 (bf16, N % 256 == 0): 256x256 tile, K-tile of 64 elements = 128-BYTE operand rows, LDS = 2 K-tiles x 4 half-tile slots
// {A0, A1, B0, B1} of 16 KiB (128 rows x 128 B).  tools/micro/glds_bw.hip: the LDS-DMA path delivers 30-50 % more bytes/s when
// each row request is a full 128-byte line than with the 64-byte rows of the v4 ring, and v4 sits exactly on that limit.
// A K-tile is computed as 4 phases, one output quadrant each -- (A0,B0) (A0,B1) (A1,B1) (A1,B0) -- so a slot is free again
// after at most two phases and is refilled with the same half of the K-tile two steps ahead:
//   P0: issue B0(i+2)            P1: issue A0(i+2)            P3: issue A1(i+2), B1(i+2)       (A0 and B0 have three slots, mod 3)
// The A fragments of a half (8 x 16 B per lane) stay in registers for its two phases and are refilled in place (A1 during P1, the
// next K-tile's A0 during P3); B fragments stream through a 4-deep register ring, three k-steps ahead (P3 re-uses the B0 fragments
// and P2 the B1 fragments of P1 from registers: 24 LDS fragment reads per 32 MFMAs).  Two counted waits and two barriers per K-tile.
// Wave (wm, wn) of the 2 x 4 grid owns rows {h*128 + wm*64 ..+63} and columns {h*128 + wn*32 ..+31} of both halves h.
// split-K: the wave's accumulators (half-tile layout: tile row = half*128 + wm*64 + i2*32 + lane&31, column = j*128 + wn*32 + 8q + 4*(lane>>5) .. +3) as fp32 to
// ws [part][M][N]; 32 16-byte stores per lane.  tile16: row block bm is a 16 x 16-pixel tile.
__device__ __forceinline__ void conv_store_partial(const ConvGemmParams& p, f32x16_t (&acc)[4][2], int part, int bm, int bn, int wm, int wn, int lane) {
  const int l31 = lane & 31, fh = lane >> 5;
  float* base = p.ws + (size_t)part * p.M * p.N + bn * 256 + wn * 32 + 4 * fh;
  int tbase = 0;
  if (p.tile16) { const int tx = p.Wd >> 4, ty = p.Hd >> 4; tbase = ((bm / (tx * ty)) * p.Hd + ((bm / tx) % ty) * 16) * p.Wd + (bm % tx) * 16; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int R = (i >> 1) * 128 + wm * 64 + (i & 1) * 32 + l31;
    const int m = p.tile16 ? tbase + (R >> 4) * p.Wd + (R & 15) : bm * 256 + R;
    if (m < p.M) {
      float* row = base + (size_t)m * p.N;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(float4*)(row + j * 128 + 8 * q) = make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
    }
  }
}

// out = act((sum over parts + pre_addend) * scale + bias + addend): the epilogue of the generic store phase (same operation order) behind a split-K launch
template <typename T>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(ConvGemmParams p) {
  constexpr int EPC = 16 / sizeof(T);
  const size_t nvec = (size_t)p.M * p.N / EPC, slab = (size_t)p.M * p.N;
  const int nvc = p.N / EPC;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (size_t)gridDim.x * 256) {
    float a[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) a[e] = 0.f;
    for (int s = 0; s < p.ksplit; ++s) {
      const float4* src = (const float4*)(p.ws + s * slab + v * EPC);
#pragma unroll
      for (int h = 0; h < EPC / 4; ++h) { const float4 t = src[h]; a[4 * h + 0] += t.x; a[4 * h + 1] += t.y; a[4 * h + 2] += t.z; a[4 * h + 3] += t.w; }
    }
    // the tile kernels round the accumulators to T when they stage the tile and apply the epilogue to the rounded values: the same here, so that a layer gives the
    // same result whichever way it is dispatched (up to the order of the K sum)
    float r[EPC];
    unpack16<T>(pack16<T>(a), r);
    const int c = (int)(v % nvc) * EPC;
    if (p.pre_addend) {
      float t[EPC];
      unpack16<T>(((const uint4*)p.pre_addend)[v], t);
#pragma unroll
      for (int e = 0; e < EPC; ++e) r[e] += t[e];
    }
    if (p.bias || p.scale) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) r[e] = r[e] * (p.scale ? p.scale[c + e] : 1.f) + (p.bias ? p.bias[c + e] : 0.f);
    }
    if (p.addend) {
      float t[EPC];
      unpack16<T>(((const uint4*)p.addend)[v], t);
#pragma unroll
      for (int e = 0; e < EPC; ++e) r[e] += t[e];
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) r[e] = r[e] > 0.f ? r[e] : 0.f;
    }
    ((uint4*)p.out)[v] = pack16<T>(r);
  }
}

constexpr int P8_SLOT = 128 * 128;
constexpr int P8_RING = 10 * P8_SLOT;         // A0 x3, A1 x2, B1 x2, B0 x3 = the whole 160 KiB
constexpr int P8_LDS = P8_RING;
static_assert(EpiGeom<bf16_t, 256, 256, 2, 4, true>::TILE_BYTES / 2 <= 6 * P8_SLOT, "the half-tile staging area must fit below the four slots of a prefetched K-tile");

// PERSISTENT: the grid is min(tiles, 256) blocks (one per CU, 160 KiB of LDS each) and a block walks over tiles b, b + grid, ...  tools/p8_trace.py (s_memtime per
// block) showed where a one-tile block spends its time on the short-K layers: 512 -> 2048 forward 15 % prologue (address set-up + the HBM latency of the first K-tile)
// / 61 % main loop / 24 % epilogue, 2048 -> 512 data gradient with gated addend 15 / 46 / 40, 1024 -> 2048 10 / 75 / 15, 3x3 512 -> 512 6 / 90 / 4, plus ~800 ticks
// between two blocks on a CU.  Here the NEXT tile's set-up and first K-tile (LDS-DMA into four slots) are issued right after the main loop, so that latency runs
// under the epilogue; the epilogue stages the tile in two half-tile passes in the six slots the prefetch leaves free.  Physical slot order (16 KiB each):
//   0,1 = A0 of K-tiles 1,2 (mod 3)   2 = A1 odd   3 = B1 odd   4,5 = B0 of K-tiles 1,2   | 6 = A0 of K-tile 0   7 = A1 even   8 = B1 even   9 = B0 of K-tile 0
// vmcnt is in order over loads AND stores: the first wait of the next tile (all but the 8 LDS-DMA of its second K-tile) also covers the epilogue's stores, which by
// then have had the set-up of the second K-tile to drain.
__device__ __forceinline__ int p8_slot_a0(int j) { return (j == 0 ? 6 : j - 1) * P8_SLOT; }
__device__ __forceinline__ int p8_slot_a1(int par) { return (par ? 2 : 7) * P8_SLOT; }
__device__ __forceinline__ int p8_slot_b1(int par) { return (par ? 3 : 8) * P8_SLOT; }
__device__ __forceinline__ int p8_slot_b0(int j) { return (j == 0 ? 9 : j + 3) * P8_SLOT; }

template <int EPI>       // 0: the store phases without MODE 3, 1: with the gated-statistics store phase (MODE 3), 2: with the affine store phases (inference convs, biased Linears), 3: MODE 3 + bias only
__global__ __launch_bounds__(512) void conv_gemm_p8_kernel(ConvGemmParams p) {
  constexpr bool GATE = EPI == 1, AFF = EPI == 2, GATEB = EPI == 3;
  using T = bf16_t;
  constexpr int BM = 256, BN = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
  if (p.trace) tr0 = __builtin_amdgcn_s_memtime();
  const int ntiles = p.gridM * p.gridN;
  const int taps = p.KH * p.KW;
  const int CT = p.C1 + p.C2;
  const int nk = taps * (CT / 64);
  const int sgn = p.mode == 0 ? 1 : -1;

  // ---- load side: instruction j of this wave fills rows wave*16 + j*8 + (lane>>3) of a half-tile slot, 16 B per lane
  const int lr = lane >> 3, lpos = lane & 7;
  int rbase[2][2]; unsigned vmask[2][2]; int rsw[2];
  const unsigned char* wptr[2][2];
  const size_t wpitch = (size_t)taps * CT * sizeof(T);
#pragma unroll
  for (int j = 0; j < 2; ++j) rsw[j] = (lpos ^ (((j * 8 + lr) >> 1) & 7)) * 16;
  auto tile_of = [&](int t, int& bm, int& bn) {                        // XCD-aware order: the tiles a CU group of one XCD works on concurrently share their A rows
    const int q = ntiles >> 3, r = ntiles & 7, xcd = t & 7, idx = t >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bm = bid / p.gridN; bn = bid % p.gridN;
  };

  auto setup = [&](int bm, int bn) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = h * 128 + wave * 16 + j * 8 + lr;
        const int m = bm * BM + row;
        rbase[h][j] = 0; vmask[h][j] = 0;
        if (m < p.M) {
          const int b = m / (p.Hd * p.Wd), rem = m - b * (p.Hd * p.Wd);
          const int yd = rem / p.Wd, xd = rem - yd * p.Wd;
          const int ry = p.mode == 0 ? yd * p.stride - p.pad : yd + p.pad;
          const int rx = p.mode == 0 ? xd * p.stride - p.pad : xd + p.pad;
          rbase[h][j] = (b * p.Hs + ry) * p.Ws + rx;
          unsigned mk = 0;
          for (int t = 0; t < taps; ++t) {
            const int ky = t / p.KW, kx = t - ky * p.KW;
            const int ys = ry + sgn * ky * p.dil, xs = rx + sgn * kx * p.dil;
            if ((unsigned)ys < (unsigned)p.Hs && (unsigned)xs < (unsigned)p.Ws) mk |= 1u << t;
          }
          vmask[h][j] = mk;
        }
        wptr[h][j] = (const unsigned char*)p.wt + (size_t)(bn * BN + row) * wpitch + rsw[j];
      }
  };
  const unsigned char* zsrc = g_zero_page + lpos * 16;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  auto issueA = [&](int h, int tap, int ct, int slot_off) {
    const unsigned dst = lds_base + slot_off + wave * 2048;
    const int c0 = ct * 64;
    const unsigned char* base; unsigned pitchb;
    if (c0 < p.C1) { base = (const unsigned char*)p.src1 + (size_t)c0 * sizeof(T); pitchb = p.C1 * (unsigned)sizeof(T); }
    else           { base = (const unsigned char*)p.src2 + (size_t)(c0 - p.C1) * sizeof(T); pitchb = p.C2 * (unsigned)sizeof(T); }
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int delta = sgn * (ky * p.dil * p.Ws + kx * p.dil);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = (vmask[h][j] >> tap) & 1u;
      const unsigned char* src = base + (size_t)((unsigned)(rbase[h][j] + delta)) * pitchb + rsw[j];
      glds16_asm(ok ? src : zsrc, dst + j * 1024);
    }
  };
  auto issueB = [&](int h, int tap, int ct, int slot_off) {
    const unsigned dst = lds_base + slot_off + wave * 2048;
    const size_t koff = ((size_t)tap * CT + ct * 64) * sizeof(T);
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16_asm(wptr[h][j] + koff, dst + j * 1024);
  };
  auto adv = [&](int& tap, int& ct) { if (++tap == taps) { tap = 0; ++ct; } };
  auto issue_first = [&]() { issueB(0, 0, 0, p8_slot_b0(0)); issueA(0, 0, 0, p8_slot_a0(0)); issueA(1, 0, 0, p8_slot_a1(0)); issueB(1, 0, 0, p8_slot_b1(0)); };

  // ---- fragment side: lane (l31, fh) reads row base + l31, 16-byte chunk 2*ks + fh (swizzled) of a slot
  const int l31 = lane & 31, fh = lane >> 5;
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = l31 * 128 + (((2 * ks + fh) ^ ((l31 >> 1) & 7)) << 4);
  const unsigned char* fa = smem + wm * (64 * 128);
  const unsigned char* fb = smem + wn * (32 * 128);
  auto ldA = [&](int slot_off, int i2, int ks) { return *(const uint4*)(fa + slot_off + i2 * 4096 + foff[ks]); };
  auto ldB = [&](int slot_off, int ks) { return *(const uint4*)(fb + slot_off + foff[ks]); };

  int bm, bn;
  tile_of(blockIdx.x, bm, bn);
  setup(bm, bn);
  issue_first();
  int younger = 0;                               // vector-memory instructions issued after the current K-tile 0 was requested (see the first wait)
  const bool counted = p.flags & 1;
#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    f32x16_t acc[4][2];                         // [half*2 + 32-row block][column half]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue.  Global issue order is K-tile by K-tile: B0(s), A0(s), A1(s), B1(s) (K-tile 0 is already in flight); the loop continues it with
    // B0(i+2) in P0(i), A0(i+2) in P1(i), A1(i+2) and B1(i+2) in P3(i).
    int tap2 = 0, ct2 = 0;                      // K-tile i+2 (after the prologue)
    adv(tap2, ct2);
    if (nk > 1) {
      issueB(0, tap2, ct2, p8_slot_b0(1)); issueA(0, tap2, ct2, p8_slot_a0(1)); issueA(1, tap2, ct2, p8_slot_a1(1)); issueB(1, tap2, ct2, p8_slot_b1(1));
      // K-tile 0 must have landed.  Younger than its loads: the 8 LDS-DMA just issued and the `younger` loads / stores of the previous tile's epilogue, which need not be waited for
      switch (younger) { case 16: wait_vmcnt<24>(); break; case 17: wait_vmcnt<25>(); break; case 32: wait_vmcnt<40>(); break; case 48: wait_vmcnt<56>(); break; case 49: wait_vmcnt<57>(); break; default: wait_vmcnt<8>(); }
    } else wait_vmcnt<0>();
    adv(tap2, ct2);
    __builtin_amdgcn_s_barrier();
    if (p.trace && tile == (int)blockIdx.x) tr1 = __builtin_amdgcn_s_memtime();
    // B fragments live in two register sets that are loaded IN PLACE four k-steps before their first use: b0k (B0 half: used in P0 and P3, reloaded for the next K-tile
    // right after its P3 use) and b1k (B1 half: loaded in P0, used in P1 and P2).  (An earlier form streamed them through a 4-deep ring and copied them into keep registers:
    // 32 v_mov per K-tile next to 32 MFMAs.)
    uint4 a[4][2], b0k[4], b1k[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { a[ks][0] = ldA(p8_slot_a0(0), 0, ks); a[ks][1] = ldA(p8_slot_a0(0), 1, ks); b0k[ks] = ldB(p8_slot_b0(0), ks); }

    int s3 = 0;                                 // i mod 3
    for (int i = 0; i < nk; ++i) {
      const int par = i & 1;
      const int s3n = s3 == 2 ? 0 : s3 + 1, s3nn = s3 == 0 ? 2 : s3 - 1;                                   // (i+1) % 3, (i+2) % 3
      const int b0nxt = p8_slot_b0(s3n), b0nn = p8_slot_b0(s3nn);                                         // B0 slots of K-tiles i+1, i+2
      const int a1cur = p8_slot_a1(par), b1cur = p8_slot_b1(par), a0nxt = p8_slot_a0(s3n), a0nn = p8_slot_a0(s3nn);
      const bool more2 = i + 2 < nk;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int ph = q >> 2, ks = q & 3;
        const int ih = ph >> 1, jh = (ph == 1 || ph == 2) ? 1 : 0;
        // the two waves of a SIMD (w, w + 4) run the same phase; their address/issue sections are placed two k-steps apart so that
        // one wave's MFMAs cover the other's VALU + LDS-DMA issue (same per-wave issue order, so the vmcnt arithmetic is unchanged)
        if ((ks == 0 || ks == 2) && more2 && (ks == 2) == (wm == 1)) {
          if (ph == 0) issueB(0, tap2, ct2, b0nn);
          if (ph == 1) issueA(0, tap2, ct2, a0nn);
          if (ph == 3) { issueA(1, tap2, ct2, a1cur); issueB(1, tap2, ct2, b1cur); }
        }
        if (ph == 0) b1k[ks] = ldB(b1cur, ks);                                           // B1 of this K-tile, used from P1 on
        const uint4 bq = (ph == 0 || ph == 3) ? b0k[ks] : b1k[ks];
        Mma<T>::run(bq, a[ks][0], acc[ih * 2 + 0][jh]);
        Mma<T>::run(bq, a[ks][1], acc[ih * 2 + 1][jh]);
        if (ph == 1) { a[ks][0] = ldA(a1cur, 0, ks); a[ks][1] = ldA(a1cur, 1, ks); }   // A1 of this K-tile
        if (ph == 3) { a[ks][0] = ldA(a0nxt, 0, ks); a[ks][1] = ldA(a0nxt, 1, ks); b0k[ks] = ldB(b0nxt, ks); }   // A0 and B0 of the next K-tile
        if (ks == 3) {
          // end of P2: B0(i+1), A0(i+1) must have landed (read in P3); end of P3: A1(i+1), B1(i+1) (read from P0(i+1) on)
          if (ph == 2) { if (more2) wait_vmcnt<8>(); else if (i + 1 < nk) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
          if (ph == 3) { if (more2) wait_vmcnt<8>(); else wait_vmcnt<0>(); }
          if (ph >= 2) __builtin_amdgcn_s_barrier();
        }
      }
      adv(tap2, ct2);
      s3 = s3 == 2 ? 0 : s3 + 1;
    }
    if (p.trace && tile == (int)blockIdx.x) tr2 = __builtin_amdgcn_s_memtime();
    if constexpr (GATEB) {
      // the per-column bias joins the fp32 accumulators (accumulator block [.][j], registers 4 q + e = columns j 128 + wn 32 + 8 q + 4 (lane >> 5) + e, as epi_stage_acc stages
      // them): loaded HERE, in front of the next tile's first LDS-DMA, so the counted waits of the next prologue are unchanged
      const int fh = lane >> 5;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b4 = *(const float4*)(p.bias + bn * BN + j * (BN / 2) + wn * (BN / 8) + 8 * q + 4 * fh);
#pragma unroll
          for (int i = 0; i < 4; ++i) { acc[i][j][4 * q + 0] += b4.x; acc[i][j][4 * q + 1] += b4.y; acc[i][j][4 * q + 2] += b4.z; acc[i][j][4 * q + 3] += b4.w; }
        }
    }
    lds_barrier();
    // every slot is idle: the next tile's row map, weight rows and first K-tile go out now and land under the epilogue (slots 6-9; the staging passes use 0-5)
    const int cbm = bm, cbn = bn;
    if (tile + (int)gridDim.x < ntiles) {
      tile_of(tile + gridDim.x, bm, bn);
      setup(bm, bn);
      issue_first();
    }
    younger = conv_epilogue_lds<T, BM, BN, 2, 4, true, GATE, AFF, false, GATEB>(p, acc, cbm, cbn, wm, wn, lane, tid, smem);
    if (!counted) younger = 0;
    lds_barrier();                              // statistic partials are read from the staging area: the next tile's second K-tile goes to slots inside it
  }
  if (p.trace && tid == 0) {
    unsigned long long* t = p.trace + (size_t)blockIdx.x * 8;
    t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = __builtin_amdgcn_s_memtime(); t[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); t[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));      // HW_ID, XCC_ID
  }
}

int launch_splitk_finish(ConvGemmParams& p, hipStream_t st) {
  const size_t nvec = (size_t)p.M * p.N / 8;
  const int blocks = (int)((nvec + 255) / 256 < 4096 ? (nvec + 255) / 256 : 4096);
  hipLaunchKernelGGL(conv_splitk_finish_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, p);
  SL_LAUNCH_CHECK("conv_splitk_finish_kernel");
  return 0;
}

}  // namespace

int slconv::launch_p8(ConvGemmParams& p, hipStream_t st) {
  p.gridM = cdiv(p.M, 256);
  p.gridN = p.N / 256;
  p.trace = g_sl_debug.p8_trace;
  p.flags |= 1;                                 // the next tile's first wait is counted past the epilogue's own loads and stores
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_gemm_p8_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8_LDS);
    (void)hipFuncSetAttribute((const void*)conv_gemm_p8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8_LDS);
    (void)hipFuncSetAttribute((const void*)conv_gemm_p8_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8_LDS);
    (void)hipFuncSetAttribute((const void*)conv_gemm_p8_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8_LDS);
    attr_set = true;
  }
  const int ntiles = p.gridM * p.gridN;
  // persistent: min(tiles, 256) blocks walk over the tiles; the instantiation with the gated-statistics store phase only where it is used
  if (p.gate && p.bias) {
    if (p.M % 256 || p.addend || p.scale || p.relu || p.mask_src || p.pre_addend || p.row_scale || p.out2) { sl_set_error("conv p8: gated data gradient with a bias: whole 256-row tiles and no other store-phase operand"); return SL_EINVAL; }
    hipLaunchKernelGGL(conv_gemm_p8_kernel<3>, dim3(ntiles > 256 ? 256 : ntiles), dim3(512), P8_LDS, st, p);
  } else if (p.gate) hipLaunchKernelGGL(conv_gemm_p8_kernel<1>, dim3(ntiles > 256 ? 256 : ntiles), dim3(512), P8_LDS, st, p);
  else if ((p.bias || p.scale) && !p.stat_partial) hipLaunchKernelGGL(conv_gemm_p8_kernel<2>, dim3(ntiles > 256 ? 256 : ntiles), dim3(512), P8_LDS, st, p);
  else        hipLaunchKernelGGL(conv_gemm_p8_kernel<0>, dim3(ntiles > 256 ? 256 : ntiles), dim3(512), P8_LDS, st, p);
  SL_LAUNCH_CHECK("conv_gemm_p8_kernel");
  return 0;
}

namespace {
// ---------------------------------------------------------------------------------------------------------------
// 3x3, stride 1 layers of the dilated trunk (pad = dilation, N % 256 == 0, H and W multiples of 16): the half-tile kernel above fetches the pixel operand once per TAP --
// nine shifted copies of nearly the same rows -- and its main loop is co-limited by the LDS fill rate.  Here a block owns a 16 x 16-pixel output tile:
// per 64-channel chunk the input patch WITH its dilation halo ((16 + 2d)^2 pixels x 128 B: 41 / 50 / 72 KiB for d = 1 / 2 / 4) goes to the LDS once and the nine taps read
// their fragments from shifted patch rows; only the weight rows (two 16 KiB halves per tap) stream through a two-K-tile ring.  Same wave grid, accumulator layout, phases
// (A0,B0) (A0,B1) (A1,B1) (A1,B0) and epilogues as the half-tile kernel (A0 / A1 = image rows 0-7 / 8-15 of the tile); one barrier per tap.
constexpr int P9_PATCH = 576 * 128;                                     // largest patch (d = 4)
constexpr int P9_LDS = P9_PATCH + 4 * P8_SLOT;                          // + B0 / B1 of two K-tiles = 136 KiB
constexpr int P9_PATCH1 = 42 * 1024;                                    // d = 1: 324 rows -> two patch buffers (the next chunk's patch lands under the current chunk's taps)
constexpr int P9_LDS1 = 2 * P9_PATCH1 + 4 * P8_SLOT;                    // 148 KiB
template <int EPI>       // 0 / 1 / 2 as in conv_gemm_p8_kernel; split-K parts are ranges of 64-channel chunks (all nine taps of a chunk stay together: one patch per chunk)
__global__ __launch_bounds__(512) void conv_gemm_p9_kernel(ConvGemmParams p) {
  using T = bf16_t;
  constexpr int BM = 256, BN = 256;
  constexpr bool GATE = EPI == 1, SPLITK = EPI == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int KS = SPLITK ? p.ksplit : 1;
  const int part = SPLITK ? (int)(blockIdx.x % KS) : 0;               // neighbouring blocks share the tile: the same patch rows and weight rows pass through the L2 together
  int bid = blockIdx.x / KS;
  {
    const int nwg = p.gridM * p.gridN, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // block order: column tiles in groups of two, row tiles inside a group -- with many column tiles (PPM data gradient: 8 x 2.4 MB of weight rows) an XCD's contiguous share of
  // the blocks then covers ONE group, whose weights stay in its L2, instead of streaming all 18.9 MB once per round of its 32 CUs
  const int GN = p.gridN > 2 && p.gridN % 2 == 0 && !(p.flags & 16) ? 2 : p.gridN;
  const int grp = bid / (p.gridM * GN), rem = bid - grp * (p.gridM * GN);
  const int bm = rem / GN, bn = grp * GN + rem % GN;
  const int d = p.dil, PW = 16 + 2 * d, PP = PW * PW;
  const int CT = p.C1;
  const int cbeg = SPLITK ? (CT / 64) * part / KS : 0, nchunk = SPLITK ? (CT / 64) * (part + 1) / KS : CT / 64;      // chunks [cbeg, nchunk)
  const int tx = p.Ws >> 4, ty = p.Hs >> 4;
  const int bx = bm % tx, by = (bm / tx) % ty, bb = bm / (tx * ty);
  const int y0 = by * 16 - d, x0 = bx * 16 - d;                          // image position of patch pixel (0, 0)
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  const int lr = lane >> 3, lpos = lane & 7;
  const unsigned char* zsrc = g_zero_page + lpos * 16;

  // ---- patch fill: wave-instruction g = j * 8 + wave covers patch rows g * 8 .. + 7 (lane -> row lr, 16-byte position lpos; the swizzle is applied to the source piece)
  constexpr int NPI = 9;                                                  // instructions per wave: 9 x 8 waves x 8 rows = 576 rows (rows >= PP are skipped)
  const unsigned char* psrc[NPI];
#pragma unroll
  for (int j = 0; j < NPI; ++j) {
    const int pr = (j * 8 + wave) * 8 + lr;
    const int py = pr / PW, px = pr - py * PW;
    const int iy = y0 + py, ix = x0 + px;
    const bool ok = pr < PP && (unsigned)iy < (unsigned)p.Hs && (unsigned)ix < (unsigned)p.Ws;
    psrc[j] = ok ? (const unsigned char*)p.src1 + ((size_t)(bb * p.Hs + iy) * p.Ws + ix) * CT * sizeof(T) + ((lpos ^ ((px >> 1) & 7)) << 4) : nullptr;      // swizzle by the patch COLUMN (see ldA)
  }
  const bool dbuf = d == 1 && !(p.flags & 8);                             // two patch buffers fit (flags bit 3 forces one: unused since round 4)
  const int boff = dbuf ? 2 * P9_PATCH1 : P9_PATCH;                       // weight ring behind the patch area
  auto issue_patch = [&](int chunk) {
    const unsigned dst = lds_base + (dbuf && (chunk & 1) ? P9_PATCH1 : 0);
#pragma unroll
    for (int j = 0; j < NPI; ++j) {
      if ((j * 8 + wave) * 8 < PP)                                        // wave-uniform
        glds16_asm(psrc[j] ? psrc[j] + (size_t)chunk * 128 : zsrc, dst + (j * 8 + wave) * 1024);
    }
  };
  // ---- weight rows: half h, rows h*128 + wave*16 + j*8 + lr of the block's 256 output channels; K-tile (tap, chunk) at byte offset (tap * CT + chunk * 64) * 2
  const size_t wpitch = (size_t)9 * CT * sizeof(T);
  int rsw[2];
  const unsigned char* wptr[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) rsw[j] = (lpos ^ (((j * 8 + lr) >> 1) & 7)) * 16;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) wptr[h][j] = (const unsigned char*)p.wt + (size_t)(bn * BN + h * 128 + wave * 16 + j * 8 + lr) * wpitch + rsw[j];
  auto issueB = [&](int tap, int chunk, int par) {
    const size_t koff = ((size_t)tap * CT + chunk * 64) * sizeof(T);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned dst = lds_base + boff + (par * 2 + h) * P8_SLOT + wave * 2048;
#pragma unroll
      for (int j = 0; j < 2; ++j) glds16_asm(wptr[h][j] + koff, dst + j * 1024);
    }
  };
  // ---- fragments.  B: as in the half-tile kernel.  A: lane (l31, fh) of row block (h, i2) is tile pixel (ty, tx) = (h*8 + wm*4 + i2*2 + (l31 >> 4), l31 & 15) -> patch row
  // (ty + ky d) PW + tx + kx d, 16-byte piece (2 ks + fh) ^ ((row >> 1) & 7)
  const int l31 = lane & 31, fh = lane >> 5;
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = l31 * 128 + (((2 * ks + fh) ^ ((l31 >> 1) & 7)) << 4);
  const unsigned char* fb = smem + boff + wn * (32 * 128);
  auto ldB = [&](int slot_off, int ks) { return *(const uint4*)(fb + slot_off + foff[ks]); };
  int prow[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) prow[h][i2] = (h * 8 + wm * 4 + i2 * 2 + (l31 >> 4)) * PW + (l31 & 15);
  // The XOR swizzle is a function of the patch COLUMN px, not of the LDS row: a ds_read_b128 is serviced in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... --
  // with lane = (image row l31 >> 4, pixel l31 & 15) a group holds 16 DIFFERENT columns of two image rows, and since the patch width is even the row parity (address bit 7)
  // is the column parity: (px & 1, (px >> 1) & 7) is distinct for 16 consecutive columns, whatever the tap shift.  (Swizzled by the LDS row, the second image row of a group
  // landed on the first one's banks: SQ_LDS_BANK_CONFLICT = 40 % of the LDS cycles.)
  auto ldA = [&](int h, int i2, int toff, int ks, int pbase = 0) {
    const int pr = prow[h][i2] + (toff >> 8);                             // toff = ((ky PW + kx) d) << 8 | kx d
    const int px = (l31 & 15) + (toff & 255);
    return *(const uint4*)(smem + pbase + pr * 128 + (((2 * ks + fh) ^ ((px >> 1) & 7)) << 4));
  };

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // K-tile k = (chunk, tap), weight slot pair k & 1.  One barrier per K-tile, at the end of P2: by then every wave has read both halves of slot pair k & 1 (B0(k) in P3 of
  // K-tile k - 1, B1(k) in P0), so B(k + 2) is issued into it right there and has a whole K-tile to land; B(k + 1) is waited for at the same point and P3 already loads the next
  // K-tile's first fragments (B0 from the other slot pair, the first eight image rows of the patch at the next tap's offset), so no K-tile starts with an empty pipeline.
  const int NK = 9 * (nchunk - cbeg);
  auto toff_of = [&](int tap) { const int t2 = p.mode ? 8 - tap : tap; return ((((t2 / 3) * PW + (t2 % 3)) * d) << 8) | ((t2 % 3) * d); };      // patch row offset << 8 | column offset      // data gradient: the correlation with the flipped window
  issue_patch(cbeg);
  issueB(0, cbeg, 0);
  issueB(1, cbeg, 1);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  uint4 a[4][2], b0k[4], b1k[4];
  {
    const int toff = toff_of(0), pb0 = dbuf && (cbeg & 1) ? P9_PATCH1 : 0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { a[ks][0] = ldA(0, 0, toff, ks, pb0); a[ks][1] = ldA(0, 1, toff, ks, pb0); b0k[ks] = ldB(0, ks); }
  }
  int tap = 0, chunk = cbeg;
#pragma unroll 1
  for (int k = 0; k < NK; ++k) {
    const int par = k & 1;
    const int b1s = (par * 2 + 1) * P8_SLOT, b0n = ((par ^ 1) * 2 + 0) * P8_SLOT;
    const int toff = toff_of(tap);
    const bool last_tap = tap == 8;
    const int toffn = toff_of(last_tap ? 0 : tap + 1);
    const int pb = dbuf && (chunk & 1) ? P9_PATCH1 : 0, pbn = dbuf && last_tap ? (pb ? 0 : P9_PATCH1) : pb;      // patch buffer of this / of the next K-tile
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int ph = q >> 2, ks = q & 3;
      const int ih = ph >> 1, jh = (ph == 1 || ph == 2) ? 1 : 0;
      if (ph == 0) b1k[ks] = ldB(b1s, ks);
      const uint4 bq = (ph == 0 || ph == 3) ? b0k[ks] : b1k[ks];
      Mma<T>::run(bq, a[ks][0], acc[ih * 2 + 0][jh]);
      Mma<T>::run(bq, a[ks][1], acc[ih * 2 + 1][jh]);
      if (ph == 1) { a[ks][0] = ldA(1, 0, toff, ks, pb); a[ks][1] = ldA(1, 1, toff, ks, pb); }   // image rows 8-15 of the tile
      if (ph == 2 && ks == 3) {
        wait_vmcnt<0>();                                                  // B(k + 1) (and, with two patch buffers, the next chunk's patch once it has been requested)
        __builtin_amdgcn_s_barrier();
        if (dbuf && tap == 0 && chunk + 1 < nchunk) issue_patch(chunk + 1);      // the other buffer: every wave is past the previous chunk
        if (k + 2 < NK) {
          int t2 = tap + 2, c2 = chunk;
          if (t2 >= 9) { t2 -= 9; ++c2; }
          issueB(t2, c2, par);
        }
      }
      if (ph == 3) {
        b0k[ks] = ldB(b0n, ks);                                           // B0 of the next K-tile
        if (!last_tap || dbuf) { a[ks][0] = ldA(0, 0, toffn, ks, pbn); a[ks][1] = ldA(0, 1, toffn, ks, pbn); }
      }
    }
    if (last_tap) {
      tap = 0; ++chunk;
      if (chunk < nchunk && !dbuf) {
        __builtin_amdgcn_s_barrier();                                     // every wave is past its last read of the patch
        issue_patch(chunk);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { a[ks][0] = ldA(0, 0, toffn, ks); a[ks][1] = ldA(0, 1, toffn, ks); }
      }
    } else ++tap;
  }
  if constexpr (SPLITK) { conv_store_partial(p, acc, part, bm, bn, wm, wn, lane); return; }
  lds_barrier();
  conv_epilogue_lds<T, BM, BN, 2, 4, true, GATE>(p, acc, bm, bn, wm, wn, lane, tid, smem);
}

}  // namespace

bool slconv::p9_on() { return g_sl_debug.conv_p9 != 0; }      // test hook sl_debug_conv_p9
bool slconv::p9_shape(const ConvGemmParams& p) {
  return p9_on() && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == p.dil && (p.dil == 1 || p.dil == 2 || p.dil == 4) && p.C2 == 0 && p.C1 % 64 == 0 && p.N % 256 == 0 &&
         p.Hs == p.Hd && p.Ws == p.Wd && p.Hs % 16 == 0 && p.Ws % 16 == 0 && ((long long)p.M >= 32768 || p.ksplit > 1) &&
         !(p.out2 || p.row_scale);                                       // every epilogue with the tile16 row map (fast: store / statistics / gated addend; generic: bias, folded BN, ReLU, pre-addend)
}
int slconv::launch_p9(ConvGemmParams& p, hipStream_t st) {
  p.gridM = p.M / 256; p.gridN = p.N / 256; p.tile16 = 1;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_gemm_p9_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(P9_LDS1 > P9_LDS ? P9_LDS1 : P9_LDS));
    (void)hipFuncSetAttribute((const void*)conv_gemm_p9_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(P9_LDS1 > P9_LDS ? P9_LDS1 : P9_LDS));
    (void)hipFuncSetAttribute((const void*)conv_gemm_p9_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(P9_LDS1 > P9_LDS ? P9_LDS1 : P9_LDS));
    attr_set = true;
  }
  if (p.ksplit > 1) {
    hipLaunchKernelGGL(conv_gemm_p9_kernel<2>, dim3(p.gridM * p.gridN * p.ksplit), dim3(512), p.dil == 1 ? P9_LDS1 : P9_LDS, st, p);
    SL_LAUNCH_CHECK("conv_gemm_p9_kernel (split-K)");
    return launch_splitk_finish(p, st);
  }
  if (p.gate) hipLaunchKernelGGL(conv_gemm_p9_kernel<1>, dim3(p.gridM * p.gridN), dim3(512), p.dil == 1 ? P9_LDS1 : P9_LDS, st, p);
  else        hipLaunchKernelGGL(conv_gemm_p9_kernel<0>, dim3(p.gridM * p.gridN), dim3(512), p.dil == 1 ? P9_LDS1 : P9_LDS, st, p);
  SL_LAUNCH_CHECK("conv_gemm_p9_kernel");
  return 0;
}
