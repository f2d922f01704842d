// Tile kernels of the implicit-GEMM convolution: the two-stage LDS-DMA kernel (family 2) and the NST-stage ring (family 4).  See conv_gemm_common.h.
#include <type_traits>
#include "conv_gemm_common.h"

#ifndef SL_RING192_NST
#define SL_RING192_NST 4
#endif

namespace {

// ---------------------------------------------------------------------------------------------------------------
// v2: operands go HBM -> LDS directly (global_load_lds, 16 B per lane, no VGPR staging, no ds_write pass).
// The LDS destination of one wave-instruction is lane-linear (base + lane*16 = 8 rows x 128 B), so the bank swizzle is
// applied to the SOURCE chunk index: LDS position p of row r receives global chunk p ^ ((r>>1)&7) -- still inside the
// same 128-byte line of that row, so coalescing is untouched.  Lanes whose tap falls into the padding (or rows >= M)
// read a zero page instead.
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_gemm_glds_kernel(ConvGemmParams p) {
  constexpr int EPC = 16 / sizeof(T);
  constexpr int BKE = 8 * EPC;
  constexpr int NW = WM * WN;                   // waves per block (4 or 8)
  constexpr int AR = BM / 8 / NW, BR = BN / 8 / NW;     // 8-row (1 KiB) groups loaded per wave
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(AR >= 1 && BR >= 1 && TM >= 1 && TN >= 1, "tile/wave shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* lds_a = smem;
  unsigned char* lds_b = smem + 2 * BM * 128;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  int bid = blockIdx.x;
  {
    const int nwg = p.gridM * p.gridN, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bm = bid / p.gridN, bn = bid % p.gridN;

  const int lrow = lane >> 3, lpos = lane & 7;
  // rows handled by this lane: A: (wave*AR + j)*8 + lrow, B: (wave*BR + j)*8 + lrow
  int rb[AR], ry[AR], rx[AR], rsw[AR];
#pragma unroll
  for (int j = 0; j < AR; ++j) {
    const int row = (wave * AR + j) * 8 + lrow;
    rsw[j] = (lpos ^ ((row >> 1) & 7)) * EPC;      // source chunk (element offset) feeding this LDS position
    const int m = bm * BM + row;
    if (m < p.M) {
      const int b = m / (p.Hd * p.Wd), rem = m - b * (p.Hd * p.Wd);
      const int yd = rem / p.Wd, xd = rem - yd * p.Wd;
      rb[j] = b;
      if (p.mode == 0) { ry[j] = yd * p.stride - p.pad; rx[j] = xd * p.stride - p.pad; }
      else             { ry[j] = yd + p.pad;            rx[j] = xd + p.pad; }
    } else { rb[j] = -1; ry[j] = 0; rx[j] = 0; }
  }
  const int CT = p.C1 + p.C2;
  const int ctiles = CT / BKE;
  const int taps = p.KH * p.KW;
  const int nk = taps * ctiles;
  const T* wrow[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    const int row = (wave * BR + j) * 8 + lrow;
    wrow[j] = (const T*)p.wt + (size_t)(bn * BN + row) * taps * CT + (lpos ^ ((row >> 1) & 7)) * EPC;
  }

  // K-tile order: channel tile OUTER, tap INNER -- the taps of one 64-channel slice touch the same cache lines (3x3
  // neighbourhoods overlap), so they hit in L2 instead of re-streaming the slab once per tap (measured: FETCH_SIZE was
  // 7.6x the algorithmic bytes with the tap loop outside).  Per row we keep the source pixel index of tap (0,0) and a
  // validity bit per tap; the per-tap displacement is wave-uniform.  (dgrad through a stride > 1 is not affine in
  // the tap: that rare case recomputes the row per K-tile.)
  const bool affine = (p.mode == 0) || (p.stride == 1);
  const int sgn = p.mode == 0 ? 1 : -1;
  int rbase[AR]; unsigned vmask[AR];
#pragma unroll
  for (int j = 0; j < AR; ++j) {
    rbase[j] = rb[j] >= 0 ? (rb[j] * p.Hs + ry[j]) * p.Ws + rx[j] : 0;     // may be "outside": only used with a valid bit
    unsigned m = 0;
    if (rb[j] >= 0 && affine)
      for (int t = 0; t < taps; ++t) {
        const int ky = t / p.KW, kx = t - ky * p.KW;
        const int ys = ry[j] + sgn * ky * p.dil, xs = rx[j] + sgn * kx * p.dil;
        if ((unsigned)ys < (unsigned)p.Hs && (unsigned)xs < (unsigned)p.Ws) m |= 1u << t;
      }
    vmask[j] = m;
  }
  auto slow_pix = [&](int j, int tap_) -> int {        // dgrad, stride > 1
    const int ky = tap_ / p.KW, kx = tap_ - ky * p.KW;
    const int ty = ry[j] - ky * p.dil, tx = rx[j] - kx * p.dil;
    if (rb[j] < 0 || ty < 0 || tx < 0) return -1;
    const int ys = ty / p.stride, xs = tx / p.stride;
    if (ys * p.stride != ty || xs * p.stride != tx || ys >= p.Hs || xs >= p.Ws) return -1;
    return (rb[j] * p.Hs + ys) * p.Ws + xs;
  };
  int tap = 0, ct = 0;
  const unsigned char* zsrc = g_zero_page + lpos * 16;
  auto issue = [&](int buf) {
    const int c0 = ct * BKE;
    const unsigned char* base; unsigned pitchb;
    if (c0 < p.C1) { base = (const unsigned char*)p.src1 + (size_t)c0 * sizeof(T); pitchb = p.C1 * (unsigned)sizeof(T); }
    else           { base = (const unsigned char*)p.src2 + (size_t)(c0 - p.C1) * sizeof(T); pitchb = p.C2 * (unsigned)sizeof(T); }
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int delta = sgn * (ky * p.dil * p.Ws + kx * p.dil);
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      int pix; bool ok;
      if (affine) { pix = rbase[j] + delta; ok = (vmask[j] >> tap) & 1u; }
      else { pix = slow_pix(j, tap); ok = pix >= 0; }
      const unsigned char* src = base + (size_t)((unsigned)pix) * pitchb + rsw[j] * (int)sizeof(T);
      src = ok ? src : zsrc;
      glds16(src, lds_a + buf * BM * 128 + (wave * AR + j) * 1024);
    }
    const size_t koff = (size_t)tap * CT + c0;
#pragma unroll
    for (int j = 0; j < BR; ++j) glds16(wrow[j] + koff, lds_b + buf * BN * 128 + (wave * BR + j) * 1024);
    if (++tap == taps) { tap = 0; ++ct; }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issue(0);
  __syncthreads();      // the barrier's release waits for the LDS-DMA (vmcnt) of every wave

  const int frow = lane & 31, fhalf = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) issue(buf ^ 1);
    const unsigned char* la = lds_a + buf * BM * 128;
    const unsigned char* lb = lds_b + buf * BN * 128;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *(const uint4*)(la + lds_off(wm * (BM / WM) + i * 32 + frow, 2 * s + fhalf));
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *(const uint4*)(lb + lds_off(wn * (BN / WN) + j * 32 + frow, 2 * s + fhalf));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(bf[j], af[i], acc[i][j]);     // swapped roles: D[n][m]
    }
    __syncthreads();
  }

  conv_epilogue_lds<T, BM, BN, WM, WN>(p, acc, bm, bn, wm, wn, lane, tid, smem);
}

#ifndef SL_RING_LATE
#define SL_RING_LATE 1          // 0: the round-3..5 schedule (A/B builds: tools/ab_libs.sh)
#endif

template <typename T, int BM, int BN, int WM, int WN, int RBYTES, int NST>
struct RingGeom {
  static constexpr int STAGE = (BM + BN) * RBYTES;
  static constexpr int RING_BYTES = NST * STAGE;
  static constexpr int EBN = BN == 192 ? 64 : BN;      // the 192-column tile is stored as three 64-column tiles
  static constexpr int EPI = EpiGeom<T, BM, EBN, WM, WN>::TILE_BYTES / EpiGeom<T, BM, EBN, WM, WN>::NPASS;
  static constexpr int LDS_BYTES = RING_BYTES > EPI ? RING_BYTES : EPI;
};

// SUBP: one parity plane of a stride-2 data gradient (ConvGemmParams::sub): rows are half-resolution positions, only the taps that exist for the plane's parity are walked
template <typename T, int BM, int BN, int WM, int WN, int RBYTES, int NST, bool GATE = false, bool SUBP = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_gemm_ring_kernel(ConvGemmParams p) {
  constexpr int EPC = 16 / sizeof(T);
  constexpr int CPRW = RBYTES / 16;             // 16-byte chunks per stage row
  constexpr int BKE = CPRW * EPC;               // K elements per stage
  constexpr int RPI = 1024 / RBYTES;            // rows per 1 KiB wave-instruction
  constexpr int NW = WM * WN;
  constexpr int AR = BM / RPI / NW, BR = BN / RPI / NW;
  constexpr int L = AR + BR;                    // LDS-DMA instructions per wave per stage
  constexpr int KS = CPRW / 2;                  // MFMA k-steps per stage
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  using RG = RingGeom<T, BM, BN, WM, WN, RBYTES, NST>;
  static_assert(AR >= 1 && BR >= 1, "ring shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  int bid = blockIdx.x;
  {
    const int nwg = p.gridM * p.gridN, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bm = bid / p.gridN, bn = bid % p.gridN;
  const int lrow = lane / CPRW, lpos = lane % CPRW;
  const int taps = p.KH * p.KW;
  const int CT = p.C1 + p.C2;
  const int ctiles = CT / BKE;
  // SUBP: the taps whose source index (yd + pad - ky dil) / 2 is an integer for this plane's parity, four bits each (block-uniform); every other launch walks all taps
  unsigned long long taplist = 0; int ntaps = taps;
  if constexpr (SUBP) {
    ntaps = 0;
    for (int t = 0; t < taps; ++t) {
      const int ky = t / p.KW, kx = t - ky * p.KW;
      if (((p.sub_py + p.pad - ky * p.dil) & 1) == 0 && ((p.sub_px + p.pad - kx * p.dil) & 1) == 0) { taplist |= (unsigned long long)t << (4 * ntaps); ++ntaps; }
    }
  }
  const int nk = ntaps * ctiles;
  const bool affine = !SUBP && ((p.mode == 0) || (p.stride == 1));
  const int sgn = p.mode == 0 ? 1 : -1;

  int rb[AR], ry[AR], rx[AR], rbase[AR], rsw[AR]; unsigned vmask[AR];
#pragma unroll
  for (int j = 0; j < AR; ++j) {
    const int row = (wave * AR + j) * RPI + lrow;
    rsw[j] = (lpos ^ ring_swz<RBYTES>(row)) * 16;             // byte offset of the source chunk inside the K slice
    const int m = bm * BM + row;
    rb[j] = -1; ry[j] = 0; rx[j] = 0;
    if constexpr (SUBP) {
      if (m < p.M) {
        const int Wh = p.Wd >> 1, hw = (p.Hd >> 1) * Wh;
        const int b = m / hw, rem = m - b * hw, i = rem / Wh, jj = rem - i * Wh;
        rb[j] = b; ry[j] = 2 * i + p.sub_py + p.pad; rx[j] = 2 * jj + p.sub_px + p.pad;
      }
    } else if (m < p.M) {
      const int b = m / (p.Hd * p.Wd), rem = m - b * (p.Hd * p.Wd);
      const int yd = rem / p.Wd, xd = rem - yd * p.Wd;
      rb[j] = b;
      if (p.mode == 0) { ry[j] = yd * p.stride - p.pad; rx[j] = xd * p.stride - p.pad; }
      else             { ry[j] = yd + p.pad;            rx[j] = xd + p.pad; }
    }
    rbase[j] = rb[j] >= 0 ? (rb[j] * p.Hs + ry[j]) * p.Ws + rx[j] : 0;
    unsigned mk = 0;
    if (rb[j] >= 0 && affine)
      for (int t = 0; t < taps; ++t) {
        const int ky = t / p.KW, kx = t - ky * p.KW;
        const int ys = ry[j] + sgn * ky * p.dil, xs = rx[j] + sgn * kx * p.dil;
        if ((unsigned)ys < (unsigned)p.Hs && (unsigned)xs < (unsigned)p.Ws) mk |= 1u << t;
      }
    vmask[j] = mk;
  }
  auto slow_pix = [&](int j, int tap_) -> int {
    const int ky = tap_ / p.KW, kx = tap_ - ky * p.KW;
    const int ty = ry[j] - ky * p.dil, tx = rx[j] - kx * p.dil;
    if (rb[j] < 0 || ty < 0 || tx < 0) return -1;
    const int ys = ty / p.stride, xs = tx / p.stride;
    if (ys * p.stride != ty || xs * p.stride != tx || ys >= p.Hs || xs >= p.Ws) return -1;
    return (rb[j] * p.Hs + ys) * p.Ws + xs;
  };
  const unsigned char* wrow[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    const int row = (wave * BR + j) * RPI + lrow;
    wrow[j] = (const unsigned char*)p.wt + ((size_t)(bn * BN + row) * taps * CT) * sizeof(T) + (lpos ^ ring_swz<RBYTES>(row)) * 16;
  }
  int tap = 0, ct = 0;
  const unsigned char* zsrc = g_zero_page + lpos * 16;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  auto issue = [&](int slot) {
    const unsigned la = lds_base + slot * RG::STAGE;
    const unsigned lb = la + BM * RBYTES;
    const int c0 = ct * BKE;
    const unsigned char* base; unsigned pitchb;
    if (c0 < p.C1) { base = (const unsigned char*)p.src1 + (size_t)c0 * sizeof(T); pitchb = p.C1 * (unsigned)sizeof(T); }
    else           { base = (const unsigned char*)p.src2 + (size_t)(c0 - p.C1) * sizeof(T); pitchb = p.C2 * (unsigned)sizeof(T); }
    const int rt = SUBP ? (int)((taplist >> (4 * tap)) & 15) : tap;      // the tap this K-tile multiplies
    const int ky = rt / p.KW, kx = rt - ky * p.KW;
    const int delta = sgn * (ky * p.dil * p.Ws + kx * p.dil);
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      int pix; bool ok;
      if constexpr (SUBP) {
        // the parity is right by construction: (ry - ky dil, rx - kx dil) are even
        const int ty = ry[j] - ky * p.dil, tx = rx[j] - kx * p.dil;
        const int ys = ty >> 1, xs = tx >> 1;
        ok = rb[j] >= 0 && ty >= 0 && tx >= 0 && ys < p.Hs && xs < p.Ws;
        pix = (rb[j] * p.Hs + ys) * p.Ws + xs;
      } else if (affine) { pix = rbase[j] + delta; ok = (vmask[j] >> rt) & 1u; }
      else { pix = slow_pix(j, rt); ok = pix >= 0; }
      const unsigned char* src = base + (size_t)((unsigned)pix) * pitchb + rsw[j];
      glds16_asm(ok ? src : zsrc, la + (wave * AR + j) * 1024);
    }
    const size_t koff = ((size_t)rt * CT + c0) * sizeof(T);
#pragma unroll
    for (int j = 0; j < BR; ++j) glds16_asm(wrow[j] + koff, lb + (wave * BR + j) * 1024);
    if (++tap == ntaps) { tap = 0; ++ct; }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Pipeline invariant at the top of iteration i: stages i and i+1 are complete and visible to every wave, and the
  // fragments of (stage i, k-step 0) are already in registers.  Inside the iteration the fragment loads of the NEXT
  // k-step (the last one reaches into stage i+1) are issued before the MFMAs of the current one, so LDS latency hides
  // behind the matrix pipe; the LDS-DMA of stages i+2 / i+3 stays in flight across the barrier.
  static_assert(NST >= 3 && NST <= 6 && (KS % 2) == 0, "ring schedule: 3..6 stages and an even number of k-steps");
  constexpr int D = NST - 1;                    // stages issued ahead of the one being consumed
  // wait until at most `fl` of the most recently issued stages are still in flight (fl is block-uniform)
  auto wait_stages = [&](int fl) {
    if constexpr (D >= 4 && 3 * L <= 63) { if (fl >= 3) { wait_vmcnt<3 * L>(); return; } }
    if constexpr (D >= 3 && 2 * L <= 63) { if (fl >= 2) { wait_vmcnt<2 * L>(); return; } }
    if (fl >= 1) wait_vmcnt<L>(); else wait_vmcnt<0>();
  };
  const int frow = lane & 31, fhalf = lane >> 5;
  auto ldfrag = [&](uint4* af, uint4* bf, int slot_, int s2) {
    const unsigned char* la = smem + slot_ * RG::STAGE;
    const unsigned char* lb = la + BM * RBYTES;
#pragma unroll
    for (int ii = 0; ii < TM; ++ii) af[ii] = *(const uint4*)(la + ring_off<RBYTES>(wm * (BM / WM) + ii * 32 + frow, 2 * s2 + fhalf));
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) bf[jj] = *(const uint4*)(lb + ring_off<RBYTES>(wn * (BN / WN) + jj * 32 + frow, 2 * s2 + fhalf));
  };
  auto mma = [&](const uint4* af, const uint4* bf) {
#pragma unroll
    for (int ii = 0; ii < TM; ++ii)
#pragma unroll
      for (int jj = 0; jj < TN; ++jj) Mma<T>::run(bf[jj], af[ii], acc[ii][jj]);
  };
  uint4 afA[TM], bfA[TN], afB[TM], bfB[TN];

#pragma unroll
  for (int st = 0; st < D; ++st)
    if (st < nk) issue(st);
#if SL_RING_LATE
  // Round 6 schedule: a stage is waited for where its first fragments are read -- before the LAST k-step of the iteration in front of it -- not one iteration earlier.
  // Invariant at the top of iteration i: stage i is complete and visible to every wave, the fragments of (stage i, k-step 0) are in registers, stages i+1 .. i+D-1 fly;
  // the slot stage i+D goes to (the one of stage i-1) was released by the barrier inside iteration i-1, behind that iteration's last fragment reads of it.  Same MFMA
  // order, same results; a stage's latency budget grows from D - 1 to D - 1/KS iterations with the same LDS (small-M layers -- 8 192 tokens on 64 x 128 tiles with a
  // three-slot ring -- were bound by exactly that: one iteration of 256 MFMA clocks against ~2 000 clocks of load latency).
  wait_stages(min(nk, D) - 1);                            // stage 0 landed
  __builtin_amdgcn_s_barrier();
  ldfrag(afA, bfA, 0, 0);

  int slot = 0;
  for (int i = 0; i < nk; ++i) {
    if (i + D < nk) { int ns = slot + D; if (ns >= NST) ns -= NST; issue(ns); }
    int nslot = slot + 1; if (nslot == NST) nslot = 0;
#pragma unroll
    for (int s2 = 0; s2 < KS; s2 += 2) {
      ldfrag(afB, bfB, slot, s2 + 1);
      mma(afA, bfA);
      if (s2 + 2 < KS) ldfrag(afA, bfA, slot, s2 + 2);
      else if (i + 1 < nk) {
        wait_stages(min(i + D, nk - 1) - (i + 1));         // stage i+1 landed; i+2 .. i+D may stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's last fragment reads of slot i are done: behind the barrier the slot may be refilled
        __builtin_amdgcn_s_barrier();
        ldfrag(afA, bfA, nslot, 0);                        // first k-step of the next stage
      }
      mma(afB, bfB);
    }
    slot = nslot;
  }
#else
  wait_stages(min(nk, D) - 2);                            // stages 0 and 1 landed (later ones may still fly)
  __builtin_amdgcn_s_barrier();
  ldfrag(afA, bfA, 0, 0);

  int slot = 0;
  for (int i = 0; i < nk; ++i) {
    if (i + D < nk) { int ns = slot + D; if (ns >= NST) ns -= NST; issue(ns); }
    int nslot = slot + 1; if (nslot == NST) nslot = 0;
#pragma unroll
    for (int s2 = 0; s2 < KS; s2 += 2) {
      ldfrag(afB, bfB, slot, s2 + 1);
      mma(afA, bfA);
      if (s2 + 2 < KS) ldfrag(afA, bfA, slot, s2 + 2);
      else             ldfrag(afA, bfA, nslot, 0);          // first k-step of the next stage (complete by the invariant)
      mma(afB, bfB);
    }
    // make stage i+2 complete before anyone starts iteration i+1; stages i+3 .. i+D (already issued) may stay in flight
    wait_stages(min(i + D, nk - 1) - (i + 2));
    __builtin_amdgcn_s_barrier();
    slot = nslot;
  }
#endif
  __syncthreads();
  if constexpr (BN == 192) {
    // 192-column tile (a wave owns 32 rows x all 192 columns): the store phases map a 64 * WM * WN-thread block onto power-of-two row widths, so the tile leaves as three
    // 64-column tiles of the same rows
    static_assert(WN == 1 && TM == 1 && TN == 6, "192-column tile: one wave per 32 rows");
    auto store64 = [&](auto jc) {
      constexpr int jj = decltype(jc)::value;
      f32x16_t sub[1][2] = {{acc[0][2 * jj], acc[0][2 * jj + 1]}};
      conv_epilogue_lds<T, BM, 64, WM, 1, false, GATE>(p, sub, bm, bn * 3 + jj, wm, 0, lane, tid, smem);
      __syncthreads();
    };
    store64(std::integral_constant<int, 0>{}); store64(std::integral_constant<int, 1>{}); store64(std::integral_constant<int, 2>{});
  } else {
    conv_epilogue_lds<T, BM, BN, WM, WN, false, GATE, true, SUBP>(p, acc, bm, bn, wm, wn, lane, tid, smem);
  }
}

// one parity plane of a stride-2 data gradient on the 256-row ring tiles (bf16; conv_gemm.hip: launch_parity_planes has checked the shape)
template <int BN, int WM, int WN>
int launch_ring_subp(ConvGemmParams& p, hipStream_t st) {
  using RG = RingGeom<bf16_t, 256, BN, WM, WN, 64, 4>;
  p.gridM = p.M / 256;
  p.gridN = p.N / BN;
  const size_t lds = RG::LDS_BYTES;
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_gemm_ring_kernel<bf16_t, 256, BN, WM, WN, 64, 4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)conv_gemm_ring_kernel<bf16_t, 256, BN, WM, WN, 64, 4, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  if (p.gate) hipLaunchKernelGGL((conv_gemm_ring_kernel<bf16_t, 256, BN, WM, WN, 64, 4, true, true>), dim3(p.gridM * p.gridN), dim3(64 * WM * WN), lds, st, p);
  else        hipLaunchKernelGGL((conv_gemm_ring_kernel<bf16_t, 256, BN, WM, WN, 64, 4, false, true>), dim3(p.gridM * p.gridN), dim3(64 * WM * WN), lds, st, p);
  SL_LAUNCH_CHECK("conv_gemm_ring_kernel (parity plane)");
  return 0;
}

template <typename T, int BM, int BN, int WM, int WN, int RBYTES, int NST>
int launch_ring(ConvGemmParams& p, hipStream_t st) {
  using RG = RingGeom<T, BM, BN, WM, WN, RBYTES, NST>;
  p.gridM = cdiv(p.M, BM);
  p.gridN = p.N / BN;
  const size_t lds = RG::LDS_BYTES;
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_gemm_ring_kernel<T, BM, BN, WM, WN, RBYTES, NST, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)conv_gemm_ring_kernel<T, BM, BN, WM, WN, RBYTES, NST, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  // like the half-tile and patch kernels: the gated-statistics store phase lives in an instantiation of its own
  if (p.gate) hipLaunchKernelGGL((conv_gemm_ring_kernel<T, BM, BN, WM, WN, RBYTES, NST, true>), dim3(p.gridM * p.gridN), dim3(64 * WM * WN), lds, st, p);
  else        hipLaunchKernelGGL((conv_gemm_ring_kernel<T, BM, BN, WM, WN, RBYTES, NST, false>), dim3(p.gridM * p.gridN), dim3(64 * WM * WN), lds, st, p);
  SL_LAUNCH_CHECK("conv_gemm_ring_kernel");
  return 0;
}


template <typename T, int BM, int BN, int WM, int WN>
int launch_glds(ConvGemmParams& p, hipStream_t st) {
  p.gridM = cdiv(p.M, BM);
  p.gridN = p.N / BN;
  const size_t lds = EpiGeom<T, BM, BN, WM, WN>::LDS_BYTES;
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_gemm_glds_kernel<T, BM, BN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_gemm_glds_kernel<T, BM, BN, WM, WN>), dim3(p.gridM * p.gridN), dim3(64 * WM * WN), lds, st, p);
  SL_LAUNCH_CHECK("conv_gemm_glds_kernel");
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------
// At most 32 rows (the +-prototype rows of the POP head's classifier MLP, pspnet_pop.py:46-52 on [2K, 512]: 14 / 8 rows per launch): the tile kernels run such a launch as
// one 128-row tile per 128 columns -- 8 K-tiles of prologue / barrier / LDS round trips for 3.7 MFLOP, 17 us.  Here one wave owns 32 output columns: both operands are K-major
// rows, so a lane reads its 16-byte fragment pieces straight from global memory (rows >= M: zeros), sixteen k-steps of loads in flight, one accumulator chain over ascending k
// (the same MFMA sequence per output element as the tile kernels: bit-identical), then ReLU / ReLU-mask and 8-byte stores.  Epilogues: none, relu, mask_src.
__global__ __launch_bounds__(64) void conv_rows_small_kernel(ConvGemmParams p) {
  const int lane = threadIdx.x, row = lane & 31, half = lane >> 5;
  const int n0 = blockIdx.x * 32;
  const int K = p.C1;
  const bf16_t* xr = (const bf16_t*)p.src1 + (size_t)(row < p.M ? row : 0) * K + half * 8;
  const bf16_t* wr = (const bf16_t*)p.wt + (size_t)(n0 + row) * K + half * 8;
  const bool live = row < p.M;
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int U = 16;
  for (int k0 = 0; k0 < K; k0 += 16 * U) {
    uint4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 16 * u;
      if (k < K) { a[u] = live ? *(const uint4*)(xr + k) : make_uint4(0, 0, 0, 0); b[u] = *(const uint4*)(wr + k); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (k0 + 16 * u < K) Mma<bf16_t>::run(b[u], a[u], acc);          // operand roles as in the tile kernels: A = weight rows, B = pixel rows
  }
  // lane holds, for pixel m = lane & 31, channels n0 + 8 (r >> 2) + 4 half + (r & 3)
  if (!live) return;
  const bf16_t* ms = p.mask_src ? (const bf16_t*)p.mask_src + (size_t)row * p.N : nullptr;
  bf16_t* o = (bf16_t*)p.out + (size_t)row * p.N;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = n0 + 8 * q + 4 * half;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = bf2f(f2bf(acc[4 * q + e]));          // the store phases of the tile kernels work on the rounded accumulators
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    }
    if (ms) {
      const uint2 mk = *(const uint2*)(ms + n);
      const float k4[4] = {__uint_as_float(mk.x << 16), __uint_as_float(mk.x & 0xffff0000u), __uint_as_float(mk.y << 16), __uint_as_float(mk.y & 0xffff0000u)};
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = k4[e] > 0.f ? v[e] : 0.f;
    }
    uint2 r;
    r.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
    r.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    *(uint2*)(o + n) = r;
  }
}

}  // namespace

int slconv::launch_tile(int cfg, int dtype, ConvGemmParams& p, hipStream_t st) {
  if (p.sub) {
    if (dtype == SL_BF16 && cfg == 4256256) return launch_ring_subp<256, 2, 4>(p, st);
    if (dtype == SL_BF16 && cfg == 4256128) return launch_ring_subp<128, 4, 2>(p, st);
    sl_set_error("conv: no parity-plane kernel for configuration %d", cfg);
    return SL_EINVAL;
  }
  if (dtype == SL_BF16 && cfg == 4064128) {
    // 64 x 128 tiles of the few-tile inference layers: 128-byte stage rows (half as many stages per K, twice the bytes in flight per wave).  Up to 256 tiles one block
    // per CU is resident anyway: four stages (96 KiB); beyond, three stages (72 KiB) keep two blocks per CU.  profiles/r5_ab_ring64_geom.txt
    if ((long long)cdiv(p.M, 64) * (p.N / 128) <= 256) return launch_ring<bf16_t, 64, 128, 2, 2, 128, 4>(p, st);
    return launch_ring<bf16_t, 64, 128, 2, 2, 128, 3>(p, st);
  }
  if (dtype == SL_BF16 && cfg == 3032032) {
    p.gridM = 1; p.gridN = p.N / 32;
    hipLaunchKernelGGL(conv_rows_small_kernel, dim3(p.N / 32), dim3(64), 0, st, p);
    SL_LAUNCH_CHECK("conv_rows_small_kernel");
    return 0;
  }
  if (dtype == SL_BF16 && cfg == 4128064) return launch_ring<bf16_t, 128, 64, 4, 1, 64, 4>(p, st);
  if (dtype == SL_BF16 && cfg == 4128192) return launch_ring<bf16_t, 128, 192, 4, 1, 64, SL_RING192_NST>(p, st);
#define SL_TILE_CASES(T)                                                           \
  switch (cfg) {                                                                   \
    case 4256256: return launch_ring<T, 256, 256, 2, 4, 64, 4>(p, st);             \
    case 4256128: return launch_ring<T, 256, 128, 4, 2, 64, 4>(p, st);             \
    case 2256064: return launch_glds<T, 256, 64, 8, 1>(p, st);                     \
    case 4128128: return launch_ring<T, 128, 128, 2, 2, 64, 4>(p, st);             \
    case 2128128: return launch_glds<T, 128, 128, 2, 2>(p, st);                    \
    case 2128064: return launch_glds<T, 128, 64, 2, 2>(p, st);                     \
    default: break;                                                                \
  }
  if (dtype == SL_BF16) { SL_TILE_CASES(bf16_t) } else { SL_TILE_CASES(float) }
#undef SL_TILE_CASES
  sl_set_error("conv: no kernel for configuration %d", cfg);
  return SL_EINVAL;
}

