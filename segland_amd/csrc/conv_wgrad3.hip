// Weight gradient of the 3x3 stride-1 layers (pad == dilation) on MFMA, all nine taps from ONE pass over dy and x (gfx950, bf16).
//
//   dw[n][ky][kx][c] = sum_{b,y,x} dy[b][y][x][n] * x[b][y + (ky-1) d][x + (kx-1) d][c]          (resnet.py:46,96-97, pspnet_pop.py:19 backward)
//
// The per-tap kernel (conv_wgrad_glds_kernel) gives every tap its own blocks: dy and x travel L2 -> LDS -> registers nine times and a 256 x 256 tile
// needs 32 B of LDS-DMA per clock and CU at the MFMA rate (the path delivers ~20).  Here:
//   * POLYPHASE: a dilation-d layer is d*d independent dilation-1 problems on the sub-images (rows / columns of one residue class mod d); NHWC pixels are
//     >= 128 contiguous bytes, so a strided pixel gather costs nothing.  The halo is one pixel whatever d is.
//   * a block owns a 128 (n) x 64 (c) tile of dw for ALL nine taps (8 waves as 4 x 2, 32 x 32 x 9 = 144 accumulator registers per lane) and streams 16-pixel-wide
//     column strips of sub-images row by row: a STEP is one strip row = 16 pixels of dy (the MFMA k index) and the 18-pixel x row one image row ABOVE it.
//   * per step a wave reads ONE dy fragment (kept for two more steps in a rolling register window: rows t, t-1, t-2 = taps ky 0, 1, 2 against x row t-1) and
//     THREE x fragments (kx = 0, 1, 2: the same LDS rows shifted by one pixel) for NINE MFMAs: 0.44 KiB of LDS reads per MFMA (per-tap kernel: 0.75),
//     6.25 KiB of LDS-DMA per step and CU = 11 B/clk at the MFMA rate.
//   * rows outside a piece come from a zero page, so piece boundaries need no special case in the multiply loop: a piece of L rows runs L + 1 steps (whole
//     strips) or L + 2 (a strip cut into segments: the segment's last dy row still meets the x row below it).
//   * the accumulators leave as 36 x 1 KiB float4 stores per wave in REGISTER order (slab [split][tile][wave][q][tap][lane][4]); wgrad3_reduce_kernel sums the
//     slabs in a fixed order and writes OIHW rows of 288 contiguous floats -- bit-stable run to run.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

struct Wg3Params {
  const bf16_t* x1; const bf16_t* x2; int C1, C2;   // x (virtual concat): pixel pitch C1 / C2
  const bf16_t* dy; int Cout;
  float* ws;
  int H, W, d, Hs, Ws, nstrips;
  int L, SP, ppu, pieces, ppb;                      // rows / steps per piece, pieces per unit (strip), total pieces, pieces per block
  int tilesN, tilesC;
  unsigned long long* trace;                        // debug (tools/wgrad_trace.py): per block {s_memtime at entry, after the prologue, after the main loop, at the end, HW_ID, XCC_ID}
};

constexpr int W3_BN = 128, W3_BC = 64;
constexpr int W3_DYB = 64 * 256;                  // dy part of a stage: 4 steps x 16 pixels x 256 B
constexpr int W3_XB = 72 * 128;                   // x part: 4 steps x 18 pixels x 128 B
constexpr int W3_STAGE = W3_DYB + W3_XB;          // 25 600 B
constexpr int W3_NST = 5;
constexpr int W3_SLAB = 8 * 4 * 9 * 64 * 4;       // floats per (split, tile)
constexpr int W3_MAX_PPB = 2048;

__device__ __attribute__((aligned(256))) unsigned char g_w3zero[256];

typedef __attribute__((ext_vector_type(4))) short w3s16x4_t;
__device__ __forceinline__ uint2 w3_tr16(const unsigned char* p) {       // ds_read_b64_tr_b16; p 8-byte aligned
  const w3s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w3s16x4_t*)(p));
  return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ void w3_glds16(const void* g, unsigned lds_addr) {      // see conv_gemm_common.h: glds16_asm
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}
// the same under a lane mask (exec is set inside the statement: no divergent branch around it, the caller's basic block stays whole); LDS address = M0 + 16 * lane id
__device__ __forceinline__ void w3_glds16_masked(const void* g, unsigned lds_addr, unsigned long long mask) {
  unsigned keep; unsigned long long ex;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %3\n\ts_mov_b64 exec, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep), "=&s"(ex) : "v"(g), "s"(lds_addr), "s"(mask) : "memory");
}
template <int N> __device__ __forceinline__ void w3_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(512) void conv_wgrad3_kernel(Wg3Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int2* tbl = (int2*)(smem + W3_NST * W3_STAGE);          // per piece of this block: {pixel index of (row y0, strip column 0), y0 | strip column << 16}; x < 0: no such piece
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wc = wave & 1;
  unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
  if (p.trace) tr0 = __builtin_amdgcn_s_memtime();

  // XCD-aware order: the 32 CUs of an XCD take consecutive logical blocks = the tiles of ONE pixel range, so dy and x are fetched into that L2 once
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles = p.tilesN * p.tilesC;
  const int tile = bid % tiles, split = bid / tiles;
  const int tn = tile % p.tilesN, tc = tile / p.tilesN;
  const int n0 = tn * W3_BN, c0 = tc * W3_BC;
  const bf16_t* xb; int xpitch;
  if (c0 < p.C1) { xb = p.x1 + c0; xpitch = p.C1; } else { xb = p.x2 + (c0 - p.C1); xpitch = p.C2; }
  const bf16_t* dyb = p.dy + n0;

  for (int i = tid; i < p.ppb; i += 512) {
    const int P = split * p.ppb + i;
    int2 e = make_int2(-1, 0);
    if (P < p.pieces) {
      const int unit = P / p.ppu, seg = P - unit * p.ppu, y0 = seg * p.L;
      const int strip = unit % p.nstrips; int r = unit / p.nstrips;
      const int rx = r % p.d; r /= p.d;
      const int ry = r % p.d, b = r / p.d;
      e.x = (b * p.H + ry + y0 * p.d) * p.W + rx + strip * 16 * p.d;
      e.y = y0 | ((strip * 16) << 16);
    }
    tbl[i] = e;
  }
  if (tid == 0) tbl[p.ppb] = make_int2(-1, 0);            // sentinel: rows behind the last piece
  __syncthreads();
  const int nst = (p.ppb * p.SP + 3) >> 2;                 // stages of four steps

  // ---- fill side.  Stage = steps 4j .. 4j+3.  dy: 64 pixel rows of 256 B, one KiB instruction = 4 pixels; wave w issues instructions 2w, 2w+1 (step w>>1,
  // columns 8(w&1) + 4i + (lane>>4)): its step is wave-uniform.  x: 72 pixel rows of 128 B, instruction = 8 pixels; wave w issues instruction w (pixels 8w..8w+7,
  // per-lane step / patch column) and one eighth of instruction 8 (pixel 64 + w = step 3, patch column 10 + w; lanes 8w..8w+7 under exec).  Chunk swizzles on the
  // SOURCE side: dy chunk ^ ((pixel&3)<<2), x chunk ^ (((pixel>>1)&1)<<2) -- the LDS image of an instruction stays lane-linear.
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((const __attribute__((address_space(3))) unsigned char*)smem));
  const unsigned char* zsrc = g_w3zero + (lane & 15) * 16;
  // The fill code is BRANCH-FREE (invalid rows select the zero page, the piece table has a sentinel entry, the partial instruction sets exec inside its asm
  // statement) and is placed in four parts behind the MFMAs of the four steps of a stage: one basic block per stage, so its ~150 VALU instructions issue in the
  // shadow of the matrix pipe instead of in front of it (first version: 800 ticks per step against 576 MFMA-issue cycles, profiles/r5_wgrad_trace.txt).
  int dy_t = wave >> 1, dy_pc = 0;
  const int dy_col = 8 * (wave & 1) + (lane >> 4);
  const int dy_ch = ((lane & 15) ^ (((lane >> 4) & 3) << 2)) * 8;
  const int xq = 8 * wave + (lane >> 3);
  int x_t = xq / 18, x_pc = 0;
  const int x_pcol = xq - 18 * (xq / 18);
  const int x_ch = ((lane & 7) ^ ((((lane >> 3) >> 1) & 1) << 2)) * 8;
  int xp_t = 3, xp_pc = 0;
  const int xp_pcol = 10 + wave;
  const unsigned long long xp_mask = 0xffull << (8 * wave);
  auto advance = [&](int& t, int& pc) {
    t += 4;
    const bool wrap = t >= p.SP;
    t = wrap ? t - p.SP : t;
    pc = wrap ? pc + 1 : pc;
    pc = pc < p.ppb ? pc : p.ppb;                 // tbl[ppb] is the sentinel (no such piece)
  };
  auto xsrc = [&](int t, int pc, int pcol) -> const void* {
    const int2 e = tbl[pc];
    const int y0 = e.y & 0xffff, xs = e.y >> 16;
    const int row = y0 + t - 1, col = xs + pcol - 1;
    const bool ok = (e.x >= 0) & ((unsigned)row < (unsigned)p.Hs) & ((unsigned)col < (unsigned)p.Ws);
    const unsigned char* q = (const unsigned char*)(xb + (size_t)(unsigned)(e.x + ((t - 1) * p.W + pcol - 1) * p.d) * (unsigned)xpitch + x_ch);
    return (const void*)(ok ? q : zsrc);
  };
  auto issue_dy = [&](int slot) {
    const unsigned sb = lds_base + slot * W3_STAGE;
    const int2 e = tbl[dy_pc];
    const int y0 = e.y & 0xffff;
    const bool ok = (e.x >= 0) & (dy_t < p.L) & (y0 + dy_t < p.Hs);
    const unsigned char* q = (const unsigned char*)(dyb + (size_t)(unsigned)(e.x + (dy_t * p.W + dy_col) * p.d) * (unsigned)p.Cout + dy_ch);
    const unsigned char* s0 = ok ? q : zsrc;
    const unsigned char* s1 = ok ? q + (size_t)(8 * p.d) * p.Cout : zsrc;          // 4 pixels to the right, bytes
    w3_glds16((const void*)s0, sb + (2 * wave) * 1024);
    w3_glds16((const void*)s1, sb + (2 * wave + 1) * 1024);
    advance(dy_t, dy_pc);
  };
  auto issue_x = [&](int slot) {
    w3_glds16(xsrc(x_t, x_pc, x_pcol), lds_base + slot * W3_STAGE + W3_DYB + wave * 1024);
    advance(x_t, x_pc);
  };
  auto issue_xp = [&](int slot) {
    w3_glds16_masked(xsrc(xp_t, xp_pc, xp_pcol), lds_base + slot * W3_STAGE + W3_DYB + 8 * 1024, xp_mask);
    advance(xp_t, xp_pc);
  };
  auto issue = [&](int slot) { issue_dy(slot); issue_x(slot); issue_xp(slot); };

  // ---- multiply side
  f32x16_t acc[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  // transpose-read geometry (conv_wgrad.hip): 16-lane group g supplies pixel rows 8(g>>1) + (l>>2) (+4: second read), channels 16(g&1) + 4(l&3)
  const int g = lane >> 4, l = lane & 15;
  const int rsub = 8 * (g >> 1) + (l >> 2);
  const int eA = 32 * wn + 16 * (g & 1) + 4 * (l & 3);
  const unsigned offA = rsub * 256 + (((eA >> 3) ^ (((l >> 2) & 3) << 2)) << 4) + (eA & 7) * 2;
  const int eB = 32 * wc + 16 * (g & 1) + 4 * (l & 3);
  const unsigned EB = ((eB >> 3) << 4) + (eB & 7) * 2;
  auto ldA = [&](const unsigned char* st, int i) -> uint4 {
    const unsigned char* a0 = st + i * 4096 + offA;
    const uint2 lo = w3_tr16(a0), hi = w3_tr16(a0 + 1024);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
  auto ldB = [&](const unsigned char* st, int i, int kx) -> uint4 {
    const int px = rsub + i * 18 + kx;
    const unsigned char* b0 = st + W3_DYB + px * 128 + (EB ^ ((px & 2) << 5));
    const uint2 lo = w3_tr16(b0), hi = w3_tr16(b0 + 512);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
  auto mm = [&](f32x16_t& c, const uint4& a, const uint4& b) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  };

  // ring protocol: at the top of iteration `it` stages it and it+1 are complete (every wave waited for its own pieces, then the barrier), it+2 and it+3 are in
  // flight and it+4 is issued into the slot stage it-1 left; the fragments of the next step (reaching into stage it+1 at the last step) are read before the
  // MFMAs of the current one.
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (s < nst) issue(s);
  if (nst >= 4) w3_wait_vmcnt<8>(); else if (nst == 3) w3_wait_vmcnt<4>(); else w3_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (p.trace) tr1 = __builtin_amdgcn_s_memtime();

  uint4 A1 = make_uint4(0, 0, 0, 0), A2 = make_uint4(0, 0, 0, 0);
  uint4 An = ldA(smem, 0), Bn0 = ldB(smem, 0, 0), Bn1 = ldB(smem, 0, 1), Bn2 = ldB(smem, 0, 2);
  int slot = 0;
  // MORE (compile time): stage it+4 exists and its fill code rides behind the MFMAs of steps 0, 1, 2 -- no branch inside the stage body
  auto stage = [&](auto MORE, int it) {
    int s4 = slot + 4; if (s4 >= W3_NST) s4 -= W3_NST;
    int nslot = slot + 1; if (nslot == W3_NST) nslot = 0;
    const unsigned char* st = smem + slot * W3_STAGE;
    const unsigned char* stn = smem + nslot * W3_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint4 A0 = An, B0 = Bn0, B1 = Bn1, B2 = Bn2;
      if (i < 3) { An = ldA(st, i + 1); Bn0 = ldB(st, i + 1, 0); Bn1 = ldB(st, i + 1, 1); Bn2 = ldB(st, i + 1, 2); }
      else       { An = ldA(stn, 0);    Bn0 = ldB(stn, 0, 0);    Bn1 = ldB(stn, 0, 1);    Bn2 = ldB(stn, 0, 2); }
      mm(acc[0][0], A0, B0); mm(acc[1][0], A1, B0); mm(acc[2][0], A2, B0);
      mm(acc[0][1], A0, B1); mm(acc[1][1], A1, B1); mm(acc[2][1], A2, B1);
      mm(acc[0][2], A0, B2); mm(acc[1][2], A1, B2); mm(acc[2][2], A2, B2);
      A2 = A1; A1 = A0;
      if constexpr (decltype(MORE)::value) { if (i == 0) issue_dy(s4); else if (i == 1) issue_x(s4); else if (i == 2) issue_xp(s4); }
    }
    // stage it+2 complete before the next iteration: younger stages (it+3, it+4) may stay in flight
    const int younger = (it + 4 < nst ? it + 4 : nst - 1) - (it + 2);
    if (younger >= 2) w3_wait_vmcnt<8>(); else if (younger == 1) w3_wait_vmcnt<4>(); else w3_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    slot = nslot;
  };
  int it = 0;
  for (; it + 4 < nst; ++it) stage(std::true_type{}, it);
  for (; it < nst; ++it) stage(std::false_type{}, it);
  if (p.trace) tr2 = __builtin_amdgcn_s_memtime();

  // slab [split][tile][wave][q][tap][lane][4]: acc[ky][kx][4q + j] = dw[n0 + 32 wn + 8q + 4(lane>>5) + j][tap][c0 + 32 wc + (lane&31)]
  float* out = p.ws + ((size_t)split * tiles + tile) * W3_SLAB + (size_t)wave * (4 * 9 * 256) + lane * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const f32x16_t& a = acc[ky][kx];
        *(f32x4_t*)(out + (q * 9 + ky * 3 + kx) * 256) = (f32x4_t){a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
      }
  if (p.trace && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the stamp counts the slab stores of this wave as accepted
    unsigned long long* t = p.trace + (size_t)blockIdx.x * 8;
    t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = __builtin_amdgcn_s_memtime();
    t[4] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));        // HW_ID
    t[5] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));       // XCC_ID
    t[6] = (unsigned long long)nst; t[7] = (unsigned long long)bid;
  }
}

// dw_oihw[n][c][tap] = sum_s slab[s][tile][wave][q][tap][lane][.]: one block per (tile, wave, q) = 8 output channels x 32 input channels x 9 taps; thread = (tap, lane)
// sums its float4 over the splits in a fixed order, the 8 x 288 result rows go through the LDS and leave as contiguous float4 runs.
__global__ __launch_bounds__(576) void wgrad3_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int tilesN, int tilesC, int splits, int dw_cin_total, int dw_ci_off) {
  __shared__ __attribute__((aligned(16))) float tilebuf[8 * 288];
  const int tiles = tilesN * tilesC;
  const int tile = blockIdx.x >> 5, wq = blockIdx.x & 31, wave = wq >> 2, q = wq & 3;
  const int tid = threadIdx.x, tap = tid >> 6, lane = tid & 63;
  const f32x4_t* src = (const f32x4_t*)(ws + (size_t)tile * W3_SLAB + (size_t)(wq * 9 + tap) * 256 + lane * 4);
  const size_t stride = (size_t)tiles * W3_SLAB / 4;
  f32x4_t s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0}, s3 = {0, 0, 0, 0};
  int k = 0;
  for (; k + 3 < splits; k += 4) {
    const f32x4_t a = src[(size_t)k * stride], b = src[(size_t)(k + 1) * stride], c = src[(size_t)(k + 2) * stride], d = src[(size_t)(k + 3) * stride];
    s0 += a; s1 += b; s2 += c; s3 += d;
  }
  for (; k < splits; ++k) s0 += src[(size_t)k * stride];
  const f32x4_t s = (s0 + s1) + (s2 + s3);
  const int cl = lane & 31, nl = 4 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < 4; ++j) tilebuf[((nl + j) * 32 + cl) * 9 + tap] = s[j];
  __syncthreads();
  const int tn = tile % tilesN, tc = tile / tilesN;
  const int nb = tn * W3_BN + 32 * (wave >> 1) + 8 * q, cb = tc * W3_BC + 32 * (wave & 1);
  const int row = tid / 72, k4 = tid - row * 72;
  float* o = dw + ((size_t)(nb + row) * dw_cin_total + dw_ci_off + cb) * 9 + k4 * 4;
  *(f32x4_t*)o = *(const f32x4_t*)(tilebuf + row * 288 + k4 * 4);
}

struct Wg3Plan { int ok, d, Hs, Ws, nstrips, L, SP, ppu, pieces, ppb, splits, tilesN, tilesC; size_t ws_bytes; };

Wg3Plan wg3_plan(const SlConvDesc* d) {
  Wg3Plan pl{};
  const int c2 = d->Cin - d->C1;
  if (!g_sl_debug.wgrad3 || d->dtype != SL_BF16 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != d->dil || d->dil < 1) return pl;
  if (d->Cout % W3_BN || d->C1 % W3_BC || c2 % W3_BC || d->H % d->dil || d->W % d->dil) return pl;
  const int Hs = d->H / d->dil, Ws = d->W / d->dil;
  if (Ws % 16 || Hs < 4 || Hs > 4096 || Ws > 4096) return pl;
  if ((long long)d->B * d->H * d->W < 8192 || (long long)d->B * d->H * d->W >= (1ll << 30)) return pl;
  pl.d = d->dil; pl.Hs = Hs; pl.Ws = Ws; pl.nstrips = Ws / 16;
  pl.tilesN = d->Cout / W3_BN; pl.tilesC = d->Cin / W3_BC;
  const int tiles = pl.tilesN * pl.tilesC;
  const long long units = (long long)d->B * d->dil * d->dil * pl.nstrips;
  // cost in steps: block waves x (steps of a block + store phase) + slab reduce per split
  const double EPI = 40.0, RED = 0.21 * tiles;
  double best = 1e30;
  for (int ppu = 1; ppu <= 16; ppu *= 2) {
    const int L = (Hs + ppu - 1) / ppu;
    if (L < 4 && ppu > 1) break;
    const int SP = L + (ppu == 1 ? 1 : 2);
    const long long pieces = units * ppu;
    if (pieces > (1ll << 24)) break;
    for (long long s = 1; s <= pieces && s <= 1024; ++s) {
      const long long ppb = (pieces + s - 1) / s;
      if (ppb > W3_MAX_PPB) continue;
      const long long se = (pieces + ppb - 1) / ppb;
      if (se != s) continue;
      const double waves = (double)((tiles * se + 255) / 256);
      const double t = waves * (ppb * SP + EPI) + se * RED;
      if (t < best) { best = t; pl.ppu = ppu; pl.L = L; pl.SP = SP; pl.pieces = (int)pieces; pl.ppb = (int)ppb; pl.splits = (int)se; }
    }
  }
  if (best >= 1e30) return pl;
  pl.ws_bytes = (size_t)pl.splits * tiles * W3_SLAB * sizeof(float);
  if (pl.ws_bytes > (size_t)2 << 30) return pl;
  pl.ok = 1;
  return pl;
}

}  // namespace

// test hook (include/segland_hip_debug.h): the plan of a shape: out[0..7] = {served, pieces per strip, rows per piece, steps per piece, pieces per block, splits, tiles, blocks}
extern "C" int sl_debug_wgrad3_plan(const SlConvDesc* d, int* out) {
  const Wg3Plan pl = wg3_plan(d);
  out[0] = pl.ok; out[1] = pl.ppu; out[2] = pl.L; out[3] = pl.SP; out[4] = pl.ppb; out[5] = pl.splits; out[6] = pl.tilesN * pl.tilesC; out[7] = pl.tilesN * pl.tilesC * pl.splits;
  return pl.ok;
}

// internal (conv_wgrad.hip): does the nine-tap kernel take this layer, and with how much workspace
bool sl_wgrad3_eligible(const SlConvDesc* d, size_t* ws_bytes) {
  const Wg3Plan pl = wg3_plan(d);
  if (ws_bytes) *ws_bytes = pl.ok ? pl.ws_bytes : 0;
  return pl.ok != 0;
}

int sl_wgrad3_run(const SlConvDesc* d, const void* x, const void* x2, const void* dy, float* dw, int dw_cin_total, int dw_ci_off, void* workspace, size_t workspace_bytes,
                  hipStream_t st) {
  const Wg3Plan pl = wg3_plan(d);
  SL_REQUIRE(pl.ok, "conv bwd_weight (nine-tap kernel): shape not served");
  if (workspace_bytes < pl.ws_bytes) { sl_set_error("conv bwd_weight: workspace %zu < %zu", workspace_bytes, pl.ws_bytes); return SL_EWORKSPACE; }
  SL_REQUIRE(dw_cin_total % 4 == 0 && dw_ci_off % 4 == 0, "conv bwd_weight: dw channel window must be 4-aligned");
  Wg3Params p{};
  p.x1 = (const bf16_t*)x; p.x2 = (const bf16_t*)x2; p.C1 = d->C1; p.C2 = d->Cin - d->C1; p.dy = (const bf16_t*)dy; p.Cout = d->Cout; p.ws = (float*)workspace;
  p.H = d->H; p.W = d->W; p.d = pl.d; p.Hs = pl.Hs; p.Ws = pl.Ws; p.nstrips = pl.nstrips;
  p.L = pl.L; p.SP = pl.SP; p.ppu = pl.ppu; p.pieces = pl.pieces; p.ppb = pl.ppb; p.tilesN = pl.tilesN; p.tilesC = pl.tilesC;
  p.trace = g_sl_debug.wgrad3_trace;
  const int tiles = pl.tilesN * pl.tilesC;
  const size_t lds = (size_t)W3_NST * W3_STAGE + (size_t)(pl.ppb + 1) * sizeof(int2);
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv_wgrad3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
  hipLaunchKernelGGL(conv_wgrad3_kernel, dim3(tiles * pl.splits), dim3(512), lds, st, p);
  SL_LAUNCH_CHECK("conv_wgrad3_kernel");
  hipLaunchKernelGGL(wgrad3_reduce_kernel, dim3(tiles * 32), dim3(576), 0, st, (const float*)workspace, dw, pl.tilesN, pl.tilesC, pl.splits, dw_cin_total, dw_ci_off);
  SL_LAUNCH_CHECK("wgrad3_reduce_kernel");
  return 0;
}
