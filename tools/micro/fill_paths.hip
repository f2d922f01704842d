// Micro-benchmark (round 6): how many operand bytes per clock reach a CU's LDS when the fill of a 256 x 256 tile's K-tile
// (256 A rows + 256 B rows x 128 B = 64 KiB; stages of 256 or 128 rows per operand) is issued
//   MODE 0  entirely as LDS-DMA (global_load_lds_dwordx4, what conv_gemm_p8_kernel / conv_wgrad_glds_kernel do),
//   MODE 1  entirely through registers (global_load_dwordx4 -> VGPR -> ds_write_b128, the vector-L1 path),
//   MODE 2  A rows as LDS-DMA, B rows through registers (the split the review proposes),
//   MODE 3  no fill at all (with CONSUME: what the fragment reads + MFMAs + barriers of the loop cost on their own),
//   MODE 4  all LDS-DMA, but issued by waves 0-3 only (one wave per SIMD issues for two; its SIMD partner, waves 4-7, never touches the vector-memory pipe),
//   MODE 5  all LDS-DMA, issued by waves 0-1 only (four waves' worth each),
// each alone (CONSUME 0) and beside the fragment reads + MFMAs of such a tile (CONSUME 1: 24 ds_read_b128 + 32 v_mfma_f32_32x32x16_bf16 per wave and K-tile;
// CONSUME 2, 128-row stages only: the fragment reads of conv_wgrad_glds_kernel<256, 256> -- the 32 KiB stage read as [32 pixel rows][512 B dy | 512 B x] with
// ds_read_b64_tr_b16, 24 per wave and stage for the same 16 MFMAs, software-pipelined one k-step ahead like that kernel).
// A is `mtiles` distinct 256-row tiles (HBM / Infinity-Cache sourced when large), B is one 256-row tile every block re-reads (L2 sourced).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/fill_paths.hip -o tools/micro/fill_paths ; run on the GPU box:  fill_paths [pitch_bytes] [mtiles] [nblocks]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NW: waves per block (8, or 4 for two independent blocks per CU: each wave then issues twice the LDS-DMA and twice the MFMAs per stage, same bytes per FLOP per CU)
template <int ROWS, int NST, int FLY, int MODE, int CONSUME, int NW = 8>
__global__ __launch_bounds__(64 * NW) void fill_kernel(const unsigned char* A, const unsigned char* B, size_t pitch, int ksteps, int mtiles, float* sink, unsigned long long* ticks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int RB = 128, HALF = ROWS * RB;       // bytes per operand and stage (ROWS = 256: 32 KiB, a whole K-tile of a 256 x 256 tile; 128: half of one)
  constexpr int STAGE = 2 * HALF;
  constexpr int LH = ROWS / (8 * NW);             // 1 KiB wave-instructions per wave, operand and stage (ROWS rows / 8 rows per instruction / NW waves)
  constexpr int LA = (MODE == 1 || MODE == 3) ? 0 : LH, LB_DMA = (MODE == 0 || MODE >= 4) ? LH : 0;     // LDS-DMA instructions per wave and stage
  constexpr int VA = MODE == 1 ? LH : 0, VB = (MODE == 0 || MODE >= 3) ? 0 : LH;         // register-path loads per wave and stage
  constexpr int NV = VA + VB, ND = LA + LB_DMA;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane >> 3, lpos = lane & 7;
  const int bm = blockIdx.x % mtiles;
  const unsigned char* srcA[LH]; const unsigned char* srcB[LH];
#pragma unroll
  for (int j = 0; j < LH; ++j) {
    const int row = (wave * LH + j) * 8 + lrow;
    srcA[j] = A + (size_t)(bm * ROWS + row) * pitch + lpos * 16;
    srcB[j] = B + (size_t)row * pitch + lpos * 16;
  }
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((const __attribute__((address_space(3))) unsigned char*)smem));
  auto dma = [&](const unsigned char* g, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
  };
  u32x4 regs[FLY][NV > 0 ? NV : 1];
  // issue order inside iteration i: LDS-DMA of stage i + FLY, then register loads of stage i + FLY + 1 (they are written to the LDS one iteration before they are read)
  constexpr int SHARE = MODE == 4 ? 2 : (MODE == 5 ? 4 : 1);          // how many waves' instructions a loader wave issues
  const bool loader = wave < 8 / SHARE;
  auto issue_dma = [&](int k, int slot) {
    if constexpr (MODE >= 4) {
      if (loader) {
#pragma unroll
        for (int w = 0; w < SHARE; ++w) {
          const int vw = wave * SHARE + w;                                // the wave whose rows these are
          const size_t roff = (size_t)(vw - wave) * LH * 8 * pitch + (size_t)k * RB;
#pragma unroll
          for (int j = 0; j < LH; ++j) dma(srcA[j] + roff, lds_base + slot * STAGE + (vw * LH + j) * 1024);
#pragma unroll
          for (int j = 0; j < LH; ++j) dma(srcB[j] + roff, lds_base + slot * STAGE + HALF + (vw * LH + j) * 1024);
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < LA; ++j) dma(srcA[j] + (size_t)k * RB, lds_base + slot * STAGE + (wave * LH + j) * 1024);
#pragma unroll
    for (int j = 0; j < LB_DMA; ++j) dma(srcB[j] + (size_t)k * RB, lds_base + slot * STAGE + HALF + (wave * LH + j) * 1024);
  };
  auto issue_reg = [&](int k, u32x4* r) {
#pragma unroll
    for (int j = 0; j < VA; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[j]) : "v"(srcA[j] + (size_t)k * RB) : "memory");
#pragma unroll
    for (int j = 0; j < VB; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[VA + j]) : "v"(srcB[j] + (size_t)k * RB) : "memory");
  };
  auto write_reg = [&](int slot, const u32x4* r) {
#pragma unroll
    for (int j = 0; j < VA; ++j) *(u32x4*)(smem + slot * STAGE + (wave * LH + j) * 1024 + lane * 16) = r[j];
#pragma unroll
    for (int j = 0; j < VB; ++j) *(u32x4*)(smem + slot * STAGE + HALF + (wave * LH + j) * 1024 + lane * 16) = r[VA + j];
  };
  f32x16 acc[CONSUME == 2 ? 8 : 4];
#pragma unroll
  for (int i = 0; i < (CONSUME == 2 ? 8 : 4); ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // CONSUME 2: conv_wgrad_glds_kernel<bf16, 256, 256, 2, 4>'s fragment geometry (TM = 4, TN = 2, row bytes 512, source-side swizzle)
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  auto tr = [](const unsigned char* q) -> uint2 { return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q))); };
  auto ldfrag_tr = [&](u32x4* af, u32x4* bf, const unsigned char* st, int ks) {
    const int g = lane >> 4, l = lane & 15, wm = wave >> 2, wn = wave & 3;
    const int r = 16 * ks + 8 * (g >> 1) + (l >> 2), sw = (l >> 2) << 2, cofs = 16 * (g & 1) + 4 * (l & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = wm * 128 + i * 32 + cofs;
      const unsigned char* a0 = st + r * 512 + (((e >> 3) ^ sw) << 4) + (e & 7) * 2;
      const uint2 lo = tr(a0), hi = tr(a0 + 4 * 512);
      af[i] = (u32x4){lo.x, lo.y, hi.x, hi.y};
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = wn * 64 + j * 32 + cofs;
      const unsigned char* b0 = st + 16384 + r * 512 + (((e >> 3) ^ sw) << 4) + (e & 7) * 2;
      const uint2 lo = tr(b0), hi = tr(b0 + 4 * 512);
      bf[j] = (u32x4){lo.x, lo.y, hi.x, hi.y};
    }
  };
  auto mma_tr = [&](const u32x4* af, const u32x4* bf) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]), acc[i * 2 + j], 0, 0, 0);
  };
  u32x4 afA[4], bfA[2], afB[4], bfB[2];

  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  // prologue: DMA stages 0 .. FLY-1; register stages 0 .. FLY (stage 0 written before the loop)
#pragma unroll
  for (int s = 0; s < FLY; ++s) issue_dma(s, s % NST);
  if constexpr (NV > 0) {
    u32x4 r0[NV];
    issue_reg(0, r0);
    wait_vmcnt<0>();
    write_reg(0, r0);
#pragma unroll
    for (int s = 0; s < FLY; ++s) issue_reg(s + 1, regs[s]);
  }
  // outstanding now (oldest first): DMA(0..FLY-1), REG(1..FLY)  -- per later iteration the order is DMA(k+FLY), REG(k+FLY+1)
  int slot = 0;
  for (int k0 = 0; k0 < ksteps; k0 += FLY) {
#pragma unroll
    for (int f = 0; f < FLY; ++f) {
      const int k = k0 + f;
      // needed now: DMA(k) landed, REG(k+1) in registers.  In the steady state the queue is DMA(k) REG(k+1) DMA(k+1) REG(k+2) ... DMA(k+FLY-1) REG(k+FLY); the prologue's
      // queue (all DMA first) only makes the first waits stricter.
      if (k == 0) wait_vmcnt<0>(); else if (MODE >= 4) { if (loader) wait_vmcnt<(FLY - 1) * ND * SHARE>(); } else wait_vmcnt<(FLY - 1) * (ND + NV)>();
      int nslot = slot + 1; if (nslot == NST) nslot = 0;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the previous iteration's ds_write of stage k
      __builtin_amdgcn_s_barrier();
      if constexpr (NV > 0) write_reg(nslot, regs[f]);        // stage k + 1: its slot's last readers (stage k + 1 - NST) are behind this barrier
      int fslot = slot + FLY; if (fslot >= NST) fslot -= NST;
      issue_dma(k + FLY, fslot);                      // (reads past the end of the row range stay inside the allocation: see main)
      if constexpr (NV > 0) issue_reg(k + FLY + 1, regs[f]);
      if constexpr (CONSUME == 2) {
        static_assert(CONSUME != 2 || ROWS == 128, "the weight-gradient fragment geometry reads a 32 KiB stage");
        const unsigned char* st = smem + slot * STAGE;
        ldfrag_tr(afA, bfA, st, 0);
        ldfrag_tr(afB, bfB, st, 1);
        mma_tr(afA, bfA);
        mma_tr(afB, bfB);
      } else if constexpr (CONSUME) {
        const unsigned char* st = smem + slot * STAGE;
#pragma unroll
        for (int q = 0; q < ROWS / 64 * (8 / NW); ++q) {
          u32x4 fr[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) fr[i] = *(const u32x4*)(st + ((wave * 6 + i + q * 7) % (2 * ROWS / 8)) * 1024 + lane * 16);
#pragma unroll
          for (int i = 0; i < 8; ++i)
            acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[i % 3]), __builtin_bit_cast(bf16x8, fr[3 + (i % 3)]), acc[i & 3], 0, 0, 0);
        }
      }
      slot = nslot;
    }
  }
  wait_vmcnt<0>();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < (CONSUME == 2 ? 8 : 4); ++i) s += acc[i][0];
  if (lane == 0) sink[(blockIdx.x * 8 + wave) % 8192] = s + (float)((unsigned*)smem)[wave];
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int ROWS, int NST, int FLY, int MODE, int CONSUME, int NW = 8>
void run(const char* name, const unsigned char* A, const unsigned char* B, size_t pitch, int mtiles, int nblocks, float* sink, unsigned long long* ticks, size_t lds_pad = 0) {
  const int ksteps = (int)(pitch / 128) / FLY * FLY - 2 * FLY;           // the loop issues up to FLY + 1 stages past its last one: keep them inside the rows
  const size_t lds = (size_t)NST * ROWS * 256 + lds_pad;      // lds_pad: unused bytes that keep a second block off the CU
  auto kern = fill_kernel<ROWS, NST, FLY, MODE, CONSUME, NW>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64 * NW), lds, 0, A, B, pitch, ksteps, mtiles, sink, ticks);
  hipEventRecord(e0);
  const int it = 10;
  for (int w = 0; w < it; ++w) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64 * NW), lds, 0, A, B, pitch, ksteps, mtiles, sink, ticks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
  unsigned long long* h = (unsigned long long*)malloc(nblocks * 8);
  hipMemcpy(h, ticks, nblocks * 8, hipMemcpyDeviceToHost);
  double tk = 0; for (int i = 0; i < nblocks; ++i) tk += (double)h[i]; tk /= nblocks; free(h);
  const double bytes = (double)nblocks * ksteps * ROWS * 256;
  // s_memtime ticks at 100 MHz on gfx950: convert with the wall time of the launch instead -- B/clk/CU = bytes per block / (block time in 2.4 GHz clocks); blocks run `rounds` deep per CU
  const double rounds = (double)nblocks / 256.0;      // (two blocks per CU: the per-CU figure counts both)
  printf("%-58s %8.3f ms  %6.2f TB/s into LDS  %5.1f B/clk/CU at 2.4 GHz  (%.0f memtime ticks per block; %s)\n", name, ms, bytes / ms / 1e9,
         bytes / nblocks * rounds / (ms * 1e-3 * 2.4e9), tk, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
  const size_t pitch = argc > 1 ? atol(argv[1]) : 16384;       // bytes per row: 128 stages of 128 B
  const int mtiles = argc > 2 ? atoi(argv[2]) : 256;           // distinct A row-tiles (A bytes = mtiles * 256 * pitch)
  const int nblocks = argc > 3 ? atoi(argv[3]) : 1024;
  unsigned char *A, *B; float* sink; unsigned long long* ticks;
  hipMalloc(&A, (size_t)mtiles * 256 * pitch + 65536); hipMalloc(&B, 256 * pitch + 65536); hipMalloc(&sink, 8192 * 4); hipMalloc(&ticks, 2 * nblocks * 8);
  hipMemset(A, 0x3c, (size_t)mtiles * 256 * pitch + 65536); hipMemset(B, 0x3c, 256 * pitch + 65536);
  printf("pitch %zu B, %d A tiles (%.1f MB), %d blocks of 512 threads, one per CU (64-128 KiB of LDS)\n", pitch, mtiles, mtiles * 256.0 * pitch / 1e6, nblocks);
  const char* nm[6] = {"all LDS-DMA           ", "all through registers ", "A LDS-DMA, B registers", "NO fill (loop only)   ", "LDS-DMA by waves 0-3  ", "LDS-DMA by waves 0-1  "};
  char buf[128];
#define RUN3(ROWS, NST, FLY, C) \
  snprintf(buf, sizeof buf, "%s %s %3d-row stages, %d slots, %d in flight", C ? "with MFMA" : "fill only", nm[0], ROWS, NST, FLY); run<ROWS, NST, FLY, 0, C>(buf, A, B, pitch, mtiles, nblocks, sink, ticks); \
  snprintf(buf, sizeof buf, "%s %s %3d-row stages, %d slots, %d in flight", C ? "with MFMA" : "fill only", nm[1], ROWS, NST, FLY); run<ROWS, NST, FLY, 1, C>(buf, A, B, pitch, mtiles, nblocks, sink, ticks); \
  snprintf(buf, sizeof buf, "%s %s %3d-row stages, %d slots, %d in flight", C ? "with MFMA" : "fill only", nm[2], ROWS, NST, FLY); run<ROWS, NST, FLY, 2, C>(buf, A, B, pitch, mtiles, nblocks, sink, ticks); \
  if (C) { snprintf(buf, sizeof buf, "%s %s %3d-row stages, %d slots, %d in flight", "with MFMA", nm[3], ROWS, NST, FLY); run<ROWS, NST, FLY, 3, C>(buf, A, B, pitch, mtiles, nblocks, sink, ticks); } \
  snprintf(buf, sizeof buf, "%s %s %3d-row stages, %d slots, %d in flight", C ? "with MFMA" : "fill only", nm[4], ROWS, NST, FLY); run<ROWS, NST, FLY, 4, C>(buf, A, B, pitch, mtiles, nblocks, sink, ticks); \
  snprintf(buf, sizeof buf, "%s %s %3d-row stages, %d slots, %d in flight", C ? "with MFMA" : "fill only", nm[5], ROWS, NST, FLY); run<ROWS, NST, FLY, 5, C>(buf, A, B, pitch, mtiles, nblocks, sink, ticks);
  RUN3(256, 2, 1, 0)
  RUN3(128, 4, 2, 0)
  RUN3(128, 4, 3, 0)
  RUN3(256, 2, 1, 1)
  RUN3(128, 4, 2, 1)
  RUN3(128, 4, 3, 1)
  printf("-- one block of 8 waves per CU vs two independent blocks per CU (same bytes and MFMAs per CU; 128-row stages, 2 slots = 64 KiB per block, 1 in flight)\n");
  run<128, 2, 1, 0, 1, 8>("with MFMA all LDS-DMA   ONE 8-wave block per CU (LDS padded)          ", A, B, pitch, mtiles, nblocks, sink, ticks, 40 * 1024);
  run<128, 2, 1, 0, 1, 8>("with MFMA all LDS-DMA   TWO 8-wave blocks per CU                      ", A, B, pitch, mtiles, nblocks, sink, ticks);
  run<128, 2, 1, 0, 1, 4>("with MFMA all LDS-DMA   TWO 4-wave blocks per CU (2x work per wave)   ", A, B, pitch, mtiles, 2 * nblocks, sink, ticks);
  run<128, 2, 1, 3, 1, 8>("with MFMA NO fill       ONE 8-wave block per CU (LDS padded)          ", A, B, pitch, mtiles, nblocks, sink, ticks, 40 * 1024);
  run<128, 2, 1, 3, 1, 8>("with MFMA NO fill       TWO 8-wave blocks per CU                      ", A, B, pitch, mtiles, nblocks, sink, ticks);
  run<128, 2, 1, 3, 1, 4>("with MFMA NO fill       TWO 4-wave blocks per CU (2x work per wave)   ", A, B, pitch, mtiles, 2 * nblocks, sink, ticks);
  printf("-- the same with the weight-gradient kernel's transpose reads (24 ds_read_b64_tr_b16 per wave and stage instead of 12 ds_read_b128)\n");
  RUN3(128, 4, 2, 2)
  RUN3(128, 4, 3, 2)
  return 0;
}
