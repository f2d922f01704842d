// Micro-benchmark: how fast can a CU pull operand tiles HBM/L2 -> LDS with global_load_lds (16 B/lane), as a function of the
// contiguous bytes fetched per matrix row per request (64 vs 128) and of the bytes kept in flight.  No MFMA, no ds_read.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/glds_bw.hip -o tools/micro/glds_bw ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// each block streams `rows` matrix rows (pitch bytes apart) of A and `rows` rows of B, RB bytes per row per stage
template <int RB, int NST, int FLY>   // FLY = stages allowed in flight
__global__ __launch_bounds__(512) void stream_kernel(const unsigned char* A, const unsigned char* B, size_t pitch, int ksteps, int mtiles, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ROWS = 512;                       // 256 A rows + 256 B rows
  constexpr int STAGE = ROWS * RB;
  constexpr int RPI = 1024 / RB;                  // rows per wave-instruction
  constexpr int L = ROWS / RPI / 8;               // instructions per wave per stage
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane / (RB / 16), lpos = lane % (RB / 16);
  const int bm = blockIdx.x % mtiles;
  const unsigned char* src[L];
#pragma unroll
  for (int j = 0; j < L; ++j) {
    const int row = (wave * L + j) * RPI + lrow;
    src[j] = (row < 256 ? A + (size_t)(bm * 256 + row) * pitch : B + (size_t)(row - 256) * pitch) + lpos * 16;
  }
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((const __attribute__((address_space(3))) unsigned char*)smem));
  auto issue = [&](int k, int slot) {
#pragma unroll
    for (int j = 0; j < L; ++j) {
      const unsigned dst = lds_base + slot * STAGE + (wave * L + j) * 1024;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src[j] + (size_t)k * RB), "s"(dst) : "memory");
    }
  };
  for (int s = 0; s < FLY && s < ksteps; ++s) issue(s, s % NST);
  int slot = 0;
  for (int k = 0; k < ksteps; ++k) {
    if (k + FLY < ksteps) { issue(k + FLY, (slot + FLY) % NST); wait_vmcnt<FLY * L>(); } else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (++slot == NST) slot = 0;
  }
  __syncthreads();
  if (tid == 0) sink[blockIdx.x] = ((unsigned*)smem)[0];
}

template <int RB, int NST, int FLY>
void run(const char* name, const unsigned char* A, const unsigned char* B, size_t pitch, int mtiles, int nblocks, unsigned* sink) {
  const int ksteps = (int)(pitch / RB);
  const size_t lds = (size_t)NST * 512 * RB;
  hipFuncSetAttribute((const void*)stream_kernel<RB, NST, FLY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream_kernel<RB, NST, FLY>), dim3(nblocks), dim3(512), lds, 0, A, B, pitch, ksteps, mtiles, sink);
  hipEventRecord(e0);
  const int it = 10;
  for (int w = 0; w < it; ++w) hipLaunchKernelGGL((stream_kernel<RB, NST, FLY>), dim3(nblocks), dim3(512), lds, 0, A, B, pitch, ksteps, mtiles, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
  const double bytes = (double)nblocks * 512 * pitch;
  printf("%-34s lds %3zu KiB  in flight %3d KiB  %8.3f ms  %7.2f TB/s into LDS  (%s)\n", name, lds >> 10, FLY * 512 * RB >> 10, ms, bytes / ms / 1e9,
         hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
  const size_t pitch = argc > 1 ? atol(argv[1]) : 9216;        // bytes per row (K * 2)
  const int mtiles = argc > 2 ? atoi(argv[2]) : 256;           // distinct A row-tiles (A bytes = mtiles*256*pitch)
  const int nblocks = argc > 3 ? atoi(argv[3]) : 1024;
  unsigned char *A, *B; unsigned* sink;
  hipMalloc(&A, (size_t)mtiles * 256 * pitch); hipMalloc(&B, 256 * pitch); hipMalloc(&sink, nblocks * 4);
  hipMemset(A, 1, (size_t)mtiles * 256 * pitch); hipMemset(B, 1, 256 * pitch);
  printf("pitch %zu B, %d A tiles (%.1f MB), %d blocks\n", pitch, mtiles, mtiles * 256.0 * pitch / 1e6, nblocks);
  run<64, 4, 2>("64 B/row, 4 stages, 2 in flight", A, B, pitch, mtiles, nblocks, sink);
  run<64, 4, 3>("64 B/row, 4 stages, 3 in flight", A, B, pitch, mtiles, nblocks, sink);
  run<64, 5, 4>("64 B/row, 5 stages, 4 in flight", A, B, pitch, mtiles, nblocks, sink);
  run<128, 2, 1>("128 B/row, 2 stages, 1 in flight", A, B, pitch, mtiles, nblocks, sink);
  run<128, 2, 2>("128 B/row, 2 stages, 2 in flight", A, B, pitch, mtiles, nblocks, sink);
  run<256, 1, 1>("256 B/row, 1 stage, 1 in flight", A, B, pitch, mtiles, nblocks, sink);
  return 0;
}
