// LDS-DMA (global_load_lds_dwordx4) rate per CU against the address pattern of one wave-instruction (gfx950), source
// L2-resident: (a) 1 KiB contiguous, (b) 8 rows x 128 B (whole cache lines, rows 2 KiB apart), (c) 16 rows x 64 B (half
// lines: what a GEMM stage with a 32-element bf16 K step fetches per row).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
extern __shared__ __attribute__((aligned(16))) char smem[];

template <int PATTERN>
__global__ __launch_bounds__(512) void dma_kernel(const char* src, int steps, int* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // the workgroup's tile: 512 rows x 2 KiB (1 MiB, shared by all workgroups of an XCD -> L2 hits)
  const char* base = src + (size_t)(blockIdx.x & 7) * (1 << 20);
  for (int s = 0; s < steps; ++s) {
    const int k = s & 31;                                   // 32 steps of 64 B along a row
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int g = wave * 4 + q;                           // instruction 0..31 of the step
      const char* p;
      if (PATTERN == 0) p = base + ((size_t)(s & 31) * 32 + g) * 1024 + lane * 16;
      else if (PATTERN == 1) p = base + (size_t)(g * 8 + (lane >> 3)) * 2048 * 2 + (k >> 1) * 128 + (lane & 7) * 16;   // 256 rows used twice as wide
      else p = base + (size_t)(g * 16 + (lane >> 2)) * 2048 + k * 64 + (lane & 3) * 16;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((s & 3) * 32768 + g * 1024));
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && smem[5] == 77) *sink = 1;
}

template <int PATTERN>
static void run(const char* src, int* sink, const char* name) {
  hipFuncSetAttribute((const void*)dma_kernel<PATTERN>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int steps = 4096;
  for (int wgs : {256}) {
    hipLaunchKernelGGL(dma_kernel<PATTERN>, dim3(wgs), dim3(512), 131072, 0, src, steps, sink);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(dma_kernel<PATTERN>, dim3(wgs), dim3(512), 131072, 0, src, steps, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)steps * 32768 * wgs;
    printf("%-34s wgs=%3d : %6.2f TB/s, %6.1f GB/s per CU, %5.0f ns per 32 KiB step\n", name, wgs, bytes / (ms * 1e-3) / 1e12,
           bytes / wgs / (ms * 1e-3) / 1e9, ms * 1e6 / steps);
  }
}

int main() {
  char* src; int* sink;
  hipMalloc(&src, 16 << 20); hipMalloc(&sink, 4);
  hipMemset(src, 1, 16 << 20);
  run<0>(src, sink, "1 KiB contiguous");
  run<1>(src, sink, "8 rows x 128 B (whole lines)");
  run<2>(src, sink, "16 rows x 64 B (half lines)");
  return 0;
}
