// Per-CU VMEM throughput probes (gfx950): what one CU can (a) write to HBM with 16-byte non-temporal
// stores and (b) stream from an L2-resident buffer into LDS with global_load_lds_dwordx4 -- the two
// streams the training MLP kernels run side by side.  Build: hipcc --offload-arch=gfx950 -O3 -o vmem_probe vmem_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) char smem[];

// every WG writes `bytes_per_wg` (its own contiguous region), 512 threads, 16 B per lane per store
__global__ __launch_bounds__(512) void store_kernel(char* out, size_t bytes_per_wg, int nt) {
  char* base = out + (size_t)blockIdx.x * bytes_per_wg + threadIdx.x * 16;
  const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  for (size_t off = 0; off < bytes_per_wg; off += 512 * 16) {
    if (nt) __builtin_nontemporal_store(v, (u32x4*)(base + off));
    else *(u32x4*)(base + off) = v;
  }
}
// every WG streams the same `src_bytes` buffer `reps` times into a 64 KiB LDS ring (1 KiB per wave-instruction)
__global__ __launch_bounds__(512) void dma_kernel(const char* src, size_t src_bytes, int reps, int* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  for (int r = 0; r < reps; ++r)
    for (size_t off = 0; off < src_bytes; off += 8 * 1024) {
      const char* g = src + off + wave * 1024 + lane * 16;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((off + wave * 1024) & 65535));
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
      if (((off >> 13) & 7) == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && smem[5] == 77) *sink = 1;
}
static float run(void (*launch)(hipStream_t), int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(0); hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) launch(0);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}
static char* g_out; static const char* g_src; static int* g_sink; static int g_wgs, g_nt, g_reps; static size_t g_bpw, g_sb;
int main() {
  hipMalloc(&g_out, (size_t)4 << 30); hipMalloc((void**)&g_src, 2 << 20); hipMalloc(&g_sink, 4);
  hipMemset((void*)g_src, 1, 2 << 20);
  hipFuncSetAttribute((const void*)dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const double clk = 2.0e9;
  for (int nt = 0; nt < 2; ++nt)
    for (int wgs : {32, 64, 128, 256, 512}) {
      g_wgs = wgs; g_nt = nt; g_bpw = ((size_t)2 << 30) / wgs;
      float ms = run([](hipStream_t s) { hipLaunchKernelGGL(store_kernel, dim3(g_wgs), dim3(512), 0, s, g_out, g_bpw, g_nt); }, 5);
      double bps = (double)g_bpw * wgs / (ms * 1e-3);
      printf("store nt=%d wgs=%3d : %7.2f TB/s total, %6.1f B/clk per WG (@2.0 GHz)\n", nt, wgs, bps / 1e12, bps / wgs / clk);
    }
  for (size_t sb : {(size_t)1 << 20, (size_t)1228800})
    for (int wgs : {64, 256}) {
      g_wgs = wgs; g_sb = sb; g_reps = 64;
      float ms = run([](hipStream_t s) { hipLaunchKernelGGL(dma_kernel, dim3(g_wgs), dim3(512), 65536, s, g_src, g_sb, g_reps, g_sink); }, 5);
      double bps = (double)g_sb * g_reps * wgs / (ms * 1e-3);
      printf("lds-dma src=%7zu B (L2-resident) wgs=%3d : %7.2f TB/s total, %6.1f B/clk per WG (@2.0 GHz)\n", sb, wgs, bps / 1e12, bps / wgs / clk);
    }
  return 0;
}
