// Infinity-Cache (MALL) retention probe (gfx950): a producer kernel writes W MB with 16-byte stores (plain or
// non-temporal), a consumer kernel then streams the same bytes into LDS with global_load_lds_dwordx4 (the path the
// weight-gradient GEMMs use), either in the producer's order or in reverse.  If the consumer's rate rises above the
// HBM ceiling (~6.3 TB/s) for small W, recently written activations can be consumed out of the 256 MiB cache.
// Build: hipcc --offload-arch=gfx950 -O3 -o mall_probe mall_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) char smem[];

constexpr size_t CHUNK = 32 * 1024;        // bytes one workgroup handles per turn (8 waves x 4 KiB)

__global__ __launch_bounds__(512) void store_kernel(char* out, size_t n_chunks, int nt) {
  const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  for (size_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    char* base = out + c * CHUNK + threadIdx.x * 16;
    for (int k = 0; k < 4; ++k) {
      if (nt) __builtin_nontemporal_store(v, (u32x4*)(base + k * 8192));
      else *(u32x4*)(base + k * 8192) = v;
    }
  }
}
// order 0: chunks ascending; 1: descending (most recently written first)
__global__ __launch_bounds__(512) void dma_kernel(const char* src, size_t n_chunks, int reverse, int nt, int* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  int turn = 0;
  for (size_t c = blockIdx.x; c < n_chunks; c += gridDim.x, ++turn) {
    const size_t cc = reverse ? n_chunks - 1 - c : c;
    const uint32_t slot = (uint32_t)(turn & 3) * CHUNK;
    for (int k = 0; k < 4; ++k) {
      const char* g = src + cc * CHUNK + k * 8192 + wave * 1024 + lane * 16;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + slot + k * 8192 + wave * 1024);
      uint32_t keep;
      if (nt)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
      else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");     // three chunks (12 instructions) in flight
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && smem[5] == 77) *sink = 1;
}

int main() {
  char* buf; int* sink;
  const size_t CAP = (size_t)3 << 30;
  hipMalloc(&buf, CAP); hipMalloc(&sink, 4);
  char* trash; hipMalloc(&trash, (size_t)1 << 30);
  hipFuncSetAttribute((const void*)dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  printf("%8s %6s %6s %8s | %10s %10s\n", "MB", "st_nt", "ld_nt", "reverse", "write TB/s", "read TB/s");
  for (size_t mb : {32, 64, 128, 192, 256, 384, 512, 1024, 2048})
    for (int st_nt = 0; st_nt < 2; ++st_nt)
      for (int ld_nt = 0; ld_nt < 2; ++ld_nt)
        for (int rev = 0; rev < 2; ++rev) {
          const size_t bytes = mb << 20, n_chunks = bytes / CHUNK;
          double wsum = 0, rsum = 0; const int reps = 4;
          for (int it = 0; it < reps + 1; ++it) {
            // evict: overwrite 1 GiB elsewhere
            hipLaunchKernelGGL(store_kernel, dim3(256), dim3(512), 0, 0, trash, ((size_t)1 << 30) / CHUNK, 0);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(store_kernel, dim3(256), dim3(512), 0, 0, buf, n_chunks, st_nt);
            hipEventRecord(e1, 0);
            hipLaunchKernelGGL(dma_kernel, dim3(256), dim3(512), 131072, 0, buf, n_chunks, rev, ld_nt, sink);
            hipEventRecord(e2, 0); hipEventSynchronize(e2);
            float w, r; hipEventElapsedTime(&w, e0, e1); hipEventElapsedTime(&r, e1, e2);
            if (it) { wsum += w; rsum += r; }
          }
          printf("%8zu %6d %6d %8d | %10.2f %10.2f\n", mb, st_nt, ld_nt, rev, bytes / (wsum / reps * 1e-3) / 1e12,
                 bytes / (rsum / reps * 1e-3) / 1e12);
        }
  return 0;
}
