// HBM write rate of the training kernels' SAVE pattern in isolation (no compute): 256 workgroups x 8 waves, every wave
// owns 32-row tiles and writes, per "layer" (8 tensors of [rows][256] bf16), its 16 one-KiB chunk blocks.
//   mode 0: fragment-major (round 3): block (tile32, c) is 1 KiB contiguous; a wave's 16 blocks of a layer are 16 KiB contiguous
//   mode 1: row-major 128-byte segments (rounds 1-2): a wave-store = 8 rows x 128 B at 512-byte row stride, 4 column passes
//   mode 2: plain streaming (each wave a private contiguous region), the fill ceiling
// Build: hipcc --offload-arch=gfx950 -O3 -o store_pattern_probe store_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(char* out, size_t rows, int ntile256, int mode, int nt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  const size_t tensor_bytes = rows * 512;
  for (int t = blockIdx.x; t < ntile256; t += gridDim.x) {
    const size_t tile32 = (size_t)t * 8 + wave;
    for (int l = 0; l < 8; ++l) {
      char* T = out + (size_t)l * tensor_bytes;
      for (int c = 0; c < 16; ++c) {
        char* p;
        if (mode == 0) p = T + (tile32 * 16 + c) * 1024 + lane * 16;
        else if (mode == 1) {   // pass = c / 4 (128-byte column segment), it = c % 4: rows {4it..4it+3, 16+4it..}
          const int row = (lane >> 5) * 2 + ((lane >> 3) & 1) + 16 * ((lane >> 4) & 1) + 4 * (c & 3);
          p = T + (tile32 * 32 + row) * 512 + (c >> 2) * 128 + (lane & 7) * 16;
        } else p = out + ((((size_t)t * 8 + wave) * 8 + l) * 16 + c) * 1024 + lane * 16;
        if (nt) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
      }
    }
  }
}
int main() {
  const size_t rows = (size_t)8192 * 192;            // N_rand 8192, level 1, one net
  char* out; (void)hipMalloc(&out, rows * 512 * 8);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int ntile = (int)(rows / 256);
  for (int mode = 0; mode < 3; ++mode)
    for (int nt = 0; nt < 2; ++nt) {
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, rows, ntile, mode, nt);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(a, 0);
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, rows, ntile, mode, nt);
      (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 5;
      printf("mode %d nt %d : %.3f ms for %.2f GB = %.2f TB/s\n", mode, nt, ms, rows * 4096.0 / 1e9, rows * 4096.0 / (ms * 1e-3) / 1e12);
    }
  return 0;
}
