// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  Each lane passes the address of "its" 8-byte
// piece (piece = lane); prints which source element each (lane, elem) received.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short s[256];
  for (int i = threadIdx.x; i < 256; i += 64) s[i] = (short)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + threadIdx.x * 4));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int expect = 64 * (l >> 4) + j * 16 + (l & 15);
      if (h[l * 4 + j] != expect) ++bad;
    }
  printf("model mismatches: %d\n", bad);
  for (int l = 0; l < 20; ++l) printf("lane %2d: %3d %3d %3d %3d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
