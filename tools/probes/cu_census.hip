// Census for CU-masked streams (tools/probes/cu_mask_probe.py): every workgroup records the XCC id and the HW_ID register
// of its first wave, so that the probe can print which CUs a hipExtStreamCreateWithCUMask mask selects.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probes/cu_census.hip -o tools/probes/libcu_census.so
#include <hip/hip_runtime.h>
#include <stdint.h>
// s_getreg_b32 immediate: (size - 1) << 11 | offset << 6 | register id   (HW_ID = 4, XCC_ID = 20 on gfx94x / gfx950)
#define GETREG(id, off, size) __builtin_amdgcn_s_getreg((((size) - 1) << 11) | ((off) << 6) | (id))
__global__ void census_kernel(uint32_t* out, int spin) {
  if (threadIdx.x == 0) {
    const uint32_t hw = GETREG(4, 0, 32), xcc = GETREG(20, 0, 4);
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  // hold the CU for a while so that the blocks of one launch spread over every CU the mask allows
  const uint64_t t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (uint64_t)spin) __builtin_amdgcn_s_sleep(8);
}
extern "C" int cu_census(void* stream, int n_blocks, uint32_t* out_dev, int spin_cycles) {
  hipLaunchKernelGGL(census_kernel, dim3(n_blocks), dim3(64), 65536, (hipStream_t)stream, out_dev, spin_cycles);
  return (int)hipGetLastError();
}
extern "C" int cu_mask_stream(void** stream_out, int n_words, const uint32_t* mask) {
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask);
  *stream_out = (void*)s;
  return (int)e;
}
