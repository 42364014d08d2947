// Does the matrix pipe need time to reach full rate after an idle gap?  Every CU runs 8 waves (2 per SIMD) that
// alternate a burst of bf16 MFMAs with a pause of P microseconds (s_sleep); wave 0 of each workgroup records the
// wall-clock time (s_memrealtime, 100 MHz) of every eighth of a burst.  Output: ns per MFMA per SIMD in each eighth,
// averaged over bursts and workgroups, against the pause length.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int EIGHTH = 128;                   // MFMAs per wave per eighth (2 waves per SIMD -> 256 x 32 cycles = 4 us at 2 GHz)
constexpr int BURSTS = 12;

__global__ __launch_bounds__(512) void ramp_kernel(int pause_sleeps, int warm, uint64_t* stamps, float* sink) {
  const int wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(0.01f * (threadIdx.x + k)); b[k] = (__bf16)(0.02f * (threadIdx.x - k)); }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  uint64_t* out = stamps + (size_t)blockIdx.x * BURSTS * 9;
  for (int burst = 0; burst < BURSTS; ++burst) {
    __syncthreads();
    for (int e = 0; e < 8; ++e) {
      if (wave == 0 && (threadIdx.x & 63) == 0) out[burst * 9 + e] = __builtin_readcyclecounter() * 0 + __builtin_amdgcn_s_memrealtime();
#pragma unroll 8
      for (int i = 0; i < EIGHTH / 4; ++i) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[3], 0, 0, 0);
      }
    }
    if (wave == 0 && (threadIdx.x & 63) == 0) out[burst * 9 + 8] = __builtin_amdgcn_s_memrealtime();
    for (int p = 0; p < pause_sleeps; ++p) {                                   // 32 x 64 cycles ~ 1 us at 2 GHz per iteration
      if (warm == 0) __builtin_amdgcn_s_sleep(32);
      else {                                                                   // `warm` MFMAs per wave per microsecond of pause (1 us = 64 MFMA slots per SIMD)
        for (int q = 0; q < warm; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q & 3], 0, 0, 0);
        __builtin_amdgcn_s_sleep(24);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) sink[threadIdx.x] = s;
}

int main() {
  const int wgs = 256;
  uint64_t* d; float* sink;
  hipMalloc(&d, sizeof(uint64_t) * wgs * BURSTS * 9); hipMalloc(&sink, 4096);
  std::vector<uint64_t> h(wgs * BURSTS * 9);
  printf("pause_us / MFMAs per wave per us of pause | ns per MFMA per SIMD in each eighth of a burst (2 waves per SIMD; 16.0 = 32 cycles at 2.0 GHz)\n");
  const int cases[][2] = {{0, 0}, {1, 0}, {2, 0}, {4, 0}, {8, 0}, {16, 0}, {32, 0}, {64, 0}, {8, 1}, {8, 2}, {8, 4}, {8, 8}, {8, 16}};
  for (auto& cs : cases) {
    const int pause = cs[0], warm = cs[1];
    hipLaunchKernelGGL(ramp_kernel, dim3(wgs), dim3(512), 0, 0, pause, warm, d, sink);
    hipLaunchKernelGGL(ramp_kernel, dim3(wgs), dim3(512), 0, 0, pause, warm, d, sink);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double e8[8] = {0};
    int n = 0;
    for (int w = 0; w < wgs; ++w)
      for (int b = 2; b < BURSTS; ++b) {                   // skip the first two bursts
        const uint64_t* t = &h[(size_t)(w * BURSTS + b) * 9];
        for (int e = 0; e < 8; ++e) e8[e] += (double)(t[e + 1] - t[e]) * 10.0;       // 100 MHz ticks -> ns
        ++n;
      }
    printf("%5d/%2d |", pause, warm);
    for (int e = 0; e < 8; ++e) printf(" %6.2f", e8[e] / n / (2.0 * EIGHTH));
    printf("\n");
  }
  return 0;
}
