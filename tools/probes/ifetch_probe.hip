// Is straight-line code instruction-fetch bound?  The unit-pipelined split-bf16 MLP kernels are ~300 KB of straight-line code
// per tile (64 KB of instruction cache per CU pair): every tile streams the whole kernel through the cache.  Here: a pass over
// 8 DIFFERENT 32 KiB blocks (256 KiB: never warm) against 8 trips through ONE 32 KiB block (warm after the first), one workgroup
// of four waves per CU (160 KiB of LDS requested), for VALU-only code and for MFMA + n VALU mixes (4 independent accumulators).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/ifetch_probe tools/probes/ifetch_probe.hip && tools/probes/ifetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define PASSES 4
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// one 32 KiB block.  KIND 0: 4096 8-byte VALU.  KIND n > 0: groups of 4 MFMAs (independent accumulators), each followed by n 8-byte VALU
#define MF(n) ".rept " #n "\n v_mov_b32_e64 v1, v2\n .endr\n"
#define GROUP(n) "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n" MF(n) "v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n" MF(n) \
                 "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n" MF(n) "v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n" MF(n)
template <int KIND>
__device__ __forceinline__ void block(f32x16 (&acc)[4], bf16x8 a, bf16x8 b) {
  if constexpr (KIND == 0) asm volatile(".rept 4096\n v_mov_b32_e64 v1, v2\n .endr" ::: "v1");
  if constexpr (KIND == 4)  asm volatile(".rept 205\n" GROUP(4) ".endr" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]) : "v"(a), "v"(b) : "v1");   // 40 B per MFMA
  if constexpr (KIND == 6)  asm volatile(".rept 146\n" GROUP(6) ".endr" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]) : "v"(a), "v"(b) : "v1");   // 56 B
  if constexpr (KIND == 10) asm volatile(".rept 93\n" GROUP(10) ".endr" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]) : "v"(a), "v"(b) : "v1");   // 88 B
}
#define RF(n) ".rept " #n "\n v_accvgpr_read_b32 v1, a255\n .endr\n"
#define RGROUP(n) "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n" RF(n) "v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n" RF(n) \
                  "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n" RF(n) "v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n" RF(n)
template <int KIND>
__device__ __forceinline__ void rblock(f32x16 (&acc)[4], bf16x8 a, bf16x8 b) {
  if constexpr (KIND == 100) asm volatile(".rept 4096\n v_accvgpr_read_b32 v1, a255\n .endr" ::: "v1", "a255");
  if constexpr (KIND == 104) asm volatile(".rept 205\n" RGROUP(4) ".endr" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]) : "v"(a), "v"(b) : "v1", "a255");
  if constexpr (KIND == 106) asm volatile(".rept 146\n" RGROUP(6) ".endr" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]) : "v"(a), "v"(b) : "v1", "a255");
  if constexpr (KIND == 110) asm volatile(".rept 93\n" RGROUP(10) ".endr" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]) : "v"(a), "v"(b) : "v1", "a255");
}
template <int KIND> constexpr int mfmas() { return KIND % 100 == 4 ? 820 : KIND % 100 == 6 ? 584 : KIND % 100 == 10 ? 372 : 0; }

template <int KIND, bool COLD>
__global__ __launch_bounds__(256) void k(uint32_t* out) {
  f32x16 acc[4] = {};
  bf16x8 a = {0}, b = {0};
  for (int it = 0; it < PASSES; ++it) {
    const uint32_t t0 = (uint32_t)__builtin_readcyclecounter();
    if constexpr (KIND >= 100) {
#pragma nounroll
      for (int r = 0; r < 8; ++r) rblock<KIND>(acc, a, b);
    } else if constexpr (COLD) {
      block<KIND>(acc, a, b); block<KIND>(acc, a, b); block<KIND>(acc, a, b); block<KIND>(acc, a, b);
      block<KIND>(acc, a, b); block<KIND>(acc, a, b); block<KIND>(acc, a, b); block<KIND>(acc, a, b);
    } else {
#pragma nounroll
      for (int r = 0; r < 8; ++r) block<KIND>(acc, a, b);
    }
    const uint32_t t1 = (uint32_t)__builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * PASSES + it] = t1 - t0;
  }
  if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 123.f) out[0] = 0;
}

template <int KIND, bool COLD>
static void run(const char* what, int grid) {
  uint32_t* d;
  (void)hipMalloc(&d, (size_t)grid * 4 * PASSES * 4);
  (void)hipFuncSetAttribute((const void*)k<KIND, COLD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, COLD>), dim3(grid), dim3(256), 160 * 1024, 0, d);
  (void)hipDeviceSynchronize();
  std::vector<uint32_t> h((size_t)grid * 4 * PASSES);
  (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  const int per = KIND % 100 == 0 ? 8 * 4096 : 8 * mfmas<KIND>();
  printf("%-34s %-22s grid %5d: cycles per %s, passes 0..%d:", what, COLD ? "8 x 32 KiB straight" : "one 32 KiB block x 8", grid, KIND % 100 == 0 ? "instr" : "MFMA", PASSES - 1);
  for (int it = 0; it < PASSES; ++it) {
    std::vector<uint32_t> v;
    for (int w = 0; w < grid * 4; ++w) v.push_back(h[(size_t)w * PASSES + it]);
    std::sort(v.begin(), v.end());
    printf("  %.2f", (double)v[v.size() / 2] / per);
  }
  printf("\n");
  (void)hipFree(d);
}

int main() {
  run<100, false>("v_accvgpr_read only", 256);
  run<104, false>("MFMA + 4 v_accvgpr_read", 256);
  run<106, false>("MFMA + 6 v_accvgpr_read", 256);
  run<110, false>("MFMA + 10 v_accvgpr_read", 256);
  for (int grid : {256}) {
    run<0, false>("VALU only (8-byte)", grid);  run<0, true>("VALU only (8-byte)", grid);
    run<4, false>("MFMA + 4 VALU", grid);       run<4, true>("MFMA + 4 VALU", grid);
    run<6, false>("MFMA + 6 VALU", grid);       run<6, true>("MFMA + 6 VALU", grid);
    run<10, false>("MFMA + 10 VALU", grid);     run<10, true>("MFMA + 10 VALU", grid);
  }
  return 0;
}
