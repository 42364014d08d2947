// Diagnostic variants of linear_fm_kernel (mip360_fm.hip includes this ONLY under -DNERFPP_PROBES, after its glds16();
// DESIGN.md section 9.1).  Component removal produces GARBAGE results by design; nothing here is reachable from build.py.
//   FM_EXP_NOSTORE / FM_EXP_HALFDMA / FM_EXP_NODMA / FM_EXP_NOREAD / FM_EXP_NOMFMA   component removal
//   FM_GLDS_MODE   1: M0 left clobbered (no save / restore) | 2: two blocks per M0 value through the instruction offset
//   FM_NSLOT       LDS ring slots (4 or 5)
//   FM_ST_FLAVOR   output stores: 0 default | 1 nt | 2 sc1 | 3 sc0 sc1 | 4 sc1 nt
//   FM_SETPRIO     s_setprio around the MFMA bursts
//   FM_STAGGER     start-up classes per XCD, FM_STAGGER x 3.4 us apart
#pragma once
#ifndef FM_GLDS_MODE
#define FM_GLDS_MODE 0
#endif
#ifndef FM_NSLOT
#define FM_NSLOT 5
#endif
#ifndef FM_ST_FLAVOR
#define FM_ST_FLAVOR 1
#endif
#ifndef FM_STAGGER
#define FM_STAGGER 0
#endif
#if FM_ST_FLAVOR == 1
#define FM_ST " nt"
#elif FM_ST_FLAVOR == 2
#define FM_ST " sc1"
#elif FM_ST_FLAVOR == 3
#define FM_ST " sc0 sc1"
#elif FM_ST_FLAVOR == 4
#define FM_ST " sc1 nt"
#else
#define FM_ST ""
#endif
namespace mip360fm { namespace probe {
constexpr int NSLOT = FM_NSLOT, STAGGER = FM_STAGGER;
#ifdef FM_EXP_NOSTORE
constexpr int NSTORE = 0;
#else
constexpr int NSTORE = 16;
#endif
#if defined(FM_EXP_HALFDMA)
constexpr int DMA_PER = 2;
#elif defined(FM_EXP_NODMA)
constexpr int DMA_PER = 0;
#else
constexpr int DMA_PER = 4;
#endif
#ifdef FM_EXP_NOREAD
constexpr bool NOREAD = true;
#else
constexpr bool NOREAD = false;
#endif
#ifdef FM_EXP_NOMFMA
constexpr bool NOMFMA = true;
#else
constexpr bool NOMFMA = false;
#endif
#ifdef FM_SETPRIO
constexpr bool SETPRIO = true;
#else
constexpr bool SETPRIO = false;
#endif
__device__ __forceinline__ void glds16_m0(const void* sbase, uint32_t voff, uint32_t lds_abs) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_abs);
  const uint64_t b = (uint64_t)(uintptr_t)sbase;
  const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(base), "s"(dst) : "memory", "m0");
}
__device__ __forceinline__ void glds16_pair(const void* sbase, uint32_t voff, uint32_t lds_abs) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_abs);
  const uint64_t b = (uint64_t)(uintptr_t)sbase;
  const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024"
               :: "v"(voff), "s"(base), "s"(dst) : "memory", "m0");
}
__device__ __forceinline__ void issue_half_step(const char* iA, const char* iW, uint32_t voff, uint32_t d, uint32_t woff) {
#if defined(FM_EXP_HALFDMA)
  glds16(iA, voff, d);
  glds16(iA + 1024, voff, d + 1024u);
#elif defined(FM_EXP_NODMA)
  (void)d;
#elif FM_GLDS_MODE == 2
  glds16_pair(iA, voff, d);
  glds16_pair(iW, voff, d + woff);
#elif FM_GLDS_MODE == 1
  glds16_m0(iA, voff, d);
  glds16_m0(iA + 1024, voff, d + 1024u);
  glds16_m0(iW, voff, d + woff);
  glds16_m0(iW + 1024, voff, d + woff + 1024u);
#else
  glds16(iA, voff, d);
  glds16(iA + 1024, voff, d + 1024u);
  glds16(iW, voff, d + woff);
  glds16(iW + 1024, voff, d + woff + 1024u);
#endif
}
}}  // namespace mip360fm::probe
