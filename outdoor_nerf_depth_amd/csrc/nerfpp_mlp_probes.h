// Diagnostic variants of the fused MLP kernels (nerfpp_mlp.hip includes this ONLY under -DNERFPP_PROBES; the shipped
// libraries are built without it: outdoor_nerf_depth_amd/csrc/build.py).  tools/probes/variant.sh builds them:
//   tools/probes/variant.sh <name> "-DNERFPP_PROBES -DNERFPP_DBG=1"
// Most of these produce GARBAGE results by design (component removal for timing); none is reachable from build.py.
//
//   NERFPP_DBG bits: 1 drops the activation stores, 2 the saves altogether, 4 plain instead of
//     non-temporal stores, 16 drops the loader hand-off, 32 folds every activation store into a 2 MiB window of out_raw
//     (forward kernel; no HBM write traffic), 64 keeps the address math of the saves but drops the store instructions,
//     128 drops the ReLU sign words (gather + stores; bf16 forward)
//   NERFPP_DBG_NO_DMA      no weight DMA (LDS holds stale bytes)
//   NERFPP_DBG_NO_MFMA     no MFMAs (the weight fragment reads stay)
//   NERFPP_STORE_FLAVOR    1 sc1 | 2 sc0 sc1 | 3 sc1 nt | 4 sc0 sc1 nt   (default: nt)
//   NERFPP_HOOK_ORDER      0 the two waves of a SIMD save at opposite ends of a block | 1 after the MFMAs | 2 before
//   NERFPP_LDS_PREFETCH    weight fragments in flight ahead of their MFMA (default 4)
//   NERFPP_WAVES_P1        waves per workgroup of the bf16 kernels (default 8; 4 = 128-sample tiles)
//   NERFPP_LDS_REUSE=n     one weight-fragment LDS read per n MFMAs (build with NERFPP_LDS_PREFETCH=0): what would halving the
//                          LDS reads per MFMA -- 64-row waves -- buy?
//   NERFPP_CHAIN_GROUP=g   two-plane precisions: the dependent MFMA chains (3 per out-block in split-bf16) of g out-blocks interleaved
//                          (shipped: 4; 1 = one chain after the other, the round-4 order); NERFPP_LDS_PREFETCH_SPLIT: their prefetch depth
//   NERFPP_SKIP_H=mask     bit l: the bf16 training forward does not write H_l (its sign words still go out) -- garbage gradients;
//                          -1: the mask is read per launch from the environment variable NERFPP_SKIP_H_RT (nerfpp_api.hip)
//   NERFPP_LOADER_SLEEP=n  the loader wave idles 64 n cycles per weight block (does added latency cost time, or only cycles?)
//   NERFPP_STAMPS=k        per-block cycle stamps (s_memtime at arrival at / release from every block barrier, per wave) of
//                          workgroups 0-3 and 400-403 (fg tiles) of kernel instantiation k (NERFPP_MLP_PART numbering: 2 = bf16
//                          training forward, 4 = bf16 backward), kept in LDS and copied out at the end of the kernel;
//                          read back with nerfpp_probe_stamps() (tools/probes/stamps_probe.py); 1 / 3 / 5 = the split-bf16 inference
//                          forward / training forward / backward (4 waves)
#pragma once
#include <hip/hip_runtime.h>

#ifndef NERFPP_DBG
#define NERFPP_DBG 0
#endif
#ifndef NERFPP_HOOK_ORDER
#define NERFPP_HOOK_ORDER 1
#endif
#ifndef NERFPP_LDS_PREFETCH
#define NERFPP_LDS_PREFETCH 4
#endif
#ifndef NERFPP_WAVES_P1
#define NERFPP_WAVES_P1 8
#endif
#ifndef NERFPP_STORE_FLAVOR
#define NERFPP_STORE_FLAVOR 0
#endif
#ifndef NERFPP_LOADER_SLEEP
#define NERFPP_LOADER_SLEEP 0
#endif
#ifndef NERFPP_LDS_REUSE
#define NERFPP_LDS_REUSE 1
#endif
#ifndef NERFPP_SKIP_H
#define NERFPP_SKIP_H 0
#endif

namespace nerfpp { namespace probe {

constexpr int DBG = NERFPP_DBG;
#ifdef NERFPP_DBG_NO_DMA
constexpr bool NO_DMA = true;
#else
constexpr bool NO_DMA = false;
#endif
#ifdef NERFPP_DBG_NO_MFMA
constexpr bool NO_MFMA = true;
#else
constexpr bool NO_MFMA = false;
#endif
constexpr int LDS_PREFETCH = NERFPP_LDS_PREFETCH;
constexpr int HOOK_ORDER = NERFPP_HOOK_ORDER;
constexpr int WAVES_P1 = NERFPP_WAVES_P1;
constexpr int LOADER_SLEEP = NERFPP_LOADER_SLEEP;
#ifndef NERFPP_LDS_PREFETCH_SPLIT
#define NERFPP_LDS_PREFETCH_SPLIT 4
#endif
constexpr int LDS_PREFETCH_SPLIT = NERFPP_LDS_PREFETCH_SPLIT;
#ifndef NERFPP_CHAIN_GROUP
#define NERFPP_CHAIN_GROUP 4
#endif
constexpr int CHAIN_GROUP = NERFPP_CHAIN_GROUP;
#ifndef NERFPP_SKEW_INFER
#define NERFPP_SKEW_INFER 0
#endif
constexpr int SKEW_INFER = NERFPP_SKEW_INFER;
#ifndef NERFPP_EXP
#define NERFPP_EXP 0
#endif
constexpr int EXP = NERFPP_EXP;                   // timing experiments, garbage results: 1 the weight DMA is never waited for (ring pipe), 2 no block barrier (ring pipe), 4 the split-bf16 epilogue without its conversion work, 8 the training forward on the ring pipe (with NERFPP_DBG & 2: no saves), 16 zero accumulators instead of the bias reads, 32 (with 16) one extra MFMA per out-block and stage (the bias as a 17th k-chunk)
#ifndef NERFPP_TRICKLE
#define NERFPP_TRICKLE 1
#endif
constexpr int TRICKLE = NERFPP_TRICKLE;           // bit 0 / 1: the ring / roles pipe issues a block's weight DMA in pieces between the step's MFMAs
#ifndef NERFPP_UNIT_VALU
#define NERFPP_UNIT_VALU 4
#endif
constexpr int UNIT_VALU = NERFPP_UNIT_VALU;       // unit-pipelined split-bf16 kernels: VALU instructions dealt out behind each MFMA of a unit (0: the compiler's own order)
#ifndef NERFPP_V2T_NBUF
#define NERFPP_V2T_NBUF 3
#endif
#ifndef NERFPP_V2T_NBUF_BWD
#define NERFPP_V2T_NBUF_BWD 3
#endif
constexpr int V2T_NBUF = NERFPP_V2T_NBUF;         // ring slots of the unit-pipelined split-bf16 training forward (2: full drain per block; 3: counted wait, DMA in pieces)
constexpr int V2T_NBUF_BWD = NERFPP_V2T_NBUF_BWD; // ... and of the dX chain (2, 3 or 4)
#ifndef NERFPP_SPLIT_V2
#define NERFPP_SPLIT_V2 7
#endif
constexpr int SPLIT_V2 = NERFPP_SPLIT_V2;         // bit 0 / 1 / 2: the split-bf16 inference forward / training forward / backward runs the unit-pipelined body (0: the stage-at-a-time bodies)
constexpr int SKIP_H = NERFPP_SKIP_H;             // bit l: the bf16 training forward leaves H_l unsaved (what would one-layer recompute in dw_kernel buy?)
constexpr int LDS_REUSE = NERFPP_LDS_REUSE;       // 2: one weight-fragment read per two MFMAs (what 64-row waves would need); with NERFPP_LDS_PREFETCH=0

#if (NERFPP_DBG & 32)
__device__ char* dbg_sink;      // timing experiment: every activation store folded into a 2 MiB window of out_raw
__device__ __forceinline__ void kernel_prologue(float* out_raw) {
  if (threadIdx.x == 0) dbg_sink = (char*)out_raw;
  __syncthreads();
}
#else
__device__ __forceinline__ void kernel_prologue(float*) {}
#endif

__device__ __forceinline__ void store16(char* gptr, const uint4 v) {
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
  if constexpr ((NERFPP_DBG & 1) != 0) return;
#if (NERFPP_DBG & 32)
  gptr = dbg_sink + ((uintptr_t)gptr & 0x1FFFF0);
#endif
#if (NERFPP_DBG & 64)
  asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(gptr));   // keep the address math, drop the store
  return;
#endif
  if constexpr ((NERFPP_DBG & 4) != 0) { *(uint4*)gptr = v; return; }                                     // plain (temporal) store
  const u32x4_ vv = {v.x, v.y, v.z, v.w};
#if NERFPP_STORE_FLAVOR == 1
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(gptr), "v"(vv) : "memory");
#elif NERFPP_STORE_FLAVOR == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(gptr), "v"(vv) : "memory");
#elif NERFPP_STORE_FLAVOR == 3
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(gptr), "v"(vv) : "memory");
#elif NERFPP_STORE_FLAVOR == 4
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(gptr), "v"(vv) : "memory");
#else
  __builtin_nontemporal_store(vv, (u32x4_*)gptr);
#endif
}


#if defined(NERFPP_STAMPS) && defined(NERFPP_MLP_PART) && NERFPP_STAMPS == NERFPP_MLP_PART
// (split-bf16 kernels: 4 waves; their training kernels stream 8-fragment blocks -- 134 / 124 per fg tile)
#if NERFPP_STAMPS == 3 || NERFPP_STAMPS == 5
constexpr int STAMP_BLKS = 152, STAMP_WAVES = 4;
#elif NERFPP_STAMPS == 1
constexpr int STAMP_BLKS = 96, STAMP_WAVES = 4;
#else
constexpr int STAMP_BLKS = 96, STAMP_WAVES = 8;
#endif
constexpr int STAMP_WGS = 8;
constexpr int STAMP_BYTES = STAMP_WAVES * STAMP_BLKS * 2 * 4;
static __device__ uint32_t g_stamps[STAMP_WGS][STAMP_WAVES][STAMP_BLKS][2];
extern __shared__ __attribute__((aligned(16))) char probe_smem[];
__device__ __forceinline__ void stamp(int which, int blk, int wave, int lane, uint32_t lds_off) {
  if (blk >= STAMP_BLKS) return;
  const uint32_t t = (uint32_t)__builtin_readcyclecounter();
  if (lane == 0) *(uint32_t*)(probe_smem + lds_off + ((wave * STAMP_BLKS + blk) * 2 + which) * 4) = t;
}
__device__ __forceinline__ void dump_stamps(uint32_t lds_off, int wave, int lane) {
  const int b = blockIdx.x;
  const int slot = b < 4 ? b : (b >= 400 && b < 404 ? b - 396 : -1);
  if (slot < 0) return;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int i = lane; i < STAMP_BLKS * 2; i += 64)
    (&g_stamps[slot][wave][0][0])[i] = *(const uint32_t*)(probe_smem + lds_off + (wave * STAMP_BLKS * 2 + i) * 4);
}
#define NERFPP_STAMPS_READER 1
#else
constexpr int STAMP_BYTES = 0;
__device__ __forceinline__ void stamp(int, int, int, int, uint32_t) {}
__device__ __forceinline__ void dump_stamps(uint32_t, int, int) {}
#endif

}}  // namespace nerfpp::probe

#ifdef NERFPP_STAMPS_READER
extern "C" int nerfpp_probe_stamps(void* host_dst, int bytes) {
  if (bytes != (int)sizeof(nerfpp::probe::g_stamps)) return (int)sizeof(nerfpp::probe::g_stamps);
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(nerfpp::probe::g_stamps), sizeof(nerfpp::probe::g_stamps));
}
#endif
