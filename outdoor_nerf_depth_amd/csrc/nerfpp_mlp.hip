// Fused NeRF++ MLP kernels for gfx950 (see nerfpp_common.h for the execution model).
//
//   mlp_fwd_kernel : positional encoding (+ inverted-sphere parametrisation for the background)
//                    -> 8x256 trunk with skip -> sigma / remap / colour heads, activations chained
//                    in registers, weights streamed L2 -> LDS (global_load_lds, 16-fragment blocks)
//                    -> v_mfma_f32_32x32x16_bf16.
//                    Reference: nerf_network.py:42-60,120-142; ddp_model.py:16-45,86-94,107-120.
//   mlp_bwd_kernel : the dX chain of the same network (closed-form backward of the above; the
//                    reference relies on autograd), again chained in registers; writes the dZ tensors
//                    for the weight-gradient GEMMs (nerfpp_dw.hip; in bf16 all but dZ7, which its job recomputes,
//                    as the forward leaves H0 to the job that needs it).
//
// Precision P: 1 = single-pass bf16 operands / f32 accumulate ("speed" mode);
//              2 = split-bf16 (x = hi + lo, 3 MFMA passes hi*hi + hi*lo + lo*hi), which holds
//                  ~1e-5 relative error against the float32 reference ("parity" mode);
//              3 = (forward kernels only) fp16x2w: weights hi + lo in fp16, activations rounded to fp16 once, 2 passes of
//                  v_mfma_f32_32x32x16_f16 -- an intermediate precision (nerfpp_common.h: precision ids).
//
// Split-bf16 (P = 2) since round 6: the kernels that SHIP are mlp_fwd_body_split / mlp_bwd_body_split (nerfpp_mlp_split.h: 12-MFMA
// units read one unit ahead, lazily converted epilogue, ring pipe without wave roles in training) -- same arithmetic, bit-identical
// results.  The P = 2 paths of mlp_fwd_body / mlp_bwd_body below are their stage-at-a-time predecessors: compiled only into the
// diagnostic build (-DNERFPP_PROBES -DNERFPP_SPLIT_V2=0) that tools/probes/split_dump.py compares the shipped kernels against.
//
// Weight-pipe modes (WeightPipe<P, NW, MODE, NBUF, BF>): blocks of BF fragments x P planes (16 KiB in the bf16 training kernels)
// through an LDS ring, four fragments per DMA set-up (glds16xN_saddr), COUNTED vmcnt waits, one raw s_barrier per block.
//   PIPE_RING    : inference forward (no stores in flight): every wave fetches its share of a block; all outstanding VMEM
//                  ops of a wave are same-type loads, in order, so "at most k blocks' worth outstanding" is exact.
//   PIPE_ROLES   : training kernels (bf16 and, since round 4, split-bf16).  Activation stores and the weight DMA share
//                  vmcnt and may retire out of order, so a wave that stores can only wait for its DMA with a full
//                  drain -- which serialises "compute" and "write 16 KB per wave" at every layer.
//                  Here wave 0 (the LOADER) is the only wave that issues and waits for the DMA, and it never stores: it
//                  hands its tile to helper waves through an LDS region, and they write it out after the next barrier.
//                  The other waves never touch vmcnt, so their stores drain under the following MFMA work.  Biases come
//                  from LDS and (backward) the ReLU sign words are DMA'd into LDS by the loader, so no wave issues a
//                  global LOAD in steady state either.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <type_traits>
#include <utility>
#include "nerfpp_common.h"
#include "nerfpp_kernels.h"

// Experiment switches (component removal, store cache policies, tile sizes: DESIGN.md sections 6-6.3) exist only in
// diagnostic builds: -DNERFPP_PROBES pulls their variants in from nerfpp_mlp_probes.h (tools/probes/variant.sh).  The
// shipped translation units see the constants and the one store flavour below and nothing else.
#ifdef NERFPP_PROBES
#include "nerfpp_mlp_probes.h"
#else
namespace nerfpp { namespace probe {
constexpr int DBG = 0;                 // component-removal bits (probes only)
constexpr bool NO_DMA = false, NO_MFMA = false;
constexpr int LDS_PREFETCH = 4;        // weight fragments in flight ahead of the MFMA that consumes them
constexpr int LDS_PREFETCH_SPLIT = 4;  // two-plane precisions: two-plane weight fragments in flight ahead of their 3 (2) MFMAs
constexpr int CHAIN_GROUP = 4;         // two-plane precisions: out-blocks whose dependent MFMA chains are interleaved (1 = one after the other)
constexpr int HOOK_ORDER = 1;          // saves issued after a block's MFMAs by every wave
constexpr int WAVES_P1 = 8;            // waves per workgroup of the bf16 kernels (256-sample tiles)
constexpr int LOADER_SLEEP = 0;        // (probes: idle cycles / 64 added to the loader wave per weight block)
constexpr int LDS_REUSE = 1;           // (probes: MFMAs per weight-fragment read)
constexpr int SKIP_H = 0;              // (probes: bit l = the training forward does not write H_l out)
constexpr int SKEW_INFER = 0;          // weight blocks by which waves NW/2.. lag waves 0..NW/2-1 (inference forward, bf16)
constexpr int EXP = 0;                 // (probes: timing experiments with garbage results -- nerfpp_mlp_probes.h)
constexpr int TRICKLE = 1;             // bit 0 / 1: ring / roles pipe issues a block's weight DMA in pieces between the MFMAs of the step
constexpr int UNIT_VALU = 4;           // unit-pipelined split-bf16 kernels: VALU instructions dealt out behind each MFMA of a unit (0: the compiler's own order)
constexpr int V2T_NBUF = 3;            // ring slots of the unit-pipelined split-bf16 training forward (nerfpp_mlp_split.h)
constexpr int V2T_NBUF_BWD = 3;        // ... and of the dX chain
constexpr int SPLIT_V2 = 7;            // bit 0 / 1 / 2: the split-bf16 inference forward / training forward / backward runs the unit-pipelined body (nerfpp_mlp_split.h)
__device__ __forceinline__ void store16(char* gptr, const uint4 v) {       // activation saves: non-temporal 16-byte stores
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
  const u32x4_ vv = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(vv, (u32x4_*)gptr);
}
__device__ __forceinline__ void kernel_prologue(float*) {}
constexpr int STAMP_BYTES = 0;         // per-block cycle stamps (probes only)
__device__ __forceinline__ void stamp(int, int, int, int, uint32_t) {}
__device__ __forceinline__ void dump_stamps(uint32_t, int, int) {}
}}  // namespace nerfpp::probe
#endif

namespace nerfpp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define LDS_AS __attribute__((address_space(3)))

// one activation chunk of a lane: a_planes(P) 16-byte register images (precision 3: fp16 bits in the bf16x8 container)
template <int P> struct Frag { bf16x8 v[a_planes(P)]; };

extern __shared__ __attribute__((aligned(16))) char smem[];

enum { PIPE_RING = 1, PIPE_ROLES = 2 };

__device__ __forceinline__ uint32_t lds_base_addr() { return (uint32_t)(uintptr_t)(LDS_AS char*)smem; }
// N (<= 4) consecutive 1 KiB fragments with ONE M0 / address set-up: the instruction offset advances the global and the
// LDS address alike.  Global address = wave-uniform SGPR base + per-lane byte offset (lane * 16): no 64-bit VALU add, no
// M0 save / restore per fragment (round 4: the loader wave of the roles pipe was the last wave at 3 of 4 block barriers,
// with 7 instructions per DMA -- profiles/r04_block_stamps.md).
template <int N>
__device__ __forceinline__ void glds16xN_saddr(const char* sbase, uint32_t voff, uint32_t lds_abs) {
  static_assert(N >= 1 && N <= 4, "the 13-bit instruction offset reaches 3 x 1024");
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_abs);
  const uint64_t b = (uint64_t)(uintptr_t)sbase;
  const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
  uint32_t keep;                         // M0 is saved / restored around the group (the compiler does not track it)
#define GLDS_HEAD "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
#define GLDS_TAIL "\n\ts_mov_b32 m0, %0"
  if constexpr (N == 4)
    asm volatile(GLDS_HEAD "\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:3072" GLDS_TAIL : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
  else if constexpr (N == 3)
    asm volatile(GLDS_HEAD "\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048" GLDS_TAIL
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
  else if constexpr (N == 2)
    asm volatile(GLDS_HEAD "\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" GLDS_TAIL : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
  else
    asm volatile(GLDS_HEAD GLDS_TAIL : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
#undef GLDS_HEAD
#undef GLDS_TAIL
}

// fragments per weight block: the split-bf16 training kernels use 8 (x 2 planes = the same 16 KiB per block as bf16), which
// buys a 4-deep ring next to the doubled hand-off region and encoded-point stash in 160 KiB of LDS
template <int P, bool TRAIN>
constexpr int blk_frags_of() { return (P >= 2 && TRAIN) ? 8 : BLK_FRAGS; }
// the split-bf16 training forward in its unit-pipelined form (nerfpp_mlp_split.h): ring pipe, 2 slots of 16-fragment blocks, no roles
template <int P, bool TRAIN>
constexpr bool split_v2_train() { return P == 2 && TRAIN && (probe::SPLIT_V2 & 2) != 0; }

// SKEW > 0: the waves NW/2.. ("lagging") consume block t - SKEW in the step in which the waves 0..NW/2-1 consume block t (one wave
// of each half per SIMD): a stage's epilogue -- conversion VALU with nothing for the matrix pipe -- of one half then coincides
// with MFMA blocks of the other half instead of with its epilogue.  Every wave takes part in every step (barrier, DMA issue):
// the lagging waves run SKEW empty steps first (lead_in), the others SKEW empty steps last (lead_out); a slot is re-filled
// SKEW steps later than without skew, i.e. the ring runs NBUF - 1 - SKEW blocks ahead.
template <int P, int NW, int MODE, int NBUF, int BF = BLK_FRAGS, int SKEW = 0>
struct WeightPipe {
  static constexpr int BLKF = BF;
  static constexpr int BLK_BYTES = BF * P * FRAG_BYTES;
  // DMA wave-instructions per block of the waves that wait for it (roles: the loader issues them all)
  static constexpr int PER_BLK = MODE == PIPE_ROLES ? BF * P : BF * P / NW;
  static constexpr int AHEAD = NBUF - 1 - SKEW;                // blocks in flight ahead of the step's block
  static_assert(AHEAD >= 1 && AHEAD <= 3 && (AHEAD - 1) * PER_BLK < 63 && SKEW >= 0, "ring depth");
  const char* g;
  uint32_t stamp_off = 0;                                      // (probes: LDS offset of the cycle stamps)
  int nblk, cur, step, wave, lane;                             // cur: blocks this wave has consumed, step: barriers passed
  int slot_cur, slot_issue, next_issue;                        // ring positions (NBUF need not be a power of 2)
  uint32_t lds_base;
  __device__ __forceinline__ void init(const void* stream, int nblk_, int wave_, int lane_) {
    g = (const char*)stream; nblk = nblk_; cur = 0; step = 0; wave = wave_; lane = lane_;
    slot_cur = 0; slot_issue = 0; next_issue = 0;
    lds_base = lds_base_addr();
#pragma unroll
    for (int b = 0; b < AHEAD; ++b) issue();
  }
  // fragments per block this wave issues (roles: the loader issues them all, the others none)
  static constexpr int SHARE = MODE == PIPE_ROLES ? BF * P : BF * P / NW;
  // (not with a 2-slot ring: the block is waited for one step after its DMA goes out and needs the whole step to land --
  // measured on the unit-pipelined split-bf16 training forward: 1.31 ms at once, 1.37 ms in pieces)
  static constexpr bool TRICKLE = ((MODE == PIPE_RING ? probe::TRICKLE : probe::TRICKLE >> 1) & 1) != 0 && NBUF - 1 - SKEW >= 2;
  int pend_blk = -1, pend_slot = 0;                            // TRICKLE: the block whose DMA the current step issues piecewise
  // fragments [f0, f1) of this wave's share of block `blk` -> ring slot `slot`, up to four per M0 / address set-up
  __device__ __forceinline__ void issue_frags(int blk, int slot, int f0, int f1) {
    if constexpr (probe::NO_DMA) return;
    if (MODE == PIPE_ROLES && wave != 0) return;               // the loader wave issues the whole block
    const int w0 = MODE == PIPE_ROLES ? 0 : wave * SHARE;
    const char* sb = g + (size_t)blk * BLK_BYTES + (size_t)w0 * FRAG_BYTES;
    const uint32_t dst = lds_base + slot * BLK_BYTES + w0 * FRAG_BYTES;
    if constexpr (MODE == PIPE_ROLES && probe::LOADER_SLEEP > 0) { if (f0 == 0) __builtin_amdgcn_s_sleep(probe::LOADER_SLEEP); }
#pragma unroll
    for (int f = f0; f < f1; f += 4) {
      const int n = f1 - f < 4 ? f1 - f : 4;                   // (compile-time after unrolling)
      const char* a = sb + f * FRAG_BYTES;
      const uint32_t d = dst + f * FRAG_BYTES;
      if (n == 4) glds16xN_saddr<4>(a, (uint32_t)lane * 16u, d);
      else if (n == 3) glds16xN_saddr<3>(a, (uint32_t)lane * 16u, d);
      else if (n == 2) glds16xN_saddr<2>(a, (uint32_t)lane * 16u, d);
      else glds16xN_saddr<1>(a, (uint32_t)lane * 16u, d);
    }
  }
  // claim the next block of the stream and the next ring slot; false past the end of the stream
  __device__ __forceinline__ bool claim(int& blk, int& slot) {
    blk = next_issue; slot = slot_issue;
    ++next_issue;
    slot_issue = slot_issue + 1 == NBUF ? 0 : slot_issue + 1;
    return blk < nblk;
  }
  __device__ __forceinline__ void issue() {                    // next block of the stream -> next ring slot, all at once
    int blk, slot;
    if (!claim(blk, slot)) return;
    static_assert(SHARE == 2 || SHARE % 4 == 0, "per-wave fragment count");
    issue_frags(blk, slot, 0, SHARE);
  }
  // TRICKLE: piece i of n of the block claimed at this step's barrier (call n times per step, between the step's MFMAs: a
  // DMA instruction that follows another one back to back waits for the texture addresser to take its predecessor)
  __device__ __forceinline__ void trickle(int i, int n) {
    if constexpr (TRICKLE) {
      if (pend_blk < 0) return;
      issue_frags(pend_blk, pend_slot, SHARE * i / n, SHARE * (i + 1) / n);
    }
  }
  // wait until block `step` has landed, leaving up to AHEAD-1 younger blocks in flight.  Valid because
  // every VMEM op the waiting wave has outstanding is a load (they retire in order); extra loads in
  // between (sign words, rays) only make the count conservative.
  __device__ __forceinline__ void wait_counted() {
    if constexpr ((probe::EXP & 1) != 0) return;                 // (probes: the weight DMA is never waited for)
    const int younger = nblk - 1 - step < AHEAD - 1 ? nblk - 1 - step : AHEAD - 1;
    if (AHEAD >= 3 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_BLK) : "memory");
    else if (AHEAD >= 2 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_BLK) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // one step: block `step` readable by everybody, start fetching into the slot the barrier freed
  __device__ __forceinline__ void sync_step() {
    probe::stamp(0, cur, wave, lane, stamp_off);                 // arrival at the block boundary
    if constexpr (MODE == PIPE_RING) {
      wait_counted();
      if constexpr ((probe::EXP & 2) == 0) __builtin_amdgcn_s_barrier();
    } else {
      // only the loader has DMA to wait for; everybody: LDS writes of the hand-off region must have landed
      if (wave == 0) wait_counted();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    probe::stamp(1, cur, wave, lane, stamp_off);                 // released by the barrier
    if constexpr (TRICKLE) { int b_, s_; pend_blk = claim(b_, s_) ? b_ : -1; pend_slot = s_; }
    else issue();
    ++step;
  }
  __device__ __forceinline__ bool lagging() const { return SKEW > 0 && wave >= NW / 2; }
  __device__ __forceinline__ void lead_in() {
    if constexpr (SKEW > 0) if (lagging()) for (int s_ = 0; s_ < SKEW; ++s_) sync_step();
  }
  __device__ __forceinline__ void lead_out() {
    if constexpr (SKEW > 0) if (!lagging()) for (int s_ = 0; s_ < SKEW; ++s_) sync_step();
  }
  // make this wave's next block readable, return its LDS address
  __device__ __forceinline__ const char* acquire() {
    sync_step();
    const char* l = smem + slot_cur * BLK_BYTES + lane * 16;
    slot_cur = slot_cur + 1 == NBUF ? 0 : slot_cur + 1;
    ++cur;
    return l;
  }
};

template <int P>
__device__ __forceinline__ void mfma_p(f32x16& acc, const char* lfrag, const Frag<P>& b) {
  const bf16x8 a_hi = *(const bf16x8*)(lfrag);
  if constexpr (probe::NO_MFMA) { asm volatile("" ::"v"(a_hi)); return; }
  if constexpr (P == 3) {          // fp16: (Wh + Wl) * A, the activation rounded once
    const bf16x8 a_lo = *(const bf16x8*)(lfrag + FRAG_BYTES);
    const f16x8 bb = __builtin_bit_cast(f16x8, b.v[0]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_hi), bb, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_lo), bb, acc, 0, 0, 0);
    return;
  }
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b.v[0], acc, 0, 0, 0);
  if constexpr (P == 2) {
    const bf16x8 a_lo = *(const bf16x8*)(lfrag + FRAG_BYTES);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b.v[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b.v[0], acc, 0, 0, 0);
  }
}

struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
#define HOOK(...) [&](int blk) __attribute__((always_inline)) { __VA_ARGS__; }
#define IC(n) std::integral_constant<int, (n)>{}

// acc[ob] += W_stage[ob-block, :] * B   for one stage of NKC k-chunks x NOB out-blocks.
// `hook(blk)` runs in every block, after the block's barrier (before or after the block's MFMAs).  The bf16
// training kernels use it to write out the PREVIOUS stage's output (= this stage's B operand, still in
// registers) one quarter-tile per block pair, so the LDS transpose and the global stores sit in the
// shadow of this stage's MFMAs instead of in an epilogue where every wave of the CU idles the matrix
// pipe at once; wave 1 flushes the loader's tile there and the loader queues the next sign-word DMA.
// LIVE < NKC: the stage's k-chunks from LIVE on are block-alignment padding whose operand fragments are zero (the 3 colour
// gradients occupy one chunk of the 4 that make B0 a whole block, the colour head's 18 live chunks are padded to 20, ...): their
// weight fragments travel with the block, their MFMAs are not issued.
template <int NOB, int NKC, int P, int LIVE = NKC, typename Pipe, typename Hook>
__device__ __forceinline__ void stage_gemm(Pipe& pipe, f32x16 (&acc)[NOB], const Frag<P> (&b)[NKC], const Hook& hook) {
  constexpr int KPB = Pipe::BLKF / NOB;           // k-chunks per block
  static_assert(NKC % KPB == 0, "stage must be block aligned");
  static_assert(LIVE >= 1 && LIVE <= NKC, "live k-chunks");
#pragma unroll
  for (int blk = 0; blk < NKC / KPB; ++blk) {
    const char* l = pipe.acquire();
    // Where the hook (the saves) runs in a block.  With the LDS-staged saves of rounds 1-2 the two waves of a SIMD ran it at
    // opposite ends (0), so that one always had MFMAs to issue while the other waited on LDS; with direct register stores
    // there is nothing to wait for and "after the MFMAs" for every wave (1) measures 1.4 % faster in the forward
    // (0.633 vs 0.642 ms at N_rand 1024), "before" (2) the same as (0).
    if (probe::HOOK_ORDER == 2 || (probe::HOOK_ORDER == 0 && pipe.wave >= 4)) hook(blk);
    // (TRICKLE pipes: the DMA of the block this step's barrier freed a slot for goes out in UPB * KPB pieces between the MFMAs)
    constexpr bool CHAINS = P >= 2 && probe::CHAIN_GROUP > 1 && NOB % probe::CHAIN_GROUP == 0 && !(probe::LDS_REUSE > 1 && P == 1);
    constexpr int UPB = CHAINS ? NOB / probe::CHAIN_GROUP : (NOB == 8 ? 2 : 1);       // trickle points per k-chunk
#pragma unroll
    for (int kl = 0; kl < KPB; ++kl) {
      if (blk * KPB + kl >= LIVE) {                  // (compile-time after unrolling)
#pragma unroll
        for (int u = 0; u < UPB; ++u) pipe.trickle(kl * UPB + u, KPB * UPB);
        continue;
      }
      if constexpr (probe::LDS_REUSE > 1 && P == 1) {      // (probes: one weight-fragment read per LDS_REUSE MFMAs -- garbage results)
        bf16x8 w{};
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
          if (ob % probe::LDS_REUSE == 0) w = *(const bf16x8*)(l + (kl * NOB + ob) * FRAG_BYTES);
          acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, b[blk * KPB + kl].v[0], acc[ob], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < UPB; ++u) pipe.trickle(kl * UPB + u, KPB * UPB);
      } else if constexpr (CHAINS) {
        // split-bf16 / fp16x2w: the 3 (2) MFMAs of an out-block are a dependent chain on its accumulator; the chains of CHAIN_GROUP
        // out-blocks interleaved keep the order inside each chain (bit-identical) and give every MFMA an independent predecessor
        constexpr int G = probe::CHAIN_GROUP;
#pragma unroll
        for (int ob = 0; ob < NOB; ob += G) {
          const char* f0 = l + (kl * NOB + ob) * 2 * FRAG_BYTES;
          bf16x8 wh[G], wl[G];
#pragma unroll
          for (int g = 0; g < G; ++g) { wh[g] = *(const bf16x8*)(f0 + 2 * g * FRAG_BYTES); wl[g] = *(const bf16x8*)(f0 + (2 * g + 1) * FRAG_BYTES); }
          const Frag<P>& bb = b[blk * KPB + kl];
          if constexpr (P == 2) {
#pragma unroll
            for (int g = 0; g < G; ++g) acc[ob + g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[g], bb.v[0], acc[ob + g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[ob + g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[g], bb.v[1], acc[ob + g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[ob + g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[g], bb.v[0], acc[ob + g], 0, 0, 0);
          } else {
            const f16x8 bf = __builtin_bit_cast(f16x8, bb.v[0]);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[ob + g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wh[g]), bf, acc[ob + g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[ob + g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wl[g]), bf, acc[ob + g], 0, 0, 0);
          }
          pipe.trickle(kl * UPB + ob / G, KPB * UPB);
        }
      } else {
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
          mfma_p<P>(acc[ob], l + (kl * NOB + ob) * w_planes(P) * FRAG_BYTES, b[blk * KPB + kl]);
          if (UPB == 2 && (ob & 3) == 3) pipe.trickle(kl * 2 + (ob >> 2), KPB * 2);
        }
        if (UPB == 1) pipe.trickle(kl, KPB);
      }
    }
    if (probe::HOOK_ORDER == 1 || (probe::HOOK_ORDER == 0 && pipe.wave < 4)) hook(blk);
    if constexpr (P >= 2 && probe::LDS_PREFETCH_SPLIT > 0) {
      // split-bf16: a unit = the two planes of a weight fragment (2 LDS reads) and its 3 MFMAs; LDS_PREFETCH_SPLIT units in flight
      const int live_kl = LIVE - blk * KPB < KPB ? (LIVE - blk * KPB < 0 ? 0 : LIVE - blk * KPB) : KPB;
      const int D = probe::LDS_PREFETCH_SPLIT, N = NOB * live_kl;
      if (N < D) continue;
#pragma unroll
      for (int i = 0; i < D; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int i = 0; i < N - D; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, P == 2 ? 3 : 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
#pragma unroll
      for (int i = 0; i < D; ++i) __builtin_amdgcn_sched_group_barrier(0x008, P == 2 ? 3 : 2, 0);
    }
    if constexpr (P == 1 && probe::LDS_PREFETCH > 0) {
      // shape the block's schedule: LDS_PREFETCH weight fragments in flight ahead of the MFMA
      // that consumes them (LDS latency is ~2-4 MFMA slots; the default schedule keeps only 1-2 ahead)
      const int live_kl = LIVE - blk * KPB < KPB ? (LIVE - blk * KPB < 0 ? 0 : LIVE - blk * KPB) : KPB;
      const int D = probe::LDS_PREFETCH, N = NOB * live_kl;
      static_assert(probe::LDS_REUSE == 1 || probe::LDS_PREFETCH == 0, "the reuse probe runs on the default schedule");
      if (N < D) continue;
#pragma unroll
      for (int i = 0; i < D; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
      for (int i = 0; i < N - D; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < D; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  }
}

template <int P>
__device__ __forceinline__ void set_slot(Frag<P>& f, int t, float v) {
  if constexpr (P == 3) {
    f16x8 q = __builtin_bit_cast(f16x8, f.v[0]);
    q[t] = (_Float16)v;
    f.v[0] = __builtin_bit_cast(bf16x8, q);
    return;
  }
  const __bf16 h = (__bf16)v;
  f.v[0][t] = h;
  if constexpr (P == 2) f.v[1][t] = (__bf16)(v - (float)h);
}
// what goes to a saved tensor: the register image itself, or (precision 3) its fp16 values re-rounded to bf16 -- the saved
// tensors feed the bf16 weight-gradient GEMMs (nerfpp_dw.hip) whatever the forward's operand format was
template <int P>
__device__ __forceinline__ uint4 saved_image(const uint4 regs) {
  if constexpr (P == 3) {
    const f32x8 f = __builtin_convertvector(__builtin_bit_cast(f16x8, regs), f32x8);
    return __builtin_bit_cast(uint4, __builtin_convertvector(f, bf16x8));
  } else return regs;
}
template <int P>
__device__ __forceinline__ Frag<P> zero_frag() {
  Frag<P> f;
#pragma unroll
  for (int p = 0; p < a_planes(P); ++p)
#pragma unroll
    for (int t = 0; t < 8; ++t) f.v[p][t] = (__bf16)0.f;
  return f;
}

enum { ACT_NONE = 0, ACT_RELU = 1 };

// accumulator (C/D layout) -> B operand fragments of the next stage
template <int NOB, int P, int ACT>
__device__ __forceinline__ void acc_to_frags(const f32x16 (&acc)[NOB], Frag<P> (&h)[2 * NOB]) {
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float v = acc[ob][8 * hh + t];
        if (ACT == ACT_RELU) v = fmaxf(v, 0.f);
        set_slot<P>(h[2 * ob + hh], t, v);
      }
}

// The encoded point is needed again by layer 5 (skip connection): park its fragments in LDS
// (lane-linear 16-byte slots, conflict-free) instead of holding 16-24 VGPRs through layers 1-4.
template <int N, int P>
__device__ __forceinline__ void stash_frags(char* base, int lane, const Frag<P> (&f)[N]) {
#pragma unroll
  for (int c = 0; c < N; ++c)
#pragma unroll
    for (int p = 0; p < a_planes(P); ++p) *(uint4*)(base + ((c * a_planes(P) + p) * 64 + lane) * 16) = *(const uint4*)&f[c].v[p];
}
template <int N, int P>
__device__ __forceinline__ void unstash_frags(const char* base, int lane, Frag<P> (&f)[N]) {
#pragma unroll
  for (int c = 0; c < N; ++c)
#pragma unroll
    for (int p = 0; p < a_planes(P); ++p) *(uint4*)&f[c].v[p] = *(const uint4*)(base + ((c * a_planes(P) + p) * 64 + lane) * 16);
}

// two float32 -> one dword of packed bf16 (fp16 in precision 3): ONE v_cvt_pk_* instruction.  (Element-wise `q[t] = (__bf16) x`
// into an 8-vector made the compiler convert every element on its own and merge pairs with v_perm_b32 -- three VALU
// instructions per dword in the per-stage epilogues, where all waves of the CU sit at once.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <int P>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const f32x2 v = {a, b};
  if constexpr (P == 3) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
  else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// ReLU + conversion + sign words in one pass over the accumulators.
// Sign-word layout (uint4 per lane per stage, consumed by mask_to_frags in the backward kernel): element
// (ob, r = 8*hh + 2*w + e) lives in word ob>>1 at bit (e ? 31 : 15) - j, j = (ob&1)*8 + hh*4 + w, and holds
// the SIGN BIT of the pre-activation (set = unit inactive, gradient 0).  In that layout the 16 packed
// bf16 dwords of a word are gathered with 2 VALU ops each, and ReLU is one packed signed-int16 max per
// dword (a bf16 with the sign bit set is a negative int16), instead of compare/select/or per element.
// PK (split-bf16 only): the packed form of the hi / lo split; the inference forward (no sign words) keeps the element-wise
// one, which measures 2 % faster there (1.09 vs 1.12 ms per level-1 launch), training is 2-3 % faster packed.
template <int NOB, int P, bool PK = true>
__device__ __forceinline__ uint4 acc_to_frags_relu_bits(const f32x16 (&acc)[NOB], Frag<P> (&h)[2 * NOB]) {
  uint32_t m[4] = {0, 0, 0, 0};
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if constexpr (P != 2) {
        // (bf16 and fp16 alike: the sign is bit 15 of the 16-bit pattern, a value with it set is a negative int16)
        u32x4 d;
#pragma unroll
        for (int w = 0; w < 4; ++w) d[w] = pack2<P>(acc[ob][8 * hh + 2 * w], acc[ob][8 * hh + 2 * w + 1]);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int j = (ob & 1) * 8 + hh * 4 + w;
          // shift + ONE v_and_or_b32 per dword (left to itself the compiler pairs the ORs with v_or3_b32: 2.6 per dword)
          if constexpr ((probe::DBG & 128) == 0)        // (probes: no sign words -- the backward of such a forward is garbage)
            asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(m[ob >> 1]) : "v"(d[w] >> j), "s"(0x80008000u >> j));
        }
        const s16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        h[2 * ob + hh].v[0] = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, d), zero));
      } else if constexpr ((probe::EXP & 4) != 0) {
        // (probes: no conversion work at all -- the accumulator bits go on as the operand; what would a hidden epilogue buy?)
        u32x4 d0, d1;
#pragma unroll
        for (int w = 0; w < 4; ++w) { d0[w] = __float_as_uint(acc[ob][8 * hh + w]) & 0x3f803f80u; d1[w] = __float_as_uint(acc[ob][8 * hh + 4 + w]) & 0x3f803f80u; }
        h[2 * ob + hh].v[0] = __builtin_bit_cast(bf16x8, d0);
        h[2 * ob + hh].v[1] = __builtin_bit_cast(bf16x8, d1);
      } else if constexpr (!PK) {
#pragma unroll
        for (int t = 0; t < 8; ++t) set_slot<P>(h[2 * ob + hh], t, fmaxf(acc[ob][8 * hh + t], 0.f));
      } else {
        // split-bf16, packed (round 5; bit-identical to the element-wise form it replaces -- sign of bf16(v) = sign of v,
        // max_i16(bf16(v), 0) = bf16(max(v, 0)) -- at half its VALU instructions: with one wave per SIMD nothing hides them)
        u32x4 dh, dl;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float a = acc[ob][8 * hh + 2 * w], b = acc[ob][8 * hh + 2 * w + 1];
          const uint32_t d = pack2<1>(a, b);
          const int j = (ob & 1) * 8 + hh * 4 + w;
          m[ob >> 1] |= (d >> j) & (0x80008000u >> j);
          // lo = bf16(v - bf16(v)) where v > 0, else 0: formed on the pre-activation pair and masked by the smeared signs
          const uint32_t lo = pack2<1>(a - __uint_as_float(d << 16), b - __uint_as_float(d & 0xffff0000u));
          const uint32_t neg = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, d) >> (s16x2){15, 15});
          dh[w] = d & ~neg;
          dl[w] = lo & ~neg;
        }
        h[2 * ob + hh].v[0] = __builtin_bit_cast(bf16x8, dh);
        h[2 * ob + hh].v[1] = __builtin_bit_cast(bf16x8, dl);
      }
    }
  return make_uint4(m[0], m[1], m[2], m[3]);
}

// rows past the end of the batch (last tile only): their fragments are written out as zeros
template <int N, int P>
__device__ __forceinline__ void zero_invalid(Frag<P> (&f)[N], bool valid) {
#pragma unroll
  for (int c = 0; c < N; ++c)
#pragma unroll
    for (int p = 0; p < a_planes(P); ++p) {
      u32x4 d = __builtin_bit_cast(u32x4, f[c].v[p]);
#pragma unroll
      for (int w = 0; w < 4; ++w) d[w] = valid ? d[w] : 0u;
      f[c].v[p] = __builtin_bit_cast(bf16x8, d);
    }
}

// biases initialise the accumulators; the bias stream has been copied to LDS (keeps compiler-tracked global loads out of
// the steady state of the ring / roles pipes)
template <int NOB>
__device__ __forceinline__ void init_bias_lds(f32x16 (&acc)[NOB], uint32_t lds_off_bytes, int hi) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  LDS_AS char* base = (LDS_AS char*)smem;
  if constexpr ((probe::EXP & 16) != 0) {
    // (probes, garbage results: what would the bias cost as a 17th k-chunk -- zero accumulators (free: the first MFMA takes the
    // inline constant) and, with EXP bit 5, one more MFMA per out-block fed by ONE 1 KiB fragment read instead of four reads)
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ob][r] = 0.f;
      if constexpr ((probe::EXP & 32) != 0) {
        const bf16x8 w = *(const bf16x8*)(smem + (lds_off_bytes & ~1023u) % 8192 + ob * 1024 + (threadIdx.x & 63) * 16);
        bf16x8 one;
#pragma unroll
        for (int t = 0; t < 8; ++t) one[t] = (__bf16)(t == 0 ? 1.f : 0.f);
        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, one, acc[ob], 0, 0, 0);
      }
    }
    return;
  }
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *(LDS_AS f32x4*)(base + lds_off_bytes + (ob * 32 + hi * 16 + 4 * q) * 4);
      acc[ob][4 * q] = v[0]; acc[ob][4 * q + 1] = v[1]; acc[ob][4 * q + 2] = v[2]; acc[ob][4 * q + 3] = v[3];
    }
  }
}
template <int NOB>
__device__ __forceinline__ void init_zero(f32x16 (&acc)[NOB]) {
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ob][r] = 0.f;
}

__device__ __forceinline__ void lds_wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void store_nt16(char* gptr, const uint4 v) { probe::store16(gptr, v); }
// (The ReLU sign words -- 16 B per lane per stage, 113 MB per level-1 launch -- keep the DEFAULT policy on both sides: written
// non-temporally they come back from HBM instead of the Infinity Cache and the backward kernel loses 0.035 ms per level-1
// launch to its sign-word DMA; a non-temporal DMA of default-policy words measures the same as the default.  gpurun_out/r04ae.)

// Saved tensors are FRAGMENT-MAJOR (nerfpp_common.h, "saved tensors"): the 16 bytes lane (j, hi) holds of chunk c of a
// wave's 32-row tile go to byte ((tile32 * (ld / 16) + c) * 1024 + (2 j + hi) * 16) of the tensor (hi plane, then the
// lo plane at +plane elements).  A chunk of a tile is therefore ONE 1-KiB-contiguous wave-store straight from the
// accumulator-layout registers: no transposition through LDS, no lgkmcnt wait in the save path, and the weight-gradient
// kernel DMAs whole 1 KiB blocks (nerfpp_dw.hip gathers its sample-major MFMA fragments from them with
// ds_read_b64_tr_b16 and per-lane addresses).  (Until round 3 the tensors were row-major and every tile went through a
// per-wave LDS transposition image: 36 KiB of LDS, 8 ds_write_b64 + 8 ds_read_b64 + a wait per 4 chunks, and wave-stores
// made of eight 128-byte row segments.)
// Rows past the end of the batch (tile tail, < rows_padded) must be written as zeros so the weight-gradient GEMMs can
// run over whole 32-row chunks without masking: the kernels zero those lanes' fragments -- zero_invalid, last tile only
// -- before they get here.
constexpr int region_mask(int P) { return 16 * a_planes(P) * FRAG_BYTES; }   // hand-off region: 16 chunk blocks per activation plane, then 64 x 16 B of sign words
constexpr int region_bytes(int P) { return region_mask(P) + 1024; }

__device__ __forceinline__ char* frag_addr(__bf16* base, int ld, size_t wave_row0, int c, int lane) {
  const size_t blk = (wave_row0 >> 5) * (size_t)(ld >> 4) + (size_t)c;
  return (char*)base + blk * FRAG_BYTES + ((((lane & 31) << 1) | (lane >> 5)) << 4);
}
template <int P>
// np < P: only the first np planes are written (a split-bf16 forward whose backward is single-pass bf16 reads the hi planes only)
__device__ __forceinline__ void store_chunk(__bf16* base, size_t plane, int ld, size_t wave_row0, int lane, int c, const Frag<P>& f, int np = a_planes(P)) {
#pragma unroll
  for (int p = 0; p < a_planes(P); ++p)
    if (p < np) store_nt16(frag_addr(base + p * plane, ld, wave_row0, c, lane), saved_image<P>(*(const uint4*)&f.v[p]));
}
template <int NCH, int P>
__device__ __forceinline__ void save_frags(__bf16* base, size_t plane, int ld, size_t wave_row0, int lane, const Frag<P> (&h)[NCH], int np = a_planes(P)) {
  if constexpr ((probe::DBG & 2) != 0) return;
  // Scheduling fences on both sides: the LDS-staged save this replaces was a fence by construction (wave barriers);
  // without one the scheduler starts the next stage's accumulator set while this stage's is still being converted and
  // stored (the split-bf16 backward went from 371 to 512 registers + spills).
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < NCH; ++c) store_chunk<P>(base, plane, ld, wave_row0, lane, c, h[c], np);
  __builtin_amdgcn_sched_barrier(0);
}

// ---- PIPE_ROLES hand-off: the loader never stores; its tile goes to an LDS region (lane-linear chunk blocks,
// conflict-free 16-byte accesses) and helper waves write it out after the next barrier --------------------------------
template <int NCH, int P>
__device__ __forceinline__ void handoff_write(char* region, int lane, const Frag<P> (&h)[NCH]) {
  if constexpr ((probe::DBG & (2 | 16)) != 0) return;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int p = 0; p < a_planes(P); ++p) *(uint4*)(region + ((c * a_planes(P) + p) * 64 + lane) * 16) = *(const uint4*)&h[c].v[p];
}
// chunks [c0, c0 + n) of the loader's tile (tile rows row0 .. row0 + 31): region -> HBM (plane p at base + p * plane elements)
template <int P>
__device__ __forceinline__ void handoff_flush_chunks(const char* region, int lane, __bf16* base, size_t plane, int ld, size_t row0, int c0, int n, int np = a_planes(P)) {
  if constexpr ((probe::DBG & 16) != 0) return;
#pragma unroll 4
  for (int c = c0; c < c0 + n; ++c)
#pragma unroll
    for (int p = 0; p < a_planes(P); ++p)
      if (p < np) store_nt16(frag_addr(base + p * plane, ld, row0, c, lane), saved_image<P>(*(const uint4*)(region + ((c * a_planes(P) + p) * 64 + lane) * 16)));
}

// region chunk cr -> tensor chunk ct (a hand-off that carries the chunks of two tensors back to back)
template <int P>
__device__ __forceinline__ void handoff_flush_one(const char* region, int lane, __bf16* base, size_t plane, int ld, size_t row0, int cr, int ct, int np = a_planes(P)) {
  if constexpr ((probe::DBG & 16) != 0) return;
#pragma unroll
  for (int p = 0; p < a_planes(P); ++p)
    if (p < np) store_nt16(frag_addr(base + p * plane, ld, row0, ct, lane), saved_image<P>(*(const uint4*)(region + ((cr * a_planes(P) + p) * 64 + lane) * 16)));
}

// dH (accumulators) masked by the forward sign words (see acc_to_frags_relu_bits) -> dZ fragments
template <int NOB, int P>
__device__ __forceinline__ void mask_to_frags(const f32x16 (&acc)[NOB], const uint4 bits, Frag<P> (&dz)[2 * NOB]) {
  const uint32_t act[4] = {~bits.x, ~bits.y, ~bits.z, ~bits.w};       // bit set = unit active
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if constexpr (P == 1) {
        u32x4 d;
#pragma unroll
        for (int w = 0; w < 4; ++w) d[w] = pack2<1>(acc[ob][8 * hh + 2 * w], acc[ob][8 * hh + 2 * w + 1]);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int j = (ob & 1) * 8 + hh * 4 + w;
          // bits 31 / 15 of (act << j) are this dword's two flags: smear each over its half
          const s16x2 keep = __builtin_bit_cast(s16x2, act[ob >> 1] << j) >> (s16x2){15, 15};
          d[w] &= __builtin_bit_cast(uint32_t, keep);
        }
        dz[2 * ob + hh].v[0] = __builtin_bit_cast(bf16x8, d);
      } else {
        // split-bf16, packed like the branch above (bit-identical to `on ? v : 0` split element by element)
        u32x4 dh, dl;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float a = acc[ob][8 * hh + 2 * w], b = acc[ob][8 * hh + 2 * w + 1];
          const int j = (ob & 1) * 8 + hh * 4 + w;
          const uint32_t keep = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, act[ob >> 1] << j) >> (s16x2){15, 15});
          const uint32_t hi = pack2<1>(a, b);
          const uint32_t lo = pack2<1>(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
          dh[w] = hi & keep;
          dl[w] = lo & keep;
        }
        dz[2 * ob + hh].v[0] = __builtin_bit_cast(bf16x8, dh);
        dz[2 * ob + hh].v[1] = __builtin_bit_cast(bf16x8, dl);
      }
    }
}

// ------------------------------------------------------------------------------------------------
// input geometry + encodings for one sample (one lane-half)
// ------------------------------------------------------------------------------------------------
template <int NET>
__device__ __forceinline__ void sample_point(const MlpGeom& gm, size_t row, int S, float (&x)[4], float (&vd)[3],
                                             float* depth_real) {
  // No implicit FMA contraction in the point / encoding arithmetic: which mul + add pairs the compiler fuses depends on the code
  // around the inlined copy, and the stage-at-a-time and the unit-pipelined split-bf16 bodies (nerfpp_mlp_split.h) then differ in
  // the last bit of a few encoded values.  Uncontracted float32 is also what the reference's torch ops compute.
#pragma clang fp contract(off)
  const int ray = (int)(row / S);
  const float ox = gm.ray_o[ray * 3], oy = gm.ray_o[ray * 3 + 1], oz = gm.ray_o[ray * 3 + 2];
  const float dx = gm.ray_d[ray * 3], dy = gm.ray_d[ray * 3 + 1], dz = gm.ray_d[ray * 3 + 2];
  const float z = gm.z[row];
  const float dd = dx * dx + dy * dy + dz * dz;
  const float inv_n = 1.f / sqrtf(dd);
  vd[0] = dx * inv_n; vd[1] = dy * inv_n; vd[2] = dz * inv_n;       // ddp_model.py:82-83
  if (NET == 0) {
    x[0] = ox + z * dx; x[1] = oy + z * dy; x[2] = oz + z * dz; x[3] = 0.f;   // ddp_model.py:91
    *depth_real = 0.f;
  } else {                                                           // ddp_model.py:16-45
    const float d1 = -(dx * ox + dy * oy + dz * oz) / dd;
    const float mx = ox + d1 * dx, my = oy + d1 * dy, mz = oz + d1 * dz;
    const float pmn = sqrtf(mx * mx + my * my + mz * mz);
    const float d2 = sqrtf(1.f - pmn * pmn) * inv_n;
    const float sx = ox + (d1 + d2) * dx, sy = oy + (d1 + d2) * dy, sz = oz + (d1 + d2) * dz;
    float ax = oy * sz - oz * sy, ay = oz * sx - ox * sz, az = ox * sy - oy * sx;
    const float an = 1.f / sqrtf(ax * ax + ay * ay + az * az);
    ax *= an; ay *= an; az *= an;
    const float phi = asinf(pmn), theta = asinf(pmn * z);
    float sn, cs;
    sincosf(phi - theta, &sn, &cs);
    const float cx = ay * sz - az * sy, cy = az * sx - ax * sz, cz = ax * sy - ay * sx;
    const float dt = (ax * sx + ay * sy + az * sz) * (1.f - cs);
    float nx = sx * cs + cx * sn + ax * dt, ny = sy * cs + cy * sn + ay * dt, nz = sz * cs + cz * sn + az * dt;
    const float nn = 1.f / sqrtf(nx * nx + ny * ny + nz * nz);
    x[0] = nx * nn; x[1] = ny * nn; x[2] = nz * nn; x[3] = z;
    *depth_real = 1.f / (z + 1e-6f) * cosf(theta) * inv_n + d1;
  }
}

// sin / cos of one encoding argument.  Split-bf16 (parity) precision: libm-accurate sincosf.  Single-pass
// bf16: the features are rounded to 8 mantissa bits anyway, so the hardware v_sin_f32 / v_cos_f32 (argument in
// revolutions, reduced with v_fract; |arg| <= 512 rad = 82 rev keeps the reduction error below 2e-5 rad)
// replace ~60 VALU instructions + a Payne-Hanek slow path per call with 4.
// (Precision 3 rounds the features to fp16 -- 2.4e-4 at |v| ~ 1 -- an order of magnitude above the hardware path's error.)
template <int P>
__device__ __forceinline__ void pe_sincos(float arg, float* sn, float* cs) {
#pragma clang fp contract(off)          // (the FMAs below are explicit)
  if constexpr (P != 2) {
    const float r = __builtin_amdgcn_fractf(arg * 0.15915494309189535f);
    *sn = __builtin_amdgcn_sinf(r);
    *cs = __builtin_amdgcn_cosf(r);
  } else {
    // split-bf16 (parity) precision: float32-accurate sin / cos for |arg| <= 1024 (points of the unit sphere x 2^9), branch-free:
    // three-term Cody-Waite reduction by pi / 2 with FMAs (n <= 652 has 10 bits, n * HI is exact inside the FMA and the first
    // difference is exactly representable), degree-9 / degree-8 minimax kernels on [-pi/4, pi/4], quadrant swap.  Measured
    // against float64 over 2 M arguments up to 512 (tests/test_layout_emulation.py carries the numpy twin): max error 7.3e-8
    // = 1.2 ulp, against 0.5 ulp for a correctly rounded result -- the reference's torch.sin is itself ~1 ulp.  libm's sincosf
    // (~70 instructions with a Payne-Hanek branch, 20-26 calls per lane and tile) took 2-3 % of the split-bf16 forward.
    const float n = __builtin_rintf(arg * 0.6366197723675814f);
    float r = __builtin_fmaf(-n, 1.5707962512969971f, arg);
    r = __builtin_fmaf(-n, 7.5497901264043321e-08f, r);
    r = __builtin_fmaf(-n, -1.7763568394002505e-15f, r);
    const float z = r * r;
    float ps = __builtin_fmaf(z, 2.7183114e-6f, -0.00019839335f);
    ps = __builtin_fmaf(z, ps, 0.008333329f);
    ps = __builtin_fmaf(z, ps, -0.16666667f);
    const float s = __builtin_fmaf(r * z, ps, r);
    float pc = __builtin_fmaf(z, 2.4390449e-5f, -0.0013886763f);
    pc = __builtin_fmaf(z, pc, 0.04166662f);
    pc = __builtin_fmaf(z, pc, -0.5f);
    const float c = __builtin_fmaf(z, pc, 1.0f);
    const int q = (int)n;
    const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
  }
}

// positional encoding of this lane-half's share (nerfpp_common.h: pe_ref_of_lane_slot)
template <int NET, int P>
__device__ __forceinline__ void encode_point(const float (&x)[4], int hi, Frag<P> (&pe)[kpe(NET)]) {
#pragma clang fp contract(off)
  constexpr int D = pe_dim(NET), NV = kpe(NET) * 8;
  float v[NV];
#pragma unroll
  for (int kk = 0; kk < 5; ++kk) {
    const float scale = __int_as_float((127 + 5 * hi + kk) << 23);       // 2^(5hi+kk), exact
#pragma unroll
    for (int d = 0; d < D; ++d) {
      float sn, cs;
      pe_sincos<P>(x[d] * scale, &sn, &cs);
      v[(kk * D + d) * 2] = sn;
      v[(kk * D + d) * 2 + 1] = cs;
    }
  }
  v[10 * D] = hi ? x[2] : x[0];
  v[10 * D + 1] = hi ? (D == 4 ? x[3] : 0.f) : x[1];
#pragma unroll
  for (int m = 10 * D + 2; m < NV; ++m) v[m] = 0.f;
#pragma unroll
  for (int c = 0; c < kpe(NET); ++c)
#pragma unroll
    for (int t = 0; t < 8; ++t) set_slot<P>(pe[c], t, v[8 * c + t]);
}

template <int P>
__device__ __forceinline__ void encode_dir(const float (&vd)[3], int hi, Frag<P> (&df)[2]) {
#pragma clang fp contract(off)
  float v[16];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const float scale = __int_as_float((127 + 2 * hi + kk) << 23);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float sn, cs;
      pe_sincos<P>(vd[d] * scale, &sn, &cs);
      v[(kk * 3 + d) * 2] = sn;
      v[(kk * 3 + d) * 2 + 1] = cs;
    }
  }
  v[12] = hi ? vd[2] : vd[0];
  v[13] = hi ? 0.f : vd[1];
  v[14] = 0.f; v[15] = 0.f;
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 8; ++t) set_slot<P>(df[c], t, v[8 * c + t]);
}

// LDS carve-up shared by kernel and launcher
template <int NET, int P, int NW, bool TRAIN>
struct FwdLds {
  static constexpr bool V2T = split_v2_train<P, TRAIN>();
  // (probes, EXP bit 3: the training forward on the ring pipe -- only meaningful with the saves compiled out, NERFPP_DBG & 2)
  static constexpr int MODE = (!TRAIN || V2T || (probe::EXP & 8) != 0) ? PIPE_RING : PIPE_ROLES;
  static constexpr bool ROLES = MODE == PIPE_ROLES;
  static constexpr int BF = V2T ? BLK_FRAGS : blk_frags_of<P, TRAIN>();
  // ring depth: as deep as the 160 KiB of LDS allow
  static constexpr int SKEW = (MODE == PIPE_RING && P == 1 && NW >= 2) ? probe::SKEW_INFER : 0;
  // (V2T: every wave stores AND fetches -- probe::V2T_NBUF slots, see nerfpp_mlp_split.h on what a counted wait guarantees there)
  static constexpr int NBUF = V2T ? probe::V2T_NBUF : MODE == PIPE_RING ? (P == 1 ? 4 + SKEW : 3) : 4;
  static constexpr int W = NBUF * BF * w_planes(P) * FRAG_BYTES;
  static constexpr int REGION = W;
  static constexpr int STASH = REGION + (ROLES ? region_bytes(P) : 0);
  static constexpr int BIAS = STASH + NW * kpe(NET) * a_planes(P) * 1024;
  static constexpr int TOTAL = BIAS + FWD_BIAS_FLOATS * 4;
};
template <int P, int NW>
struct BwdLds {
  static constexpr int MODE = PIPE_ROLES;
  static constexpr bool ROLES = true;
  static constexpr int BF = blk_frags_of<P, true>();
  static constexpr int NBUF = 4;
  static constexpr int W = NBUF * BF * P * FRAG_BYTES;
  static constexpr int REGION = W;
  static constexpr int MASKS = REGION + (ROLES ? region_bytes(P) : 0);       // 2 x NW KiB of sign words (ROLES)
  static constexpr int TOTAL = MASKS + (ROLES ? 2 * NW * 1024 : 0);
};

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// bid: this workgroup's tile among the net's tiles (the pair kernel below runs both nets of a level in one launch)
template <int NET, int P, int NW, bool TRAIN>
__device__ __forceinline__ void mlp_fwd_body(const MlpFwdArgs& a, const int bid) {
  using LD = FwdLds<NET, P, NW, TRAIN>;
  constexpr int KPE = kpe(NET);
  constexpr bool ROLES = LD::ROLES;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5;
  const size_t row_raw = (size_t)bid * (NW * 32) + wave * 32 + (lane & 31);
  const bool valid = row_raw < (size_t)a.rows;
  const size_t row = valid ? row_raw : (size_t)a.rows - 1;
  const size_t plane_rows = a.rows_padded;
  const size_t wrow0 = (size_t)bid * (NW * 32) + wave * 32;                 // this wave's first tile row
  const bool loader = ROLES && wave == 0, partner = ROLES && wave == 1;
  const int NPS = (P == 2 && !a.save_lo) ? 1 : a_planes(P);  // planes of the saved tensors that are written out (wave-uniform)
  probe::kernel_prologue(a.out_raw);
  const bool tail = wrow0 + 32 > (size_t)a.rows;                            // wave-uniform
  char* region = smem + LD::REGION;
  char* pe_stash = smem + LD::STASH + wave * (KPE * a_planes(P) * 1024);
  const size_t nblk32 = a.rows_padded / 32;
  uint4* mask_out = a.masks + (wrow0 / 32) * 64 + lane;                     // + stage * nblk32 * 64

  WeightPipe<w_planes(P), NW, LD::MODE, LD::NBUF, LD::BF, LD::SKEW> pipe;
  pipe.stamp_off = LD::TOTAL;
  pipe.init(a.w_stream, fwd_frags(NET) / LD::BF, wave, lane);
  // The loader's tile is written out by the helper waves 1..H.  CPB chunks of a storer's own tile go out per weight block
  // (a 16-chunk stage has 16 / CPB blocks).  bf16: every helper writes its share of the loader's tile in block 2; split-bf16
  // (two planes per chunk, one wave per SIMD): one chunk per block from block 2 on.
  constexpr int H = NW >= 5 ? 4 : NW - 1, Q = (16 + H - 1) / H, CPB = LD::BF / 8;
  constexpr int RMASK = region_mask(P);
  for (int i = threadIdx.x; i < FWD_BIAS_FLOATS / 4; i += NW * 64)
    *(float4*)(smem + LD::BIAS + i * 16) = ((const float4*)a.bias)[i];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  pipe.lead_in();
  auto bias_init8 = [&](f32x16 (&acc_)[8], int off) { init_bias_lds<8>(acc_, LD::BIAS + off * 4, hi); };
  auto bias_init4 = [&](f32x16 (&acc_)[4], int off) { init_bias_lds<4>(acc_, LD::BIAS + off * 4, hi); };
  auto bias_init1 = [&](f32x16 (&acc_)[1], int off) { init_bias_lds<1>(acc_, LD::BIAS + off * 4, hi); };
  const size_t tile_row0 = wrow0 - (size_t)wave * 32;                       // = the loader's rows
  // after the first barrier of a stage: helper waves 1..4 write out what the loader handed over at the end of the
  // previous stage (chunk c by wave 1 + c % 4; statically known per stage; mask_stage < 0: no sign words)
  auto flush = [&](int blk, auto nch_c, __bf16* base, int ld, int mask_stage) __attribute__((always_inline)) {
    if constexpr (ROLES) {
      constexpr int NCH = decltype(nch_c)::value;
      if (blk == 0 && wave >= 1 && wave <= H) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          if ((c % H) == wave - 1) handoff_flush_chunks<P>(region, lane, base, plane_rows * ld, ld, tile_row0, c, 1, NPS);
        if (partner && mask_stage >= 0)
          mask_out[(size_t)mask_stage * nblk32 * 64 - 64] = *(const uint4*)(region + RMASK + lane * 16);
      }
    }
  };
  // [rows,256] trunk activations.  finish_h: what stays in the producing stage's epilogue (sign words, tail zeroing; the
  // whole save when the pipe has no roles).  psave_h: runs in every block of the CONSUMING stage (the tile is its B
  // operand, still in registers): a storer writes two chunks per block straight from them (16 wave-stores of 1 KiB over
  // the stage's first 8 blocks); the loader hands its tile over in block 0 and waves 1..4 write a quarter of it each in
  // block 2 (wave 1 alone used to: 32 wave-stores per layer on one wave against 16 on the others, and the slowest wave
  // sets the pace at every barrier).
  auto finish_h = [&](__bf16* base, Frag<P> (&frags)[16], uint4 bits, int mask_stage) __attribute__((always_inline)) {
    if constexpr (!TRAIN) return;
    if (tail) zero_invalid(frags, valid);
    if constexpr (ROLES) {
      if constexpr ((probe::DBG & 128) != 0) return;
      if (loader) *(uint4*)(region + RMASK + lane * 16) = bits;
      else mask_out[(size_t)mask_stage * nblk32 * 64] = bits;
    } else {
      if constexpr ((probe::DBG & 128) == 0) mask_out[(size_t)mask_stage * nblk32 * 64] = bits;
      save_frags<16, P>(base, plane_rows * 256, 256, wrow0, lane, frags, NPS);
    }
  };
  // (cpb_c: chunks per block of the storers; the colour head that consumes h7 has 5 / 10 blocks, not 8 / 16)
  auto psave_hc = [&](int blk, const Frag<P> (&frags)[16], __bf16* base, int mask_stage, auto cpb_c) __attribute__((always_inline)) {
    constexpr int CPBH = decltype(cpb_c)::value;
    if constexpr (TRAIN && ROLES) {
      // (probes: H_l stays unsaved, its sign words go out; SKIP_H < 0: the mask comes with the launch, bits 8.. of save_lo)
      const bool skip = (P == 1 && (((probe::SKIP_H < 0 ? a.save_lo >> 8 : probe::SKIP_H) >> mask_stage) & 1)) ||
                        (mask_stage == 0 && a.skip_h0);      // H0: recomputed by its weight-gradient job (nerfpp_dw.hip: rc_job)
      if (skip) {
        if (partner && blk == 1) mask_out[(size_t)mask_stage * nblk32 * 64 - 64] = *(const uint4*)(region + RMASK + lane * 16);
      } else if (loader) {
        if (blk == 0) handoff_write<16, P>(region, lane, frags);
      } else {
        if constexpr ((probe::DBG & 2) == 0) {
          if (blk * CPBH < 16) {
#pragma unroll
            for (int i = 0; i < CPBH; ++i)
              if (CPBH * blk + i < 16)
                store_chunk<P>(base, plane_rows * 256, 256, wrow0, lane, CPBH * blk + i, frags[CPBH * blk + i], NPS);
          }
        }
        if constexpr (P == 1) {
          if (wave <= H && blk == 2) handoff_flush_chunks<P>(region, lane, base, plane_rows * 256, 256, tile_row0, Q * (wave - 1), Q, NPS);
        } else {
          if (wave <= H && blk >= 2 && blk < 2 + Q && Q * (wave - 1) + blk - 2 < 16)
            handoff_flush_chunks<P>(region, lane, base, plane_rows * 256, 256, tile_row0, Q * (wave - 1) + blk - 2, 1, NPS);
        }
        if constexpr ((probe::DBG & 128) == 0) {
          if (partner && blk == 1)
            mask_out[(size_t)mask_stage * nblk32 * 64 - 64] = *(const uint4*)(region + RMASK + lane * 16);
        }
      }
    }
  };
  auto psave_h = [&](int blk, const Frag<P> (&frags)[16], __bf16* base, int mask_stage) __attribute__((always_inline)) {
    psave_hc(blk, frags, base, mask_stage, std::integral_constant<int, CPB>{});
  };
  // save one tensor of this stage: storer waves write their own tile, the loader hands its tile over
  auto save = [&](auto nch_c, __bf16* base, int ld, auto& frags, bool has_mask, uint4 bits, int mask_stage) __attribute__((always_inline)) {
    constexpr int NCH = decltype(nch_c)::value;
    if constexpr (!TRAIN) return;
    if (tail) zero_invalid(frags, valid);       // wave-uniform, last tile only; those rows' values are never used
    if constexpr (ROLES) {
      if (loader) {
        handoff_write<NCH, P>(region, lane, frags);
        if (has_mask) *(uint4*)(region + RMASK + lane * 16) = bits;
      } else {
        if (has_mask) mask_out[(size_t)mask_stage * nblk32 * 64] = bits;
        save_frags<NCH, P>(base, plane_rows * ld, ld, wrow0, lane, frags, NPS);
      }
    } else {
      if (has_mask) mask_out[(size_t)mask_stage * nblk32 * 64] = bits;
      save_frags<NCH, P>(base, plane_rows * ld, ld, wrow0, lane, frags, NPS);
    }
  };

  float x[4], vd[3], depth_real;
  sample_point<NET>(a.geom, row, a.S, x, vd, &depth_real);
  Frag<P> pe[KPE];
  encode_point<NET, P>(x, hi, pe);
  if constexpr (TRAIN) {
    // The encoded point and the encoded view direction are saved together: storers write both tensors, the loader hands
    // its KPE + 2 chunks over in one piece and the helper waves write them out in L0's first block (flush_xd).  (The view
    // direction used to go out next to the colour head, in the slot of the hand-off region that h7 needs there now.)
    Frag<P> df0[2];
    encode_dir<P>(vd, hi, df0);
    if (tail) { zero_invalid(pe, valid); zero_invalid(df0, valid); }
    bool direct = true;
    if constexpr (ROLES) {
      if (loader) {
        Frag<P> xd[KPE + 2];
#pragma unroll
        for (int c = 0; c < KPE; ++c) xd[c] = pe[c];
        xd[KPE] = df0[0]; xd[KPE + 1] = df0[1];
        handoff_write<KPE + 2, P>(region, lane, xd);
        direct = false;
      }
    }
    if (direct) {
      save_frags<KPE, P>(a.ws.t[T_X], plane_rows * kpew(NET), kpew(NET), wrow0, lane, pe, NPS);
      save_frags<2, P>(a.ws.t[T_DIRX], plane_rows * 32, 32, wrow0, lane, df0, NPS);
    }
  }
  auto flush_xd = [&](int blk) __attribute__((always_inline)) {
    if constexpr (TRAIN && ROLES) {
      if (blk == 0 && wave >= 1 && wave <= H) {
#pragma unroll
        for (int c = 0; c < KPE + 2; ++c)
          if ((c % H) == wave - 1) {
            if (c < KPE) handoff_flush_one<P>(region, lane, a.ws.t[T_X], plane_rows * kpew(NET), kpew(NET), tile_row0, c, c, NPS);
            else handoff_flush_one<P>(region, lane, a.ws.t[T_DIRX], plane_rows * 32, 32, tile_row0, c, c - KPE, NPS);
          }
      }
    }
  };
  stash_frags<KPE, P>(pe_stash, lane, pe);

  f32x16 acc[8];
  Frag<P> h[16];
  // L0
  bias_init8(acc, fs_bias_off(FS_L0));
  stage_gemm<8, KPE, P>(pipe, acc, pe, HOOK(flush_xd(blk)));
  {
    const uint4 bits = acc_to_frags_relu_bits<8, P, TRAIN>(acc, h);
    finish_h(a.ws.t[T_H0], h, bits, 0);
  }
  // L1..L4
  for (int l = 1; l <= 4; ++l) {
    bias_init8(acc, fs_bias_off(FS_L0) + l * 256);
    stage_gemm<8, 16, P>(pipe, acc, h, HOOK(psave_h(blk, h, a.ws.t[T_H0 + l - 1], l - 1)));
    const uint4 bits = acc_to_frags_relu_bits<8, P, TRAIN>(acc, h);
    finish_h(TRAIN ? a.ws.t[T_H0 + l] : nullptr, h, bits, l);
  }
  // L5: input = cat(encoded point, h4)                                 nerf_network.py:127-129
  {
    Frag<P> in5[KPE + 16];
    {
      Frag<P> pe2[KPE];
      unstash_frags<KPE, P>(pe_stash, lane, pe2);
#pragma unroll
      for (int c = 0; c < KPE; ++c) in5[c] = pe2[c];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) in5[KPE + c] = h[c];
    bias_init8(acc, fs_bias_off(FS_L5));
    stage_gemm<8, KPE + 16, P>(pipe, acc, in5, HOOK(psave_h(blk, h, a.ws.t[T_H0 + 4], 4)));
    const uint4 bits = acc_to_frags_relu_bits<8, P, TRAIN>(acc, h);
    finish_h(a.ws.t[T_H0 + 5], h, bits, 5);
  }
  // L6, L7
  for (int l = 6; l <= 7; ++l) {
    bias_init8(acc, fs_bias_off(FS_L0) + l * 256);
    stage_gemm<8, 16, P>(pipe, acc, h, HOOK(psave_h(blk, h, a.ws.t[T_H0 + l - 1], l - 1)));
    const uint4 bits = acc_to_frags_relu_bits<8, P, TRAIN>(acc, h);
    finish_h(TRAIN ? a.ws.t[T_H0 + l] : nullptr, h, bits, l);
  }
  // sigma from h7                                                        nerf_network.py:131-136
  // (the remap layer is folded into the colour head: nerfpp_common.h, forward stages.  R is not a tensor here.)
  f32x16 acc1[1];
  bias_init1(acc1, fs_bias_off(FS_SIG));
  // h7 goes out under the two stages that consume it, three chunks per block: sigma (1 block) and the colour head (5 blocks).
  // A/B on one box (gpurun_out/r04v): 2.267 ms per step against 2.280 with four chunks per block under the colour head alone;
  // split-bf16 training (2 + 10 blocks of 8 fragments, two chunks per block) measured 0.5 % faster with the colour head alone.
  constexpr int SIG_BLKS = P == 1 ? 1 : 0;
  stage_gemm<1, 16, P>(pipe, acc1, h, HOOK(if constexpr (P == 1) psave_hc(blk, h, a.ws.t[T_H0 + 7], 7, IC(3))));
  const float sigma_raw = acc1[0][0];
  // colour head: relu(Wc h7 + Wrgb0[:, 256:] dirs + bc)                  nerf_network.py:131,137-138
  Frag<P> g[8];
  {
    Frag<P> in[20];
#pragma unroll
    for (int c = 0; c < 16; ++c) in[c] = h[c];
    Frag<P> df[2];
    encode_dir<P>(vd, hi, df);
    in[16] = df[0]; in[17] = df[1]; in[18] = zero_frag<P>(); in[19] = zero_frag<P>();
    f32x16 acc4[4];
    bias_init4(acc4, fs_bias_off(FS_RGB0));
    stage_gemm<4, 20, P, 18>(pipe, acc4, in, HOOK(psave_hc(blk + SIG_BLKS, h, a.ws.t[T_H0 + 7], 7, IC(P == 1 ? 3 : 2))));
    const uint4 bits = acc_to_frags_relu_bits<4, P, TRAIN>(acc4, g);
    save(std::integral_constant<int, 8>{}, a.ws.t[T_G], 128, g, true, bits, 8);
  }
  {
    Frag<P> in[16];
#pragma unroll
    for (int c = 0; c < 8; ++c) in[c] = g[c];
#pragma unroll
    for (int c = 8; c < 16; ++c) in[c] = zero_frag<P>();
    bias_init1(acc1, fs_bias_off(FS_RGB1));
    stage_gemm<1, 16, P, 8>(pipe, acc1, in, HOOK(flush(blk, IC(8), a.ws.t[T_G], 128, 8)));
  }
  pipe.lead_out();
  if (valid && hi == 0) {
    float4 o;
    o.x = 1.f / (1.f + expf(-acc1[0][0]));
    o.y = 1.f / (1.f + expf(-acc1[0][1]));
    o.z = 1.f / (1.f + expf(-acc1[0][2]));
    o.w = sigma_raw;
    ((float4*)a.out_raw)[row] = o;
    if (NET == 1) a.depth_real[row] = depth_real;
  }
  probe::dump_stamps(LD::TOTAL, wave, lane);
}

// ------------------------------------------------------------------------------------------------
// backward (dX chain).  d_out[row] = (d rgb_pre-sigmoid[3], d sigma_raw)
// ------------------------------------------------------------------------------------------------
template <int NET, int P, int NW>
__device__ __forceinline__ void mlp_bwd_body(const MlpBwdArgs& a, const int bid) {
  using LD = BwdLds<P, NW>;
  constexpr bool ROLES = LD::ROLES;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5;
  const size_t row_raw = (size_t)bid * (NW * 32) + wave * 32 + (lane & 31);
  const bool valid = row_raw < (size_t)a.rows;
  const size_t row = valid ? row_raw : (size_t)a.rows - 1;
  const size_t plane_rows = a.rows_padded;
  const size_t wrow0 = (size_t)bid * (NW * 32) + wave * 32;
  const bool loader = ROLES && wave == 0, partner = ROLES && wave == 1;
  char* region = smem + LD::REGION;
  const size_t tile_row0 = wrow0 - (size_t)wave * 32;
  const size_t nblk32 = a.rows_padded / 32;
  const uint4* mask_in = a.masks + (wrow0 / 32) * 64 + lane;
  const uint32_t lds0 = lds_base_addr();

  // ReLU sign words: PIPE_ROLES has the loader DMA the whole tile's words (NW KiB per stage) into a
  // 2-slot LDS buffer (slot = stage & 1) two uses ahead; the classic pipe loads them per lane.
  auto issue_masks = [&](int mstage) __attribute__((always_inline)) {
    if constexpr (ROLES) {
      if (loader && mstage >= 0) {
        static_assert(NW % 4 == 0, "groups of four");
        const char* sb = (const char*)(a.masks + ((size_t)mstage * nblk32 + (size_t)bid * NW) * 64);
#pragma unroll
        for (int w = 0; w < NW; w += 4) glds16xN_saddr<4>(sb + w * 1024, (uint32_t)lane * 16u, lds0 + LD::MASKS + ((mstage & 1) * NW + w) * 1024);
      }
    }
  };
  auto get_mask = [&](int mstage) __attribute__((always_inline)) -> uint4 {
    if constexpr (ROLES) return *(const uint4*)(smem + LD::MASKS + ((mstage & 1) * NW + wave) * 1024 + lane * 16);
    else return mask_in[(size_t)mstage * nblk32 * 64];
  };
  // dZ tensors are written out while the NEXT stage consumes them (see stage_gemm): a storer writes its chunks straight
  // from the operand registers, spread over the stage's blocks; the loader hands its tile over in block 0 and waves
  // 1..NCH/4 write four chunks of it each in block 2.  Without roles (split-bf16) the tile is saved in the epilogue.
  auto psave = [&](int blk, auto nch_c, const auto& frags, __bf16* base, int ld) __attribute__((always_inline)) {
    // helper waves 1..HN write the loader's tile (Q chunks each); CPB chunks of a storer's own tile per weight block
    constexpr int NCH = decltype(nch_c)::value, HMAX = NW >= 5 ? 4 : NW - 1, HN = NCH / 4 < HMAX ? NCH / 4 : HMAX;
    constexpr int Q = (NCH + HN - 1) / HN, CPB = LD::BF / 8;
    if constexpr (ROLES) {
      if (P == 1 && a.skip_dz7 && base == a.ws.t[T_DZ0 + 7]) return;      // recomputed by its weight-gradient job (nerfpp_dw.hip: rc7_job)
      if (loader) {
        if (blk == 0) handoff_write<NCH, P>(region, lane, frags);
      } else {
        if constexpr ((probe::DBG & 2) == 0) {
          if (CPB * blk < NCH) {
#pragma unroll
            for (int i = 0; i < CPB; ++i)
              store_chunk<P>(base, plane_rows * ld, ld, wrow0, lane, CPB * blk + i, frags[CPB * blk + i]);
          }
        }
        if constexpr (P == 1) {
          if (wave <= HN && blk == 2) handoff_flush_chunks<P>(region, lane, base, plane_rows * ld, ld, tile_row0, Q * (wave - 1), Q);
        } else {
          if (wave <= HN && blk >= 2 && blk < 2 + Q && Q * (wave - 1) + blk - 2 < NCH)
            handoff_flush_chunks<P>(region, lane, base, plane_rows * ld, ld, tile_row0, Q * (wave - 1) + blk - 2, 1);
        }
      }
    }
  };
  auto save = [&](auto nch_c, __bf16* base, int ld, const auto& frags) __attribute__((always_inline)) {
    constexpr int NCH = decltype(nch_c)::value;
    if constexpr (!ROLES) save_frags<NCH, P>(base, plane_rows * ld, ld, wrow0, lane, frags);
  };

  // sign words 8 and 7 first: they are needed after the first barriers, and the loader's counted waits
  // only guarantee what is OLDER than the weight blocks they count
  issue_masks(8);
  issue_masks(7);
  WeightPipe<P, NW, LD::MODE, LD::NBUF, LD::BF> pipe;
  pipe.stamp_off = LD::TOTAL;
  pipe.init(a.w_stream, BWD_FRAGS / LD::BF, wave, lane);

  float4 d = ((const float4*)a.d_out)[row];
  if (!valid) d = make_float4(0.f, 0.f, 0.f, 0.f);
  // dP [rows,32] and dS [rows,32] come straight from d_out: every storer writes its own rows, and
  // wave 1 also builds and writes the loader's rows (the loader never stores)
  auto save_dp_ds = [&](const float4& dd, size_t r0) __attribute__((always_inline)) {   // dd == 0 on rows past the end
    Frag<P> t[2];
    t[0] = zero_frag<P>(); t[1] = zero_frag<P>();
    if (hi == 0) { set_slot<P>(t[0], 0, dd.x); set_slot<P>(t[0], 1, dd.y); set_slot<P>(t[0], 2, dd.z); }
    save_frags<2, P>(a.ws.t[T_DP], plane_rows * 32, 32, r0, lane, t);
    t[0] = zero_frag<P>();
    if (hi == 0) set_slot<P>(t[0], 0, dd.w);
    save_frags<2, P>(a.ws.t[T_DS], plane_rows * DSG_LD, DSG_LD, r0, lane, t);   // chunks 0, 1 of [dS | dG]
  };
  {
    const size_t rr = row_raw - 32;                         // the loader's row of this lane (partner only)
    const bool ok = partner && rr < (size_t)a.rows;
    float4 d0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) d0 = ((const float4*)a.d_out)[rr];
    if (!loader) save_dp_ds(d, wrow0);
    if (partner) save_dp_ds(d0, wrow0 - 32);
  }

  // B0: dG = Wrgb1^T dP, masked by G > 0
  Frag<P> dg[8];
  {
    Frag<P> in[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) in[c] = zero_frag<P>();
    if (hi == 0) { set_slot<P>(in[0], 0, d.x); set_slot<P>(in[0], 1, d.y); set_slot<P>(in[0], 2, d.z); }
    f32x16 acc4[4];
    init_zero<4>(acc4);
    stage_gemm<4, 4, P, 1>(pipe, acc4, in, NoHook{});
    mask_to_frags<4, P>(acc4, get_mask(8), dg);
    save(std::integral_constant<int, 8>{}, a.ws.t[T_DG], DSG_LD, dg);
  }
  f32x16 acc[8];
  Frag<P> dz[16];
  // B2: dH7 = Wc^T dG + wsigma * dsigma, masked by H7 > 0  (B1, dR = Wrgb0[:, :256]^T dG, is folded into Wc: nerfpp_common.h)
  {
    Frag<P> in[10];
#pragma unroll
    for (int c = 0; c < 8; ++c) in[c] = dg[c];
    in[8] = zero_frag<P>(); in[9] = zero_frag<P>();
    if (hi == 0) set_slot<P>(in[8], 0, d.w);
    init_zero<8>(acc);
    // (the barrier just passed ends every wave's use of sign words 8: their slot takes words 6)
    stage_gemm<8, 10, P, 9>(pipe, acc, in, HOOK(psave(blk, IC(8), dg, a.ws.t[T_DG], DSG_LD); if (blk == 0) issue_masks(6)));
    mask_to_frags<8, P>(acc, get_mask(7), dz);
    save(std::integral_constant<int, 16>{}, a.ws.t[T_DZ0 + 7], 256, dz);
  }
  // B3..B9: dH_{l-1} = W_l^T dZ_l, l = 7..1
  for (int l = 7; l >= 1; --l) {
    init_zero<8>(acc);
    stage_gemm<8, 16, P>(pipe, acc, dz, HOOK(psave(blk, IC(16), dz, a.ws.t[T_DZ0 + l], 256); if (blk == 0) issue_masks(l - 2)));
    mask_to_frags<8, P>(acc, get_mask(l - 1), dz);
    save(std::integral_constant<int, 16>{}, a.ws.t[T_DZ0 + l - 1], 256, dz);
  }
  if constexpr (ROLES) {
    // dZ0 is the last tensor and no barrier follows: every wave (the loader too -- its DMA is done)
    // writes its own tile
    save_frags<16, P>(a.ws.t[T_DZ0], plane_rows * 256, 256, wrow0, lane, dz);
  }
  probe::dump_stamps(LD::TOTAL, wave, lane);
}

}  // namespace nerfpp
#include "nerfpp_mlp_split.h"
namespace nerfpp {

// Both nets of a cascade level in ONE launch: the fg net's tiles first, the bg net's as CUs free up.  They are independent
// (ddp_model.py:86-120 evaluates the two MLPs on different points), so one launch has one start-up and one tail where two
// launches had two (profiles/r04_pair_launch.md: ~20 us per launch boundary at N_rand = 1024, 12 MLP / weight-gradient
// launches per step before, 6 now).  tiles0 = 0 or grid = tiles0 runs one net alone (probes: the two-launch form).
template <int P, int NW, bool TRAIN>
__global__ __launch_bounds__(NW * 64, (NW == 8 ? 2 : 1)) void mlp_fwd_pair_kernel(MlpFwdArgs a0, MlpFwdArgs a1, int tiles0) {
  if constexpr (P == 2 && ((probe::SPLIT_V2 >> (TRAIN ? 1 : 0)) & 1) != 0) {
    // (training: one body per save mode -- both planes of the saved tensors, or the hi planes alone for a bf16 backward)
    const bool fg = (int)blockIdx.x < tiles0;
    const bool hi_only = TRAIN && !(fg ? a0.save_lo : a1.save_lo);
    if (fg) { if (hi_only) mlp_fwd_body_split<0, NW, TRAIN, TRAIN>(a0, (int)blockIdx.x); else mlp_fwd_body_split<0, NW, TRAIN, false>(a0, (int)blockIdx.x); }
    else { if (hi_only) mlp_fwd_body_split<1, NW, TRAIN, TRAIN>(a1, (int)blockIdx.x - tiles0); else mlp_fwd_body_split<1, NW, TRAIN, false>(a1, (int)blockIdx.x - tiles0); }
  } else {
    if ((int)blockIdx.x < tiles0) mlp_fwd_body<0, P, NW, TRAIN>(a0, (int)blockIdx.x);
    else mlp_fwd_body<1, P, NW, TRAIN>(a1, (int)blockIdx.x - tiles0);
  }
}
template <int P, int NW>
__global__ __launch_bounds__(NW * 64, (P == 1 ? 2 : 1)) void mlp_bwd_pair_kernel(MlpBwdArgs a0, MlpBwdArgs a1, int tiles0) {
  if constexpr (P == 2 && (probe::SPLIT_V2 & 4) != 0) {
    if ((int)blockIdx.x < tiles0) mlp_bwd_body_split<0, NW>(a0, (int)blockIdx.x);
    else mlp_bwd_body_split<1, NW>(a1, (int)blockIdx.x - tiles0);
  } else {
    if ((int)blockIdx.x < tiles0) mlp_bwd_body<0, P, NW>(a0, (int)blockIdx.x);
    else mlp_bwd_body<1, P, NW>(a1, (int)blockIdx.x - tiles0);
  }
}

}  // namespace nerfpp

using namespace nerfpp;

// waves per workgroup: 8 (two per SIMD, <= 256 VGPRs) where a lane holds ONE register image per activation chunk, 4 in split-bf16
#define MLP_WAVES(P) ((P) == 1 ? probe::WAVES_P1 : (P) == 3 ? 8 : 4)

// which: 0 = both nets, 1 = fg only, 2 = bg only
template <int P, bool TRAIN>
static void launch_fwd_t(hipStream_t st, const MlpFwdArgs& a0, const MlpFwdArgs& a1, int which) {
  constexpr int NW = MLP_WAVES(P);
  const int tile = NW * 32;
  const int t0 = which == 2 ? 0 : (int)((a0.rows + tile - 1) / tile), t1 = which == 1 ? 0 : (int)((a1.rows + tile - 1) / tile);
  constexpr size_t l0 = FwdLds<0, P, NW, TRAIN>::TOTAL, l1 = FwdLds<1, P, NW, TRAIN>::TOTAL;
  constexpr size_t lds = (l0 > l1 ? l0 : l1) + probe::STAMP_BYTES;
  static_assert(lds <= 160 * 1024, "LDS budget");
  hipLaunchKernelGGL((mlp_fwd_pair_kernel<P, NW, TRAIN>), dim3(t0 + t1), dim3(NW * 64), lds, st, a0, a1, t0);
}
template <int P>
static void launch_bwd_t(hipStream_t st, const MlpBwdArgs& a0, const MlpBwdArgs& a1, int which) {
  constexpr int NW = MLP_WAVES(P);
  const int tile = NW * 32;
  const int t0 = which == 2 ? 0 : (int)((a0.rows + tile - 1) / tile), t1 = which == 1 ? 0 : (int)((a1.rows + tile - 1) / tile);
  constexpr size_t lds = ((P == 2 && (probe::SPLIT_V2 & 4) != 0) ? (size_t)BwdLdsV2<NW>::TOTAL : (size_t)BwdLds<P, NW>::TOTAL) + probe::STAMP_BYTES;
  static_assert(lds <= 160 * 1024, "LDS budget");
  hipLaunchKernelGGL((mlp_bwd_pair_kernel<P, NW>), dim3(t0 + t1), dim3(NW * 64), lds, st, a0, a1, t0);
}

// The kernels are fully unrolled instruction streams (2 x 1200 MFMAs each) and take minutes to compile, so the in-tree build
// compiles this file once per kernel instantiation: -DNERFPP_MLP_PART=k emits instantiation k only (0 / 1: inference forward
// bf16 / split-bf16, 2 / 3: training forward, 4 / 5: backward, 6: the dispatchers, 7 / 8: fp16x2w inference / training forward),
// no define = everything in one translation unit.
#ifndef NERFPP_MLP_PART
#define NERFPP_MLP_PART -1
#endif
#define MLP_PART(k) (NERFPP_MLP_PART == -1 || NERFPP_MLP_PART == (k))
#define FWD_ENTRY(P, TRAIN) void launch_fwd_##P##_##TRAIN(hipStream_t st, const MlpFwdArgs& a0, const MlpFwdArgs& a1, int which)
#define BWD_ENTRY(P) void launch_bwd_##P(hipStream_t st, const MlpBwdArgs& a0, const MlpBwdArgs& a1, int which)
FWD_ENTRY(1, 0); FWD_ENTRY(2, 0); FWD_ENTRY(1, 1); FWD_ENTRY(2, 1); FWD_ENTRY(3, 0); FWD_ENTRY(3, 1);
BWD_ENTRY(1); BWD_ENTRY(2);
#if MLP_PART(0)
FWD_ENTRY(1, 0) { launch_fwd_t<1, false>(st, a0, a1, which); }
#endif
#if MLP_PART(1)
FWD_ENTRY(2, 0) { launch_fwd_t<2, false>(st, a0, a1, which); }
#endif
#if MLP_PART(2)
FWD_ENTRY(1, 1) { launch_fwd_t<1, true>(st, a0, a1, which); }
#endif
#if MLP_PART(3)
FWD_ENTRY(2, 1) { launch_fwd_t<2, true>(st, a0, a1, which); }
#endif
#if MLP_PART(4)
BWD_ENTRY(1) { launch_bwd_t<1>(st, a0, a1, which); }
#endif
#if MLP_PART(5)
BWD_ENTRY(2) { launch_bwd_t<2>(st, a0, a1, which); }
#endif

#if MLP_PART(7)
FWD_ENTRY(3, 0) { launch_fwd_t<3, false>(st, a0, a1, which); }
#endif
#if MLP_PART(8)
FWD_ENTRY(3, 1) { launch_fwd_t<3, true>(st, a0, a1, which); }
#endif

#if MLP_PART(6)
void launch_mlp_fwd_pair(hipStream_t st, int P, bool train, const MlpFwdArgs& a0, const MlpFwdArgs& a1, int which) {
  if (P == 1)      { if (train) launch_fwd_1_1(st, a0, a1, which); else launch_fwd_1_0(st, a0, a1, which); }
  else if (P == 3) { if (train) launch_fwd_3_1(st, a0, a1, which); else launch_fwd_3_0(st, a0, a1, which); }
  else             { if (train) launch_fwd_2_1(st, a0, a1, which); else launch_fwd_2_0(st, a0, a1, which); }
}
void launch_mlp_bwd_pair(hipStream_t st, int P, const MlpBwdArgs& a0, const MlpBwdArgs& a1, int which) {
  if (P == 1) launch_bwd_1(st, a0, a1, which); else launch_bwd_2(st, a0, a1, which);
}
#endif
