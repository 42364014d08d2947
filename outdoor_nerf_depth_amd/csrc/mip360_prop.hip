// The proposal MLP of MipNeRF-360 (models.py:436-606 with configs/360.gin:12-13 -- 4 x 256, no skip, density head only) as ONE
// launch per level: IPE features [rows, 512] -> relu(Dense) x 4 -> density, the activations between the layers in registers.
//
// Design = the NeRF++ forward (nerfpp_mlp.hip, "samples on lanes") on the fm tensors of mip360_fm.hip:
//   * a workgroup owns 256 rows (one linear_fm tile), a wave 32 of them; lane (row, hi) of v_mfma_f32_32x32x16_bf16 carries the
//     sample as the B operand, the weights are the A operand, so a lane's 16 accumulator registers of out-block ob are the
//     columns 32 ob + {8 (r >> 2) + 4 hi + (r & 3)} of ITS row: registers 8 b .. 8 b + 7 packed to bf16 are at once the 16-byte
//     unit of the fm block (32 ob + 16 b) that is stored for the backward pass and the B operand of the next layer's k step.
//   * the weights stream L2 -> LDS with global_load_lds_dwordx4 through a ring of 16-KiB blocks (2 k steps x 8 out-blocks); a
//     fragment of the stream is one 1-KiB block of the fm operand copy mip360_pack_weight_fm already writes (no repacking: the
//     DMA gathers them), read back by unit index -- conflict-free by the construction of the unit order.
//   * the first layer's B operand (K = 512) are the blocks cast_encode wrote, loaded 16 bytes per lane straight from global
//     memory into a register ring that travels with the weight ring (the two k steps of a block with the block's DMA).
//   * training: H_l leaves as 16 wave-contiguous 1-KiB stores spread over the NEXT layer's weight blocks (its B operand, still in
//     registers), the ReLU bits in linear_fm's mask format (word = row block of the tile, bit 8 j + p + 16 e), so the
//     backward pass (masked dX chain, weight gradients) runs unchanged on what this kernel saved.
//   * the density head is the row-dot of rowdot_fm_kernel<2> on the registers of the last layer: same products in the same
//     order, so density == mip360_rowdot_fm of the saved H3 bit for bit.
// Counted vmcnt: loads and stores share vmcnt and retire in order; the kernel is one unrolled instruction stream, so every
// wait is the exact number of VMEM instructions (DMA, operand loads, saves) issued after the block it waits for.
#include "probe_env.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

namespace mip360prop {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int DEPTH = 4, WIDTH = 256, NOB = 8;          // layers, columns, 32-column out-blocks
constexpr int K0 = 512;                                 // columns of the first layer's operand (the IPE features, padded)
constexpr int NW = 8;                                   // waves per workgroup = 32-row blocks per 256-row tile
constexpr int BLK_FRAGS = 16, BLK_BYTES = BLK_FRAGS * 1024, KPB = BLK_FRAGS / NOB;
#ifndef PROP_NBUF
#define PROP_NBUF 6
#endif
#ifndef PROP_LDS_PREFETCH
#define PROP_LDS_PREFETCH 4
#endif
constexpr int NBUF = PROP_NBUF, AHEAD = NBUF - 1;       // ring slots / blocks in flight ahead of the one being consumed
constexpr int LDS_BIAS = NBUF * BLK_BYTES, LDS_HEAD = LDS_BIAS + DEPTH * WIDTH * 4, LDS_TOTAL = LDS_HEAD + WIDTH * 2;

struct Args {
  int rows;                                             // multiple of 256
  const char* x; int x_bpr, x_blk0;                     // fm input: blocks per row block (ld / 16), first column block
  const char* w[DEPTH]; int w_bpr[DEPTH];               // fm weights [256, ld]: blocks per row block
  const float* bias[DEPTH];
  char* h[DEPTH];                                       // fm outputs [rows, 256] (training)
  uint32_t* mask[DEPTH];                                // linear_fm mask words (training)
  const uint16_t* wd; const float* bd; float act_param; float* density;   // head (density == nullptr: none)
};

extern __shared__ __attribute__((aligned(16))) char smem[];

__device__ __forceinline__ uint32_t unit_of(int row, int hi) { return 8u * (row >> 2) + 4u * (hi ^ (row >> 4)) + (row & 3); }

// LDS-DMA of one 1-KiB fragment: global address = wave-uniform base + 16 * lane, LDS address = dst + 16 * lane
__device__ __forceinline__ void glds_frag(const char* sbase, uint32_t voff, uint32_t lds_abs) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_abs);
  const uint64_t b = (uint64_t)(uintptr_t)sbase;
  const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}

__device__ __forceinline__ void wait_vmcnt(int n) {        // n is a compile-time value after unrolling: one case survives
  switch (n < 0 ? 0 : n) {
#define W_(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    W_(0) W_(1) W_(2) W_(3) W_(4) W_(5) W_(6) W_(7) W_(8) W_(9) W_(10) W_(11) W_(12) W_(13) W_(14) W_(15) W_(16) W_(17) W_(18) W_(19)
    W_(20) W_(21) W_(22) W_(23) W_(24) W_(25) W_(26) W_(27) W_(28) W_(29) W_(30) W_(31) W_(32) W_(33) W_(34) W_(35) W_(36) W_(37) W_(38) W_(39)
    W_(40) W_(41) W_(42) W_(43) W_(44) W_(45) W_(46) W_(47) W_(48)
#undef W_
    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
  }
}

template <bool TRAIN>
__global__ __launch_bounds__(NW * 64, 1) void prop_mlp_fwd_kernel(const Args a) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = lane & 31, hi = lane >> 5;
  const uint32_t u16 = unit_of(row, hi) * 16u;
  const size_t tile = blockIdx.x, rb = tile * NW + wave;                      // this wave's 32-row block
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  constexpr int NBLK0 = K0 / 16 / KPB, LBLK = WIDTH / 16 / KPB, NBLK = NBLK0 + (DEPTH - 1) * LBLK;

  // Every VMEM instruction of a wave is counted (vm_issued; all of it compile-time after unrolling): loads and stores share
  // vmcnt and retire in order, so "block t has landed" = at most (vm_issued - vm_mark[t]) younger operations outstanding.
  int vm_issued = 0, vm_mark[NBLK];
  // ---- the weight stream: block t -> (layer, pair of k steps); every wave fetches two fragments of a block, and (layer 0) the
  // two fragments of ITS rows' operand the block's k steps multiply, into a register ring as deep as the LDS ring
  const uint64_t xbase = (uint64_t)(uintptr_t)(a.x + (rb * (size_t)a.x_bpr + a.x_blk0) * 1024);
  const uint64_t xb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(xbase >> 32)) << 32) |
                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)xbase);
  u32x4 xr[NBUF][KPB];
  int issue_t = 0, slot_issue = 0;
  auto issue = [&]() {
    const int t = issue_t++, slot = slot_issue;
    slot_issue = slot_issue + 1 == NBUF ? 0 : slot_issue + 1;
    if (t >= NBLK) return;
    const int l = t < NBLK0 ? 0 : 1 + (t - NBLK0) / LBLK;
    const int lb = t < NBLK0 ? t : (t - NBLK0) % LBLK;
    if (l == 0) {
#pragma unroll
      for (int kl = 0; kl < KPB; ++kl)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(xr[slot][kl]) : "v"(u16), "s"(xb + (uint64_t)(lb * KPB + kl) * 1024) : "memory");
      vm_issued += KPB;
    }
    const int kc = lb * KPB + (wave >> 2), ob = 2 * (wave & 3);                // fragments (kl, ob), (kl, ob + 1), kl = wave / 4
    const char* src = a.w[l] + ((size_t)ob * a.w_bpr[l] + kc) * 1024;
    const uint32_t dst = lds0 + slot * BLK_BYTES + ((wave >> 2) * NOB + ob) * 1024;
    glds_frag(src, (uint32_t)lane * 16u, dst);
    glds_frag(src + (size_t)a.w_bpr[l] * 1024, (uint32_t)lane * 16u, dst + 1024u);
    vm_issued += 2;
    vm_mark[t] = vm_issued;
  };
  int step = 0, slot_cur = 0;
  auto acquire = [&]() -> int {                                                 // returns the slot of block `step`
    wait_vmcnt(vm_issued - vm_mark[step]);
    __builtin_amdgcn_s_barrier();
    issue();                                                                    // into the slot the barrier freed
    const int s_ = slot_cur;
    slot_cur = slot_cur + 1 == NBUF ? 0 : slot_cur + 1;
    ++step;
    return s_;
  };
#pragma unroll
  for (int b = 0; b < AHEAD; ++b) issue();

  // ---- biases and the head's weights to LDS
  for (int i = threadIdx.x; i < DEPTH * WIDTH / 4; i += NW * 64)
    *(float4*)(smem + LDS_BIAS + i * 16) = ((const float4*)a.bias[i / (WIDTH / 4)])[i % (WIDTH / 4)];
  if (a.density && threadIdx.x < WIDTH / 8) *(uint4*)(smem + LDS_HEAD + threadIdx.x * 16) = ((const uint4*)a.wd)[threadIdx.x];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  f32x16 acc[NOB];
  auto init_bias = [&](int l) {
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *(const float4*)(smem + LDS_BIAS + (l * WIDTH + ob * 32 + 8 * q + 4 * hi) * 4);
        acc[ob][4 * q] = v.x; acc[ob][4 * q + 1] = v.y; acc[ob][4 * q + 2] = v.z; acc[ob][4 * q + 3] = v.w;
      }
  };
  auto mfma_block = [&](int slot, int kl, const bf16x8 b) {
    const char* l = smem + slot * BLK_BYTES + u16;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
      acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(l + (kl * NOB + ob) * 1024), b, acc[ob], 0, 0, 0);
  };
  // shape a block's schedule: PROP_LDS_PREFETCH weight fragments in flight ahead of the MFMA that consumes them
  auto shape_block = [&]() {
    if constexpr (PROP_LDS_PREFETCH > 0) {
      constexpr int D = PROP_LDS_PREFETCH, NF = KPB * NOB;
#pragma unroll
      for (int i = 0; i < D; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
      for (int i = 0; i < NF - D; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < D; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  };
  // ReLU, bf16, mask bits: accumulators -> the 16 fragments of the layer's output (and the layer's mask words to global memory)
  u32x4 h[WIDTH / 16];
  auto epilogue = [&](int l) {
    uint32_t words[4];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
      uint32_t word = (ob & 1) ? words[ob >> 1] : 0u;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const f32x2 f = {acc[ob][2 * p], acc[ob][2 * p + 1]};
        uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
        w = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), (s16x2){0, 0}));
        uint32_t nz;
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz) : "v"(w), "v"(0x00010001u));
        word |= nz << (8 * (ob & 1) + p);
        h[2 * ob + (p >> 2)][p & 3] = w;
      }
      words[ob >> 1] = word;
    }
    if constexpr (TRAIN) {
      // linear_fm's layout: one uint4 per (tile, its wave 4 wm + wn, lane), word i = row block i of that wave's 128 rows
      uint32_t* m = a.mask[l] + (((size_t)tile * 8 + (wave >> 2) * 4) * 64 + lane) * 4 + (wave & 3);
#pragma unroll
      for (int wn = 0; wn < 4; ++wn) m[(size_t)wn * 64 * 4] = words[wn];
      vm_issued += 4;
    }
    // (the next layer's bias reads must not be hoisted above the conversions: 128 old + 128 new accumulators would not fit)
    __builtin_amdgcn_sched_barrier(0);
  };
  auto save_frag = [&](int l, int c) {
    if constexpr (TRAIN) {
      __builtin_nontemporal_store(h[c], (u32x4*)(a.h[l] + (rb * (WIDTH / 16) + c) * 1024 + u16));
      vm_issued += 1;
    }
  };

  // ---- layer 0: the B operand arrives with the weight block
  init_bias(0);
#pragma unroll
  for (int b = 0; b < NBLK0; ++b) {
    const int slot = acquire();
#pragma unroll
    for (int kl = 0; kl < KPB; ++kl) asm volatile("" : "+v"(xr[slot][kl]));      // (consumed here, behind the wait that covered the loads)
#pragma unroll
    for (int kl = 0; kl < KPB; ++kl) mfma_block(slot, kl, __builtin_bit_cast(bf16x8, xr[slot][kl]));
    shape_block();
    __builtin_amdgcn_sched_barrier(0);                          // (nothing moves across a block boundary: left alone, the
  }                                                             //  scheduler sinks MFMAs below the next barrier and spills fragments)
  epilogue(0);

  // ---- layers 1 .. 3: B operand = the previous layer's fragments, which leave for global memory two per weight block
#pragma unroll
  for (int l = 1; l < DEPTH; ++l) {
    init_bias(l);
#pragma unroll
    for (int b = 0; b < LBLK; ++b) {
      const int slot = acquire();
#pragma unroll
      for (int kl = 0; kl < KPB; ++kl) mfma_block(slot, kl, __builtin_bit_cast(bf16x8, h[KPB * b + kl]));
      shape_block();
#pragma unroll
      for (int kl = 0; kl < KPB; ++kl) save_frag(l - 1, KPB * b + kl);
      __builtin_amdgcn_sched_barrier(0);
    }
    epilogue(l);
  }
#pragma unroll
  for (int c = 0; c < WIDTH / 16; ++c) save_frag(DEPTH - 1, c);

  // ---- density head: rowdot_fm_kernel<2>'s products in its order
  if (a.density) {
    float d = 0.f;
    const char* wl = smem + LDS_HEAD + 8 * hi;
#pragma unroll
    for (int c = 0; c < WIDTH / 16; ++c) {
      const uint2 w0 = *(const uint2*)(wl + c * 32), w1 = *(const uint2*)(wl + c * 32 + 16);
      const uint32_t ws[4] = {w0.x, w0.y, w1.x, w1.y};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        d += __builtin_bit_cast(float, h[c][q] << 16) * __builtin_bit_cast(float, ws[q] << 16);
        d += __builtin_bit_cast(float, h[c][q] & 0xFFFF0000u) * __builtin_bit_cast(float, ws[q] & 0xFFFF0000u);
      }
    }
    d += __shfl_xor(d, 32, 64);
    if (hi == 0) {
      const float x = (d + (a.bd ? a.bd[0] : 0.f)) + a.act_param;
      a.density[rb * 32 + row] = fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The dX chain of the same MLP (jax.grad through the four ReLU layers) as one launch:
//   dZ_3 = mask_3 . bf16(z (x) w_density)          (mip360_outer_masked_fm)
//   dZ_{l-1} = mask_{l-1} . bf16(dZ_l W_l)          (mip360_linear_fm act 2), l = 3, 2, 1
// with dZ_l in registers between the layers; every dZ_l is written (fm) for the weight-gradient GEMMs, two fragments per weight
// block of the layer that consumes it.  Same MFMA sequence per output element as linear_fm (k ascending, no bias): bit-identical.
struct BwdArgs {
  int rows;                                             // multiple of 256
  const uint16_t* z;                                    // d loss / d raw density, bf16 [rows]
  const uint16_t* wd;                                   // density kernel, bf16 [256]
  const uint32_t* mask[DEPTH];                          // mask words of H_l (written by the forward)
  const char* wb[DEPTH]; int wb_bpr[DEPTH];             // l = 1 .. 3: fm [256 inputs, ld] backward operand of layer l
  char* dz[DEPTH];                                      // fm [rows, 256]
};

__global__ __launch_bounds__(NW * 64, 1) void prop_mlp_bwd_kernel(const BwdArgs a) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = lane & 31, hi = lane >> 5;
  const uint32_t u16 = unit_of(row, hi) * 16u;
  const size_t tile = blockIdx.x, rb = tile * NW + wave;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  constexpr int LBLK = WIDTH / 16 / KPB, NBLK = (DEPTH - 1) * LBLK;

  // side data first (older than every counted operation): this wave's mask words of the four layers, its rows' z
  uint32_t mw[DEPTH][4];
#pragma unroll
  for (int l = 0; l < DEPTH; ++l) {
    const uint32_t* m = a.mask[l] + (((size_t)tile * 8 + (wave >> 2) * 4) * 64 + lane) * 4 + (wave & 3);
#pragma unroll
    for (int wn = 0; wn < 4; ++wn) mw[l][wn] = m[(size_t)wn * 64 * 4];
  }
  const float zv = __builtin_bit_cast(float, (uint32_t)a.z[rb * 32 + row] << 16);
  if (threadIdx.x < WIDTH / 8) *(uint4*)(smem + LDS_HEAD + threadIdx.x * 16) = ((const uint4*)a.wd)[threadIdx.x];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int vm_issued = 0, vm_mark[NBLK];
  int issue_t = 0, slot_issue = 0;
  auto issue = [&]() {
    const int t = issue_t++, slot = slot_issue;
    slot_issue = slot_issue + 1 == NBUF ? 0 : slot_issue + 1;
    if (t >= NBLK) return;
    const int l = DEPTH - 1 - t / LBLK, lb = t % LBLK;                          // layers 3, 2, 1
    const int kc = lb * KPB + (wave >> 2), ob = 2 * (wave & 3);
    const char* src = a.wb[l] + ((size_t)ob * a.wb_bpr[l] + kc) * 1024;
    const uint32_t dst = lds0 + slot * BLK_BYTES + ((wave >> 2) * NOB + ob) * 1024;
    glds_frag(src, (uint32_t)lane * 16u, dst);
    glds_frag(src + (size_t)a.wb_bpr[l] * 1024, (uint32_t)lane * 16u, dst + 1024u);
    vm_issued += 2;
    vm_mark[t] = vm_issued;
  };
  int step = 0, slot_cur = 0;
  auto acquire = [&]() -> int {
    wait_vmcnt(vm_issued - vm_mark[step]);
    __builtin_amdgcn_s_barrier();
    issue();
    const int s_ = slot_cur;
    slot_cur = slot_cur + 1 == NBUF ? 0 : slot_cur + 1;
    ++step;
    return s_;
  };
#pragma unroll
  for (int b = 0; b < AHEAD; ++b) issue();

  // dZ_3: the masked outer product, fragment by fragment
  u32x4 dzf[WIDTH / 16];
  auto mask_of = [&](int l, int ob, int p) { return (mw[l][ob >> 1] >> (8 * (ob & 1) + p)) & 0x00010001u; };
  {
    const char* wl = smem + LDS_HEAD + 8 * hi;
#pragma unroll
    for (int c = 0; c < WIDTH / 16; ++c) {
      const uint2 w0 = *(const uint2*)(wl + c * 32), w1 = *(const uint2*)(wl + c * 32 + 16);
      const uint32_t ws[4] = {w0.x, w0.y, w1.x, w1.y};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2 f = {zv * __builtin_bit_cast(float, ws[q] << 16), zv * __builtin_bit_cast(float, ws[q] & 0xFFFF0000u)};
        uint32_t v = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
        v &= mask_of(DEPTH - 1, c >> 1, 4 * (c & 1) + q) * 0xFFFFu;
        dzf[c][q] = v;
      }
    }
  }
  auto save_frag = [&](int l, int c) {
    __builtin_nontemporal_store(dzf[c], (u32x4*)(a.dz[l] + (rb * (WIDTH / 16) + c) * 1024 + u16));
    vm_issued += 1;
  };
  f32x16 acc[NOB];
#pragma unroll
  for (int l = DEPTH - 1; l >= 1; --l) {
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ob][r] = 0.f;
#pragma unroll
    for (int b = 0; b < LBLK; ++b) {
      const int slot = acquire();
      const char* lw = smem + slot * BLK_BYTES + u16;
#pragma unroll
      for (int kl = 0; kl < KPB; ++kl)
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
          acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(lw + (kl * NOB + ob) * 1024),
                                                            __builtin_bit_cast(bf16x8, dzf[KPB * b + kl]), acc[ob], 0, 0, 0);
      if constexpr (PROP_LDS_PREFETCH > 0) {
        constexpr int D = PROP_LDS_PREFETCH, NF = KPB * NOB;
#pragma unroll
        for (int i = 0; i < D; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int i = 0; i < NF - D; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < D; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
#pragma unroll
      for (int kl = 0; kl < KPB; ++kl) save_frag(l, KPB * b + kl);
      __builtin_amdgcn_sched_barrier(0);
    }
    // bf16, then the ReLU mask of the layer below: halves times their bit (v_pk_mul_lo_u16), as linear_fm act 2
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const f32x2 f = {acc[ob][2 * p], acc[ob][2 * p + 1]};
        uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        w = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, w) * __builtin_bit_cast(u16x2, mask_of(l - 1, ob, p)));
        dzf[2 * ob + (p >> 2)][p & 3] = w;
      }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int c = 0; c < WIDTH / 16; ++c) save_frag(0, c);
}

}  // namespace mip360prop

// hipFuncSetAttribute is per device: remember which devices of this process have had it applied (one bit per device id)
static inline bool prop_first_launch_on_this_device(std::atomic<uint64_t>& done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  return (done.fetch_or(bit) & bit) == 0;
}

// x_fm [rows, ldx] from column x_col0 (512 columns), w_fm[l] [256, ldw[l]] (l = 0: 512 live columns, else 256), bias[l] [256];
// h_fm / masks != nullptr: training (H_l [rows, 256] fm and linear_fm mask words written for every layer);
// density != nullptr: softplus(H_3 . wd + bd[0] + act_param) per row.  Returns 1 for shapes the kernel does not take.
int mip360_launch_prop_mlp_fm(hipStream_t st, int rows, const void* x_fm, int ldx, int x_col0, const void* const* w_fm, const int* ldw,
                              const float* const* bias, void* const* h_fm, void* const* masks, const void* wd, const float* bd,
                              float act_param, float* density) {
  using namespace mip360prop;
  if (rows <= 0 || rows % 256 || ldx % 16 || x_col0 % 16 || x_col0 < 0 || ldx < x_col0 + K0 || !x_fm || !w_fm || !ldw || !bias) return 1;
  if ((h_fm == nullptr) != (masks == nullptr) || (density && !wd) || (!density && !h_fm)) return 1;
  Args a{};
  a.rows = rows;
  a.x = (const char*)x_fm; a.x_bpr = ldx / 16; a.x_blk0 = x_col0 / 16;
  for (int l = 0; l < DEPTH; ++l) {
    if (!w_fm[l] || !bias[l] || ldw[l] % 16 || ldw[l] < (l == 0 ? K0 : WIDTH)) return 1;
    a.w[l] = (const char*)w_fm[l]; a.w_bpr[l] = ldw[l] / 16; a.bias[l] = bias[l];
    if (h_fm) {
      if (!h_fm[l] || !masks[l]) return 1;
      a.h[l] = (char*)h_fm[l]; a.mask[l] = (uint32_t*)masks[l];
    }
  }
  a.wd = (const uint16_t*)wd; a.bd = bd; a.act_param = act_param; a.density = density;
  const bool train = h_fm != nullptr;
  static std::atomic<uint64_t> done_t{0}, done_i{0};
  if (train) {
    if (prop_first_launch_on_this_device(done_t))
      if (hipFuncSetAttribute((const void*)prop_mlp_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess) return 3;
    hipLaunchKernelGGL(prop_mlp_fwd_kernel<true>, dim3(rows / 256), dim3(NW * 64), LDS_TOTAL, st, a);
  } else {
    if (prop_first_launch_on_this_device(done_i))
      if (hipFuncSetAttribute((const void*)prop_mlp_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess) return 3;
    hipLaunchKernelGGL(prop_mlp_fwd_kernel<false>, dim3(rows / 256), dim3(NW * 64), LDS_TOTAL, st, a);
  }
  return 0;
}

// z bf16 [rows], wd bf16 [256], masks[l] of the forward, wb_fm[l] (l = 1 .. 3) fm [256, ldwb[l]], dz_fm[l] fm [rows, 256] out
int mip360_launch_prop_mlp_bwd_fm(hipStream_t st, int rows, const void* z, const void* wd, const void* const* masks,
                                  const void* const* wb_fm, const int* ldwb, void* const* dz_fm) {
  using namespace mip360prop;
  if (rows <= 0 || rows % 256 || !z || !wd || !masks || !wb_fm || !ldwb || !dz_fm) return 1;
  BwdArgs a{};
  a.rows = rows; a.z = (const uint16_t*)z; a.wd = (const uint16_t*)wd;
  for (int l = 0; l < DEPTH; ++l) {
    if (!masks[l] || !dz_fm[l]) return 1;
    a.mask[l] = (const uint32_t*)masks[l]; a.dz[l] = (char*)dz_fm[l];
    if (l >= 1) {
      if (!wb_fm[l] || ldwb[l] % 16 || ldwb[l] < WIDTH) return 1;
      a.wb[l] = (const char*)wb_fm[l]; a.wb_bpr[l] = ldwb[l] / 16;
    }
  }
  static std::atomic<uint64_t> done{0};
  if (prop_first_launch_on_this_device(done))
    if (hipFuncSetAttribute((const void*)prop_mlp_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess) return 3;
  hipLaunchKernelGGL(prop_mlp_bwd_kernel, dim3(rows / 256), dim3(NW * 64), LDS_TOTAL, st, a);
  return 0;
}
