// Fragment-major dense layers of the MipNeRF-360 MLPs (models.py:436-606 nn.Dense + nn.relu, and the dX chain of
// jax.grad): the activations between the wide layers never exist in row-major form.
//
// Layout ("fm").  A [rows, ld] bf16 tensor (rows % 32 == 0, ld % 16 == 0) is stored as 1 KiB blocks of 32 rows x 16
// columns, block (r / 32, c / 16) at ((r / 32) * (ld / 16) + c / 16) * 1024.  Inside a block the 16-byte unit of
// (row, hi) -- hi = ((c % 16) / 4) % 2 -- holds the 8 columns {4 hi + 0..3, 8 + 4 hi + 0..3} (element t = 4 ((c % 16) / 8)
// + c % 4) and sits at unit index  u = 8 (row >> 2) + 4 (hi ^ (row >> 4)) + (row & 3).
//   * A unit is exactly what one lane of v_mfma_f32_32x32x16_bf16 supplies for a 16-wide k step (lane = row + 32 hi; the
//     k order inside the step is the same permutation for both operands, so the contraction is exact), AND what one lane
//     of the 32 x 32 accumulator block owns of a 16-column block of the output (registers 8 b .. 8 b + 7 of lane
//     (row, hi) are columns 16 b + {4 hi + 0..3, 8 + 4 hi + 0..3}).  So a layer's epilogue stores its accumulators with
//     plain 16-byte stores, 1 KiB contiguous per wave instruction, no LDS staging, and the next layer's operand DMA
//     (global_load_lds_dwordx4) is a linear copy of whole blocks whose LDS image needs no swizzle.
//   * The unit order makes BOTH read patterns conflict-free: ds_read_b128 by lane (row, hi) (the 16-lane service groups
//     of MI355X_MICROARCH.md "LDS" see 16 distinct units mod 16), and the ds_read_b64_tr_b16 patches of the weight-
//     gradient kernel (4 rows x 16 columns = 8 consecutive units).
//
// Kernel (linear_fm_kernel): C = act(A W^T + b) on 256 x 256 tiles, 8 waves of 128 x 64 (4 x 2 MFMA blocks, operands
// swapped so that the accumulator block is C^T: lane = row).  Persistent workgroups; the operand stream is ONE ring of
// five 32 KiB slots (a slot = 32 k-elements of both operands) that runs across tile boundaries: the first half-steps of
// the next tile are in flight while the current tile finishes, and the epilogue (registers -> global) does not touch
// LDS.  The two wave groups run half a step apart (one multiplies while the other reads fragments / issues DMA / stores)
// as in linear_bf16_pp64_kernel.  Every VMEM instruction is inline asm: loads and stores share vmcnt and return in
// order, the counted waits below rely on the exact instruction sequence (and a compiler that saw the stores would drain
// vmcnt at every barrier).
#include "probe_env.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

// hipFuncSetAttribute is per device: remember which devices of this process have had it applied (one bit per device id)
#include <atomic>
static inline bool first_launch_on_this_device(std::atomic<uint64_t>& done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  return (done.fetch_or(bit) & bit) == 0;
}

namespace mip360fm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__host__ __device__ __forceinline__ uint32_t unit_of(int row, int hi) { return 8u * (row >> 2) + 4u * (hi ^ (row >> 4)) + (row & 3); }

extern __shared__ __attribute__((aligned(16))) char fm_smem[];

// LDS-DMA of 64 x 16 bytes: global address = wave-uniform base + per-lane offset, LDS address = dst + 16 * lane
__device__ __forceinline__ void glds16(const void* sbase, uint32_t voff, uint32_t lds_abs) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_abs);
  const uint64_t b = (uint64_t)(uintptr_t)sbase;
  const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}
}  // namespace mip360fm
// Experiment switches (component removal, DMA issue forms, store cache policies, ring depth, start-up stagger) exist only
// in diagnostic builds: -DNERFPP_PROBES takes their variants from mip360_fm_probes.h (tools/probes/mip360_variant.sh); the
// shipped library sees the constants, the one DMA issue form and the one store policy below.
#ifdef NERFPP_PROBES
#include "mip360_fm_probes.h"
#else
namespace mip360fm { namespace probe {
constexpr int NSLOT = 5;               // LDS ring slots of 32 KiB
constexpr int NSTORE = 16;             // output stores per wave and tile
constexpr int DMA_PER = 4;             // DMA wave-instructions per wave and half-step
constexpr int STAGGER = 0;
constexpr bool NOREAD = false, NOMFMA = false, SETPRIO = false;
__device__ __forceinline__ void issue_half_step(const char* iA, const char* iW, uint32_t voff, uint32_t d, uint32_t woff) {
  glds16(iA, voff, d);
  glds16(iA + 1024, voff, d + 1024u);
  glds16(iW, voff, d + woff);
  glds16(iW + 1024, voff, d + woff + 1024u);
}
}}  // namespace mip360fm::probe
#define FM_ST " nt"                    // cache policy of the output stores (measured best: -0.8 % per step)
#endif
namespace mip360fm {
__device__ __forceinline__ uint64_t uniform64(const void* p) {
  const uint64_t b = (uint64_t)(uintptr_t)p;
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
         (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
}

struct Cfg {
  static constexpr int SLOT = 32768, NSLOT = probe::NSLOT, WOFF = 16384, LDS = SLOT * NSLOT;
  static constexpr int AHEAD = NSLOT - 1;           // the fragment-read phase of half-step x issues the DMA of x + AHEAD
};
static_assert(Cfg::AHEAD == 3 || Cfg::AHEAD == 4, "ring depth 4 or 5");

// ACT 0: C = A W^T + b;  1: relu(A W^T + b), bit mask of the non-zero outputs to `mask` (when not null);
//     2: (A W^T) with the elements whose mask bit is clear set to zero (the dX chain; no bias)
// mask: one uint4 per (tile, wave, lane): word i = MFMA row block i of the wave, bit 8 j + p + 16 e = accumulator
// registers 2 p + e of column block j -- the same lanes own the same elements in the layer that writes it and the
// one that applies it (equal tile shapes).
template <int ACT>
__global__ __launch_bounds__(512, 1)
void linear_fm_kernel(int M, int N, int K, const char* __restrict__ A, int lda, const char* __restrict__ W, int ldw,
                      const float* __restrict__ bias, char* __restrict__ C, int ldc, u32x4* __restrict__ mask) {
  constexpr int SLOT = Cfg::SLOT;
  // VMEM instructions of one epilogue + preload (older than the DMA issued after it): output stores, mask store / load, bias loads
  constexpr int NSTORE = probe::NSTORE;
  constexpr int NE = ACT == 0 ? NSTORE + 2 : ACT == 1 ? NSTORE + 1 + 2 : NSTORE + 1;
  constexpr int DMA_PER = probe::DMA_PER;
  constexpr int AHEAD = Cfg::AHEAD, VM_STEADY = (AHEAD - 1) * DMA_PER, VM_PEEL = VM_STEADY + NE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, grp = wm;            // group 0 = waves 0-3 = tile rows 0-127
  const int tiles_n = N >> 8, tiles_m = M >> 8, tiles = tiles_m * tiles_n;
  const int nh = K >> 5;                                        // half-steps (32 k-elements) per tile, > AHEAD
  const int T = (tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)fm_smem;
  const int row = lane & 31, hi = lane >> 5;
  const uint32_t u16 = unit_of(row, hi) * 16u;
  const uint32_t pa0 = lds0 + u16 + (uint32_t)wm * 8192u, pb0 = lds0 + Cfg::WOFF + u16 + (uint32_t)wn * 4096u;
  const uint32_t voff = (uint32_t)lane * 16u;
  const size_t a_rb = (size_t)(lda >> 4) * 1024, w_rb = (size_t)(ldw >> 4) * 1024, c_rb = (size_t)(ldc >> 4) * 1024;

  auto decode = [&](int t, int& tile_m, int& tile_n) {          // XCD-aware order (gridDim.x % 8 == 0 or one tile each)
    const int vb = (int)blockIdx.x + t * (int)gridDim.x;
    const int xcd = vb & 7, id = vb >> 3;
    const int full = (tiles_m / 8) * 8;
    const int group = id / tiles_n;
    if (group * 8 + 8 <= full) { tile_m = group * 8 + xcd; tile_n = id - group * tiles_n; }
    else { const int r = vb - full * tiles_n; tile_m = full + r / tiles_n; tile_n = r - (r / tiles_n) * tiles_n; }
  };

  // ---- DMA issue cursor: wave w fetches row block w of the A tile and row block w of the W tile, 2 x 1 KiB each per half-step
  int it = 0, ih = 0;
  uint32_t islot = 0;
  const char *iA = A, *iW = W;
  auto set_issue_tile = [&](int t) {
    int tm, tn;
    decode(t, tm, tn);
    iA = A + (size_t)(tm * 8 + wave) * a_rb;
    iW = W + (size_t)(tn * 8 + wave) * w_rb;
  };
  auto issue = [&]() {
    const uint32_t d = lds0 + islot + (uint32_t)wave * 2048u;
    probe::issue_half_step(iA, iW, voff, d, Cfg::WOFF);
    islot = islot == (Cfg::NSLOT - 1) * SLOT ? 0u : islot + SLOT;
    iA += 2048; iW += 2048;
    if (++ih == nh) { ih = 0; ++it; set_issue_tile(it < T ? it : 0); }   // past the last tile: re-loads of tile 0 into free slots keep the counts uniform
  };

  bf16x8 fa0[4], fb0[2], fa1[4], fb1[2];
  f32x16 acc[4][2];
#define FM_READ(s_)                                                                                                    \
  if constexpr (!probe::NOREAD) {                                                                                      \
    const uint32_t pa = pa0 + (s_), pb = pb0 + (s_);                                                                   \
    asm volatile("ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:2048\n\tds_read_b128 %2, %12 offset:4096\n\t"    \
                 "ds_read_b128 %3, %12 offset:6144\n\tds_read_b128 %4, %13\n\tds_read_b128 %5, %13 offset:2048\n\t"     \
                 "ds_read_b128 %6, %12 offset:1024\n\tds_read_b128 %7, %12 offset:3072\n\tds_read_b128 %8, %12 offset:5120\n\t" \
                 "ds_read_b128 %9, %12 offset:7168\n\tds_read_b128 %10, %13 offset:1024\n\tds_read_b128 %11, %13 offset:3072"   \
                 : "=&v"(fa0[0]), "=&v"(fa0[1]), "=&v"(fa0[2]), "=&v"(fa0[3]), "=&v"(fb0[0]), "=&v"(fb0[1]),            \
                   "=&v"(fa1[0]), "=&v"(fa1[1]), "=&v"(fa1[2]), "=&v"(fa1[3]), "=&v"(fb1[0]), "=&v"(fb1[1])             \
                 : "v"(pa), "v"(pb) : "memory");                                                                       \
  }
#define FM_LGKM0()                                                                                                     \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                  \
               : "+v"(fa0[0]), "+v"(fa0[1]), "+v"(fa0[2]), "+v"(fa0[3]), "+v"(fb0[0]), "+v"(fb0[1]), "+v"(fa1[0]),      \
                 "+v"(fa1[1]), "+v"(fa1[2]), "+v"(fa1[3]), "+v"(fb1[0]), "+v"(fb1[1]) :: "memory")
#define FM_MFMA(a_, b_, c_) (probe::NOMFMA ? (c_) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, 0, 0, 0))
#define FM_PRIO(x_) { if constexpr (probe::SETPRIO) __builtin_amdgcn_s_setprio(x_); }
#define FM_MUL()                                                                                                       \
  {                                                                                                                    \
    FM_PRIO(1);                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                      \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                    \
        acc[i][j] = FM_MFMA(fb0[j], fa0[i], acc[i][j]);                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                      \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                    \
        acc[i][j] = FM_MFMA(fb1[j], fa1[i], acc[i][j]);                                                                 \
    FM_PRIO(0);                                                                                                        \
  }
#define FM_MUL_FIRST()                                                                                                 \
  {                                                                                                                    \
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                      \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                    \
        acc[i][j] = FM_MFMA(fb0[j], fa0[i], zero);                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                      \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                    \
        acc[i][j] = FM_MFMA(fb1[j], fa1[i], acc[i][j]);                                                                 \
  }
#define FM_VMCNT(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define FM_PHASE_END()                                                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                                                   \
  __builtin_amdgcn_s_barrier();                                                                                        \
  __builtin_amdgcn_sched_barrier(0);
  auto next_slot = [&](uint32_t s) { return s == (Cfg::NSLOT - 1) * SLOT ? 0u : s + SLOT; };

  // ---- per-tile side data, fetched one tile ahead (asm loads: covered by the counted waits of the following half-steps)
  uint32_t bias_raw[2] = {0u, 0u};                              // float bits of bias[n0 + 64 wn + 32 j + row]
  u32x4 mw = {0u, 0u, 0u, 0u};                                  // ACT 2: this lane's mask words of the tile
  auto preload = [&](int t) {
    int tm, tn;
    decode(t < T ? t : 0, tm, tn);
    if (ACT != 2) {
      const uint64_t b = uniform64(bias + tn * 256 + wn * 64);
      const uint32_t o = (uint32_t)row * 4u;
      asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %3 offset:128"
                   : "=&v"(bias_raw[0]), "=&v"(bias_raw[1]) : "v"(o), "s"(b) : "memory");
    } else {
      const uint64_t b = uniform64(mask + ((size_t)(tm * tiles_n + tn) * 8 + wave) * 64);
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(mw) : "v"(voff), "s"(b) : "memory");
    }
  };

  // ---- epilogue of tile t: bias through the matrix pipe, activation, bf16, 16 wave-contiguous stores
  auto bias_mfma = [&]() {
    if (ACT == 2) return;
    asm volatile("" : "+v"(bias_raw[0]), "+v"(bias_raw[1]));      // (consumed here, behind the waits that covered the loads)
    bf16x8 ones, bfr[2];
    {
      u32x4 o = {hi ? 0u : 0x3F803F80u, 0u, 0u, 0u};
      ones = __builtin_bit_cast(bf16x8, o);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float b = __builtin_bit_cast(float, bias_raw[j]);
      const __bf16 bh = (__bf16)b;
      const __bf16 bl = (__bf16)(b - (float)bh);
      const uint32_t w0 = (uint32_t)__builtin_bit_cast(uint16_t, bh) | ((uint32_t)__builtin_bit_cast(uint16_t, bl) << 16);
      u32x4 v = {hi ? 0u : w0, 0u, 0u, 0u};
      bfr[j] = __builtin_bit_cast(bf16x8, v);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], ones, acc[i][j], 0, 0, 0);
  };
  // (a store's data registers are read a few cycles after issue: the s_nop behind every asm store keeps the compiler's next
  // VALU write -- it cannot see that the asm is a store -- out of that window)
  auto epilogue = [&](int t) {
    int tm, tn;
    decode(t, tm, tn);
    if (ACT == 2) asm volatile("" : "+v"(mw));
    u32x4 mout = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint64_t cb = uniform64(C + (size_t)(tm * 8 + wm * 4 + i) * c_rb + (size_t)(tn * 16 + wn * 4) * 1024);
      uint32_t word = 0u;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        uint32_t pk[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const f32x2 f = {acc[i][j][2 * p], acc[i][j][2 * p + 1]};
          uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
          if (ACT == 1) {
            // ReLU on the packed pair: a negative bf16 (or -0) is a negative int16 -> v_pk_max_i16 with 0; non-zero flag per
            // half: v_pk_min_u16 with 1
            w = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), (s16x2){0, 0}));
            uint32_t nz;                                           // (asm: the compiler turns min(max(x, 0), 1) into compares and selects)
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz) : "v"(w), "v"(0x00010001u));
            word |= nz << (8 * j + p);
          }
          if (ACT == 2) {                                        // halves times their mask bit (v_pk_mul_lo_u16)
            const uint32_t m = (mw[i] >> (8 * j + p)) & 0x00010001u;
            w = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, w) * __builtin_bit_cast(u16x2, m));
          }
          pk[p] = w;
        }
        const u32x4 lo = {pk[0], pk[1], pk[2], pk[3]}, hi4 = {pk[4], pk[5], pk[6], pk[7]};
        if constexpr (probe::NSTORE == 0) asm volatile("" ::"v"(u16), "v"(lo), "v"(hi4), "s"(cb));
        else if (j == 0)
          asm volatile("global_store_dwordx4 %0, %1, %3" FM_ST "\n\tglobal_store_dwordx4 %0, %2, %3 offset:1024" FM_ST "\n\ts_nop 1" ::"v"(u16), "v"(lo), "v"(hi4), "s"(cb) : "memory");
        else
          asm volatile("global_store_dwordx4 %0, %1, %3 offset:2048" FM_ST "\n\tglobal_store_dwordx4 %0, %2, %3 offset:3072" FM_ST "\n\ts_nop 1" ::"v"(u16), "v"(lo), "v"(hi4), "s"(cb) : "memory");
      }
      mout[i] = word;
    }
    if (ACT == 1) {
      // (a null mask pointer still issues the store -- to a scratch line the launcher provides -- so that the counts hold)
      const uint64_t b = uniform64(mask + ((size_t)(tm * tiles_n + tn) * 8 + wave) * 64);
      asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(mout), "s"(b) : "memory");
    }
  };

  if constexpr (probe::STAGGER > 0)       // (probes: start-up classes 3.4 us apart per XCD)
    for (int i = 0; i < (int)((blockIdx.x >> 3) & 3) * probe::STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  // ---- prologue: half-steps 0 .. AHEAD-1 and the first tile's side data, synchronously
  set_issue_tile(0);
#pragma unroll
  for (int x = 0; x < AHEAD; ++x) issue();
  preload(0);
  FM_VMCNT(0);
  asm volatile("" : "+v"(bias_raw[0]), "+v"(bias_raw[1]), "+v"(mw));
  FM_PHASE_END();
  uint32_t cs = 0;                                              // slot of the half-step whose fragments are read next
  // Schedule (x = half-step counted over all tiles of the workgroup):
  //   group 0: [MUL(x), wait] | [read(x + 1), DMA(x + 5)]          group 1: [read(x), DMA(x + 4), wait] | [MUL(x)]
  // "wait" = own DMA of half-step x + 1 has landed (3 younger half-steps of 4 instructions may be out; + NE while the
  // epilogue's instructions are younger than it, i.e. for the first AHEAD half-steps of a tile -- the DMA of the phase that holds
  // the epilogue is issued ahead of its stores; the first tile's first AHEAD - 1 were awaited by the prologue).  The barrier after it publishes x + 1 to both groups one phase before they read it.
  // The wait of half-step AHEAD - 1 of a tile targets the first DMA issued after the tile switch: behind the previous tile's
  // epilogue in every tile but the first (whose prologue did not cover it).
#define FM_VM_LASTPEEL() { if (t > 0) { FM_VMCNT(VM_PEEL); } else { FM_VMCNT(VM_STEADY); } }
  if (grp == 0) {
    FM_READ(cs); issue(); FM_LGKM0(); cs = next_slot(cs);
    FM_PHASE_END();
    for (int t = 0; t < T; ++t) {
#define FM_G0_STEP(MUL_, WAIT_)                                                                                        \
      MUL_;                                                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      WAIT_;                                                                                                           \
      FM_PHASE_END();                                                                                                  \
      FM_READ(cs); issue(); FM_LGKM0(); cs = next_slot(cs);                                                            \
      FM_PHASE_END();
      // the last half-step of a tile: the DMA of this phase is issued BEFORE the epilogue's stores (it is then older than they
      // are: one more half-step before a wait needs them retired)
#define FM_G0_LAST(WAIT_)                                                                                              \
      FM_MUL(); bias_mfma();                                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      WAIT_;                                                                                                           \
      FM_PHASE_END();                                                                                                  \
      issue(); epilogue(t); preload(t + 1);                                                                            \
      FM_READ(cs); FM_LGKM0(); cs = next_slot(cs);                                                                     \
      FM_PHASE_END();
      FM_G0_STEP(FM_MUL_FIRST(), FM_VMCNT(VM_PEEL));
      FM_G0_STEP(FM_MUL(), FM_VMCNT(VM_PEEL));
      if constexpr (AHEAD == 4) { FM_G0_STEP(FM_MUL(), FM_VMCNT(VM_PEEL)); }
      FM_G0_STEP(FM_MUL(), FM_VM_LASTPEEL());
      for (int h = AHEAD; h < nh - 1; ++h) { FM_G0_STEP(FM_MUL(), FM_VMCNT(VM_STEADY)); }
      FM_G0_LAST(FM_VMCNT(VM_STEADY));
    }
  } else {
    FM_PHASE_END();
    for (int t = 0; t < T; ++t) {
#define FM_G1_STEP(ISSUE_, MUL_, WAIT_)                                                                                \
      FM_READ(cs); ISSUE_; FM_LGKM0(); cs = next_slot(cs);                                                             \
      WAIT_;                                                                                                           \
      FM_PHASE_END();                                                                                                  \
      MUL_;                                                                                                            \
      FM_PHASE_END();
      // (the first half-step's DMA of every tile but the first was issued ahead of the previous tile's epilogue, below)
      FM_G1_STEP({ if (t == 0) issue(); }, FM_MUL_FIRST(), FM_VMCNT(VM_PEEL));
      FM_G1_STEP(issue(), FM_MUL(), FM_VMCNT(VM_PEEL));
      if constexpr (AHEAD == 4) { FM_G1_STEP(issue(), FM_MUL(), FM_VMCNT(VM_PEEL)); }
      FM_G1_STEP(issue(), FM_MUL(), FM_VM_LASTPEEL());
      for (int h = AHEAD; h < nh - 1; ++h) { FM_G1_STEP(issue(), FM_MUL(), FM_VMCNT(VM_STEADY)); }
      FM_G1_STEP(issue(), { FM_MUL(); bias_mfma(); }, FM_VMCNT(VM_STEADY));
      issue();
      epilogue(t);
      preload(t + 1);
    }
  }
  FM_VMCNT(0);
#undef FM_G0_STEP
#undef FM_G0_LAST
#undef FM_G1_STEP
#undef FM_VM_LASTPEEL
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradients from fm operands: dK[i][o] = sum_m H[m][i] dZ[m][o] (jax.grad of nn.Dense, flax kernel layout [in, out]).
// 256 x 256 output tile, 8 waves of 128 (i) x 64 (o), split over row slices into float32 slabs (summed in a fixed order by
// slab_sum2_kernel).  A chunk = 32 rows of both operands = 2 x 16 blocks, DMA'd as they are (1 KiB per instruction) into a
// 4-deep ring with the blocks 1152 B apart; the contraction runs over the rows, so the MFMA fragments come from
// ds_read_b64_tr_b16: a 16-lane group reads the [4 rows x 16 columns] patch of one block -- 8 consecutive units = 128
// contiguous bytes -- and the neighbouring group (the next block) lands 128 B further in the bank window: conflict-free.
// Both operands use the same row -> k-slot map.
constexpr int GBLKP = 1152, GOPER = 16 * GBLKP, GCHUNK = 2 * GOPER, GNBUF = 4;      // 4 x 36 KiB = 144 KiB

__device__ __forceinline__ bf16x8 tr_frag_fm(uint32_t off) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  __attribute__((address_space(3))) char* base = (__attribute__((address_space(3))) char*)fm_smem;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + off));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + off + 128));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// (slice, b: the workgroup's row slice and its tile among the problem's (I / 256) x (O / 256))
template <bool PP>
__device__ __forceinline__ void grad_weight_fm_body(const int slice, const int b, int M, int I, int O, const char* __restrict__ H, int ldh,
                                                    const char* __restrict__ dZ, int lddz, int ksplit,
                                                    float* __restrict__ slabs, int ldc, float* __restrict__ bias_slabs) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 2, wo = wave & 3;
  const int tiles_o = O / 256;
  const int ti = b / tiles_o, to = b - ti * tiles_o;
  const int i0 = ti * 256, o0 = to * 256;
  const int64_t chunks_total = M / 32;
  const int64_t per = (chunks_total + ksplit - 1) / ksplit;
  const int64_t c_begin = (int64_t)slice * per, c_end = c_begin + per < chunks_total ? c_begin + per : chunks_total;
  const int nchunk = c_end > c_begin ? (int)(c_end - c_begin) : 0;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)fm_smem;
  const uint32_t voff = (uint32_t)lane * 16u;
  const size_t h_rb = (size_t)(ldh >> 4) * 1024, z_rb = (size_t)(lddz >> 4) * 1024;
  const char* gh = H + (size_t)(i0 >> 4) * 1024;
  const char* gz = dZ + (size_t)(o0 >> 4) * 1024;
  auto issue = [&](int c, int slot) {
    const size_t mb = (size_t)(c_begin + c);
    const uint32_t buf = lds_base + (uint32_t)slot * GCHUNK;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int id = x * 8 + wave, op = id >> 4, blk = id & 15;
      const char* src = op == 0 ? gh + mb * h_rb + (size_t)blk * 1024 : gz + mb * z_rb + (size_t)blk * 1024;
      glds16(src, voff, buf + op * GOPER + blk * GBLKP);
    }
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  const bool do_bias = bias_slabs != nullptr && ti == 0 && wi == 0;
  float bsum[2] = {0.f, 0.f};
  // transposed-read lane map: 16-lane group g: block (g & 1) of the 32-column MFMA block, k half g >> 1 (8 rows); a16: row
  // a16 >> 2 of the 4 a read covers, column quad q = a16 & 3 -> unit half hi = q & 1, byte half q >> 1.  Unit of (row, hi) =
  // 8 (row >> 2) + 4 (hi ^ (row >> 4)) + (row & 3) with row = 16 kk + 8 (g >> 1) + (a16 >> 2) (+ 4: second read, + 128 B)
  const int g = lane >> 4, a16 = lane & 15;
  uint32_t lane_off[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
    lane_off[kk] = (uint32_t)((g & 1) * GBLKP + kk * 512 + (g >> 1) * 256 + 64 * ((a16 & 1) ^ kk) + 16 * (a16 >> 2) + 8 * ((a16 >> 1) & 1));
  auto read_frags = [&](uint32_t buf, bf16x8 (&fh)[2][4], bf16x8 (&fz)[2][2]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int x = 0; x < 4; ++x) fh[kk][x] = tr_frag_fm(buf + lane_off[kk] + 2 * (4 * wi + x) * GBLKP);
#pragma unroll
      for (int y = 0; y < 2; ++y) fz[kk][y] = tr_frag_fm(buf + GOPER + lane_off[kk] + 2 * (2 * wo + y) * GBLKP);
    }
  };
  auto multiply = [&](const bf16x8 (&fh)[2][4], const bf16x8 (&fz)[2][2]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[kk][x], fz[kk][y], acc[x][y], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[y] += (float)fz[kk][y][e];
      }
    }
  };
  if constexpr (PP) {
  // Two wave groups half a chunk apart (one multiplies while the other reads fragments and issues DMA), as in
  // linear_fm_kernel:   group 0: [MUL(x), wait] | [read(x + 1), DMA(x + 4)]     group 1: [read(x), DMA(x + 3), wait] | [MUL(x)]
  // wait = own DMA of chunk x + 1 has landed (2 younger chunks of 4 instructions may be out); chunks past the end of the
  // slice are re-loads of its last chunk into free slots (uniform counts).
  constexpr int AHEAD = GNBUF - 1;
  int ic = 0;
  auto issue_next = [&]() { issue(ic < nchunk ? ic : nchunk - 1, ic % GNBUF); ++ic; };
  if (nchunk > 0) {
#pragma unroll
    for (int c = 0; c < AHEAD; ++c) issue_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FM_PHASE_END();
    bf16x8 fh[2][4], fz[2][2];
    if (wi == 0) {
      read_frags(0u, fh, fz);
      issue_next();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      FM_PHASE_END();
      for (int c = 0; c < nchunk; ++c) {
        multiply(fh, fz);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * 4) : "memory");
        FM_PHASE_END();
        read_frags((uint32_t)((c + 1) % GNBUF) * GCHUNK, fh, fz);
        issue_next();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        FM_PHASE_END();
      }
    } else {
      FM_PHASE_END();
      for (int c = 0; c < nchunk; ++c) {
        read_frags((uint32_t)(c % GNBUF) * GCHUNK, fh, fz);
        issue_next();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * 4) : "memory");
        FM_PHASE_END();
        multiply(fh, fz);
        FM_PHASE_END();
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  } else {                                          // lock-step: short slices (the ping-pong prologue is synchronous)
#pragma unroll
  for (int c = 0; c < GNBUF - 1; ++c) if (c < nchunk) issue(c, c % GNBUF);
  for (int c = 0; c < nchunk; ++c) {
    const int younger = nchunk - 1 - c < GNBUF - 2 ? nchunk - 1 - c : GNBUF - 2;
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (c + GNBUF - 1 < nchunk) issue(c + GNBUF - 1, (c + GNBUF - 1) % GNBUF);
    bf16x8 fh[2][4], fz[2][2];
    read_frags((uint32_t)(c % GNBUF) * GCHUNK, fh, fz);
    multiply(fh, fz);
  }
  }
  float* slab = slabs + (size_t)slice * I * ldc;
  const int hi = lane >> 5, j = lane & 31;
#pragma unroll
  for (int y = 0; y < 2; ++y) {
    const int o = o0 + wo * 64 + y * 32 + j;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + wi * 128 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        slab[(size_t)i * ldc + o] = acc[x][y][r];
      }
  }
  if (do_bias) {
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const float tot = bsum[y] + __shfl_xor(bsum[y], 32, 64);
      const int o = o0 + wo * 64 + y * 32 + j;
      if (hi == 0) bias_slabs[(size_t)slice * O + o] = tot;
    }
  }
}
template <bool PP>
__global__ __launch_bounds__(512) void grad_weight_fm_kernel(int M, int I, int O, const char* __restrict__ H, int ldh,
                                                             const char* __restrict__ dZ, int lddz, int ksplit,
                                                             float* __restrict__ slabs, int ldc, float* __restrict__ bias_slabs) {
  const int tiles = (I / 256) * (O / 256);
  int slice, b;                                         // all tiles of a row slice on one XCD (they share its rows through that L2)
  if ((ksplit & 7) == 0) {
    const int xcd = blockIdx.x & 7, id = blockIdx.x >> 3, per_xcd = ksplit >> 3;
    slice = xcd * per_xcd + id / tiles;
    b = id % tiles;
  } else {
    slice = blockIdx.x / tiles;
    b = blockIdx.x - slice * tiles;
  }
  grad_weight_fm_body<PP>(slice, b, M, I, O, H, ldh, dZ, lddz, ksplit, slabs, ldc, bias_slabs);
}
// Several weight-gradient problems over the same M rows in ONE launch (the four layers of the PropMLP: one tile each, two for the
// first): with the workgroups of all of them on the chip together a problem gets along with ksplit ~ 256 / (total tiles) row slices
// instead of 256 -- a quarter of the split-K slab traffic, and slices long enough for the ping-pong loop.
struct GwMulti {
  static constexpr int MAXP = 8;
  int n, M, ksplit;
  int I[MAXP], O[MAXP], ldh[MAXP], lddz[MAXP], first_tile[MAXP + 1];       // first_tile[n] = tiles of all problems
  const char* H[MAXP]; const char* dZ[MAXP];
  float* slabs[MAXP]; float* bias_slabs[MAXP];
};
// Workgroup order: XCD-major (the hardware deals workgroup g to XCD g % 8), inside it (slice, problem, tile) with the tile running
// fastest -- the tiles of a problem's row slice sit on one XCD and share its rows through that L2.
template <bool PP>
__global__ __launch_bounds__(512) void grad_weight_fm_multi_kernel(const GwMulti a) {
  const int T = a.first_tile[a.n];
  const int v = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int slice = v / T, r = v - slice * T;
  int p = 0;
  while (p + 1 < a.n && r >= a.first_tile[p + 1]) ++p;
  grad_weight_fm_body<PP>(slice, r - a.first_tile[p], a.M, a.I[p], a.O[p], a.H[p], a.ldh[p], a.dZ[p], a.lddz[p], a.ksplit,
                          a.slabs[p], a.O[p], a.bias_slabs[p]);
}

// ---------------------------------------------------------------------------------------------------------------------
// One output column from an fm operand (the density head, models.py:497 raw_density = Dense(1)(x)): a wave streams the
// K / 16 blocks of a 32-row block (1 KiB per load, lane = unit); the two units of a row sit 4 lanes apart.
__device__ __forceinline__ void unit_to_row_hi(int u, int& row, int& hi) {
  row = ((u >> 3) << 2) | (u & 3);
  hi = ((u >> 2) & 1) ^ (row >> 4);
}

template <int ACT>
__global__ __launch_bounds__(256) void rowdot_fm_kernel(int M, int K, const char* __restrict__ A, int lda, const uint16_t* __restrict__ w,
                                                        const float* __restrict__ bias, float act_param, float* __restrict__ out, int ldo) {
  const int lane = threadIdx.x & 63, rb = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rb * 32 >= M) return;
  int row, hi;
  unit_to_row_hi(lane, row, hi);
  const char* a = A + (size_t)rb * (size_t)(lda >> 4) * 1024 + lane * 16;
  const uint16_t* wl = w + 4 * hi;
  float acc = 0.f;
  const int nb = K >> 4;
  auto dot8 = [&](const uint4 x, const uint2 w0, const uint2 w1) {
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ws[4] = {w0.x, w0.y, w1.x, w1.y};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc += __builtin_bit_cast(float, xs[q] << 16) * __builtin_bit_cast(float, ws[q] << 16);
      acc += __builtin_bit_cast(float, xs[q] & 0xFFFF0000u) * __builtin_bit_cast(float, ws[q] & 0xFFFF0000u);
    }
  };
  int cb = 0;
  for (; cb + 8 <= nb; cb += 8) {
    uint4 x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = *(const uint4*)(a + (size_t)(cb + u) * 1024);
#pragma unroll
    for (int u = 0; u < 8; ++u) dot8(x[u], *(const uint2*)(wl + (cb + u) * 16), *(const uint2*)(wl + (cb + u) * 16 + 8));
  }
  for (; cb < nb; ++cb) dot8(*(const uint4*)(a + (size_t)cb * 1024), *(const uint2*)(wl + cb * 16), *(const uint2*)(wl + cb * 16 + 8));
  acc += __shfl_xor(acc, 4, 64);
  if (hi == 0) {
    float v = acc + (bias ? bias[0] : 0.f);
    if (ACT == 1) v = fmaxf(v, 0.f);
    if (ACT == 2) { const float x = v + act_param; v = fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
    out[(size_t)(rb * 32 + row) * ldo] = v;
  }
}

// d kernel[i] = sum_m H[m][i] z[m] for a one-column dZ (column zcol of an fm tensor): wave = (row slice, 16-column block of H),
// lane = unit, float32 accumulation over the slice's row blocks, then a butterfly over the 32 rows.  Slabs as the MFMA
// kernels write them (slabs[slice][i * ldc], bias slabs [slice] = sum_m z[m]).
__global__ __launch_bounds__(256) void grad_weight_col_fm_kernel(int M, int I, const char* __restrict__ H, int ldh, const char* __restrict__ dZ,
                                                                 int lddz, int zcol, int ksplit, float* __restrict__ slabs, int ldc,
                                                                 float* __restrict__ bias_slabs) {
  const int lane = threadIdx.x & 63, nb = I >> 4;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int slice = wid / nb, cb = wid - slice * nb;
  if (slice >= ksplit) return;
  int row, hi;
  unit_to_row_hi(lane, row, hi);
  const int64_t rbs = M / 32, per = (rbs + ksplit - 1) / ksplit;
  const int64_t b0 = (int64_t)slice * per, b1 = b0 + per < rbs ? b0 + per : rbs;
  const size_t h_rb = (size_t)(ldh >> 4) * 1024, z_rb = lddz == 1 ? 64 : (size_t)(lddz >> 4) * 1024;
  const int zf = zcol & 15;
  const char* hp = H + (size_t)cb * 1024 + lane * 16;
  // lddz == 1: dZ is a plain bf16 vector [M]
  const char* zp = lddz == 1 ? dZ + 2 * row : dZ + (size_t)(zcol >> 4) * 1024 + unit_of(row, (zf >> 2) & 1) * 16 + 2 * (4 * (zf >> 3) + (zf & 3));
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, zsum = 0.f;
  auto fma8 = [&](const uint4 x, const uint16_t zr) {
    const float z = __builtin_bit_cast(float, (uint32_t)zr << 16);
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[2 * q] += __builtin_bit_cast(float, xs[q] << 16) * z;
      acc[2 * q + 1] += __builtin_bit_cast(float, xs[q] & 0xFFFF0000u) * z;
    }
    zsum += z;
  };
  int64_t b = b0;
  for (; b + 8 <= b1; b += 8) {
    uint4 x[8];
    uint16_t z[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { x[u] = *(const uint4*)(hp + (size_t)(b + u) * h_rb); z[u] = *(const uint16_t*)(zp + (size_t)(b + u) * z_rb); }
#pragma unroll
    for (int u = 0; u < 8; ++u) fma8(x[u], z[u]);
  }
  for (; b < b1; ++b) fma8(*(const uint4*)(hp + (size_t)b * h_rb), *(const uint16_t*)(zp + (size_t)b * z_rb));
  // sum over the 32 rows of equal hi: row bits 0, 1 = unit bits 0, 1; row bits 2, 3 = unit bits 3, 4; row bit 4 = unit bits 5 and 2
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = acc[e];
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 36, 64);
    acc[e] = v;
  }
  zsum += __shfl_xor(zsum, 1, 64); zsum += __shfl_xor(zsum, 2, 64); zsum += __shfl_xor(zsum, 8, 64); zsum += __shfl_xor(zsum, 16, 64);
  zsum += __shfl_xor(zsum, 36, 64);
  if (row == 0) {
    float* slab = slabs + (size_t)slice * I * ldc;
#pragma unroll
    for (int e = 0; e < 8; ++e) slab[(size_t)(cb * 16 + 4 * hi + 8 * (e >> 2) + (e & 3)) * ldc] = acc[e];
    if (bias_slabs && cb == 0 && hi == 0) bias_slabs[slice] = zsum;
  }
}

// dZ of the last trunk layer when the only head is the density column (PropMLP): out[m][n] = bf16(z[m] w[n]) where the mask
// bit of (m, n) is set -- what linear_fm<2> computes from a one-column operand, without the GEMM.  A wave owns the 128 x 64
// sub-tile whose mask words it reads (same lane <-> element map as linear_fm_kernel's epilogue).
__global__ __launch_bounds__(512) void outer_masked_fm_kernel(int M, int N, const uint16_t* __restrict__ z, const uint16_t* __restrict__ w,
                                                              const u32x4* __restrict__ mask, char* __restrict__ out, int ldc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;
  const int tiles_n = N >> 8;
  const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile - tm * tiles_n;
  int row, hi;
  unit_to_row_hi(lane, row, hi);                                // (the stores below use lane = unit; the mask is indexed by the GEMM's lane)
  const int glane = row + 32 * hi;
  const u32x4 mw = mask[((size_t)tile * 8 + wave) * 64 + glane];
  float wv[2][2][8];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const uint16_t* wp = w + tn * 256 + wn * 64 + j * 32 + half * 16 + 4 * hi;
#pragma unroll
      for (int t = 0; t < 8; ++t) wv[j][half][t] = __builtin_bit_cast(float, (uint32_t)wp[(t & 3) + 8 * (t >> 2)] << 16);
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rb = tm * 8 + wm * 4 + i;
    const float zv = __builtin_bit_cast(float, (uint32_t)z[rb * 32 + row] << 16);
    char* ob = out + ((size_t)rb * (ldc >> 4) + tn * 16 + wn * 4) * 1024 + lane * 16;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        uint32_t pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x2 f = {zv * wv[j][half][2 * q], zv * wv[j][half][2 * q + 1]};
          uint32_t v = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
          v &= ((mw[i] >> (8 * j + 4 * half + q)) & 0x00010001u) * 0xFFFFu;
          pk[q] = v;
        }
        *(uint4*)(ob + (j * 2 + half) * 1024) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// row-major <-> fragment-major (bf16): one thread per 16-byte unit
__global__ __launch_bounds__(256) void to_fm_kernel(int rows, int cols, const uint16_t* __restrict__ src, int ld_src, char* __restrict__ dst,
                                                    int ld_dst, int col0_dst) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;   // unit: (row block, col block, row, hi)
  const int cb_n = cols >> 4;
  const int64_t blk = id >> 6;
  if (blk >= (int64_t)(rows >> 5) * cb_n) return;
  const int rb = (int)(blk / cb_n), cb = (int)(blk - (int64_t)rb * cb_n), row = (int)(id & 31), hi = (int)((id >> 5) & 1);
  const uint16_t* s = src + (size_t)(rb * 32 + row) * ld_src + cb * 16 + 4 * hi;
  const uint2 a = *(const uint2*)s, b = *(const uint2*)(s + 8);
  char* d = dst + ((size_t)rb * (ld_dst >> 4) + (col0_dst >> 4) + cb) * 1024 + unit_of(row, hi) * 16;
  *(uint4*)d = make_uint4(a.x, a.y, b.x, b.y);
}
__global__ __launch_bounds__(256) void from_fm_kernel(int rows, int cols, const char* __restrict__ src, int ld_src, int col0_src,
                                                      uint16_t* __restrict__ dst, int ld_dst) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cb_n = cols >> 4;
  const int64_t blk = id >> 6;
  if (blk >= (int64_t)(rows >> 5) * cb_n) return;
  const int rb = (int)(blk / cb_n), cb = (int)(blk - (int64_t)rb * cb_n), row = (int)(id & 31), hi = (int)((id >> 5) & 1);
  const uint4 v = *(const uint4*)(src + ((size_t)rb * (ld_src >> 4) + (col0_src >> 4) + cb) * 1024 + unit_of(row, hi) * 16);
  uint16_t* d = dst + (size_t)(rb * 32 + row) * ld_dst + cb * 16 + 4 * hi;
  *(uint2*)d = make_uint2(v.x, v.y);
  *(uint2*)(d + 8) = make_uint2(v.z, v.w);
}

}  // namespace mip360fm

static int fm_n_cu() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    n_cu = n_cu >= 8 ? n_cu / 8 * 8 : 8;
  }
  return n_cu;
}

// M, N multiples of 256; K a multiple of 32, >= 160 (more half-steps than the ring is deep); lda / ldw / ldc multiples of 16
int mip360_launch_linear_fm(hipStream_t st, int M, int N, int K, const void* A, int lda, const void* W, int ldw, const float* bias,
                            int act, void* C, int ldc, void* mask) {
  using namespace mip360fm;
  if (M <= 0 || N <= 0 || M % 256 || N % 256 || K % 32 || K < 32 * (Cfg::NSLOT) || lda % 16 || ldw % 16 || ldc % 16) return 1;
  if ((act == 2 && !mask) || (act != 2 && !bias) || (act == 1 && !mask) || act < 0 || act > 2) return 1;
  static std::atomic<uint64_t> attr_done{0};
  if (first_launch_on_this_device(attr_done)) {
    (void)hipFuncSetAttribute((const void*)linear_fm_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    (void)hipFuncSetAttribute((const void*)linear_fm_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    (void)hipFuncSetAttribute((const void*)linear_fm_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
  }
  const int tiles = (M / 256) * (N / 256), n_cu = fm_n_cu();
  const dim3 grid(tiles < n_cu ? tiles : n_cu), block(512);
#define FM_LAUNCH(ACT_)                                                                                                 \
  hipLaunchKernelGGL((linear_fm_kernel<ACT_>), grid, block, Cfg::LDS, st, M, N, K, (const char*)A, lda, (const char*)W, ldw, bias,  \
                     (char*)C, ldc, (u32x4*)mask)
  if (act == 0) FM_LAUNCH(0);
  else if (act == 1) FM_LAUNCH(1);
  else FM_LAUNCH(2);
#undef FM_LAUNCH
  return 0;
}

int mip360_launch_to_fm(hipStream_t st, int rows, int cols, const void* src, int ld_src, void* dst, int ld_dst, int col0_dst) {
  if (rows % 32 || cols % 16 || ld_dst % 16 || col0_dst % 16 || ld_src % 4) return 1;
  const int64_t units = (int64_t)(rows / 32) * (cols / 16) * 64;
  hipLaunchKernelGGL(mip360fm::to_fm_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, st, rows, cols, (const uint16_t*)src, ld_src,
                     (char*)dst, ld_dst, col0_dst);
  return 0;
}
int mip360_launch_from_fm(hipStream_t st, int rows, int cols, const void* src, int ld_src, int col0_src, void* dst, int ld_dst) {
  if (rows % 32 || cols % 16 || ld_src % 16 || col0_src % 16 || ld_dst % 4) return 1;
  const int64_t units = (int64_t)(rows / 32) * (cols / 16) * 64;
  hipLaunchKernelGGL(mip360fm::from_fm_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, st, rows, cols, (const char*)src, ld_src,
                     col0_src, (uint16_t*)dst, ld_dst);
  return 0;
}

// slabs: ksplit x [I, ldc] floats (+ ksplit x O bias slabs when bias_slabs != null); I, O multiples of 256, M of 32
int mip360_launch_grad_weight_fm(hipStream_t st, int M, int I, int O, const void* H, int ldh, const void* dZ, int lddz, int ksplit,
                                 float* slabs, int ldc, float* bias_slabs) {
  using namespace mip360fm;
  if (M <= 0 || M % 32 || I <= 0 || O <= 0 || I % 256 || O % 256 || ldh % 16 || lddz % 16 || ksplit < 1) return 1;
  static std::atomic<uint64_t> attr_done{0};
  if (first_launch_on_this_device(attr_done)) {
    (void)hipFuncSetAttribute((const void*)grad_weight_fm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GNBUF * GCHUNK);
    (void)hipFuncSetAttribute((const void*)grad_weight_fm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GNBUF * GCHUNK);
  }
  const dim3 grid((I / 256) * (O / 256) * ksplit);
  if (M / 32 / ksplit >= 96)                       // measured: ping-pong wins from ~100 chunks per slice (131072 rows / 16: -8 %), loses below (32 chunks: +10 %)
    hipLaunchKernelGGL(grad_weight_fm_kernel<true>, grid, dim3(512), GNBUF * GCHUNK, st, M, I, O, (const char*)H, ldh, (const char*)dZ, lddz, ksplit,
                       slabs, ldc, bias_slabs);
  else
    hipLaunchKernelGGL(grad_weight_fm_kernel<false>, grid, dim3(512), GNBUF * GCHUNK, st, M, I, O, (const char*)H, ldh, (const char*)dZ, lddz, ksplit,
                       slabs, ldc, bias_slabs);
  return 0;
}

// n problems (I[p], O[p] multiples of 256) over the same M rows; slabs[p]: ksplit x I[p] x O[p] floats followed by ksplit x O[p]
// bias slabs (as mip360_grad_weight_fm lays them out)
int mip360_launch_grad_weight_fm_multi(hipStream_t st, int n, int M, int ksplit, const int* I, const int* O, const void* const* H, const int* ldh,
                                       const void* const* dZ, const int* lddz, float* const* slabs) {
  using namespace mip360fm;
  if (n < 1 || n > GwMulti::MAXP || M <= 0 || M % 32 || ksplit < 1 || ksplit > 256) return 1;
  GwMulti a{};
  a.n = n; a.M = M; a.ksplit = ksplit;
  int tiles = 0;
  for (int p = 0; p < n; ++p) {
    if (I[p] <= 0 || O[p] <= 0 || I[p] % 256 || O[p] % 256 || ldh[p] % 16 || lddz[p] % 16 || ldh[p] < I[p] || lddz[p] < O[p] || !H[p] || !dZ[p] || !slabs[p]) return 1;
    a.I[p] = I[p]; a.O[p] = O[p]; a.ldh[p] = ldh[p]; a.lddz[p] = lddz[p]; a.H[p] = (const char*)H[p]; a.dZ[p] = (const char*)dZ[p];
    a.slabs[p] = slabs[p]; a.bias_slabs[p] = slabs[p] + (size_t)ksplit * I[p] * O[p];
    a.first_tile[p] = tiles;
    tiles += (I[p] / 256) * (O[p] / 256);
  }
  a.first_tile[n] = tiles;
  const int blocks = tiles * ksplit;
  static std::atomic<uint64_t> attr_done{0};
  if (first_launch_on_this_device(attr_done)) {
    (void)hipFuncSetAttribute((const void*)grad_weight_fm_multi_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GNBUF * GCHUNK);
    (void)hipFuncSetAttribute((const void*)grad_weight_fm_multi_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GNBUF * GCHUNK);
  }
  if (M / 32 / ksplit >= 96) hipLaunchKernelGGL(grad_weight_fm_multi_kernel<true>, dim3(blocks), dim3(512), GNBUF * GCHUNK, st, a);
  else hipLaunchKernelGGL(grad_weight_fm_multi_kernel<false>, dim3(blocks), dim3(512), GNBUF * GCHUNK, st, a);
  return 0;
}

int mip360_launch_rowdot_fm(hipStream_t st, int M, int K, const void* A, int lda, const void* w, const float* bias, int act, float act_param,
                            float* out, int ldo) {
  using namespace mip360fm;
  if (M <= 0 || M % 32 || K <= 0 || K % 16 || lda % 16 || act < 0 || act > 2) return 1;
  const dim3 grid((M / 32 + 3) / 4), block(256);
  if (act == 2) hipLaunchKernelGGL(rowdot_fm_kernel<2>, grid, block, 0, st, M, K, (const char*)A, lda, (const uint16_t*)w, bias, act_param, out, ldo);
  else if (act == 1) hipLaunchKernelGGL(rowdot_fm_kernel<1>, grid, block, 0, st, M, K, (const char*)A, lda, (const uint16_t*)w, bias, act_param, out, ldo);
  else hipLaunchKernelGGL(rowdot_fm_kernel<0>, grid, block, 0, st, M, K, (const char*)A, lda, (const uint16_t*)w, bias, act_param, out, ldo);
  return 0;
}

int mip360_launch_grad_weight_col_fm(hipStream_t st, int M, int I, const void* H, int ldh, const void* dZ, int lddz, int zcol, int ksplit,
                                     float* slabs, int ldc, float* bias_slabs) {
  using namespace mip360fm;
  if (M <= 0 || M % 32 || I <= 0 || I % 16 || ldh % 16 || (lddz != 1 && lddz % 16) || zcol < 0 || zcol >= lddz || ksplit < 1) return 1;
  const int waves = (I / 16) * ksplit;
  hipLaunchKernelGGL(grad_weight_col_fm_kernel, dim3((waves + 3) / 4), dim3(256), 0, st, M, I, (const char*)H, ldh, (const char*)dZ, lddz, zcol,
                     ksplit, slabs, ldc, bias_slabs);
  return 0;
}

int mip360_launch_outer_masked_fm(hipStream_t st, int M, int N, const void* z, const void* w, const void* mask, void* out, int ldc) {
  using namespace mip360fm;
  if (M <= 0 || N <= 0 || M % 256 || N % 256 || ldc % 16) return 1;
  hipLaunchKernelGGL(outer_masked_fm_kernel, dim3((M / 256) * (N / 256)), dim3(512), 0, st, M, N, (const uint16_t*)z, (const uint16_t*)w,
                     (const u32x4*)mask, (char*)out, ldc);
  return 0;
}
