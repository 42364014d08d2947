// Dense layer on the gfx950 matrix cores for the MipNeRF-360 MLPs (PropMLP 4 x 256, NerfMLP 8 x 1024):
//   C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]),  bf16 operands, float32 accumulation (v_mfma_f32_32x32x16_bf16).
// Replaces flax nn.Dense + nn.relu of MLP.__call__ (nerf-methods/mipnerf360/internal/models.py:436-606).
//
// The 1024-wide NerfMLP does not fit the NeRF++ "whole network in registers" design (32 samples x 1024 features of
// float32 accumulators = 512 VGPRs per lane), so its layers run as tiled GEMMs whose [rows, 1024] bf16 activations
// round-trip through L2 / Infinity Cache (268 MB per layer at 4096 rays x 32 samples).
//
// Tiling: K step 32, double-buffered LDS.  Wide layers: 512 threads = 2 x 4 waves on a 256 x 256 tile, a wave owns
// 128 x 64 = 4 x 2 MFMA blocks; narrow layers (N <= 128): 256 threads on 128 x 128.  Both operands are K-contiguous,
// so an MFMA fragment is one 16-byte LDS read per lane (row l & 31, k offset 8 * (l >> 5)); LDS rows are padded to
// 80 bytes.  The next K tile is fetched into registers while the current one is multiplied (one __syncthreads per K
// step).
#include "probe_env.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// hipFuncSetAttribute is per device: remember which devices of this process have had it applied (one bit per device id)
#include <atomic>
static inline bool first_launch_on_this_device(std::atomic<uint64_t>& done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  return (done.fetch_or(bit) & bit) == 0;
}

// Component-removal switches of the ping-pong GEMM (timing experiments, garbage results) exist only in diagnostic builds:
// -DNERFPP_PROBES takes them from mip360_gemm_probes.h (MIP360_EXP_NODMA / NOLDS / NOMFMA); the shipped library has none.
#ifdef NERFPP_PROBES
#include "mip360_gemm_probes.h"
#else
namespace mip360 { namespace probe { constexpr bool NODMA = false, NOLDS = false, NOMFMA = false; } }
#endif

namespace mip360 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32, LDS_ROW = 40;      // elements; 40 * 2 B = 80 B row stride

// Tile configuration: NWM x NWN waves, each owning FM x FN MFMA blocks of 32 x 32.
//   small (2, 2, 2, 2): 256 threads, 128 x 128 tile -- narrow layers (N <= 128: heads, view branch)
//   big   (2, 4, 4, 2): 512 threads, 256 x 256 tile, 128 x 64 per wave: 6 LDS fragment reads feed 8 MFMAs per k16 step
template <int ACT, int NWM, int NWN, int FM, int FN>
__global__ __launch_bounds__(NWM * NWN * 64) void linear_bf16_kernel(int M, int N, int K, const __bf16* __restrict__ A, int lda,
                                                                     const __bf16* __restrict__ W, int ldw,
                                                                     const float* __restrict__ bias, __bf16* __restrict__ C16,
                                                                     int ldc, float* __restrict__ C32, int ldc32, float act_param,
                                                                     const __bf16* __restrict__ aux, int ldaux) {
  constexpr int NT = NWM * NWN * 64, BM = NWM * FM * 32, BN = NWN * FN * 32;
  constexpr int QA = BM * 4 / NT, QB = BN * 4 / NT;               // 16-byte chunks per thread per operand tile
  static_assert(BM * 4 % NT == 0 && BN * 4 % NT == 0, "tile / thread mismatch");
  __shared__ __attribute__((aligned(16))) __bf16 sA[2][BM * LDS_ROW];
  __shared__ __attribute__((aligned(16))) __bf16 sB[2][BN * LDS_ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave - wm * NWN;
  // XCD-aware tile order.  Workgroup b runs on XCD b % 8 and every XCD has its own L2, so the tiles_n workgroups that
  // read the SAME A tile are given ids b, b + 8, b + 16, ... : they execute back to back on one XCD and share the A
  // tile (BM x K) through that XCD's L2 instead of fetching it eight times from HBM; W (<= 3 MB) lives in every L2.
  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  int tile_m, tile_n;
  {
    const int b = blockIdx.x, xcd = b & 7, id = b >> 3;
    const int full = (tiles_m / 8) * 8;                       // M tiles that can be dealt 8 at a time
    const int group = id / tiles_n;                           // which group of 8 M tiles
    if (group * 8 + 8 <= full) { tile_m = group * 8 + xcd; tile_n = id - group * tiles_n; }
    else {                                                    // remainder (tiles_m % 8 M tiles): plain order
      const int r = b - full * tiles_n;
      tile_m = full + r / tiles_n; tile_n = r - (r / tiles_n) * tiles_n;
    }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  uint4 ra[QA], rb[QB];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int c = tid + q * NT, row = c >> 2, kc = (c & 3) * 8;
      const int m = m0 + row;
      ra[q] = m < M ? *(const uint4*)(A + (size_t)m * lda + k0 + kc) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int c = tid + q * NT, row = c >> 2, kc = (c & 3) * 8;
      const int n = n0 + row;
      rb[q] = n < N ? *(const uint4*)(W + (size_t)n * ldw + k0 + kc) : make_uint4(0, 0, 0, 0);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int c = tid + q * NT, row = c >> 2, kc = (c & 3) * 8;
      *(uint4*)(&sA[buf][row * LDS_ROW + kc]) = ra[q];
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int c = tid + q * NT, row = c >> 2, kc = (c & 3) * 8;
      *(uint4*)(&sB[buf][row * LDS_ROW + kc]) = rb[q];
    }
  };
  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / BK;
  fetch(0);
  stash(0);
  __syncthreads();
  const int frow = lane & 31, fk = (lane >> 5) * 8;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) fetch((kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[FM], fb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) fa[i] = *(const bf16x8*)(&sA[buf][(wm * FM * 32 + i * 32 + frow) * LDS_ROW + ks * 16 + fk]);
#pragma unroll
      for (int j = 0; j < FN; ++j) fb[j] = *(const bf16x8*)(&sB[buf][(wn * FN * 32 + j * 32 + frow) * LDS_ROW + ks * 16 + fk]);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) stash(buf ^ 1);
    __syncthreads();
  }
  // epilogue: lane (j = lane & 31, hi = lane >> 5), register r -> row (r & 3) + 8 (r >> 2) + 4 hi, column j
  const int hi = lane >> 5;
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = n0 + wn * FN * 32 + j * 32 + frow;
    if (n >= N) continue;
    const float b = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * FM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (m >= M) continue;
        float v = acc[i][j][r] + b;
        if (ACT == 1) v = fmaxf(v, 0.f);
        if (ACT == 2) { const float x = v + act_param; v = fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }   // softplus(raw + density_bias)
        if (ACT == 4) v = (float)aux[(size_t)m * ldaux + n] > 0.f ? v : 0.f;                             // ReLU mask of the layer's saved output
        if (ACT == 3) v = (1.f / (1.f + expf(-v))) * (1.f + 2.f * act_param) - act_param;                // padded sigmoid
        if (C16) C16[(size_t)m * ldc + n] = (__bf16)v;
        if (C32) C32[(size_t)m * ldc32 + n] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Wide layers (N >= 192): 256 x 256 tile, 8 waves (2 x 4, 128 x 64 per wave), K step 32, operands streamed
// global -> LDS with global_load_lds_dwordx4 into a 4-deep ring (no VGPR staging, three K steps in flight, counted
// vmcnt + one raw s_barrier per step).  A DMA instruction fills 1 KiB of LDS linearly (lane l -> byte 16 l) but every
// lane supplies its own global address, so the 16-byte chunks of a 64-byte tile row are stored XOR-swizzled
// (position p of row r holds chunk p ^ ((r >> 2) & 3)): the 16 lanes one ds_read_b128 services together then hit 16
// different 16-byte slots of the 256-byte bank line (conflict-free) without padding.
// Rows beyond M / N are clamped to the last valid row on the load side (their results are never stored).
// ------------------------------------------------------------------------------------------------------------
constexpr int RBK = 32;
extern __shared__ __attribute__((aligned(16))) char ring_smem[];

__device__ __forceinline__ void glds16_asm(const void* g, uint32_t lds_abs) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_abs);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}

// the same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset (one VGPR instead of a pointer pair)
__device__ __forceinline__ void glds16_saddr(const void* sbase, uint32_t voff, uint32_t lds_abs) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_abs);
  const uint64_t b = (uint64_t)(uintptr_t)sbase;
  const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}

// WM x WN waves, each owning FM x FN MFMA blocks of 32 x 32; NBUF ring stages of (BM + BN) rows x 64 bytes:
//   (2, 4, 4, 2, 4) = 8 waves of 128 x 64 on a 256 x 256 tile, 128 KiB ring, one workgroup per CU
//   (2, 2, 4, 2, 3) = 4 waves of 128 x 64 on a 256 x 128 tile,  72 KiB ring, TWO workgroups per CU: while one of them
//                     drains its tile (epilogue) or primes its ring (prologue) the other one keeps the matrix pipe busy
//   (2, 2, 4, 4, 4) = 4 waves of 128 x 128, one per SIMD with the whole register file (measured slower, kept for probes)
template <int WM, int WN, int FM, int FN, int NBUF>
struct RingCfg {
  static constexpr int NW = WM * WN, NT = NW * 64, BM = WM * FM * 32, BN = WN * FN * 32;
  static constexpr int STAGE = (BM + BN) * RBK * 2;             // bytes per stage
  static constexpr int QW = (BM + BN) / 16 / NW;                // DMA instructions (16 rows x 64 B = 1 KiB) per wave per stage
  static constexpr int PIECES = BN / 8;                         // 16-byte pieces per output row
  static constexpr int IT = BM * PIECES / NT;                   // output rows per thread in the bf16 epilogue
  static constexpr int LDS = NBUF * STAGE > BM * BN * 2 ? NBUF * STAGE : BM * BN * 2;
  static_assert((BM + BN) / 16 % NW == 0, "DMA instructions divide among the waves");
  static_assert(IT % 16 == 0 && NT % PIECES == 0, "mask words: whole 16-byte groups per thread");
  static_assert(NBUF >= 3 && NBUF <= 5, "ring depth");
};

// Epilogue shared by the ring kernels: bias + activation on the accumulators (lane l & 31 = tile row, register r = column
// (r & 3) + 8 (r >> 2) + 4 (l >> 5) of a 32 x 32 block), float32 outputs stored directly, bf16 outputs staged through the
// (idle) ring memory and written as 16 bytes per lane; ReLU masks applied / emitted on the way out.
template <int ACT, int WM, int WN, int FM, int FN, class Cfg>
__device__ __forceinline__ void ring_epilogue(f32x16 (&acc)[FM][FN], uint32_t (&mw)[Cfg::IT / 4], uint8_t* const mask_at, const int tid,
                                              const int wm, const int wn, const int m0, const int n0, const int M, const int N,
                                              const float* __restrict__ bias, __bf16* __restrict__ C16, const int ldc,
                                              float* __restrict__ C32, const int ldc32, const float act_param,
                                              const __bf16* __restrict__ aux, const int ldaux) {
  constexpr int BN = Cfg::BN, IT = Cfg::IT, PIECES = Cfg::PIECES;
  const int lane = tid & 63, frow = lane & 31;
  const int piece = tid % PIECES, rgroup = tid / PIECES;
  const int hi = lane >> 5;
  auto activate = [&](float v) {
    if (ACT == 1 || ACT == 5) v = fmaxf(v, 0.f);
    if (ACT == 2) { const float x = v + act_param; v = fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
    if (ACT == 3) v = (1.f / (1.f + expf(-v))) * (1.f + 2.f * act_param) - act_param;
    return v;
  };
  if (C32 != nullptr) {                                       // float32 outputs (rare for wide layers): direct stores
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m0 + wm * FM * 32 + i * 32 + frow;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + wn * FN * 32 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (n >= N) continue;
          float v = activate(acc[i][j][r] + (bias ? bias[n] : 0.f));
          if (ACT == 4) v = (float)aux[(size_t)m * ldaux + n] > 0.f ? v : 0.f;
          C32[(size_t)m * ldc32 + n] = v;
          if (C16) C16[(size_t)m * ldc + n] = (__bf16)v;
        }
    }
    return;
  }
  // bf16 output.  The ring is idle now, so the tile is staged through it ([BM][BN] bf16, rows rotated by 16 bytes per row
  // against bank conflicts; 8-byte writes: a register quad is 4 consecutive columns of one row) and leaves as 16 bytes per
  // lane = whole row segments per PIECES lanes; the ReLU mask (ACT 4 / 6) is applied on the way out.
  __builtin_amdgcn_s_barrier();
  {
    __bf16* tile = (__bf16*)ring_smem;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < FN; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = wn * FN * 32 + j * 32 + 8 * q + 4 * hi;        // first of 4 consecutive columns
        const int n = n0 + nl;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) {
          if (n + 4 <= N) b4 = *(const float4*)(bias + n);
          else { if (n < N) b4.x = bias[n]; if (n + 1 < N) b4.y = bias[n + 1]; if (n + 2 < N) b4.z = bias[n + 2]; }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int ml = wm * FM * 32 + i * 32 + frow;
          const f32x2 lo = {activate(acc[i][j][4 * q] + b4.x), activate(acc[i][j][4 * q + 1] + b4.y)};
          const f32x2 hi2 = {activate(acc[i][j][4 * q + 2] + b4.z), activate(acc[i][j][4 * q + 3] + b4.w)};
          uint2 pk;
          pk.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));
          pk.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi2, bf16x2));
          *(uint2*)(tile + ml * BN + ((nl + 8 * ml) & (BN - 1))) = pk;
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    const __bf16* tile = (const __bf16*)ring_smem;
    const bool vec_ok = ((ldc & 7) == 0) && (ACT != 4 || (ldaux & 7) == 0);
    const int n = n0 + piece * 8;
#pragma unroll
    for (int q = 0; q < IT / 4; ++q) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int ml = rgroup * IT + 4 * q + b, m = m0 + ml;
        if (m >= M || n >= N) continue;
        uint4 v = *(const uint4*)(tile + ml * BN + ((piece * 8 + 8 * ml) & (BN - 1)));
        if (ACT == 5) {                                                // bit k: bf16 k is > 0 (sign clear, magnitude non-zero)
          const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
          uint32_t bits = 0u;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            bits |= (((vw[k] & 0x8000u) == 0 && (vw[k] & 0x7FFFu) != 0) ? 1u : 0u) << (2 * k);
            bits |= (((vw[k] & 0x80000000u) == 0 && (vw[k] & 0x7FFF0000u) != 0) ? 1u : 0u) << (2 * k + 1);
          }
          mw[q] |= bits << (8 * b);
        }
        if (ACT == 6) {
          const uint32_t bits = mw[q] >> (8 * b);
          uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 4; ++k)
            vw[k] &= ((0u - ((bits >> (2 * k)) & 1u)) & 0x0000FFFFu) | ((0u - ((bits >> (2 * k + 1)) & 1u)) & 0xFFFF0000u);
          v = make_uint4(vw[0], vw[1], vw[2], vw[3]);
        }
        if (vec_ok && n + 8 <= N) {
          if (ACT == 4) {
            const uint4 a4 = *(const uint4*)(aux + (size_t)m * ldaux + n);
            const uint32_t aw[4] = {a4.x, a4.y, a4.z, a4.w};
            uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              // keep a bf16 where the saved activation is > 0: positive sign and non-zero magnitude
              const uint32_t lo = ((aw[k] & 0x8000u) == 0 && (aw[k] & 0x7FFFu) != 0) ? 0x0000FFFFu : 0u;
              const uint32_t hi16 = ((aw[k] & 0x80000000u) == 0 && (aw[k] & 0x7FFF0000u) != 0) ? 0xFFFF0000u : 0u;
              vw[k] &= (lo | hi16);
            }
            v = make_uint4(vw[0], vw[1], vw[2], vw[3]);
          }
          *(uint4*)(C16 + (size_t)m * ldc + n) = v;
        } else {
          const __bf16* pv = (const __bf16*)&v;
          for (int k = 0; k < 8 && n + k < N; ++k) {
            float x = (float)pv[k];
            if (ACT == 4) x = (float)aux[(size_t)m * ldaux + n + k] > 0.f ? x : 0.f;
            C16[(size_t)m * ldc + n + k] = (__bf16)x;
          }
        }
      }
    }
    if (ACT == 5) {
#pragma unroll
      for (int q = 0; q < IT / 16; ++q) *(uint4*)(mask_at + 16 * q) = make_uint4(mw[4 * q], mw[4 * q + 1], mw[4 * q + 2], mw[4 * q + 3]);
    }
  }
}

template <int ACT, int WM, int WN, int FM, int FN, int NBUF>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 && NBUF == 3) ? 2 : 1)
void linear_bf16_ring_kernel(int M, int N, int K, const __bf16* __restrict__ A, int lda, const __bf16* __restrict__ W, int ldw,
                             const float* __restrict__ bias, __bf16* __restrict__ C16, int ldc, float* __restrict__ C32, int ldc32,
                             float act_param, const __bf16* __restrict__ aux, int ldaux, uint8_t* __restrict__ mask, int ldmask) {
  using Cfg = RingCfg<WM, WN, FM, FN, NBUF>;
  constexpr int NT = Cfg::NT, BM = Cfg::BM, BN = Cfg::BN, STAGE = Cfg::STAGE, QW = Cfg::QW, PIECES = Cfg::PIECES, IT = Cfg::IT;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  int tile_m, tile_n;
  {                                                           // XCD-aware order, see linear_bf16_kernel
    const int b = blockIdx.x, xcd = b & 7, id = b >> 3;
    const int full = (tiles_m / 8) * 8;
    const int group = id / tiles_n;
    if (group * 8 + 8 <= full) { tile_m = group * 8 + xcd; tile_n = id - group * tiles_n; }
    else { const int r = b - full * tiles_n; tile_m = full + r / tiles_n; tile_n = r - (r / tiles_n) * tiles_n; }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring_smem;
  // DMA lane map: instruction q of this wave covers stage rows [16 g, 16 g + 16), g = wave * QW + q; stage rows
  // [0, BM) are the A tile, [BM, BM + BN) the W tile
  const int drow = lane >> 2, dpos = lane & 3;
  const char* gsrc[QW];
  uint32_t ldst[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    const int srow = 16 * (wave * QW + q) + drow;
    const int chunk = dpos ^ ((srow >> 2) & 3);               // (BM is a multiple of 16: the swizzle of a W row is that of its tile row)
    if (srow < BM) {
      int m = m0 + srow; m = m < M ? m : M - 1;
      gsrc[q] = (const char*)(A + (size_t)m * lda + chunk * 8);
    } else {
      int n = n0 + srow - BM; n = n < N ? n : N - 1;
      gsrc[q] = (const char*)(W + (size_t)n * ldw + chunk * 8);
    }
    ldst[q] = (uint32_t)(16 * (wave * QW + q) * 64);          // + slot * STAGE, + lane * 16 by the hardware
  }
  const int nk = K / RBK;
  auto issue = [&](int kt) {
    if (kt >= nk) return;
    if constexpr (probe::NODMA) { if (kt >= NBUF) return; }
    const uint32_t slot = (uint32_t)(kt % NBUF) * STAGE;
#pragma unroll
    for (int q = 0; q < QW; ++q) glds16_asm(gsrc[q] + (size_t)kt * RBK * 2, lds0 + slot + ldst[q]);
  };
  // ReLU bit mask (ACT 5 writes it, ACT 6 applies it), column-byte-major: byte [(n >> 3) * ldmask + m], bit n & 7 =
  // (C[m][n] > 0).  In the bf16 epilogue a thread owns columns [8 piece, 8 piece + 8) of IT consecutive rows, i.e. IT
  // consecutive mask bytes: ACT 6 fetches them here, ahead of the K loop (IT / 4 registers), ACT 5 stores them at the end.
  const int piece = tid % PIECES, rgroup = tid / PIECES;
  uint32_t mw[IT / 4];
  uint8_t* const mask_at = mask + (size_t)((n0 >> 3) + piece) * ldmask + m0 + rgroup * IT;
#pragma unroll
  for (int q = 0; q < IT / 4; ++q) mw[q] = 0u;
  if (ACT == 6) {
#pragma unroll
    for (int q = 0; q < IT / 16; ++q) {
      const uint4 t = *(const uint4*)(mask_at + 16 * q);
      mw[4 * q] = t.x; mw[4 * q + 1] = t.y; mw[4 * q + 2] = t.z; mw[4 * q + 3] = t.w;
    }
  }
  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int s_ = 0; s_ < NBUF - 1; ++s_) issue(s_);
  // fragment read: row R = block row + (lane & 31), k chunk c = 2 ks + (lane >> 5) -> position c ^ ((R >> 2) & 3)
  const int frow = lane & 31, fkh = lane >> 5;
  auto read_frags = [&](const char* st, int ks, bf16x8 (&fa)[FM], bf16x8 (&fb)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int R = wm * FM * 32 + i * 32 + frow;
      fa[i] = *(const bf16x8*)(st + R * 64 + (((2 * ks + fkh) ^ ((R >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int R = wn * FN * 32 + j * 32 + frow;
      fb[j] = *(const bf16x8*)(st + BM * RBK * 2 + R * 64 + (((2 * ks + fkh) ^ ((R >> 2) & 3)) << 4));
    }
  };
  // operands swapped (W fragment as "A"): the accumulator block is C^T, i.e. lane (l & 31) holds ROW m of the tile and
  // register r column (r & 3) + 8 (r >> 2) + 4 (l >> 5) -- four consecutive columns per register quad, which the
  // epilogue packs into one 8-byte LDS write
  auto multiply = [&](const bf16x8 (&fa)[FM], const bf16x8 (&fb)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
  };
  auto wait_stages = [&](int younger) {                       // all but the `younger` most recent stages of this wave have landed
    if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * QW) : "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QW) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  for (int kt = 0; kt < nk; ++kt) {
    wait_stages(nk - 1 - kt < NBUF - 2 ? nk - 1 - kt : NBUF - 2);
    __builtin_amdgcn_s_barrier();
    issue(kt + NBUF - 1);
    const char* st = ring_smem + (kt % NBUF) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[FM], fb[FN];
      read_frags(st, ks, fa, fb);
      multiply(fa, fb);
    }
  }
  ring_epilogue<ACT, WM, WN, FM, FN, Cfg>(acc, mw, mask_at, tid, wm, wn, m0, n0, M, N, bias, C16, ldc, C32, ldc32, act_param, aux, ldaux);
}

// ------------------------------------------------------------------------------------------------------------
// Persistent ping-pong kernel for K % 64 == 0 (every wide layer of both MLPs except the 288-column head GEMM).
//
// What the lock-step ring kernel above is bound by (profiles/r02_g_*): (1) its DMA instructions fetch 16 rows x 64 bytes,
// i.e. HALF cache lines -- the CU's texture path then delivers 80 GB/s instead of 136 GB/s with whole lines
// (tools/probes/dma_pattern_probe.hip); (2) both waves of a SIMD leave the barrier together, issue DMA and LDS reads
// together and queue their MFMAs together, so those times add up instead of overlapping (matrix pipe busy 35 % of the
// cycles); (3) a constant of ~14 us per 256 x 256 tile (35 % of a K = 1024 tile): every CU finishes its tile at the same
// moment, 33 MB of outputs then drain to HBM while nothing computes, and the next workgroup starts with an empty ring.
// Here
//   * a stage is 64 k-elements = one 128-byte line per row (32 KiB per operand); the A ring is 3 deep, the W ring 2
//     deep: 160 KiB; a DMA instruction covers 8 rows x 128 B; the 16-byte chunks of a row are XOR-swizzled with
//     (row >> 1) & 7, which keeps ds_read_b128 conflict-free with 128-byte rows;
//   * the eight waves form two groups of four (one wave of each group per SIMD) that run half a step apart: while one
//     group is in its 16-MFMA burst the other one reads its next 12 fragments from LDS (MI355X_MICROARCH.md, "Two waves
//     per SIMD"), a barrier after every phase.  With h = half step (32 k-elements), stage st = h / 2:
//         P0(h): group 0 multiplies h                                  | group 1 reads h
//         P1(h): group 0 reads h + 1, h even: DMA A rows of stage st + 2 | group 1 multiplies h
//                                     h odd:  DMA W rows of stage st + 2 |
//   * wave roles: group 0 issues ALL the DMA and is the only one that waits on vmcnt (stage st + 1 at the end of
//     P0(2 st + 1); the barrier publishes it); group 1 issues ALL the output stores.  Stores and loads share vmcnt, so
//     a wave that does both can wait for its loads only by also waiting for its stores; with the roles apart the
//     stores of tile t drain under the K loop of tile t + 1;
//   * persistent: one workgroup per CU walks its tiles (same XCD-aware order), so there is no workgroup turnover.
//   * LDS reads are inline asm with an explicit lgkmcnt wait (complete before the wave passes the barrier that lets
//     the DMA overwrite the slot).
// ------------------------------------------------------------------------------------------------------------
struct PP64Cfg {
  static constexpr int NW = 8, NT = 512, BM = 256, BN = 256, BK = 64;
  static constexpr int HALF = BM * BK * 2;                      // one operand's stage: 256 rows x 128 B = 32 KiB
  static constexpr int NA = 3, NB = 2;                          // A ring 3 deep, W ring 2 deep
  static constexpr int NTR = 512;                               // writer threads of the epilogue (256: group 1 only -- measured slower)
  static constexpr int PIECES = BN / 8, IT = BM * PIECES / NTR; // epilogue: 32 pieces per row, 16 rows per writer thread
  static constexpr int LDS = (NA + NB) * HALF;                  // 160 KiB (the bf16 output tile is staged in the first 128)
};

template <int ACT>
__global__ __launch_bounds__(512, 1)
void linear_bf16_pp64_kernel(int M, int N, int K, const __bf16* __restrict__ A, int lda, const __bf16* __restrict__ W, int ldw,
                             const float* __restrict__ bias, __bf16* __restrict__ C16, int ldc, float* __restrict__ C32, int ldc32,
                             float act_param, const __bf16* __restrict__ aux, int ldaux, uint8_t* __restrict__ mask, int ldmask) {
  using Cfg = PP64Cfg;
  constexpr int WN = 4, FM = 4, FN = 2, BM = Cfg::BM, BN = Cfg::BN, HALF = Cfg::HALF, NA = Cfg::NA, IT = Cfg::IT, PIECES = Cfg::PIECES;
  constexpr uint32_t WBASE = NA * HALF;                         // the W ring sits behind the A ring
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN, grp = wm;           // group 0 = waves 0-3 = tile rows 0-127
  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM, tiles = tiles_m * tiles_n;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring_smem;
  const int ns = K / 64;                                        // stages
  // fragment addresses: row R = block row + (lane & 31); chunk c = 4 hh + 2 ks + (lane >> 5) at position c ^ ((R >> 1) & 7)
  const int frow = lane & 31, fkh = lane >> 5, hi = fkh, sw = (frow >> 1) & 7;
  uint32_t offA[4], offB[4];
#pragma unroll
  for (int c2 = 0; c2 < 4; ++c2) {
    offA[c2] = lds0 + (uint32_t)((wm * 128 + frow) * 128 + (((2 * c2 + fkh) ^ sw) << 4));
    offB[c2] = lds0 + WBASE + (uint32_t)((wn * 64 + frow) * 128 + (((2 * c2 + fkh) ^ sw) << 4));
  }
  // writer threads (group 1): columns [8 piece, 8 piece + 8) of rows [32 rgroup, 32 rgroup + 32) of the tile
  const int wt = tid & (Cfg::NTR - 1), piece = wt % PIECES, rgroup = wt / PIECES;
  bf16x8 fa0[FM], fb0[FN], fa1[FM], fb1[FN];
  f32x16 acc[FM][FN];
#define PP64_READ6(fa, fb, pa, pb)                                                                                     \
  if constexpr (!probe::NOLDS)                                                                                         \
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:4096\n\tds_read_b128 %2, %6 offset:8192\n\t"       \
                 "ds_read_b128 %3, %6 offset:12288\n\tds_read_b128 %4, %7\n\tds_read_b128 %5, %7 offset:4096"          \
                 : "=&v"(fa[0]), "=&v"(fa[1]), "=&v"(fa[2]), "=&v"(fa[3]), "=&v"(fb[0]), "=&v"(fb[1])                  \
                 : "v"(pa), "v"(pb) : "memory");
  // fragments of one half step (A slot / W slot byte offsets sa_ / sb_, half HH): the 12 reads are issued first, then DMA_
  // (its issue time overlaps the reads' latency), then the wait
#define PP64_LOAD(sa_, sb_, HH, DMA_)                                                                                  \
  {                                                                                                                    \
    PP64_READ6(fa0, fb0, offA[2 * (HH)] + (sa_), offB[2 * (HH)] + (sb_));                                              \
    PP64_READ6(fa1, fb1, offA[2 * (HH) + 1] + (sa_), offB[2 * (HH) + 1] + (sb_));                                      \
    DMA_;                                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                \
                 : "+v"(fa0[0]), "+v"(fa0[1]), "+v"(fa0[2]), "+v"(fa0[3]), "+v"(fb0[0]), "+v"(fb0[1]), "+v"(fa1[0]),    \
                   "+v"(fa1[1]), "+v"(fa1[2]), "+v"(fa1[3]), "+v"(fb1[0]), "+v"(fb1[1]) :: "memory");                  \
  }
#define PP64_MULTIPLY()                                                                                                \
  if constexpr (!probe::NOMFMA) {                                                                                      \
    _Pragma("unroll") for (int i = 0; i < FM; ++i)                                                                     \
      _Pragma("unroll") for (int j = 0; j < FN; ++j)                                                                   \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb0[j], fa0[i], acc[i][j], 0, 0, 0);                       \
    _Pragma("unroll") for (int i = 0; i < FM; ++i)                                                                     \
      _Pragma("unroll") for (int j = 0; j < FN; ++j)                                                                   \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb1[j], fa1[i], acc[i][j], 0, 0, 0);                       \
  }
#define PP64_PHASE_END()                                                                                               \
  __builtin_amdgcn_sched_barrier(0);                                                                                   \
  __builtin_amdgcn_s_barrier();
  auto next_a = [&](uint32_t x) { return x == (NA - 1) * HALF ? 0u : x + HALF; };
  auto activate = [&](float v) {
    if (ACT == 1 || ACT == 5) v = fmaxf(v, 0.f);
    if (ACT == 2) { const float x = v + act_param; v = fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
    if (ACT == 3) v = (1.f / (1.f + expf(-v))) * (1.f + 2.f * act_param) - act_param;
    return v;
  };

  for (int vb = blockIdx.x; vb < tiles; vb += gridDim.x) {
    int tile_m, tile_n;
    {                                                           // XCD-aware order, see linear_bf16_kernel (gridDim.x % 8 == 0 or one tile each)
      const int xcd = vb & 7, id = vb >> 3;
      const int full = (tiles_m / 8) * 8;
      const int group = id / tiles_n;
      if (group * 8 + 8 <= full) { tile_m = group * 8 + xcd; tile_n = id - group * tiles_n; }
      else { const int r = vb - full * tiles_n; tile_m = full + r / tiles_n; tile_n = r - (r / tiles_n) * tiles_n; }
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    uint32_t mw[IT / 4];                                        // ReLU bit mask words of a writer thread's 32 rows
    uint8_t* const mask_at = mask + (size_t)((n0 >> 3) + piece) * ldmask + m0 + rgroup * IT;
#pragma unroll
    for (int q = 0; q < IT / 4; ++q) mw[q] = 0u;
    if (ACT == 6 && (Cfg::NTR == 512 || grp == 1)) {            // (issued ahead of this tile's DMA: loads return in order)
#pragma unroll
      for (int q = 0; q < IT / 16; ++q) {
        const uint4 t = *(const uint4*)(mask_at + 16 * q);
        mw[4 * q] = t.x; mw[4 * q + 1] = t.y; mw[4 * q + 2] = t.z; mw[4 * q + 3] = t.w;
      }
    }
    uint32_t sa = 0, sb = 0;                                    // ring positions (byte offsets) of the current stage
    // one loop per group (straight-line bodies); both execute the same sequence of barriers
    // DMA: a stage of one operand is 32 instructions of 8 rows x 128 B; wave w issues instructions [4 w, 4 w + 4) of BOTH
    // operands, i.e. rows [32 w, 32 w + 32): four instructions in each of its load phases, so that every load phase of
    // every wave carries the same issue work (eight in one phase made that phase 1.6x as long as the MFMA burst beside it).
    // lane l -> row l >> 3, position l & 7 holding chunk (l & 7) ^ ((row >> 1) & 7).  Whole tiles only (the launcher sends
    // ragged shapes to the ring kernel), so a row's address is a wave-uniform base (SGPRs: tile origin, wave, instruction
    // q, stage) plus a per-lane offset that only depends on the parity of q: 4 VGPRs for both operands.
    uint32_t voffA[2], voffW[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int chunk = (lane & 7) ^ ((4 * par + (lane >> 4)) & 7);        // ((8 q + (lane >> 3)) >> 1) & 7 with q & 1 = par
      voffA[par] = (uint32_t)((lane >> 3) * lda * 2 + chunk * 16);
      voffW[par] = (uint32_t)((lane >> 3) * ldw * 2 + chunk * 16);
    }
    const char* const baseA = (const char*)(A + (size_t)(m0 + 32 * wave) * lda);
    const char* const baseW = (const char*)(W + (size_t)(n0 + 32 * wave) * ldw);
    const uint32_t dstA = lds0 + (uint32_t)(32 * wave * 128), dstW = dstA + WBASE;
    auto issue_a = [&](int st, uint32_t slot_bytes) {
      if (st >= ns) return;
      if constexpr (probe::NODMA) { if (st >= 2) return; }
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16_saddr(baseA + (size_t)q * 8 * lda * 2 + (size_t)st * 128, voffA[q & 1], dstA + slot_bytes + q * 1024);
    };
    auto issue_w = [&](int st, uint32_t slot_bytes) {
      if (st >= ns) return;
      if constexpr (probe::NODMA) { if (st >= 2) return; }
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16_saddr(baseW + (size_t)q * 8 * ldw * 2 + (size_t)st * 128, voffW[q & 1], dstW + slot_bytes + q * 1024);
    };
    // every wave: its rows of stages 0 and 1, then stage 0 must have landed (stage 1: 8 instructions may be out)
    issue_a(0, 0);
    issue_w(0, 0);
    issue_a(1, HALF);
    issue_w(1, HALF);
    if (ns > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // DMA schedule (slots: the A ring is 3 deep, the W ring 2 deep; a stage's slots are free once group 1 has read its second
    // half, i.e. after P0(2 st + 1)):
    //     group 0, P1(2 st):     A rows of stage st + 2      group 1, P0(2 st):     W rows of stage st + 1 (st >= 1)
    //     group 0, P1(2 st + 1): W rows of stage st + 2      group 1, P0(2 st + 1): A rows of stage st + 2
    // every wave waits for its own rows of stage st + 1 at the end of P0(2 st + 1): only its A rows of stage st + 2 may be out.
    if (grp == 0) {
      PP64_LOAD(0u, 0u, 0, (void)0);
      PP64_PHASE_END();
      for (int st = 0; st < ns; ++st) {
        const uint32_t sa1 = next_a(sa), sb1 = sb ^ HALF;      // slots of stage st + 1
        PP64_MULTIPLY();                                        // P0(2 st)
        PP64_PHASE_END();
        PP64_LOAD(sa, sb, 1, issue_a(st + 2, next_a(sa1)));     // P1(2 st)
        PP64_PHASE_END();
        PP64_MULTIPLY();                                        // P0(2 st + 1)
        __builtin_amdgcn_sched_barrier(0);
        if (st + 2 < ns) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP64_PHASE_END();
        // P1(2 st + 1): fragments of (st + 1, half 0) (last stage: a harmless re-read)
        if (st + 1 < ns) { PP64_LOAD(sa1, sb1, 0, issue_w(st + 2, sb)); } else { PP64_LOAD(sa, sb, 0, (void)0); }
        PP64_PHASE_END();
        sa = sa1; sb = sb1;
      }
    } else {
      PP64_PHASE_END();
      for (int st = 0; st < ns; ++st) {
        const uint32_t sa1 = next_a(sa), sb1 = sb ^ HALF;
        PP64_LOAD(sa, sb, 0, if (st >= 1) issue_w(st + 1, sb1));          // P0(2 st): the W slot of stage st - 1
        PP64_PHASE_END();
        PP64_MULTIPLY();                                        // P1(2 st)
        PP64_PHASE_END();
        PP64_LOAD(sa, sb, 1, issue_a(st + 2, next_a(sa1)));     // P0(2 st + 1): the A slot of stage st - 1
        if (st + 2 < ns) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP64_PHASE_END();
        PP64_MULTIPLY();                                        // P1(2 st + 1)
        PP64_PHASE_END();
        sa = sa1; sb = sb1;
      }
    }
    // ---- epilogue of the tile.  Accumulators: lane l & 31 = tile row, register r = column (r & 3) + 8 (r >> 2) + 4 (l >> 5)
    // of a 32 x 32 block.
    if (C32 != nullptr) {                                       // float32 outputs (rare for wide layers): direct stores by every wave
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * FM * 32 + i * 32 + frow;
        if (m < M) {
#pragma unroll
          for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int n = n0 + wn * FN * 32 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
              if (n >= N) continue;
              float v = activate(acc[i][j][r] + (bias ? bias[n] : 0.f));
              if (ACT == 4) v = (float)aux[(size_t)m * ldaux + n] > 0.f ? v : 0.f;
              C32[(size_t)m * ldc32 + n] = v;
              if (C16) C16[(size_t)m * ldc + n] = (__bf16)v;
            }
        }
      }
      __builtin_amdgcn_s_barrier();
      continue;
    }
    // bf16 output: every wave stages its 128 x 64 part in the (idle) ring memory ([256][256] bf16, rows rotated by 16 bytes
    // per row against bank conflicts; 8-byte writes: a register quad is 4 consecutive columns of one row); group 1 writes
    // the tile out as 16 bytes per lane = whole 512-byte rows per 32 lanes, applying / emitting the ReLU masks
    {
      __bf16* tile = (__bf16*)ring_smem;
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int j = 0; j < FN; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nl = wn * FN * 32 + j * 32 + 8 * q + 4 * hi;        // first of 4 consecutive columns
          const int n = n0 + nl;
          float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (bias) {
            if (n + 4 <= N) b4 = *(const float4*)(bias + n);
            else { if (n < N) b4.x = bias[n]; if (n + 1 < N) b4.y = bias[n + 1]; if (n + 2 < N) b4.z = bias[n + 2]; }
          }
#pragma unroll
          for (int i = 0; i < FM; ++i) {
            const int ml = wm * FM * 32 + i * 32 + frow;
            const f32x2 lo = {activate(acc[i][j][4 * q] + b4.x), activate(acc[i][j][4 * q + 1] + b4.y)};
            const f32x2 hi2 = {activate(acc[i][j][4 * q + 2] + b4.z), activate(acc[i][j][4 * q + 3] + b4.w)};
            uint2 pk;
            pk.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));
            pk.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi2, bf16x2));
            *(uint2*)(tile + ml * BN + ((nl + 8 * ml) & (BN - 1))) = pk;
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (group 0: the bias loads; group 1: bias, mask words, and the previous tile's stores -- long done)
    __builtin_amdgcn_s_barrier();
    if (Cfg::NTR == 512 || grp == 1) {
      // (whole tiles, ldc / ldaux multiples of 8: the launcher sends everything else to the ring kernel)
      const __bf16* tile = (const __bf16*)ring_smem;
      const int n = n0 + piece * 8;
#pragma unroll
      for (int q = 0; q < IT / 4; ++q) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int ml = rgroup * IT + 4 * q + b, m = m0 + ml;
          uint4 v = *(const uint4*)(tile + ml * BN + ((piece * 8 + 8 * ml) & (BN - 1)));
          if (ACT == 5) {                                              // bit k: bf16 k is > 0 (sign clear, magnitude non-zero)
            const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
            uint32_t bits = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              bits |= (((vw[k] & 0x8000u) == 0 && (vw[k] & 0x7FFFu) != 0) ? 1u : 0u) << (2 * k);
              bits |= (((vw[k] & 0x80000000u) == 0 && (vw[k] & 0x7FFF0000u) != 0) ? 1u : 0u) << (2 * k + 1);
            }
            mw[q] |= bits << (8 * b);
          }
          if (ACT == 6) {
            const uint32_t bits = mw[q] >> (8 * b);
            v.x &= ((0u - ((bits >> 0) & 1u)) & 0x0000FFFFu) | ((0u - ((bits >> 1) & 1u)) & 0xFFFF0000u);
            v.y &= ((0u - ((bits >> 2) & 1u)) & 0x0000FFFFu) | ((0u - ((bits >> 3) & 1u)) & 0xFFFF0000u);
            v.z &= ((0u - ((bits >> 4) & 1u)) & 0x0000FFFFu) | ((0u - ((bits >> 5) & 1u)) & 0xFFFF0000u);
            v.w &= ((0u - ((bits >> 6) & 1u)) & 0x0000FFFFu) | ((0u - ((bits >> 7) & 1u)) & 0xFFFF0000u);
          }
          if (ACT == 4) {                                              // keep a bf16 where the saved activation is > 0
            const uint4 a4 = *(const uint4*)(aux + (size_t)m * ldaux + n);
            auto keep = [](uint32_t aw) {
              const uint32_t lo = ((aw & 0x8000u) == 0 && (aw & 0x7FFFu) != 0) ? 0x0000FFFFu : 0u;
              const uint32_t hi16 = ((aw & 0x80000000u) == 0 && (aw & 0x7FFF0000u) != 0) ? 0xFFFF0000u : 0u;
              return lo | hi16;
            };
            v.x &= keep(a4.x); v.y &= keep(a4.y); v.z &= keep(a4.z); v.w &= keep(a4.w);
          }
          *(uint4*)(C16 + (size_t)m * ldc + n) = v;
        }
      }
      if (ACT == 5) {
#pragma unroll
        for (int q = 0; q < IT / 16; ++q) *(uint4*)(mask_at + 16 * q) = make_uint4(mw[4 * q], mw[4 * q + 1], mw[4 * q + 2], mw[4 * q + 3]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the tile has left LDS (the stores themselves drain under the next tile)
    }
    __builtin_amdgcn_s_barrier();                                 // the ring may be refilled
  }
#undef PP64_READ6
#undef PP64_LOAD
#undef PP64_MULTIPLY
#undef PP64_PHASE_END
}

}  // namespace mip360

namespace mip360 {
// One output column (the density head, models.py:497: raw_density = Dense(1)(x)): a GEMM tile would spend 128 columns of
// MFMA work on it; this is a row dot product that streams x once (HBM-bound: 268 MB at the NerfMLP shape).  16 lanes per
// row, 16 bytes per lane and step, float32 accumulation, xor-shuffle reduction; four rows per wave and pass.
template <int ACT>
__global__ __launch_bounds__(256) void rowdot_bf16_kernel(int M, int K, const __bf16* __restrict__ A, int lda, const __bf16* __restrict__ w,
                                                          const float* __restrict__ bias, float act_param, float* __restrict__ out,
                                                          int ldo) {
  const int sub = threadIdx.x & 15, rq = threadIdx.x >> 4;            // 16 row slots per block and pass
  const float b = bias ? bias[0] : 0.f;
  for (int m = blockIdx.x * 16 + rq; m < M; m += gridDim.x * 16) {
    const __bf16* row = A + (size_t)m * lda;
    float acc = 0.f;
    for (int k = sub * 8; k < K; k += 128) {
      const bf16x8 x = *(const bf16x8*)(row + k);
      const bf16x8 ww = *(const bf16x8*)(w + k);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += (float)x[e] * (float)ww[e];
    }
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 16);
    if (sub == 0) {
      float v = acc + b;
      if (ACT == 1) v = fmaxf(v, 0.f);
      if (ACT == 2) { const float x = v + act_param; v = fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
      if (ACT == 3) v = (1.f / (1.f + expf(-v))) * (1.f + 2.f * act_param) - act_param;
      out[(size_t)m * ldo] = v;
    }
  }
}
}  // namespace mip360

template <int ACT, int WM, int WN, int FM, int FN, int NBUF>
static void launch_ring(hipStream_t st, int M, int N, int K, const void* A, int lda, const void* W, int ldw, const float* bias,
                        float act_param, void* C16, int ldc, float* C32, int ldc32, const void* aux, int ldaux, void* mask, int ldmask) {
  using namespace mip360;
  using Cfg = RingCfg<WM, WN, FM, FN, NBUF>;
  static std::atomic<uint64_t> attr_done{0};
  if (first_launch_on_this_device(attr_done)) {
    (void)hipFuncSetAttribute((const void*)linear_bf16_ring_kernel<ACT, WM, WN, FM, FN, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              Cfg::LDS);
  }
  const int tiles = ((M + Cfg::BM - 1) / Cfg::BM) * ((N + Cfg::BN - 1) / Cfg::BN);
  hipLaunchKernelGGL((linear_bf16_ring_kernel<ACT, WM, WN, FM, FN, NBUF>), dim3(tiles), dim3(Cfg::NT), Cfg::LDS, st, M, N, K,
                     (const __bf16*)A, lda, (const __bf16*)W, ldw, bias, (__bf16*)C16, ldc, C32, ldc32, act_param, (const __bf16*)aux,
                     ldaux, (uint8_t*)mask, ldmask);
}

template <int ACT>
static void launch_pp64(hipStream_t st, int M, int N, int K, const void* A, int lda, const void* W, int ldw, const float* bias,
                        float act_param, void* C16, int ldc, float* C32, int ldc32, const void* aux, int ldaux, void* mask, int ldmask) {
  using namespace mip360;
  static std::atomic<uint64_t> attr_done{0};
  if (first_launch_on_this_device(attr_done)) {
    (void)hipFuncSetAttribute((const void*)linear_bf16_pp64_kernel<ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, PP64Cfg::LDS);
  }
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    n_cu = n_cu >= 8 ? n_cu / 8 * 8 : 8;                          // a multiple of 8: virtual block id % 8 stays the XCD
  }
  const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
  hipLaunchKernelGGL((linear_bf16_pp64_kernel<ACT>), dim3(tiles < n_cu ? tiles : n_cu), dim3(512), PP64Cfg::LDS, st, M, N, K, (const __bf16*)A, lda,
                     (const __bf16*)W, ldw, bias, (__bf16*)C16, ldc, C32, ldc32, act_param, (const __bf16*)aux, ldaux, (uint8_t*)mask,
                     ldmask);
}

template <int ACT>
static void launch_linear_t(hipStream_t st, int M, int N, int K, const void* A, int lda, const void* W, int ldw, const float* bias,
                            float act_param, void* C16, int ldc, float* C32, int ldc32, const void* aux, int ldaux, void* mask,
                            int ldmask) {
  using namespace mip360;
  static const bool force_small = PROBE_GETENV("MIP360_GEMM_SMALL") != nullptr;
  static const bool no_ring = PROBE_GETENV("MIP360_GEMM_NORING") != nullptr;
  static const char* ring_env = PROBE_GETENV("MIP360_GEMM_RING");     // probes: 1 = lock-step ring kernel everywhere, 4 = 4-wave 256 x 128
  static const int ring_kind = ring_env ? atoi(ring_env) : 0;
  constexpr bool MASKED = ACT == 5 || ACT == 6;                 // bit-mask variants exist in the ring kernels only
  if (MASKED || (N >= 192 && M >= 256 && !force_small && !no_ring)) {
#define MIP360_RING(...) launch_ring<ACT, __VA_ARGS__>(st, M, N, K, A, lda, W, ldw, bias, act_param, C16, ldc, C32, ldc32, aux, ldaux, mask, ldmask)
    if (ring_kind == 4) MIP360_RING(2, 2, 4, 2, 3);                  // 4 waves, 256 x 128, two workgroups per CU (measured slower)
    else if (ring_kind == 1 || K % 64 != 0 || M % 256 != 0 || N % 256 != 0 || (C16 && ldc % 8 != 0) || (ACT == 4 && ldaux % 8 != 0))
      MIP360_RING(2, 4, 4, 2, 4);
    else launch_pp64<ACT>(st, M, N, K, A, lda, W, ldw, bias, act_param, C16, ldc, C32, ldc32, aux, ldaux, mask, ldmask);
#undef MIP360_RING
  } else if constexpr (!MASKED) {
    if (N >= 192 && M >= 256 && !force_small) {
      const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
      hipLaunchKernelGGL((linear_bf16_kernel<ACT, 2, 4, 4, 2>), dim3(tiles), dim3(512), 0, st, M, N, K, (const __bf16*)A, lda,
                         (const __bf16*)W, ldw, bias, (__bf16*)C16, ldc, C32, ldc32, act_param, (const __bf16*)aux, ldaux);
    } else {
      const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
      hipLaunchKernelGGL((linear_bf16_kernel<ACT, 2, 2, 2, 2>), dim3(tiles), dim3(256), 0, st, M, N, K, (const __bf16*)A, lda,
                         (const __bf16*)W, ldw, bias, (__bf16*)C16, ldc, C32, ldc32, act_param, (const __bf16*)aux, ldaux);
    }
  }
}

// act 5: ReLU + its bit mask written to `mask`; act 6: output multiplied by the bits of `mask` (mip360_hip.h)
void mip360_launch_linear(hipStream_t st, int M, int N, int K, const void* A, int lda, const void* W, int ldw, const float* bias,
                          int act, float act_param, void* C16, int ldc, float* C32, int ldc32, const void* aux, int ldaux,
                          void* mask, int ldmask) {
  static const bool exp_nomask = PROBE_GETENV("MIP360_EXP_NOMASK") != nullptr;       // timing experiment: dX without its ReLU mask
  if (exp_nomask && (act == 4 || act == 6)) act = 0;
  static const bool no_rowdot = PROBE_GETENV("MIP360_NO_ROWDOT") != nullptr;
  if (N == 1 && C32 && !C16 && act >= 0 && act <= 3 && K % 8 == 0 && !no_rowdot) {        // single column: row dot product
    using namespace mip360;
    const int blocks = M / 16 < 4096 ? (M + 15) / 16 : 4096;
#define MIP360_ROWDOT(ACT_) hipLaunchKernelGGL((rowdot_bf16_kernel<ACT_>), dim3(blocks), dim3(256), 0, st, M, K, (const __bf16*)A, lda, \
                                               (const __bf16*)W, bias, act_param, C32, ldc32)
    if (act == 1) MIP360_ROWDOT(1);
    else if (act == 2) MIP360_ROWDOT(2);
    else if (act == 3) MIP360_ROWDOT(3);
    else MIP360_ROWDOT(0);
#undef MIP360_ROWDOT
    return;
  }
#define MIP360_LINEAR_CASE(ACT_) \
  launch_linear_t<ACT_>(st, M, N, K, A, lda, W, ldw, bias, act_param, C16, ldc, C32, ldc32, aux, ldaux, mask, ldmask)
  if (act == 1) MIP360_LINEAR_CASE(1);
  else if (act == 2) MIP360_LINEAR_CASE(2);
  else if (act == 3) MIP360_LINEAR_CASE(3);
  else if (act == 4) MIP360_LINEAR_CASE(4);
  else if (act == 5) MIP360_LINEAR_CASE(5);
  else if (act == 6) MIP360_LINEAR_CASE(6);
  else MIP360_LINEAR_CASE(0);
#undef MIP360_LINEAR_CASE
}
