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
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

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
constexpr int RT = 256, RBK = 32, RNBUF = 4;
constexpr int RSTAGE = 2 * RT * RBK * 2;            // bytes per stage: A tile + B tile = 32 KiB
extern __shared__ __attribute__((aligned(16))) char ring_smem[];

__device__ __forceinline__ void glds16_asm(const void* g, uint32_t lds_abs) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_abs);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}

// WM x WN waves, each owning FM x FN MFMA blocks: (2, 4, 4, 2) = 8 waves of 128 x 64, two per SIMD;
// (2, 2, 4, 4) = 4 waves of 128 x 128, one per SIMD with the whole register file (8 fragment reads feed 16 MFMAs)
template <int ACT, int WM, int WN, int FM, int FN>
__global__ __launch_bounds__(WM * WN * 64) void linear_bf16_ring_kernel(int M, int N, int K, const __bf16* __restrict__ A, int lda,
                                                               const __bf16* __restrict__ W, int ldw,
                                                               const float* __restrict__ bias, __bf16* __restrict__ C16, int ldc,
                                                               float* __restrict__ C32, int ldc32, float act_param,
                                                               const __bf16* __restrict__ aux, int ldaux,
                                                               uint8_t* __restrict__ mask, int ldmask) {
  static_assert(WM * FM * 32 == RT && WN * FN * 32 == RT, "256 x 256 tile");
  constexpr int NW = WM * WN, NT = NW * 64, QD = 16 / NW;        // QD: DMA instructions per operand per wave per stage
  constexpr int IT = 8192 / NT;                                  // output rows per thread in the bf16 epilogue
  static_assert(IT % 16 == 0, "mask words: whole 16-byte groups per thread");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (N + RT - 1) / RT, tiles_m = (M + RT - 1) / RT;
  int tile_m, tile_n;
  {                                                           // XCD-aware order, see linear_bf16_kernel
    const int b = blockIdx.x, xcd = b & 7, id = b >> 3;
    const int full = (tiles_m / 8) * 8;
    const int group = id / tiles_n;
    if (group * 8 + 8 <= full) { tile_m = group * 8 + xcd; tile_n = id - group * tiles_n; }
    else { const int r = b - full * tiles_n; tile_m = full + r / tiles_n; tile_n = r - (r / tiles_n) * tiles_n; }
  }
  const int m0 = tile_m * RT, n0 = tile_n * RT;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring_smem;
  // DMA lane map: instruction q of this wave covers tile rows [16 * (wave * QD + q % QD) ... + 16) of operand q / QD
  const int drow = lane >> 2, dpos = lane & 3;
  const char* gsrc[2 * QD];
  uint32_t ldst[2 * QD];
#pragma unroll
  for (int q = 0; q < 2 * QD; ++q) {
    const int op = q / QD, row = 16 * (wave * QD + q % QD) + drow;
    const int chunk = dpos ^ ((row >> 2) & 3);
    if (op == 0) {
      int m = m0 + row; m = m < M ? m : M - 1;
      gsrc[q] = (const char*)(A + (size_t)m * lda + chunk * 8);
    } else {
      int n = n0 + row; n = n < N ? n : N - 1;
      gsrc[q] = (const char*)(W + (size_t)n * ldw + chunk * 8);
    }
    ldst[q] = (uint32_t)(op * RT * RBK * 2 + 16 * (wave * QD + q % QD) * 64);      // + slot * RSTAGE, + lane * 16 by the hardware
  }
  const int nk = K / RBK;
  auto issue = [&](int kt) {
    if (kt >= nk) return;
    const uint32_t slot = (uint32_t)(kt % RNBUF) * RSTAGE;
#pragma unroll
    for (int q = 0; q < 2 * QD; ++q) glds16_asm(gsrc[q] + (size_t)kt * RBK * 2, lds0 + slot + ldst[q]);
  };
  // ReLU bit mask (ACT 5 writes it, ACT 6 applies it), column-byte-major: byte [(n >> 3) * ldmask + m], bit n & 7 =
  // (C[m][n] > 0).  In the bf16 epilogue a thread owns columns [8 piece, 8 piece + 8) of IT consecutive rows, i.e. IT
  // consecutive mask bytes: ACT 6 fetches them here, ahead of the K loop (IT / 4 registers), ACT 5 stores them at the end.
  uint32_t mw[IT / 4];
  uint8_t* const mask_at = mask + (size_t)((n0 >> 3) + (tid & 31)) * ldmask + m0 + (tid >> 5) * IT;
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
  for (int s_ = 0; s_ < RNBUF - 1; ++s_) issue(s_);
  // fragment read: row R = block row + (lane & 31), k chunk c = 2 ks + (lane >> 5) -> position c ^ ((R >> 2) & 3)
  const int frow = lane & 31, fkh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int younger = nk - 1 - kt < RNBUF - 2 ? nk - 1 - kt : RNBUF - 2;
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * QD) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(kt + RNBUF - 1);
    const char* st = ring_smem + (kt % RNBUF) * RSTAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[FM], fb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int R = wm * FM * 32 + i * 32 + frow;
        fa[i] = *(const bf16x8*)(st + R * 64 + (((2 * ks + fkh) ^ ((R >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int R = wn * FN * 32 + j * 32 + frow;
        fb[j] = *(const bf16x8*)(st + RT * RBK * 2 + R * 64 + (((2 * ks + fkh) ^ ((R >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        // operands swapped (W fragment as "A"): the accumulator block is C^T, i.e. lane (l & 31) holds ROW m of the
        // tile and register r column (r & 3) + 8 (r >> 2) + 4 (l >> 5) -- four consecutive columns per register
        // quad, which the epilogue packs into one 8-byte LDS write
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
  }
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
  // bf16 output.  The ring is idle now, so the tile is staged through it ([256][256] bf16 = its 128 KiB, rows rotated by
  // 16 bytes per row against bank conflicts; 8-byte writes: a register quad is 4 consecutive columns of one row) and
  // leaves as 16 bytes per lane = whole 512-byte rows per 32 lanes; the ReLU mask (ACT 4) is applied on the way out
  // from equally coalesced 16-byte loads of the saved activation.
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
          *(uint2*)(tile + ml * 256 + ((nl + 8 * ml) & 255)) = pk;
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    const __bf16* tile = (const __bf16*)ring_smem;
    const bool vec_ok = ((ldc & 7) == 0) && (ACT != 4 || (ldaux & 7) == 0);
    const int piece = tid & 31, n = n0 + piece * 8;                    // 32 pieces of 8 columns per row
#pragma unroll
    for (int q = 0; q < IT / 4; ++q) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int ml = (tid >> 5) * IT + 4 * q + b, m = m0 + ml;
        if (m >= M || n >= N) continue;
        uint4 v = *(const uint4*)(tile + ml * 256 + ((piece * 8 + 8 * ml) & 255));
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

}  // namespace mip360

template <int ACT>
static void launch_linear_t(hipStream_t st, int M, int N, int K, const void* A, int lda, const void* W, int ldw, const float* bias,
                            float act_param, void* C16, int ldc, float* C32, int ldc32, const void* aux, int ldaux, void* mask,
                            int ldmask) {
  using namespace mip360;
  static const bool force_small = getenv("MIP360_GEMM_SMALL") != nullptr;
  static const bool no_ring = getenv("MIP360_GEMM_NORING") != nullptr;
  static const bool four_waves = getenv("MIP360_GEMM_4WAVES") != nullptr;
  constexpr bool MASKED = ACT == 5 || ACT == 6;                 // bit-mask variants exist in the ring kernel only
  if (MASKED || (N >= 192 && M >= 256 && !force_small && !no_ring)) {
    const int tiles = ((M + RT - 1) / RT) * ((N + RT - 1) / RT);
    if (four_waves && !MASKED)
      hipLaunchKernelGGL((linear_bf16_ring_kernel<ACT, 2, 2, 4, 4>), dim3(tiles), dim3(256), RNBUF * RSTAGE, st, M, N, K, (const __bf16*)A,
                         lda, (const __bf16*)W, ldw, bias, (__bf16*)C16, ldc, C32, ldc32, act_param, (const __bf16*)aux, ldaux,
                         (uint8_t*)mask, ldmask);
    else
      hipLaunchKernelGGL((linear_bf16_ring_kernel<ACT, 2, 4, 4, 2>), dim3(tiles), dim3(512), RNBUF * RSTAGE, st, M, N, K, (const __bf16*)A,
                         lda, (const __bf16*)W, ldw, bias, (__bf16*)C16, ldc, C32, ldc32, act_param, (const __bf16*)aux, ldaux,
                         (uint8_t*)mask, ldmask);
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
  static const bool exp_nomask = getenv("MIP360_EXP_NOMASK") != nullptr;       // timing experiment: dX without its ReLU mask
  if (exp_nomask && (act == 4 || act == 6)) act = 0;
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
