// Training-side kernels of the MipNeRF-360 path: weight-gradient GEMM, bias gradients, head
// backward, gradient-norm clipping and Adam.  Upstream relies on jax.grad + optax (internal/train_utils.py:215-236,
// 303-370); the closed forms are restated in oracle/mip360_oracle.py (mlp_backward) and checked there by finite
// differences.
//
//   grad_weight_bf16 : dK[I, O] = H[M, I]^T * dZ[M, O]  (flax kernel layout [in, out]), bf16 operands, f32 result.
//     The contraction runs over the sample rows M (131 072 at 4096 rays x 32 samples) -- both operands are
//     row-major with M as the slow axis, so MFMA fragments (8 consecutive k per lane) come from
//     ds_read_b64_tr_b16 transposed reads of row-major [32 rows][128 cols] LDS tiles, as in the NeRF++ dw_kernel.
//     Split-K: blockIdx = (tile_i, tile_o, slice); every slice writes its own f32 slab and a second kernel sums the
//     slabs in a fixed order (deterministic, no atomics).
#include "probe_env.h"
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/mip360_hip.h"

namespace mip360 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GT = 128;            // output tile (both ways)
constexpr int GK = 32;             // rows per K step
constexpr int GROWB = GT * 2 + 16; // LDS row stride in bytes (272: consecutive rows 4 banks apart)

extern __shared__ __attribute__((aligned(16))) char gw_smem[];

__device__ __forceinline__ bf16x8 tr_frag(uint32_t off) {
  __attribute__((address_space(3))) char* base = (__attribute__((address_space(3))) char*)gw_smem;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + off));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + off + GROWB));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(256) void grad_weight_kernel(int M, int I, int O, const __bf16* __restrict__ H, int ldh,
                                                          const __bf16* __restrict__ dZ, int lddz, int ksplit,
                                                          float* __restrict__ slabs, int ldc, float* __restrict__ bias_slabs) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wo = wave & 1;
  const int tiles_o = (O + GT - 1) / GT, tiles_i = (I + GT - 1) / GT;
  // XCD-aware order (workgroup b runs on XCD b % 8, one L2 per XCD): all tiles of a row slice go to ONE XCD, so the
  // slice's H / dZ row tiles are fetched from HBM once and re-used by its tiles_i * tiles_o workgroups through that L2
  int slice, b;
  if ((ksplit & 7) == 0) {
    const int xcd = blockIdx.x & 7, id = blockIdx.x >> 3, per_xcd = ksplit >> 3;
    slice = xcd * per_xcd + id / (tiles_i * tiles_o);
    b = id % (tiles_i * tiles_o);
  } else {
    slice = blockIdx.x / (tiles_i * tiles_o);
    b = blockIdx.x - slice * tiles_i * tiles_o;
  }
  const int ti = b / tiles_o, to = b - ti * tiles_o;
  const int i0 = ti * GT, o0 = to * GT;
  // rows of this slice, in multiples of GK
  const int64_t steps_total = ((int64_t)M + GK - 1) / GK;
  const int64_t per = (steps_total + ksplit - 1) / ksplit;
  const int64_t s_begin = (int64_t)slice * per, s_end = s_begin + per < steps_total ? s_begin + per : steps_total;
  char* sH = gw_smem;                                   // [2][GK][GROWB]
  char* sZ = gw_smem + 2 * GK * GROWB;
  // loads: a tile is 32 rows x 256 B = 512 16-byte chunks, 2 per thread: chunk c -> row c >> 4, byte (c & 15) * 16
  uint4 rh[2], rz[2];
  auto fetch = [&](int64_t step) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = tid + q * 256, row = c >> 4, cb = (c & 15) * 8;         // cb in elements
      const int64_t m = step * GK + row;
      const bool rok = m < M;
      rh[q] = (rok && i0 + cb < I) ? *(const uint4*)(H + (size_t)m * ldh + i0 + cb) : make_uint4(0, 0, 0, 0);
      rz[q] = (rok && o0 + cb < O) ? *(const uint4*)(dZ + (size_t)m * lddz + o0 + cb) : make_uint4(0, 0, 0, 0);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = tid + q * 256, row = c >> 4, cbyte = (c & 15) * 16;
      *(uint4*)(sH + (buf * GK + row) * GROWB + cbyte) = rh[q];
      *(uint4*)(sZ + (buf * GK + row) * GROWB + cbyte) = rz[q];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  // bias gradient = column sums of dZ: the dZ fragments of the first input tile's first wave row already hold every
  // column of this output tile (lane = column, 8 rows per fragment), so they are summed on the side (VALU)
  const bool do_bias = bias_slabs != nullptr && ti == 0 && wi == 0;
  float bsum[2] = {0.f, 0.f};
  // transposed-read lane map (see nerfpp_dw.hip): 16-lane group g = lane >> 4: k half g >> 1, column sub-block g & 1;
  // lane a16 = lane & 15: row pair a16 >> 2, 4-column piece a16 & 3
  const int g = lane >> 4, a16 = lane & 15;
  const uint32_t lane_off = (uint32_t)((8 * (g >> 1) + 2 * (a16 >> 2)) * GROWB + (16 * (g & 1) + 4 * (a16 & 3)) * 2);
  if (s_begin < s_end) {
    fetch(s_begin);
    stash(0);
  }
  __syncthreads();
  for (int64_t st = s_begin; st < s_end; ++st) {
    const int buf = (int)((st - s_begin) & 1);
    if (st + 1 < s_end) fetch(st + 1);
#pragma unroll
    for (int kk = 0; kk < GK / 16; ++kk) {
      bf16x8 fh[2], fz[2];
      const uint32_t base = (uint32_t)((buf * GK + kk * 16) * GROWB) + lane_off;
#pragma unroll
      for (int x = 0; x < 2; ++x) fh[x] = tr_frag(base + (uint32_t)((wi * 64 + x * 32) * 2));
#pragma unroll
      for (int y = 0; y < 2; ++y) fz[y] = tr_frag((uint32_t)(2 * GK * GROWB) + base + (uint32_t)((wo * 64 + y * 32) * 2));
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[x], fz[y], acc[x][y], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[y] += (float)fz[y][e];
      }
    }
    if (st + 1 < s_end) stash(buf ^ 1);
    __syncthreads();
  }
  float* slab = slabs + (size_t)slice * I * ldc;
  const int hi = lane >> 5, j = lane & 31;
#pragma unroll
  for (int y = 0; y < 2; ++y) {
    const int o = o0 + wo * 64 + y * 32 + j;
    if (o >= O) continue;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + wi * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (i < I) slab[(size_t)i * ldc + o] = acc[x][y][r];
      }
  }
  if (do_bias) {
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const float tot = bsum[y] + __shfl_xor(bsum[y], 32, 64);       // the two k halves of a column
      const int o = o0 + wo * 64 + y * 32 + j;
      if (hi == 0 && o < O) bias_slabs[(size_t)slice * O + o] = tot;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Wide layers (I and O multiples of 256, M a multiple of 32): 256 x 256 output tile per workgroup, 8 waves (2 along I x
// 4 along O, 128 x 64 = 4 x 2 MFMA blocks each), operands DMA'd unchanged (row-major, 32 rows x 512 B per operand and
// K step) into a 4-deep LDS ring with global_load_lds_dwordx4 -- no VGPR staging, three K steps in flight, counted
// vmcnt, one raw s_barrier per step -- and read back through ds_read_b64_tr_b16 (same image as the NeRF++ dw_kernel:
// a 1 KiB DMA instruction covers 2 rows, every 1 KiB segment is followed by 64 B of padding so that the 4 rows one
// transposed read touches fall into disjoint bank windows).  Per K step a wave reads 6 fragments for 8 MFMAs (the
// 128 x 128 kernel above: 4 for 4) and a workgroup's operand bytes per FLOP are halved.
// ------------------------------------------------------------------------------------------------------------
constexpr int WSEG = 1024 + 64, WOPER = 16 * WSEG, WNBUF = 4;       // LDS: 4 x 2 x 17 KiB = 136 KiB

__device__ __forceinline__ void glds16_nt(const void* g, uint32_t lds_abs) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_abs);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
__device__ __forceinline__ bf16x8 tr_frag_wide(uint32_t off) {
  __attribute__((address_space(3))) char* base = (__attribute__((address_space(3))) char*)gw_smem;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + off));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + off + 512));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(512) void grad_weight_wide_kernel(int M, int I, int O, const __bf16* __restrict__ H, int ldh,
                                                               const __bf16* __restrict__ dZ, int lddz, int ksplit,
                                                               float* __restrict__ slabs, int ldc, float* __restrict__ bias_slabs) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 2, wo = wave & 3;
  const int tiles_o = O / 256, tiles = (I / 256) * tiles_o;
  int slice, b;                                         // XCD-aware order, see grad_weight_kernel
  if ((ksplit & 7) == 0) {
    const int xcd = blockIdx.x & 7, id = blockIdx.x >> 3, per_xcd = ksplit >> 3;
    slice = xcd * per_xcd + id / tiles;
    b = id % tiles;
  } else {
    slice = blockIdx.x / tiles;
    b = blockIdx.x - slice * tiles;
  }
  const int ti = b / tiles_o, to = b - ti * tiles_o;
  const int i0 = ti * 256, o0 = to * 256;
  const int64_t chunks_total = M / 32;
  const int64_t per = (chunks_total + ksplit - 1) / ksplit;
  const int64_t c_begin = (int64_t)slice * per, c_end = c_begin + per < chunks_total ? c_begin + per : chunks_total;
  const int nchunk = c_end > c_begin ? (int)(c_end - c_begin) : 0;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)gw_smem;
  // DMA: 2 operands x 16 segments per chunk = 32 wave-instructions, 4 per wave; lane -> (row of the pair, 16-byte column)
  const int dma_col = (lane & 31) * 16, dma_row = lane >> 5;
  const char* gh = (const char*)(H + i0) + (size_t)dma_row * ldh * 2 + dma_col;
  const char* gz = (const char*)(dZ + o0) + (size_t)dma_row * lddz * 2 + dma_col;
  auto issue = [&](int c) {
    if (c >= nchunk) return;
    const int64_t r0 = (c_begin + c) * 32;
    const uint32_t buf = lds_base + (uint32_t)(c % WNBUF) * (2 * WOPER);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int id = x * 8 + wave, op = id >> 4, seg = id & 15;
      const char* src = op == 0 ? gh + (size_t)(r0 + 2 * seg) * ldh * 2 : gz + (size_t)(r0 + 2 * seg) * lddz * 2;
      glds16_nt(src, buf + op * WOPER + seg * WSEG);
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
  // transposed-read lane map: 16-lane group g = lane >> 4: k half g >> 1, column sub-block g & 1; a16 = lane & 15: row
  // pair (segment) a16 >> 2 of the half, 4-column piece a16 & 3
  const int g = lane >> 4, a16 = lane & 15;
  const uint32_t lane_off = (uint32_t)((4 * (g >> 1) + (a16 >> 2)) * WSEG + (16 * (g & 1) + 4 * (a16 & 3)) * 2);
#pragma unroll
  for (int c = 0; c < WNBUF - 1; ++c) issue(c);
  for (int c = 0; c < nchunk; ++c) {
    const int younger = nchunk - 1 - c < WNBUF - 2 ? nchunk - 1 - c : WNBUF - 2;
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(c + WNBUF - 1);
    const uint32_t buf = (uint32_t)(c % WNBUF) * (2 * WOPER) + lane_off;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fh[4], fz[2];
#pragma unroll
      for (int x = 0; x < 4; ++x) fh[x] = tr_frag_wide(buf + kk * 8 * WSEG + (4 * wi + x) * 64);
#pragma unroll
      for (int y = 0; y < 2; ++y) fz[y] = tr_frag_wide(buf + WOPER + kk * 8 * WSEG + (2 * wo + y) * 64);
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[x], fz[y], acc[x][y], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[y] += (float)fz[y][e];
      }
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

// out[e] = scale * sum_s slabs[s][e] in a fixed order: a thread sums every 4th slab (group g: s = g, g + 4, ...) for 4
// consecutive elements, the 4 group sums are added in group order through LDS.  (One thread per element over all
// slabs has too few loads in flight: 64 MB of slabs took 64 us.)
__global__ __launch_bounds__(256) void slab_sum_kernel(int64_t n, int ksplit, const float* __restrict__ slabs, float scale,
                                                       float* __restrict__ out) {
  __shared__ float4 part[4][64];
  const int t = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t e = ((int64_t)blockIdx.x * 64 + t) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < n) {
    if ((n & 3) == 0) {
      for (int s_ = g; s_ < ksplit; s_ += 4) {
        const float4 v = *(const float4*)(slabs + (size_t)s_ * n + e);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    } else {
      for (int s_ = g; s_ < ksplit; s_ += 4) {
        const float* p = slabs + (size_t)s_ * n + e;
        acc.x += p[0];
        if (e + 1 < n) acc.y += p[1];
        if (e + 2 < n) acc.z += p[2];
        if (e + 3 < n) acc.w += p[3];
      }
    }
  }
  part[g][t] = acc;
  __syncthreads();
  if (g == 0 && e < n) {
    float4 r = part[0][t];
#pragma unroll
    for (int k = 1; k < 4; ++k) { r.x += part[k][t].x; r.y += part[k][t].y; r.z += part[k][t].z; r.w += part[k][t].w; }
    out[e] = r.x * scale;
    if (e + 1 < n) out[e + 1] = r.y * scale;
    if (e + 2 < n) out[e + 2] = r.z * scale;
    if (e + 3 < n) out[e + 3] = r.w * scale;
  }
}

// One output column (the density head's kernel gradient, d kernel[i] = sum_m H[m][i] dz[m]): a column dot product that
// streams H once instead of an MFMA tile with 127 idle columns.  Block = (row slice, group of 256 input columns); a thread
// owns 8 columns (16 bytes) of every 8th row of the slice, so a row of the block is 512 contiguous bytes (64-column
// groups: one cache line per row); sixteen rows in flight per thread (with four the kernel was latency-bound: 78 us for either shape); float32 accumulation, the 8
// row-threads of a column are combined through LDS in a fixed order.  Slabs as the MFMA kernels write them:
// slabs[slice][i], bias slabs [slice] = sum_m dz[m].
__global__ __launch_bounds__(256) void grad_weight_col_kernel(int M, int I, const __bf16* __restrict__ H, int ldh,
                                                              const __bf16* __restrict__ dZ, int lddz, int ksplit,
                                                              float* __restrict__ slabs, int ldc, float* __restrict__ bias_slabs) {
  __shared__ float part[8][257];
  __shared__ float bpart[8];
  const int cgroups = (I + 255) / 256;
  const int slice = blockIdx.x / cgroups, cg = blockIdx.x - slice * cgroups;
  const int sub = threadIdx.x & 31, rt = threadIdx.x >> 5;            // 32 threads x 8 columns per row, 8 rows per pass
  const int64_t per = ((int64_t)M + ksplit - 1) / ksplit;
  const int64_t r_begin = (int64_t)slice * per, r_end = r_begin + per < M ? r_begin + per : M;
  const int col = cg * 256 + sub * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  if (col < I) {
    int64_t m = r_begin + rt;
    constexpr int U = 16;                          // rows in flight per thread: a block per CU must cover ~2 us of HBM latency by itself
    for (; m + 8 * (U - 1) < r_end; m += 8 * U) {
      bf16x8 h[U];
      float z[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        h[u] = *(const bf16x8*)(H + (size_t)(m + 8 * u) * ldh + col);
        z[u] = (float)dZ[(size_t)(m + 8 * u) * lddz];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (float)h[u][e] * z[u];
        bsum += z[u];
      }
    }
    for (; m < r_end; m += 8) {
      const float z = (float)dZ[(size_t)m * lddz];
      const bf16x8 h = *(const bf16x8*)(H + (size_t)m * ldh + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)h[e] * z;
      bsum += z;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[rt][sub * 8 + e] = acc[e];
  if (sub == 0) bpart[rt] = bsum;
  __syncthreads();
  {
    float t = 0.f;
    for (int r = 0; r < 8; ++r) t += part[r][threadIdx.x];
    const int i = cg * 256 + threadIdx.x;
    if (i < I) slabs[(size_t)slice * I * ldc + (size_t)i * ldc] = t;
  }
  if (bias_slabs && cg == 0 && threadIdx.x == 0) {
    float t = 0.f;
    for (int r = 0; r < 8; ++r) t += bpart[r];
    bias_slabs[slice] = t;
  }
}

// the same for two slab sets in one launch (a layer's kernel and bias gradients): blocks [0, blocks_a) reduce set a
// (n_a elements of slabs whose stride is stride_a >= n_a), the remaining blocks set b
__global__ __launch_bounds__(256) void slab_sum2_kernel(int blocks_a, int64_t n_a, int64_t stride_a, const float* __restrict__ slabs_a,
                                                        float* __restrict__ out_a, int64_t n_b, int64_t stride_b,
                                                        const float* __restrict__ slabs_b, float* __restrict__ out_b, int ksplit,
                                                        float scale) {
  __shared__ float4 part[4][64];
  const bool second = (int)blockIdx.x >= blocks_a;
  const int64_t n = second ? n_b : n_a, stride = second ? stride_b : stride_a;
  const float* slabs = second ? slabs_b : slabs_a;
  float* out = second ? out_b : out_a;
  const int blk = second ? blockIdx.x - blocks_a : blockIdx.x;
  const int t = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t e = ((int64_t)blk * 64 + t) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < n) {
    if ((n & 3) == 0 && (stride & 3) == 0) {
      int s_ = g;
      for (; s_ + 12 < ksplit; s_ += 16) {                 // four slabs in flight per thread (same summation order)
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const float4*)(slabs + (size_t)(s_ + 4 * u) * stride + e);
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
      for (; s_ < ksplit; s_ += 4) {
        const float4 v = *(const float4*)(slabs + (size_t)s_ * stride + e);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    } else {
      for (int s_ = g; s_ < ksplit; s_ += 4) {
        const float* p = slabs + (size_t)s_ * stride + e;
        acc.x += p[0];
        if (e + 1 < n) acc.y += p[1];
        if (e + 2 < n) acc.z += p[2];
        if (e + 3 < n) acc.w += p[3];
      }
    }
  }
  part[g][t] = acc;
  __syncthreads();
  if (g == 0 && e < n) {
    float4 r = part[0][t];
#pragma unroll
    for (int k = 1; k < 4; ++k) { r.x += part[k][t].x; r.y += part[k][t].y; r.z += part[k][t].z; r.w += part[k][t].w; }
    out[e] = r.x * scale;
    if (e + 1 < n) out[e + 1] = r.y * scale;
    if (e + 2 < n) out[e + 2] = r.z * scale;
    if (e + 3 < n) out[e + 3] = r.w * scale;
  }
}

// bias gradients: partial[slice][o] = sum over the slice's rows of dZ[m][o]
__global__ __launch_bounds__(256) void col_sum_kernel(int M, int O, const __bf16* __restrict__ dZ, int ld, int nslice,
                                                      float* __restrict__ partial) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  const int slice = blockIdx.y;
  const int64_t per = ((int64_t)M + nslice - 1) / nslice;
  const int64_t m0 = (int64_t)slice * per, m1 = m0 + per < M ? m0 + per : M;
  if (o >= O) return;
  float acc = 0.f;
  for (int64_t m = m0; m < m1; ++m) acc += (float)dZ[(size_t)m * ld + o];
  partial[(size_t)slice * O + o] = acc;
}

// Heads (models.py:497, 573-594): density = softplus(raw - 1) -> d raw = g * (1 - exp(-density));
// rgb = sigmoid(pre) (1 + 2p) - p -> d pre = g (1 + 2p) s (1 - s), s = (rgb + p) / (1 + 2p).
// d_raw goes to column `raw_col` of a bf16 [rows, ld_raw] tensor (the columns next to it up to a multiple of 32 are
// zero-filled by the caller once); d_pre to columns 0..2 of a bf16 [rows, 32] tensor (3..31 zero-filled here).
__global__ void head_backward_kernel(int64_t rows, const float* __restrict__ density, const float* __restrict__ g_density,
                                     const float* __restrict__ rgb, const float* __restrict__ g_rgb, float pad,
                                     __bf16* __restrict__ d_raw, int ld_raw, int raw_col, int raw_zero_to,
                                     __bf16* __restrict__ d_pre) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float v = g_density[r] * (1.f - expf(-density[r]));
  d_raw[(size_t)r * ld_raw + raw_col] = (__bf16)v;
  for (int c = raw_col + 1; c < raw_zero_to; ++c) d_raw[(size_t)r * ld_raw + c] = (__bf16)0.f;
  if (d_pre) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float s = (rgb[r * 3 + c] + pad) / (1.f + 2.f * pad);
      d_pre[(size_t)r * 32 + c] = (__bf16)(g_rgb[r * 3 + c] * (1.f + 2.f * pad) * s * (1.f - s));
    }
    for (int c = 3; c < 32; ++c) d_pre[(size_t)r * 32 + c] = (__bf16)0.f;
  }
}

// deterministic sum of squares: partial[b] per block, then block 0 style reduce by a second launch
// (1024 threads per block and 16-byte loads: with 256 threads the 256 blocks were 4 waves per CU of dependent 4-byte loads -- 52 us
// for the NerfMLP's 50 MB.  Fixed order: thread-strided double sums, then a tree over the block.)
__global__ __launch_bounds__(1024) void sumsq_partial_kernel(int64_t n, const float* __restrict__ g, float* __restrict__ partial) {
  __shared__ double sh[1024];
  double acc = 0;
  const int64_t n4 = ((uintptr_t)g & 15) == 0 ? n >> 2 : 0;
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 1024) {
    const float4 v = ((const float4*)g)[i];
    acc += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 1024) acc += (double)g[i] * g[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int d = 512; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = (float)sh[0];
}
// clip_gradients (train_utils.py:215-236): mult = min(1, max_norm / (eps + ||g||)) over all tensors of ONE MLP;
// `partials` holds the per-tensor partial sums of that MLP back to back
__global__ __launch_bounds__(256) void clip_mult_kernel(int n_partial, const float* __restrict__ partial, float max_norm, float* __restrict__ out) {
  __shared__ double sh[256];
  double acc = 0;
  for (int i = threadIdx.x; i < n_partial; i += 256) acc += (double)partial[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x) return;
  const float norm = (float)sqrt(sh[0]);
  // jnp.minimum(1, max_norm / (eps + norm)) (train_utils.py:228-229) PROPAGATES a NaN norm (fminf would return 1)
  const float ratio = max_norm / (1.1920928955078125e-07f + norm);
  out[0] = max_norm > 0.f ? (ratio != ratio ? ratio : fminf(1.f, ratio)) : 1.f;
  out[1] = norm;
}
// optax.adam (scale_by_adam + scale(-lr)): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; update = -lr m_hat / (sqrt(v_hat) + eps)
__global__ void adam_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, const float* __restrict__ gmult, float lr, float b1, float b2, float eps,
                            float bc1, float bc2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gi = g[i] * (gmult ? gmult[0] : 1.f);
  // grad = tree_map(jnp.nan_to_num, grad) after the clipping (train_utils.py:345): NaN -> 0, +-inf -> +-FLT_MAX, so one
  // non-finite gradient (then norm = inf / NaN, multiplier 0 / NaN) costs one step instead of poisoning mu, nu, params
  if (gi != gi) gi = 0.f;
  else if (gi > 3.4028234663852886e38f) gi = 3.4028234663852886e38f;
  else if (gi < -3.4028234663852886e38f) gi = -3.4028234663852886e38f;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  p[i] = p[i] - lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
}
// f32 [rows, cols] parameter (flax kernel [in, out]) -> bf16 copies: fwd [out, in_pad] (transposed, zero padded) and
// bwd [in_pad?]: the kernel as stored, [in, out_pad], both K-contiguous for the NT dense-layer kernel
// element (r, c) of an fm tensor with ld columns (mip360_fm.hip), in elements
__device__ __forceinline__ size_t fm_elem(int r, int c, int ld) {
  const int row = r & 31, f = c & 15, hi = (f >> 2) & 1;
  return ((size_t)(r >> 5) * (ld >> 4) + (c >> 4)) * 512 + (size_t)(8 * (row >> 2) + 4 * (hi ^ (row >> 4)) + (row & 3)) * 8 + 4 * (f >> 3) + (f & 3);
}
// ... and the fm copies the fragment-major kernels read: fwd_fm [n_out, ld_fwd_fm] (element (o, i)), bwd_fm [rows, ld_bwd_fm]
// (element (i, bwd_col0 + o) for i < bwd_rows); their zero padding is the caller's (written once)
__global__ void pack_weight_kernel(int n_in, int n_out, const float* __restrict__ k, __bf16* __restrict__ fwd, int ld_fwd,
                                   __bf16* __restrict__ bwd, int ld_bwd, __bf16* __restrict__ fwd_fm, int ld_fwd_fm,
                                   __bf16* __restrict__ bwd_fm, int ld_bwd_fm, int bwd_rows, int bwd_col0) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n_in * n_out) return;
  const int i = (int)(e / n_out), o = (int)(e - (int64_t)i * n_out);
  const __bf16 v = (__bf16)k[e];
  if (fwd) fwd[(size_t)o * ld_fwd + i] = v;
  if (bwd) bwd[(size_t)i * ld_bwd + o] = v;
  if (fwd_fm) fwd_fm[fm_elem(o, i, ld_fwd_fm)] = v;
  if (bwd_fm && i < bwd_rows) bwd_fm[fm_elem(i, bwd_col0 + o, ld_bwd_fm)] = v;
}

}  // namespace mip360

using namespace mip360;

// the 256 x 256-tile kernel needs whole tiles, whole 32-row chunks and 16-byte aligned rows
bool mip360_grad_weight_is_wide(int M, int I, int O, int ldh, int lddz) {
  static const bool off = PROBE_GETENV("MIP360_DW_NARROW") != nullptr;
  return !off && M % 32 == 0 && M >= 256 && I % 256 == 0 && O % 256 == 0 && ldh % 8 == 0 && lddz % 8 == 0;
}
void mip360_launch_grad_weight_reduce(hipStream_t st, int rows, int I_slab, int O, int ksplit, const float* slabs, float* out, int ldc,
                                      float scale, float* bias_out);
void mip360_launch_grad_weight(hipStream_t st, int M, int I, int O, const void* H, int ldh, const void* dZ, int lddz, int ksplit,
                               float* slabs, float* out, int ldc, float scale, float* bias_out) {
  const int tiles = ((I + GT - 1) / GT) * ((O + GT - 1) / GT);
  const size_t lds = 4 * GK * GROWB;
  const int64_t n = (int64_t)I * ldc;
  float* bias_slabs = bias_out ? slabs + (size_t)ksplit * n : nullptr;            // [ksplit][O] after the kernel slabs
  static const bool no_col = PROBE_GETENV("MIP360_NO_COLDOT") != nullptr;
  if (O == 1 && I % 8 == 0 && !no_col)
    hipLaunchKernelGGL(grad_weight_col_kernel, dim3(((I + 255) / 256) * ksplit), dim3(256), 0, st, M, I, (const __bf16*)H, ldh,
                       (const __bf16*)dZ, lddz, ksplit, slabs, ldc, bias_slabs);
  else if (mip360_grad_weight_is_wide(M, I, O, ldh, lddz))
    hipLaunchKernelGGL(grad_weight_wide_kernel, dim3((I / 256) * (O / 256) * ksplit), dim3(512), WNBUF * 2 * WOPER, st, M, I, O,
                       (const __bf16*)H, ldh, (const __bf16*)dZ, lddz, ksplit, slabs, ldc, bias_slabs);
  else
  hipLaunchKernelGGL(grad_weight_kernel, dim3(tiles * ksplit), dim3(256), lds, st, M, I, O, (const __bf16*)H, ldh,
                     (const __bf16*)dZ, lddz, ksplit, slabs, ldc, bias_slabs);
  if (out) mip360_launch_grad_weight_reduce(st, I, I, O, ksplit, slabs, out, ldc, scale, bias_out);
}
// kernel gradient = scale * sum of the split-K slabs (first `rows` of the I_slab input rows of every slab), bias gradient
// likewise from the [ksplit][O] slabs behind them: one launch
void mip360_launch_grad_weight_reduce(hipStream_t st, int rows, int I_slab, int O, int ksplit, const float* slabs, float* out, int ldc,
                                      float scale, float* bias_out) {
  const int64_t stride = (int64_t)I_slab * ldc, n = (int64_t)rows * ldc;
  const float* bias_slabs = slabs + (size_t)ksplit * stride;
  const int blocks_a = (int)((n + 255) / 256), blocks_b = bias_out ? (O + 255) / 256 : 0;
  hipLaunchKernelGGL(slab_sum2_kernel, dim3(blocks_a + blocks_b), dim3(256), 0, st, blocks_a, n, stride, slabs, out, (int64_t)O,
                     (int64_t)O, bias_slabs, bias_out, ksplit, scale);
}
void mip360_launch_col_sum(hipStream_t st, int M, int O, const void* dZ, int ld, int nslice, float* partial, float* out,
                           float scale) {
  hipLaunchKernelGGL(col_sum_kernel, dim3((O + 255) / 256, nslice), dim3(256), 0, st, M, O, (const __bf16*)dZ, ld, nslice, partial);
  hipLaunchKernelGGL(slab_sum_kernel, dim3((O + 255) / 256), dim3(256), 0, st, (int64_t)O, nslice, partial, scale, out);
}
void mip360_launch_head_backward(hipStream_t st, int64_t rows, const float* density, const float* g_density, const float* rgb,
                                 const float* g_rgb, float pad, void* d_raw, int ld_raw, int raw_col, int raw_zero_to,
                                 void* d_pre) {
  hipLaunchKernelGGL(head_backward_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, rows, density, g_density, rgb,
                     g_rgb, pad, (__bf16*)d_raw, ld_raw, raw_col, raw_zero_to, (__bf16*)d_pre);
}
void mip360_launch_sumsq(hipStream_t st, int64_t n, const float* g, float* partial, int nblocks) {
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nblocks), dim3(1024), 0, st, n, g, partial);
}
void mip360_launch_clip_mult(hipStream_t st, int n_partial, const float* partial, float max_norm, float* out) {
  hipLaunchKernelGGL(clip_mult_kernel, dim3(1), dim3(256), 0, st, n_partial, partial, max_norm, out);
}
void mip360_launch_adam(hipStream_t st, int64_t n, float* p, const float* g, float* m, float* v, const float* gmult, float lr,
                        float b1, float b2, float eps, float bc1, float bc2) {
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, p, g, m, v, gmult, lr, b1, b2, eps, bc1, bc2);
}
// the same for up to 16 tensors in one launch: blockIdx.y = tensor, blocks beyond a tensor's elements leave at once
struct PackBatch { mip360_pack_desc d[MIP360_PACK_BATCH_MAX]; };
__global__ void pack_weight_batch_kernel(const PackBatch b) {
  const mip360_pack_desc& d = b.d[blockIdx.y];
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)d.n_in * d.n_out) return;
  const int i = (int)(e / d.n_out), o = (int)(e - (int64_t)i * d.n_out);
  const __bf16 v = (__bf16)d.kernel[e];
  if (d.fwd_bf16) ((__bf16*)d.fwd_bf16)[(size_t)o * d.ld_fwd + i] = v;
  if (d.bwd_bf16) ((__bf16*)d.bwd_bf16)[(size_t)i * d.ld_bwd + o] = v;
  if (d.fwd_fm) ((__bf16*)d.fwd_fm)[fm_elem(o, i, d.ld_fwd_fm)] = v;
  if (d.bwd_fm && i < d.bwd_rows) ((__bf16*)d.bwd_fm)[fm_elem(i, d.bwd_col0 + o, d.ld_bwd_fm)] = v;
}
void mip360_launch_pack_weight_batch(hipStream_t st, int n, const mip360_pack_desc* descs) {
  PackBatch b{};
  int64_t most = 0;
  for (int t = 0; t < n; ++t) {
    b.d[t] = descs[t];
    const int64_t e = (int64_t)descs[t].n_in * descs[t].n_out;
    most = e > most ? e : most;
  }
  hipLaunchKernelGGL(pack_weight_batch_kernel, dim3((unsigned)((most + 255) / 256), (unsigned)n), dim3(256), 0, st, b);
}
void mip360_launch_pack_weight(hipStream_t st, int n_in, int n_out, const float* k, void* fwd, int ld_fwd, void* bwd, int ld_bwd,
                               void* fwd_fm, int ld_fwd_fm, void* bwd_fm, int ld_bwd_fm, int bwd_rows, int bwd_col0) {
  const int64_t n = (int64_t)n_in * n_out;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n_in, n_out, k, (__bf16*)fwd, ld_fwd,
                     (__bf16*)bwd, ld_bwd, (__bf16*)fwd_fm, ld_fwd_fm, (__bf16*)bwd_fm, ld_bwd_fm, bwd_rows, bwd_col0);
}
