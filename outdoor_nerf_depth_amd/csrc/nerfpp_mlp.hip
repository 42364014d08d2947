// Fused NeRF++ MLP kernels for gfx950 (see nerfpp_common.h for the execution model).
//
//   mlp_fwd_kernel : positional encoding (+ inverted-sphere parametrisation for the background)
//                    -> 8x256 trunk with skip -> sigma / remap / colour heads, activations chained
//                    in registers, weights streamed L2 -> LDS (global_load_lds, double-buffered
//                    16-fragment blocks) -> v_mfma_f32_32x32x16_bf16.
//                    Reference: nerf_network.py:42-60,120-142; ddp_model.py:16-45,86-94,107-120.
//   mlp_bwd_kernel : the dX chain of the same network (closed-form backward of the above; the
//                    reference relies on autograd), again chained in registers; writes every dZ
//                    for the weight-gradient GEMMs (nerfpp_dw.hip).
//
// Precision P: 1 = single-pass bf16 operands / f32 accumulate ("speed" mode);
//              2 = split-bf16 (x = hi + lo, 3 MFMA passes hi*hi + hi*lo + lo*hi), which holds
//                  ~1e-5 relative error against the float32 reference ("parity" mode).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "nerfpp_common.h"
#include "nerfpp_kernels.h"

namespace nerfpp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int P> struct Frag { bf16x8 v[P]; };

extern __shared__ __attribute__((aligned(16))) char smem[];

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// LDS-DMA through inline asm (invisible to hipcc's waitcnt pass; completion is tracked by the counted
// s_waitcnt of the ring pipe).  M0 carries the wave-uniform absolute LDS destination.
__device__ __forceinline__ void glds16_asm(const void* g, uint32_t lds_abs) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_abs);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}

// ---- weight stream pipe: blocks of BLK_FRAGS*P KiB streamed L2 -> LDS ----------------------------
// RING = false (training kernels): double buffer, builtin DMA, vmcnt(0) + __syncthreads per block
//        (activation stores share vmcnt with the DMA and may retire out of order, so only a full
//        drain is safe there).
// RING = true  (inference forward: no stores in flight): 4-deep ring, inline-asm DMA, COUNTED vmcnt
//        (all outstanding VMEM ops are same-type loads, in order) and a raw s_barrier, so two blocks
//        stay in flight while one is consumed.
template <int P, int NW, bool RING>
struct WeightPipe {
  static constexpr int BLK_BYTES = BLK_FRAGS * P * FRAG_BYTES;
  static constexpr int NBUF = RING ? 4 : 2;
  static constexpr int PER_BLK = BLK_FRAGS * P / NW;          // DMA wave-instructions per wave per block
  const char* g;
  int nblk, cur, wave, lane;
  uint32_t lds_base;
  __device__ __forceinline__ void init(const void* stream, int nblk_, int wave_, int lane_) {
    g = (const char*)stream; nblk = nblk_; cur = 0; wave = wave_; lane = lane_;
    lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) issue(b);
  }
  __device__ __forceinline__ void issue(int blk) {
    if (blk < nblk) {
      const char* src = g + (size_t)blk * BLK_BYTES + lane * 16;
      const int slot = RING ? (blk & 3) : (blk & 1);
#pragma unroll
      for (int f = 0; f < PER_BLK; ++f) {
        const int fi = f * NW + wave;
        if constexpr (RING) glds16_asm(src + fi * FRAG_BYTES, lds_base + slot * BLK_BYTES + fi * FRAG_BYTES);
        else glds16(src + fi * FRAG_BYTES, smem + slot * BLK_BYTES + fi * FRAG_BYTES);
      }
    }
  }
  // make block `cur` readable, start fetching the next free slot, return LDS address of block cur
  __device__ __forceinline__ const char* acquire() {
    if constexpr (RING) {
      const int younger = nblk - 1 - cur < NBUF - 2 ? nblk - 1 - cur : NBUF - 2;
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_BLK) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_BLK) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue(cur + NBUF - 1);
      const char* l = smem + (cur & 3) * BLK_BYTES + lane * 16;
      ++cur;
      return l;
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      issue(cur + 1);
      const char* l = smem + (cur & 1) * BLK_BYTES + lane * 16;
      ++cur;
      return l;
    }
  }
};

template <int P>
__device__ __forceinline__ void mfma_p(f32x16& acc, const char* lfrag, const Frag<P>& b) {
  const bf16x8 a_hi = *(const bf16x8*)(lfrag);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b.v[0], acc, 0, 0, 0);
  if constexpr (P == 2) {
    const bf16x8 a_lo = *(const bf16x8*)(lfrag + FRAG_BYTES);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b.v[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b.v[0], acc, 0, 0, 0);
  }
}

// acc[ob] += W_stage[ob-block, :] * B   for one stage of NKC k-chunks x NOB out-blocks
template <int NOB, int NKC, int P, typename Pipe>
__device__ __forceinline__ void stage_gemm(Pipe& pipe, f32x16 (&acc)[NOB], const Frag<P> (&b)[NKC]) {
  constexpr int KPB = BLK_FRAGS / NOB;            // k-chunks per block
  static_assert(NKC % KPB == 0, "stage must be block aligned");
#pragma unroll
  for (int blk = 0; blk < NKC / KPB; ++blk) {
    const char* l = pipe.acquire();
#pragma unroll
    for (int kl = 0; kl < KPB; ++kl) {
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob)
        mfma_p<P>(acc[ob], l + (kl * NOB + ob) * P * FRAG_BYTES, b[blk * KPB + kl]);
    }
  }
}

template <int P>
__device__ __forceinline__ void set_slot(Frag<P>& f, int t, float v) {
  const __bf16 h = (__bf16)v;
  f.v[0][t] = h;
  if constexpr (P == 2) f.v[1][t] = (__bf16)(v - (float)h);
}
template <int P>
__device__ __forceinline__ Frag<P> zero_frag() {
  Frag<P> f;
#pragma unroll
  for (int p = 0; p < P; ++p)
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
    for (int p = 0; p < P; ++p) *(uint4*)(base + ((c * P + p) * 64 + lane) * 16) = *(const uint4*)&f[c].v[p];
}
template <int N, int P>
__device__ __forceinline__ void unstash_frags(const char* base, int lane, Frag<P> (&f)[N]) {
#pragma unroll
  for (int c = 0; c < N; ++c)
#pragma unroll
    for (int p = 0; p < P; ++p) *(uint4*)&f[c].v[p] = *(const uint4*)(base + ((c * P + p) * 64 + lane) * 16);
}

// ReLU + conversion + sign bits in one pass over the accumulators (bit ob*16 + r <-> acc[ob][r] > 0)
template <int NOB, int P>
__device__ __forceinline__ uint4 acc_to_frags_relu_bits(const f32x16 (&acc)[NOB], Frag<P> (&h)[2 * NOB]) {
  uint32_t m[4] = {0, 0, 0, 0};
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float v = acc[ob][8 * hh + t];
        m[ob >> 1] |= (v > 0.f ? 1u : 0u) << ((ob & 1) * 16 + 8 * hh + t);
        set_slot<P>(h[2 * ob + hh], t, fmaxf(v, 0.f));
      }
  return make_uint4(m[0], m[1], m[2], m[3]);
}

template <int NOB>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NOB], const float* __restrict__ bias, int hi) {
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
    const float4* p = (const float4*)(bias + ob * 32 + hi * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = p[q];
      acc[ob][4 * q] = v.x; acc[ob][4 * q + 1] = v.y; acc[ob][4 * q + 2] = v.z; acc[ob][4 * q + 3] = v.w;
    }
  }
}
// same, but the bias stream has been copied to LDS (inference forward: keeps compiler-tracked global
// loads out of the counted-vmcnt weight ring)
template <int NOB>
__device__ __forceinline__ void init_bias_lds(f32x16 (&acc)[NOB], uint32_t lds_off_bytes, int hi) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __attribute__((address_space(3))) char* base = (__attribute__((address_space(3))) char*)smem;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *(__attribute__((address_space(3))) f32x4*)(base + lds_off_bytes + (ob * 32 + hi * 16 + 4 * q) * 4);
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

// fragments -> row-major [rows][ld] bf16 tensor (hi plane, then lo plane at +plane elements).
// The accumulator layout gives every lane 8-byte pieces of 32 DIFFERENT rows, so the tile is first
// transposed through a per-wave LDS staging area ([32 rows][<=128 cols], row stride 272 B) and then
// written with 16 B per lane, whole 128..256-byte row segments per instruction.
// Rows past the end of the batch (tile tail, < rows_padded) are written as zeros so the
// weight-gradient GEMMs can run over whole 32-row chunks without masking.
constexpr int STAGE_ROW = 272;                      // 256 B of data + 16 B pad (keeps 16-B alignment)
constexpr int STAGE_BYTES = 32 * STAGE_ROW;         // per wave

__device__ __forceinline__ void lds_wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

template <int NCH, int P>
__device__ __forceinline__ void save_frags(char* stage, __bf16* base, size_t plane, int ld, size_t wave_row0,
                                           int lane, bool valid, const Frag<P> (&h)[NCH]) {
  const int j = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int p = 0; p < P; ++p) {
#pragma unroll
    for (int c0 = 0; c0 < NCH; c0 += 8) {
      constexpr int dummy = 0; (void)dummy;
      const int nc = NCH - c0 < 8 ? NCH - c0 : 8;           // chunks in this pass (compile-time after unroll)
      char* w = stage + j * STAGE_ROW + 8 * hi;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c < nc) {
          uint4 bits = *(const uint4*)&h[c0 + c].v[p];
          if (!valid) bits = make_uint4(0, 0, 0, 0);
          *(uint2*)(w + 32 * c) = make_uint2(bits.x, bits.y);
          *(uint2*)(w + 32 * c + 16) = make_uint2(bits.z, bits.w);
        }
      }
      lds_wave_sync();
      const int lpr = 2 * nc;                               // 16-byte pieces per row
      char* g = (char*)(base + p * plane + wave_row0 * ld + c0 * 16);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int idx = it * 64 + lane;
        if (it * 64 < 32 * lpr && idx < 32 * lpr) {
          const int row = idx / lpr, piece = idx - row * lpr;
          const uint4 v = *(const uint4*)(stage + row * STAGE_ROW + piece * 16);
          typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 vv = {v.x, v.y, v.z, v.w};
          __builtin_nontemporal_store(vv, (u32x4*)(g + (size_t)row * ld * 2 + piece * 16));
        }
      }
      lds_wave_sync();
    }
  }
}

// ReLU sign bits of one stage: bit ob*16 + r  <->  acc[ob][r] > 0
template <int NOB>
__device__ __forceinline__ uint4 relu_bits(const f32x16 (&acc)[NOB]) {
  uint32_t m[4] = {0, 0, 0, 0};
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) m[ob >> 1] |= (acc[ob][r] > 0.f ? 1u : 0u) << ((ob & 1) * 16 + r);
  return make_uint4(m[0], m[1], m[2], m[3]);
}

// dH (accumulators) * [forward activation > 0] -> dZ fragments
template <int NOB, int P>
__device__ __forceinline__ void mask_to_frags(const f32x16 (&acc)[NOB], const uint4 bits, Frag<P> (&dz)[2 * NOB]) {
  const uint32_t m[4] = {bits.x, bits.y, bits.z, bits.w};
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const bool on = (m[ob >> 1] >> ((ob & 1) * 16 + 8 * hh + t)) & 1u;
        set_slot<P>(dz[2 * ob + hh], t, on ? acc[ob][8 * hh + t] : 0.f);
      }
}

// ------------------------------------------------------------------------------------------------
// input geometry + encodings for one sample (one lane-half)
// ------------------------------------------------------------------------------------------------
template <int NET>
__device__ __forceinline__ void sample_point(const MlpGeom& gm, size_t row, int S, float (&x)[4], float (&vd)[3],
                                             float* depth_real) {
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

// positional encoding of this lane-half's share (nerfpp_common.h: pe_ref_of_lane_slot)
template <int NET, int P>
__device__ __forceinline__ void encode_point(const float (&x)[4], int hi, Frag<P> (&pe)[kpe(NET)]) {
  constexpr int D = pe_dim(NET), NV = kpe(NET) * 8;
  float v[NV];
#pragma unroll
  for (int kk = 0; kk < 5; ++kk) {
    const float scale = __int_as_float((127 + 5 * hi + kk) << 23);       // 2^(5hi+kk), exact
#pragma unroll
    for (int d = 0; d < D; ++d) {
      float sn, cs;
      sincosf(x[d] * scale, &sn, &cs);
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
  float v[16];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const float scale = __int_as_float((127 + 2 * hi + kk) << 23);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float sn, cs;
      sincosf(vd[d] * scale, &sn, &cs);
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

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int NET, int P, int NW, bool TRAIN>
__global__ __launch_bounds__(NW * 64, (P == 1 ? 2 : 1)) void mlp_fwd_kernel(MlpFwdArgs a) {
  constexpr int KPE = kpe(NET);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5;
  const size_t row_raw = (size_t)blockIdx.x * (NW * 32) + wave * 32 + (lane & 31);
  const bool valid = row_raw < (size_t)a.rows;
  const size_t row = valid ? row_raw : (size_t)a.rows - 1;
  const size_t plane_rows = a.rows_padded;
  const size_t wrow0 = (size_t)blockIdx.x * (NW * 32) + wave * 32;          // this wave's first tile row
  constexpr bool RINGMODE = !TRAIN && P == 1;                               // inference, bf16: counted-vmcnt ring
  constexpr int WBYTES = (RINGMODE ? 4 : 2) * BLK_FRAGS * P * FRAG_BYTES;   // weight buffers
  char* stage = smem + WBYTES + wave * STAGE_BYTES;
  const size_t nblk32 = a.rows_padded / 32;
  uint4* mask_out = a.masks + (wrow0 / 32) * 64 + lane;                     // + stage * nblk32 * 64

  const uint32_t bias_lds = WBYTES + (TRAIN ? NW * STAGE_BYTES : 0) + NW * KPE * P * 1024;   // RINGMODE only
  if constexpr (RINGMODE) {
    for (int i = threadIdx.x; i < FWD_BIAS_FLOATS / 4; i += NW * 64)
      *(float4*)(smem + bias_lds + i * 16) = ((const float4*)a.bias)[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  auto bias_init8 = [&](f32x16 (&acc_)[8], int off) {
    if constexpr (!RINGMODE) init_bias<8>(acc_, a.bias + off, hi); else init_bias_lds<8>(acc_, bias_lds + off * 4, hi);
  };
  auto bias_init4 = [&](f32x16 (&acc_)[4], int off) {
    if constexpr (!RINGMODE) init_bias<4>(acc_, a.bias + off, hi); else init_bias_lds<4>(acc_, bias_lds + off * 4, hi);
  };
  auto bias_init1 = [&](f32x16 (&acc_)[1], int off) {
    if constexpr (!RINGMODE) init_bias<1>(acc_, a.bias + off, hi); else init_bias_lds<1>(acc_, bias_lds + off * 4, hi);
  };

  WeightPipe<P, NW, RINGMODE> pipe;
  pipe.init(a.w_stream, fwd_frags(NET) / BLK_FRAGS, wave, lane);

  float x[4], vd[3], depth_real;
  sample_point<NET>(a.geom, row, a.S, x, vd, &depth_real);
  Frag<P> pe[KPE];
  encode_point<NET, P>(x, hi, pe);
  if (TRAIN) save_frags<KPE, P>(stage, a.ws.t[T_X], plane_rows * kpew(NET), kpew(NET), wrow0, lane, valid, pe);
  char* pe_stash = smem + WBYTES + (TRAIN ? NW * STAGE_BYTES : 0) + wave * (KPE * P * 1024);
  stash_frags<KPE, P>(pe_stash, lane, pe);

  f32x16 acc[8];
  Frag<P> h[16];
  // L0
  bias_init8(acc, fs_bias_off(FS_L0));
  stage_gemm<8, KPE, P>(pipe, acc, pe);
  const uint4 bits = acc_to_frags_relu_bits<8, P>(acc, h);
  if (TRAIN) {
    mask_out[0] = bits;
    save_frags<16, P>(stage, a.ws.t[T_H0], plane_rows * 256, 256, wrow0, lane, valid, h);
  }
  // L1..L4
  for (int l = 1; l <= 4; ++l) {
    bias_init8(acc, fs_bias_off(FS_L0) + l * 256);
    stage_gemm<8, 16, P>(pipe, acc, h);
    const uint4 bits = acc_to_frags_relu_bits<8, P>(acc, h);
    if (TRAIN) {
      mask_out[(size_t)l * nblk32 * 64] = bits;
      save_frags<16, P>(stage, a.ws.t[T_H0 + l], plane_rows * 256, 256, wrow0, lane, valid, h);
    }
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
    stage_gemm<8, KPE + 16, P>(pipe, acc, in5);
    const uint4 bits = acc_to_frags_relu_bits<8, P>(acc, h);
    if (TRAIN) {
      mask_out[(size_t)5 * nblk32 * 64] = bits;
      save_frags<16, P>(stage, a.ws.t[T_H0 + 5], plane_rows * 256, 256, wrow0, lane, valid, h);
    }
  }
  // L6, L7
  for (int l = 6; l <= 7; ++l) {
    bias_init8(acc, fs_bias_off(FS_L0) + l * 256);
    stage_gemm<8, 16, P>(pipe, acc, h);
    const uint4 bits = acc_to_frags_relu_bits<8, P>(acc, h);
    if (TRAIN) {
      mask_out[(size_t)l * nblk32 * 64] = bits;
      save_frags<16, P>(stage, a.ws.t[T_H0 + l], plane_rows * 256, 256, wrow0, lane, valid, h);
    }
  }
  // remap (no activation) and sigma, both from h7                       nerf_network.py:131-136
  Frag<P> rm[16];
  bias_init8(acc, fs_bias_off(FS_REMAP));
  stage_gemm<8, 16, P>(pipe, acc, h);
  acc_to_frags<8, P, ACT_NONE>(acc, rm);
  // (R is not saved: the weight gradients that need it are derived from M = dG^T H7, nerfpp_optim.hip)
  f32x16 acc1[1];
  bias_init1(acc1, fs_bias_off(FS_SIG));
  stage_gemm<1, 16, P>(pipe, acc1, h);
  const float sigma_raw = acc1[0][0];
  // colour head                                                         nerf_network.py:137-138
  Frag<P> g[8];
  {
    Frag<P> in[20];
#pragma unroll
    for (int c = 0; c < 16; ++c) in[c] = rm[c];
    Frag<P> df[2];
    encode_dir<P>(vd, hi, df);
    if (TRAIN) save_frags<2, P>(stage, a.ws.t[T_DIRX], plane_rows * 32, 32, wrow0, lane, valid, df);
    in[16] = df[0]; in[17] = df[1]; in[18] = zero_frag<P>(); in[19] = zero_frag<P>();
    f32x16 acc4[4];
    bias_init4(acc4, fs_bias_off(FS_RGB0));
    stage_gemm<4, 20, P>(pipe, acc4, in);
    const uint4 bits = acc_to_frags_relu_bits<4, P>(acc4, g);
    if (TRAIN) {
      mask_out[(size_t)8 * nblk32 * 64] = bits;
      save_frags<8, P>(stage, a.ws.t[T_G], plane_rows * 128, 128, wrow0, lane, valid, g);
    }
  }
  {
    Frag<P> in[16];
#pragma unroll
    for (int c = 0; c < 8; ++c) in[c] = g[c];
#pragma unroll
    for (int c = 8; c < 16; ++c) in[c] = zero_frag<P>();
    bias_init1(acc1, fs_bias_off(FS_RGB1));
    stage_gemm<1, 16, P>(pipe, acc1, in);
  }
  if (valid && hi == 0) {
    float4 o;
    o.x = 1.f / (1.f + expf(-acc1[0][0]));
    o.y = 1.f / (1.f + expf(-acc1[0][1]));
    o.z = 1.f / (1.f + expf(-acc1[0][2]));
    o.w = sigma_raw;
    ((float4*)a.out_raw)[row] = o;
    if (NET == 1) a.depth_real[row] = depth_real;
  }
}

// ------------------------------------------------------------------------------------------------
// backward (dX chain).  d_out[row] = (d rgb_pre-sigmoid[3], d sigma_raw)
// ------------------------------------------------------------------------------------------------
template <int NET, int P, int NW>
__global__ __launch_bounds__(NW * 64, (P == 1 ? 2 : 1)) void mlp_bwd_kernel(MlpBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5;
  const size_t row_raw = (size_t)blockIdx.x * (NW * 32) + wave * 32 + (lane & 31);
  const bool valid = row_raw < (size_t)a.rows;
  const size_t row = valid ? row_raw : (size_t)a.rows - 1;
  const size_t plane_rows = a.rows_padded;
  const size_t wrow0 = (size_t)blockIdx.x * (NW * 32) + wave * 32;
  char* stage = smem + 2 * BLK_FRAGS * P * FRAG_BYTES + wave * STAGE_BYTES;
  const size_t nblk32 = a.rows_padded / 32;
  const uint4* mask_in = a.masks + (wrow0 / 32) * 64 + lane;

  WeightPipe<P, NW, false> pipe;
  pipe.init(a.w_stream, BWD_FRAGS / BLK_FRAGS, wave, lane);

  float4 d = ((const float4*)a.d_out)[row];
  if (!valid) d = make_float4(0.f, 0.f, 0.f, 0.f);

  // B0: dG = Wrgb1^T dP, masked by G > 0
  Frag<P> dg[8];
  {
    Frag<P> in[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) in[c] = zero_frag<P>();
    if (hi == 0) { set_slot<P>(in[0], 0, d.x); set_slot<P>(in[0], 1, d.y); set_slot<P>(in[0], 2, d.z); }
    {
      Frag<P> dp[2] = {in[0], in[1]};
      save_frags<2, P>(stage, a.ws.t[T_DP], plane_rows * 32, 32, wrow0, lane, valid, dp);
    }
    f32x16 acc4[4];
    init_zero<4>(acc4);
    stage_gemm<4, 4, P>(pipe, acc4, in);
    mask_to_frags<4, P>(acc4, mask_in[(size_t)8 * nblk32 * 64], dg);
    save_frags<8, P>(stage, a.ws.t[T_DG], plane_rows * 128, 128, wrow0, lane, valid, dg);
  }
  f32x16 acc[8];
  Frag<P> dz[16];
  // B1: dR = Wrgb0[:, :256]^T dG  (no activation on the remap layer)
  init_zero<8>(acc);
  stage_gemm<8, 8, P>(pipe, acc, dg);
  acc_to_frags<8, P, ACT_NONE>(acc, dz);
  // (dR is not saved either, same reason)
  // B2: dH7 = Wremap^T dR + wsigma * dsigma, masked by H7 > 0
  {
    Frag<P> in[18];
#pragma unroll
    for (int c = 0; c < 16; ++c) in[c] = dz[c];
    in[16] = zero_frag<P>(); in[17] = zero_frag<P>();
    if (hi == 0) set_slot<P>(in[16], 0, d.w);
    {
      Frag<P> ds[2] = {in[16], in[17]};
      save_frags<2, P>(stage, a.ws.t[T_DS], plane_rows * 32, 32, wrow0, lane, valid, ds);
    }
    init_zero<8>(acc);
    stage_gemm<8, 18, P>(pipe, acc, in);
    mask_to_frags<8, P>(acc, mask_in[(size_t)7 * nblk32 * 64], dz);
    save_frags<16, P>(stage, a.ws.t[T_DZ0 + 7], plane_rows * 256, 256, wrow0, lane, valid, dz);
  }
  // B3..B9: dH_{l-1} = W_l^T dZ_l, l = 7..1
  for (int l = 7; l >= 1; --l) {
    init_zero<8>(acc);
    stage_gemm<8, 16, P>(pipe, acc, dz);
    mask_to_frags<8, P>(acc, mask_in[(size_t)(l - 1) * nblk32 * 64], dz);
    save_frags<16, P>(stage, a.ws.t[T_DZ0 + l - 1], plane_rows * 256, 256, wrow0, lane, valid, dz);
  }
}

}  // namespace nerfpp

using namespace nerfpp;

#ifndef NERFPP_WAVES_P1
#define NERFPP_WAVES_P1 8
#endif
#define MLP_WAVES(P) ((P) == 1 ? NERFPP_WAVES_P1 : 4)

template <int NET, int P, bool TRAIN>
static void launch_fwd_t(hipStream_t st, const MlpFwdArgs& a) {
  constexpr int NW = MLP_WAVES(P);
  const int tile = NW * 32;
  const int grid = (int)((a.rows + tile - 1) / tile);
  constexpr bool RINGMODE = !TRAIN && P == 1;
  const size_t lds = (RINGMODE ? 4 : 2) * BLK_FRAGS * P * FRAG_BYTES + (TRAIN ? NW * STAGE_BYTES : 0) + NW * kpe(NET) * P * 1024 +
                     (RINGMODE ? FWD_BIAS_FLOATS * 4 : 0);
  hipLaunchKernelGGL((mlp_fwd_kernel<NET, P, NW, TRAIN>), dim3(grid), dim3(NW * 64), lds, st, a);
}
template <int NET, int P>
static void launch_bwd_t(hipStream_t st, const MlpBwdArgs& a) {
  constexpr int NW = MLP_WAVES(P);
  const int tile = NW * 32;
  const int grid = (int)((a.rows + tile - 1) / tile);
  const size_t lds = 2 * BLK_FRAGS * P * FRAG_BYTES + NW * STAGE_BYTES;
  hipLaunchKernelGGL((mlp_bwd_kernel<NET, P, NW>), dim3(grid), dim3(NW * 64), lds, st, a);
}

void launch_mlp_fwd(hipStream_t st, int net, int P, bool train, const MlpFwdArgs& a) {
  if (net == 0) {
    if (P == 1) { if (train) launch_fwd_t<0, 1, true>(st, a); else launch_fwd_t<0, 1, false>(st, a); }
    else        { if (train) launch_fwd_t<0, 2, true>(st, a); else launch_fwd_t<0, 2, false>(st, a); }
  } else {
    if (P == 1) { if (train) launch_fwd_t<1, 1, true>(st, a); else launch_fwd_t<1, 1, false>(st, a); }
    else        { if (train) launch_fwd_t<1, 2, true>(st, a); else launch_fwd_t<1, 2, false>(st, a); }
  }
}
void launch_mlp_bwd(hipStream_t st, int net, int P, const MlpBwdArgs& a) {
  if (net == 0) { if (P == 1) launch_bwd_t<0, 1>(st, a); else launch_bwd_t<0, 2>(st, a); }
  else          { if (P == 1) launch_bwd_t<1, 1>(st, a); else launch_bwd_t<1, 2>(st, a); }
}
int mlp_tile_rows(int P) { return MLP_WAVES(P) * 32; }
