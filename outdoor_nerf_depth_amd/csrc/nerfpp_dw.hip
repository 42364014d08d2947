// Weight-gradient GEMMs:  GW[stage] = dZ^T [O x rows] * IN [rows x I]  (+ bias = column sums of dZ).
// The contraction runs over the SAMPLE axis (rows = n_rays*S, 65k..1.5M), the output is at most
// 256x352, so this is a split-K problem: every workgroup owns one <=128x128 output tile of one
// layer over one slice of the rows and writes its partial tile to that slice's slab; the slabs are
// summed (in a fixed order, so the result is deterministic) by unpack_grads_kernel, which also maps
// the internal feature order back to the reference's parameter layout.
//
// v1 data path: both operands are row-major [rows][ld] bf16 in HBM (written by the fused MLP
// kernels) and MFMA wants the sample axis on the k-slots, so tiles are transposed while they are
// staged into LDS (2-byte LDS writes) and read back as 16-byte fragments.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nerfpp_common.h"
#include "nerfpp_kernels.h"

namespace nerfpp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct JobTable {
  DwJob jobs[N_NET][MAX_DW_JOBS];
  int count[N_NET];
};

constexpr void add_seg_jobs(JobTable& jt, int net, int s, int o0, int n_o, int b_tensor, int segw, int icol,
                            bool first_seg) {
  const int I = gw_I(net, s);
  for (int i0 = 0; i0 < segw; i0 += 128) {
    DwJob j{};
    j.a_tensor = (int16_t)gw_dz_tensor(s);
    j.b_tensor = (int16_t)b_tensor;
    j.o0 = (int16_t)o0;
    j.i0 = (int16_t)i0;
    j.n_o = (int16_t)n_o;
    j.n_i = (int16_t)(segw - i0 < 128 ? segw - i0 : 128);
    j.gw_off = gw_off(net, s) + o0 * I + icol + i0;
    j.gw_ld = (int16_t)I;
    j.gb_off = (int16_t)((first_seg && i0 == 0) ? gb_off(s) + o0 : -1);
    jt.jobs[net][jt.count[net]++] = j;
  }
}

// The 256x256 GEMMs (both operands 256 columns wide: L1-4, the hidden part of L5, L6, L7, remap =
// 87 % of the weight-gradient FLOPs) go to dw256_kernel as ONE full-size tile each; the narrow ones
// (PE inputs, sigma, colour head) stay on the generic 128x128 kernel.
constexpr bool is_fast_stage(int s) { return (s >= 1 && s <= 7) || s == FS_REMAP; }

constexpr JobTable build_jobs(bool fast) {
  JobTable jt{};
  for (int net = 0; net < N_NET; ++net) {
    jt.count[net] = 0;
    for (int s = 0; s < FS_COUNT; ++s) {
      const int O = gw_O(s);
      if (fast) {
        if (!is_fast_stage(s)) continue;
        DwJob j{};
        j.a_tensor = (int16_t)gw_dz_tensor(s);
        j.b_tensor = (int16_t)(s == FS_REMAP ? T_H0 + 7 : T_H0 + s - 1);
        j.o0 = 0; j.i0 = 0; j.n_o = 256; j.n_i = 256;
        j.gw_off = gw_off(net, s) + (s == FS_L5 ? kpew(net) : 0);
        j.gw_ld = (int16_t)gw_I(net, s);
        j.gb_off = (int16_t)(s == FS_L5 ? -1 : gb_off(s));      // L5's bias comes with its X segment
        jt.jobs[net][jt.count[net]++] = j;
        continue;
      }
      for (int o0 = 0; o0 < O; o0 += 128) {
        const int n_o = O - o0 < 128 ? O - o0 : 128;
        if (s == FS_L0) add_seg_jobs(jt, net, s, o0, n_o, T_X, kpew(net), 0, true);
        else if (s == FS_L5) add_seg_jobs(jt, net, s, o0, n_o, T_X, kpew(net), 0, true);
        else if (is_fast_stage(s)) continue;
        else if (s == FS_SIG) add_seg_jobs(jt, net, s, o0, n_o, T_H0 + 7, 256, 0, true);
        else if (s == FS_RGB0) {
          add_seg_jobs(jt, net, s, o0, n_o, T_R, 256, 0, true);
          add_seg_jobs(jt, net, s, o0, n_o, T_DIRX, DIRW, 256, false);
        } else add_seg_jobs(jt, net, s, o0, n_o, T_G, 128, 0, true);
      }
    }
  }
  return jt;
}

constexpr JobTable H_JOBS = build_jobs(false);
constexpr JobTable H_FAST = build_jobs(true);
static_assert(H_JOBS.count[0] <= MAX_DW_JOBS && H_JOBS.count[1] <= MAX_DW_JOBS, "job table overflow");
__constant__ JobTable c_jobs = build_jobs(false);
__constant__ JobTable c_fast = build_jobs(true);

constexpr int KB = 32;             // samples per staged chunk
constexpr int LDT = KB + 8;        // transposed-tile row stride (elements): 80 B, keeps 16-B alignment

template <int P>
__device__ __forceinline__ void stage_transposed(__bf16* lds_t, const __bf16* g, size_t plane, int ld, int col0,
                                                 int n_cols, int64_t r0, int64_t r_end, int tid) {
  const int row = tid >> 3, cg = tid & 7;                 // 32 rows x 8 groups of 16 columns
  const bool ok = (r0 + row) < r_end && cg * 16 < n_cols;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
    if (ok) {
      const uint4* src = (const uint4*)(g + p * plane + (size_t)(r0 + row) * ld + col0 + cg * 16);
      v0 = src[0];
      v1 = src[1];
    }
    const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned short* dst = (unsigned short*)(lds_t + p * 128 * LDT) + (cg * 16) * LDT + row;
#pragma unroll
    for (int e = 0; e < 16; ++e) dst[e * LDT] = (unsigned short)(w[e >> 1] >> (16 * (e & 1)));
  }
}

template <int P>
__global__ __launch_bounds__(256) void dw_kernel(DwArgs a, int njobs0) {
  __shared__ __attribute__((aligned(16))) __bf16 s_a[P * 128 * LDT];
  __shared__ __attribute__((aligned(16))) __bf16 s_b[P * 128 * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, li = lane & 31;
  const int job_id = blockIdx.x / a.ksplit, split = blockIdx.x - job_id * a.ksplit;
  const int net = job_id < njobs0 ? 0 : 1;
  const DwJob job = c_jobs.jobs[net][net == 0 ? job_id : job_id - njobs0];
  const int lda = tensor_ld(net, job.a_tensor), ldb = tensor_ld(net, job.b_tensor);
  const __bf16* ga = a.ws[net].t[job.a_tensor];
  const __bf16* gb = a.ws[net].t[job.b_tensor];
  const size_t plane_a = (size_t)a.rows_padded * lda, plane_b = (size_t)a.rows_padded * ldb;

  int64_t rps = (a.rows + a.ksplit - 1) / a.ksplit;
  rps = (rps + KB - 1) / KB * KB;
  const int64_t r_begin = split * rps;
  const int64_t r_end = r_begin + rps < a.rows ? r_begin + rps : a.rows;
  // (an empty slice still writes its zero partial tile: unpack sums every slab)

  const int wm = wave >> 1, wn = wave & 1;
  const bool do_bias = job.gb_off >= 0 && wn == 0;
  f32x16 acc[2][2], accb[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[x][0][r] = 0.f; acc[x][1][r] = 0.f; accb[x][r] = 0.f; }
  }
  bf16x8 ones;
#pragma unroll
  for (int t = 0; t < 8; ++t) ones[t] = (__bf16)1.f;

  for (int64_t r0 = r_begin; r0 < r_end; r0 += KB) {
    stage_transposed<P>(s_a, ga, plane_a, lda, job.o0, job.n_o, r0, r_end, tid);
    stage_transposed<P>(s_b, gb, plane_b, ldb, job.i0, job.n_i, r0, r_end, tid);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KB / 16; ++kk) {
      bf16x8 fa[2][P], fb[2][P];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int p = 0; p < P; ++p) {
          fa[x][p] = *(const bf16x8*)(s_a + p * 128 * LDT + (64 * wm + 32 * x + li) * LDT + kk * 16 + 8 * hi);
          fb[x][p] = *(const bf16x8*)(s_b + p * 128 * LDT + (64 * wn + 32 * x + li) * LDT + kk * 16 + 8 * hi);
        }
#pragma unroll
      for (int bo = 0; bo < 2; ++bo) {
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
          acc[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bo][0], fb[bi][0], acc[bo][bi], 0, 0, 0);
          if constexpr (P == 2) {
            acc[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bo][0], fb[bi][1], acc[bo][bi], 0, 0, 0);
            acc[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bo][1], fb[bi][0], acc[bo][bi], 0, 0, 0);
          }
        }
        if (do_bias) {
          accb[bo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bo][0], ones, accb[bo], 0, 0, 0);
          if constexpr (P == 2)
            accb[bo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bo][1], ones, accb[bo], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  float* slab = a.slabs[net] + (size_t)split * gslab_floats(net);
#pragma unroll
  for (int bo = 0; bo < 2; ++bo) {
    const int ob = 64 * wm + 32 * bo;
    if (ob >= job.n_o) continue;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
      const int ib = 64 * wn + 32 * bi;
      if (ib >= job.n_i) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = ob + (r & 3) + 8 * (r >> 2) + 4 * hi;
        slab[job.gw_off + o * job.gw_ld + ib + li] = acc[bo][bi][r];
      }
    }
    if (do_bias && li == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = ob + (r & 3) + 8 * (r >> 2) + 4 * hi;
        slab[gw_floats(net) + job.gb_off + o] = accb[bo][r];
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// dw256_kernel: one 256x256 output (a whole layer) per workgroup over one row slice.
//   * 8 waves, wave (wo = w>>2, wi = w&3) owns out-blocks [4wo, 4wo+4) x in-blocks [2wi, 2wi+2):
//     8 accumulators of 32x32 = 128 VGPRs
//   * operands stay row-major [rows][256] bf16 in HBM and are DMA'd (global_load_lds_dwordx4) into
//     LDS unchanged: a 1 KiB wave-instruction = 2 rows; every 1 KiB segment is followed by 64 B of
//     padding so that the 4 rows one transposed read touches fall into 4 disjoint bank windows
//   * MFMA fragments (sample axis on the k-slots) come from ds_read_b64_tr_b16: a 16-lane group
//     reads a [4 samples x 16 features] block and receives it transposed (lane = feature)
//   * double-buffered 32-row chunks, one barrier per chunk
// Rows beyond `rows` up to the next multiple of 32 are zero in every saved tensor (the MLP
// kernels zero-fill their tile tails), so no masking is needed here.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int SEG = FRAG_BYTES + 64;
constexpr int OPER_BYTES = 16 * SEG;          // 32 rows x 512 B + padding

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ bf16x8 tr_frag(const char* p) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p + 512));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ float bf16_sum8(const bf16x8& v) {
  const uint4 w = *(const uint4*)&v;
  const uint32_t u[4] = {w.x, w.y, w.z, w.w};
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) s += __uint_as_float(u[k] << 16) + __uint_as_float(u[k] & 0xffff0000u);
  return s;
}

extern __shared__ __attribute__((aligned(16))) char dw_smem[];

template <int P>
__global__ __launch_bounds__(512) void dw256_kernel(DwArgs a, int njobs0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;
  const int job_id = blockIdx.x / a.ksplit, split = blockIdx.x - job_id * a.ksplit;
  const int net = job_id < njobs0 ? 0 : 1;
  const DwJob job = c_fast.jobs[net][net == 0 ? job_id : job_id - njobs0];
  const char* ga = (const char*)a.ws[net].t[job.a_tensor];
  const char* gb = (const char*)a.ws[net].t[job.b_tensor];
  const size_t plane = (size_t)a.rows_padded * 512;            // bytes per precision plane (ld = 256)

  const int64_t rows32 = (a.rows + 31) / 32 * 32;
  int64_t rps = (rows32 + a.ksplit - 1) / a.ksplit;
  rps = (rps + 31) / 32 * 32;
  const int64_t r_begin = split * rps;
  const int64_t r_end = r_begin + rps < rows32 ? r_begin + rps : rows32;
  const int nchunk = r_end > r_begin ? (int)((r_end - r_begin) / 32) : 0;

  const int wo = wave >> 2, wi = wave & 3;
  const bool do_bias = job.gb_off >= 0 && wi == 0;
  f32x16 acc[4][2];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[x][0][r] = 0.f; acc[x][1][r] = 0.f; }
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  // DMA: 2 operands x P planes x 16 segments per chunk, 4P per wave
  auto issue = [&](int c) {
    if (c >= nchunk) return;
    const int64_t r0 = r_begin + (int64_t)c * 32;
    char* buf = dw_smem + (c & 1) * (2 * P * OPER_BYTES);
#pragma unroll
    for (int x = 0; x < 4 * P; ++x) {
      const int id = x * 8 + wave;                 // 0 .. 32P-1
      const int op = id / (16 * P), rem = id - op * 16 * P, pl = rem >> 4, seg = rem & 15;
      const char* src = (op == 0 ? ga : gb) + pl * plane + (size_t)(r0 + 2 * seg + (lane >> 5)) * 512 + (lane & 31) * 16;
      glds16(src, buf + (op * P + pl) * OPER_BYTES + seg * SEG);
    }
  };
  // per-lane byte offset inside an operand plane for the transposed reads (see header comment)
  const int g = lane >> 4, a16 = lane & 15;
  const int lane_off = (4 * (g >> 1) + (a16 >> 2)) * SEG + (16 * (g & 1) + 4 * (a16 & 3)) * 2;

  issue(0);
  for (int c = 0; c < nchunk; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    issue(c + 1);
    const char* buf = dw_smem + (c & 1) * (2 * P * OPER_BYTES) + lane_off;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4][P], fb[2][P];
#pragma unroll
      for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int x = 0; x < 4; ++x) fa[x][p] = tr_frag(buf + p * OPER_BYTES + kk * 8 * SEG + (4 * wo + x) * 64);
#pragma unroll
        for (int x = 0; x < 2; ++x) fb[x][p] = tr_frag(buf + (P + p) * OPER_BYTES + kk * 8 * SEG + (2 * wi + x) * 64);
      }
#pragma unroll
      for (int bo = 0; bo < 4; ++bo) {
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
          acc[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bo][0], fb[bi][0], acc[bo][bi], 0, 0, 0);
          if constexpr (P == 2) {
            acc[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bo][0], fb[bi][1], acc[bo][bi], 0, 0, 0);
            acc[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bo][1], fb[bi][0], acc[bo][bi], 0, 0, 0);
          }
        }
        if (do_bias) {
          bsum[bo] += bf16_sum8(fa[bo][0]);
          if constexpr (P == 2) bsum[bo] += bf16_sum8(fa[bo][1]);
        }
      }
    }
  }

  float* slab = a.slabs[net] + (size_t)split * gslab_floats(net);
#pragma unroll
  for (int bo = 0; bo < 4; ++bo) {
    const int ob = 32 * (4 * wo + bo);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
      const int ib = 32 * (2 * wi + bi);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = ob + (r & 3) + 8 * (r >> 2) + 4 * hi;
        slab[job.gw_off + o * job.gw_ld + ib + li] = acc[bo][bi][r];
      }
    }
    if (do_bias) {                                   // lane (li, hi) holds feature ob + li, half of the samples
      const float tot = bsum[bo] + __shfl_xor(bsum[bo], 32, 64);
      if (hi == 0) slab[gw_floats(net) + job.gb_off + ob + li] = tot;
    }
  }
}

}  // namespace nerfpp

using namespace nerfpp;

int dw_jobs_total() { return H_JOBS.count[0] + H_JOBS.count[1]; }

void launch_dw(hipStream_t st, int P, const DwArgs& a) {
  {
    const int njobs = H_FAST.count[0] + H_FAST.count[1];
    dim3 grid(njobs * a.ksplit), block(512);
    const size_t lds = 2 * 2 * P * OPER_BYTES;
    if (P == 1) hipLaunchKernelGGL(dw256_kernel<1>, grid, block, lds, st, a, H_FAST.count[0]);
    else hipLaunchKernelGGL(dw256_kernel<2>, grid, block, lds, st, a, H_FAST.count[0]);
  }
  const int njobs = dw_jobs_total();
  dim3 grid(njobs * a.ksplit), block(256);
  if (P == 1) hipLaunchKernelGGL(dw_kernel<1>, grid, block, 0, st, a, H_JOBS.count[0]);
  else hipLaunchKernelGGL(dw_kernel<2>, grid, block, 0, st, a, H_JOBS.count[0]);
}
