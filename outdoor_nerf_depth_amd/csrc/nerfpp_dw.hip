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

constexpr JobTable build_jobs() {
  JobTable jt{};
  for (int net = 0; net < N_NET; ++net) {
    jt.count[net] = 0;
    for (int s = 0; s < FS_COUNT; ++s) {
      const int O = gw_O(s);
      for (int o0 = 0; o0 < O; o0 += 128) {
        const int n_o = O - o0 < 128 ? O - o0 : 128;
        if (s == FS_L0) add_seg_jobs(jt, net, s, o0, n_o, T_X, kpew(net), 0, true);
        else if (s == FS_L5) {
          add_seg_jobs(jt, net, s, o0, n_o, T_X, kpew(net), 0, true);
          add_seg_jobs(jt, net, s, o0, n_o, T_H0 + 4, 256, kpew(net), false);
        } else if (s < 8) add_seg_jobs(jt, net, s, o0, n_o, T_H0 + s - 1, 256, 0, true);
        else if (s == FS_REMAP || s == FS_SIG) add_seg_jobs(jt, net, s, o0, n_o, T_H0 + 7, 256, 0, true);
        else if (s == FS_RGB0) {
          add_seg_jobs(jt, net, s, o0, n_o, T_R, 256, 0, true);
          add_seg_jobs(jt, net, s, o0, n_o, T_DIRX, DIRW, 256, false);
        } else add_seg_jobs(jt, net, s, o0, n_o, T_G, 128, 0, true);
      }
    }
  }
  return jt;
}

constexpr JobTable H_JOBS = build_jobs();
static_assert(H_JOBS.count[0] <= MAX_DW_JOBS && H_JOBS.count[1] <= MAX_DW_JOBS, "job table overflow");
__constant__ JobTable c_jobs = build_jobs();

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

}  // namespace nerfpp

using namespace nerfpp;

int dw_jobs_total() { return H_JOBS.count[0] + H_JOBS.count[1]; }

void launch_dw(hipStream_t st, int P, const DwArgs& a) {
  const int njobs = dw_jobs_total();
  dim3 grid(njobs * a.ksplit), block(256);
  if (P == 1) hipLaunchKernelGGL(dw_kernel<1>, grid, block, 0, st, a, H_JOBS.count[0]);
  else hipLaunchKernelGGL(dw_kernel<2>, grid, block, 0, st, a, H_JOBS.count[0]);
}
