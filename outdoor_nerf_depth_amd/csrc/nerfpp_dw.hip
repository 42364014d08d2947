// Weight-gradient GEMMs:  GW[stage] = dZ^T [O x rows] * IN [rows x I]  (+ bias = column sums of dZ).
//
// The contraction runs over the SAMPLE axis (rows = n_rays*S, 65k..1.5M) and the output is at most
// 256x256 per job, so this is a split-K problem: a workgroup owns the whole output of one job over one
// slice of the rows and writes its partial result to that slice's slab; unpack_grads_kernel sums the
// slabs in a fixed order (deterministic) and maps the internal feature order back to the reference's
// parameter layout.  Every operand tensor is read exactly once per job.
//
// Data path (dw_kernel):
//   * operands stay row-major [rows][ld] bf16 in HBM (as the fused MLP kernels wrote them) and are
//     DMA'd (global_load_lds_dwordx4) into LDS as [32 rows][512 B] images: a 1 KiB wave-instruction
//     covers 2 rows; each 1 KiB segment is followed by 64 B of padding so that the 4 rows one
//     transposed read touches fall into 4 disjoint bank windows.  Tensors narrower than 256 columns
//     fill only the left part of the image (inactive DMA lanes).
//   * MFMA wants the sample axis on the k-slots: fragments come from ds_read_b64_tr_b16 -- a
//     16-lane group reads a [4 samples x 16 features] block and receives it transposed
//     (lane = feature, 4 samples per lane).  Both operands use the same sample->slot map, so the
//     contraction is exact whatever that map is.
//   * 8 waves; full 256x256 jobs: wave (wo = w>>2, wi = w&3) owns out-blocks [4wo, 4wo+4) x in-blocks
//     [2wi, 2wi+2), 8 accumulators of 32x32 (128 VGPRs); narrow jobs deal their blocks round-robin.
//     32-row chunks through a 4-deep LDS ring (inline-asm DMA, counted vmcnt), one raw barrier per chunk.
//   * the number of row slices is per job (dw_plan, nerfpp_common.h): every launch fills the 256 CUs
//     once with workgroups that move about the same number of bytes.
//   * bias gradients: VALU column sums of the A fragments, split over the waves that share them.
// Rows beyond `rows` up to the next multiple of 32 are zero in every saved tensor (the MLP kernels
// zero-fill their tile tails), so no masking is needed.
#include "probe_env.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "nerfpp_common.h"
#include "nerfpp_kernels.h"

namespace nerfpp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// FULL jobs (256 x 256, every wave has all 8 of its blocks) get an unguarded kernel instantiation
constexpr JobTable build_jobs(bool full) {
  const JobTable all = build_all_jobs();
  JobTable jt{};
  for (int net = 0; net < N_NET; ++net) {
    jt.count[net] = 0;
    for (int k = 0; k < all.count[net]; ++k) {
      if (dw_job_is_full(all.jobs[net][k]) == full) jt.jobs[net][jt.count[net]++] = all.jobs[net][k];
    }
  }
  return jt;
}
__constant__ JobTable c_full = build_jobs(true);
__constant__ JobTable c_narrow = build_jobs(false);

constexpr int SEG = FRAG_BYTES + 64;
constexpr int OPER_BYTES = 16 * SEG;          // 32 rows x 512 B + padding

// LDS-DMA through inline asm: hipcc's waitcnt pass must not see it, or it drains vmcnt to 0 before
// every LDS read of the ring (it cannot prove the transposed reads do not alias the in-flight
// destination).  Completion is tracked by the counted s_waitcnt in the main loop instead.
// M0 carries the wave-uniform LDS destination; it is saved/restored inside the statement.
__device__ __forceinline__ void glds16(const void* g, uint32_t lds_abs) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_abs);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
// LDS reads address the dynamic LDS array through an address_space(3) pointer + 32-bit byte offset
// (a flat pointer would drag a flat->LDS null check into divergent code and trips a backend bug).
extern __shared__ __attribute__((aligned(16))) char dw_smem[];
typedef uint32_t lds_addr;
// row2 = byte distance between the two rows of a segment (512 in the full-width image)
__device__ __forceinline__ bf16x8 tr_frag(lds_addr off, uint32_t row2 = 512) {
  __attribute__((address_space(3))) char* base = (__attribute__((address_space(3))) char*)dw_smem;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + off));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + off + row2));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// Narrow jobs size their LDS images by the operand width: a segment holds 2 rows of W bytes (W = the
// job's column bytes rounded up to 64 / 128 / 256 / 512) + 64 B of padding, so that consecutive segments
// sit 16 banks apart like in the full-width image; 16 segments per 32-row chunk.  Smaller chunks leave
// room for a deeper ring in the same LDS (the narrow jobs are bound by bytes in flight).
struct NarrowGeom {
  uint32_t w[2];        // image row bytes of A, B
  uint32_t segw[2];     // 2 * w + 64
  uint32_t img[2];      // 16 * segw
  uint32_t chunk;       // P * (img[0] + img[1])
  int nbuf;             // ring depth: min(8, LDS bytes / chunk)
};
__device__ __forceinline__ uint32_t pow2_width(int bytes) { return bytes <= 64 ? 64u : bytes <= 128 ? 128u : bytes <= 256 ? 256u : 512u; }
template <int P>
__device__ __forceinline__ NarrowGeom narrow_geom(const DwJob& job, uint32_t lds_bytes) {
  NarrowGeom g;
  g.w[0] = pow2_width(job.n_o * 2);
  g.w[1] = pow2_width(job.n_i * 2);
#pragma unroll
  for (int o = 0; o < 2; ++o) { g.segw[o] = 2 * g.w[o] + 64; g.img[o] = 16 * g.segw[o]; }
  g.chunk = P * (g.img[0] + g.img[1]);
  const int n = (int)(lds_bytes / g.chunk);
  g.nbuf = n > 8 ? 8 : n;
  return g;
}
__device__ __forceinline__ float bf16_sum8(const bf16x8& v) {
  const uint4 w = *(const uint4*)&v;
  const uint32_t u[4] = {w.x, w.y, w.z, w.w};
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) s += __uint_as_float(u[k] << 16) + __uint_as_float(u[k] & 0xffff0000u);
  return s;
}

// one 32-row chunk: 2 k16-steps x (<=4 out-blocks x <=2 in-blocks) MFMAs for this wave
template <int P, bool FULL>
__device__ __forceinline__ void compute_chunk(lds_addr buf, int wo, int wi, int nbo, int nbi, bool do_bias,
                                              f32x16 (&acc)[4][2], float (&bsum)[4]) {
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    bf16x8 fa[4][P], fb[2][P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
#pragma unroll
      for (int x = 0; x < 4; ++x)
        if (FULL || x < nbo) fa[x][p] = tr_frag(buf + p * OPER_BYTES + kk * 8 * SEG + (4 * wo + x) * 64);
#pragma unroll
      for (int x = 0; x < 2; ++x)
        if (FULL || x < nbi) fb[x][p] = tr_frag(buf + (P + p) * OPER_BYTES + kk * 8 * SEG + (2 * wi + x) * 64);
    }
#pragma unroll
    for (int bo = 0; bo < 4; ++bo) {
      if (!FULL && bo >= nbo) continue;
#pragma unroll
      for (int bi = 0; bi < 2; ++bi) {
        if (!FULL && bi >= nbi) continue;
        acc[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bo][0], fb[bi][0], acc[bo][bi], 0, 0, 0);
        if constexpr (P == 2) {
          acc[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bo][0], fb[bi][1], acc[bo][bi], 0, 0, 0);
          acc[bo][bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bo][1], fb[bi][0], acc[bo][bi], 0, 0, 0);
        }
      }
      if (do_bias && bo == wi) {                 // the 4 wi-waves of a wo share the A fragments: split the sums
        bsum[bo] += bf16_sum8(fa[bo][0]);
        if constexpr (P == 2) bsum[bo] += bf16_sum8(fa[bo][1]);
      }
    }
  }
}

// Narrow jobs (fewer than 8x8 blocks): blocks are dealt round-robin to the 8 waves (wave w owns blocks
// w, w+8, ...; block b = (bo = b / n_ib, bi = b % n_ib)), so every wave has work between barriers.
template <int P>
__device__ __forceinline__ void compute_chunk_rr(lds_addr buf_a, lds_addr buf_b, const NarrowGeom& gm, int wave, int n_ib,
                                                 int nblk, bool has_bias, f32x16 (&acc)[8], float (&bsum)[8]) {
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int b = wave + 8 * k;
      if (b >= nblk) break;
      const int bo = b / n_ib, bi = b - bo * n_ib;
      bf16x8 fa[P], fb[P];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        fa[p] = tr_frag(buf_a + p * gm.img[0] + kk * 8 * gm.segw[0] + bo * 64, gm.w[0]);
        fb[p] = tr_frag(buf_b + p * gm.img[1] + kk * 8 * gm.segw[1] + bi * 64, gm.w[1]);
      }
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc[k], 0, 0, 0);
      if constexpr (P == 2) {
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc[k], 0, 0, 0);
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], acc[k], 0, 0, 0);
      }
      if (has_bias && bi == 0) {
        bsum[k] += bf16_sum8(fa[0]);
        if constexpr (P == 2) bsum[k] += bf16_sum8(fa[1]);
      }
    }
  }
}

// workgroup -> (job of this launch, row slice): jobs in table order (net 0 then net 1), k slices each
struct DwSched {
  int wg_end[2 * DW_JOBS];       // exclusive prefix of workgroups per job
  int k[2 * DW_JOBS];
  int njobs, njobs0;
};
template <int P, bool FULL>
__global__ __launch_bounds__(512) void dw_kernel(DwArgs a, DwSched sc, int dbg, uint32_t lds_bytes) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;
  int job_id = 0;
  for (int j = 0; j < sc.njobs - 1; ++j) job_id += (int)blockIdx.x >= sc.wg_end[j];
  const int split = blockIdx.x - (job_id == 0 ? 0 : sc.wg_end[job_id - 1]), ksplit = sc.k[job_id];
  const int njobs0 = sc.njobs0;
  const int net = job_id < njobs0 ? 0 : 1;
  const DwJob job = (FULL ? c_full : c_narrow).jobs[net][net == 0 ? job_id : job_id - njobs0];
  const int rb_a = tensor_ld(net, job.a_tensor) * 2, rb_b = tensor_ld(net, job.b_tensor) * 2;   // row bytes
  const char* ga = (const char*)a.ws[net].t[job.a_tensor];
  const char* gb = (const char*)a.ws[net].t[job.b_tensor];
  const size_t plane_a = (size_t)a.rows_padded * rb_a, plane_b = (size_t)a.rows_padded * rb_b;

  const int64_t rows32 = (a.rows + 31) / 32 * 32;
  int64_t rps = (rows32 + ksplit - 1) / ksplit;
  rps = (rps + 31) / 32 * 32;
  const int64_t r_begin = split * rps;
  const int64_t r_end = r_begin + rps < rows32 ? r_begin + rps : rows32;
  const int nchunk = r_end > r_begin ? (int)((r_end - r_begin) / 32) : 0;

  const int wo = wave >> 2, wi = wave & 3;
  int nbo = job.n_o / 32 - 4 * wo, nbi = job.n_i / 32 - 2 * wi;       // valid blocks of this wave
  nbo = nbo < 0 ? 0 : (nbo > 4 ? 4 : nbo);
  nbi = nbi < 0 ? 0 : (nbi > 2 ? 2 : nbi);
  const bool do_bias = job.gb_off >= 0 && wi < nbo;       // wave wi owns the bias of its out-block 4wo+wi
  f32x16 acc[4][2];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[x][0][r] = 0.f; acc[x][1][r] = 0.f; }
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  // the narrow instantiation deals blocks round-robin instead (its own accumulator set)
  f32x16 acc_rr[8];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_rr[x][r] = 0.f;
  float bsum_rr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int n_ib = job.n_i / 32, nblk = (job.n_o / 32) * n_ib;

  // DMA: 2 operands x P planes x 16 segments per chunk, 4P wave-instructions per wave
  constexpr int DMA_PER_CHUNK = 4 * P;           // wave-instructions per wave per chunk
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)dw_smem;
  // per-lane position inside an operand image for the transposed reads (see header comment):
  // 16-lane group g: lane-half hi = g>>1, feature sub-block g&1; lane a16: sample a16>>2, piece a16&3
  const int g = lane >> 4, a16 = lane & 15;
  const int lane_seg = 4 * (g >> 1) + (a16 >> 2), lane_col = (16 * (g & 1) + 4 * (a16 & 3)) * 2;

  // Ring pipeline: chunks c+1 .. c+NBUF-2 stay in flight while chunk c is consumed.  All VMEM ops of
  // this kernel's main loop are LDS-DMA loads (same type, in-order), so a COUNTED vmcnt is exact:
  // "at most k*DMA_PER_CHUNK outstanding" == "chunk c has landed" when k younger chunks were issued.
  // Raw s_barrier (a __syncthreads() would drain vmcnt to 0 and kill the overlap).
  if constexpr (FULL) {
    const int dma_col = (lane & 31) * 16, dma_row = lane >> 5;
    constexpr int NBUF = P == 1 ? 4 : 2;         // LDS ring depth (P=1: 4 x 34 KiB, P=2: 2 x 68 KiB)
    auto issue = [&](int c) {
      if (c >= nchunk || dbg == 2) return;
      const int64_t r0 = r_begin + (int64_t)c * 32;
      const uint32_t buf = lds_base + (c % NBUF) * (2 * P * OPER_BYTES);
#pragma unroll
      for (int x = 0; x < 4 * P; ++x) {
        const int id = x * 8 + wave;               // 0 .. 32P-1
        const int op = id / (16 * P), rem = id - op * 16 * P, pl = rem >> 4, seg = rem & 15;
        const int rb = op == 0 ? rb_a : rb_b;
        const char* src = (op == 0 ? ga + pl * plane_a : gb + pl * plane_b) +
                          (size_t)(r0 + 2 * seg + dma_row) * rb + dma_col;
        glds16(src, buf + (op * P + pl) * OPER_BYTES + seg * SEG);
      }
    };
    const int lane_off = lane_seg * SEG + lane_col;
#pragma unroll
    for (int c = 0; c < NBUF - 1; ++c) issue(c);
    for (int c = 0; c < nchunk; ++c) {
      const int younger = nchunk - 1 - c < NBUF - 2 ? nchunk - 1 - c : NBUF - 2;
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_CHUNK) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_CHUNK) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue(c + NBUF - 1);
      if (dbg == 1) continue;
      const lds_addr buf = (c % NBUF) * (2 * P * OPER_BYTES) + lane_off;
      compute_chunk<P, true>(buf, wo, wi, nbo, nbi, do_bias, acc, bsum);
    }
  } else {
    const NarrowGeom gm = narrow_geom<P>(job, lds_bytes);
    const int NB = gm.nbuf;
    // DMA lane map: the first w/16 lanes carry row 0 of the segment, the next w/16 row 1, the rest idle
    int d_row[2], d_col[2];
    bool d_on[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int half = gm.w[o] / 16;
      d_row[o] = lane / half;
      d_col[o] = (lane - d_row[o] * half) * 16;
      d_on[o] = d_row[o] < 2 && d_col[o] < (o == 0 ? job.n_o : job.n_i) * 2;
    }
    int slot_i = 0, next_i = 0;
    auto issue = [&]() {                           // next chunk of the slice -> next ring slot
      const int c = next_i, slot = slot_i;
      ++next_i;
      slot_i = slot_i + 1 == NB ? 0 : slot_i + 1;
      if (c >= nchunk || dbg == 2) return;
      const int64_t r0 = r_begin + (int64_t)c * 32;
      const uint32_t buf = lds_base + slot * gm.chunk;
#pragma unroll
      for (int x = 0; x < 4 * P; ++x) {
        // id = x * 8 + wave (0 .. 32P-1): operand and plane depend on x only (wave < 8), so they are
        // compile-time after unrolling -- the geometry arrays must not be indexed dynamically (scratch
        // loads would sit in vmcnt between the DMA ops)
        const int op = (x * 8) / (16 * P), rem0 = x * 8 - op * 16 * P, pl = rem0 >> 4, seg = (rem0 & 15) + wave;
        if (d_on[op]) {
          const int rb = op == 0 ? rb_a : rb_b;
          const char* src = (op == 0 ? ga + pl * plane_a : gb + pl * plane_b) +
                            (size_t)(r0 + 2 * seg + d_row[op]) * rb + d_col[op];
          glds16(src, buf + (op == 0 ? 0u : P * gm.img[0]) + pl * gm.img[op] + seg * gm.segw[op]);
        }
      }
    };
    const lds_addr off_a = lane_seg * gm.segw[0] + lane_col, off_b = P * gm.img[0] + lane_seg * gm.segw[1] + lane_col;
    for (int c = 0; c < NB - 1; ++c) issue();
    int slot_c = 0;
    for (int c = 0; c < nchunk; ++c) {
      const int younger = nchunk - 1 - c < NB - 2 ? nchunk - 1 - c : NB - 2;
      switch (younger) {                           // wave-uniform; the count must be an immediate
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * DMA_PER_CHUNK) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_CHUNK) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * DMA_PER_CHUNK) : "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * DMA_PER_CHUNK) : "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * DMA_PER_CHUNK) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * DMA_PER_CHUNK) : "memory"); break;
      }
      __builtin_amdgcn_s_barrier();
      issue();
      const lds_addr buf = slot_c * gm.chunk;
      slot_c = slot_c + 1 == NB ? 0 : slot_c + 1;
      if (dbg == 1) continue;
      compute_chunk_rr<P>(buf + off_a, buf + off_b, gm, wave, n_ib, nblk, job.gb_off >= 0, acc_rr, bsum_rr);
    }
  }

  float* slab = a.slabs[net] + (size_t)split * gslab_floats(net);
  if constexpr (!FULL) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int b = wave + 8 * k;
      if (b >= nblk) break;
      const int bo = b / n_ib, bi = b - bo * n_ib;
      // output segment of this out-block (a job may feed two stages, see DwJob::o_split)
      const bool seg2 = job.o_split > 0 && bo >= job.o_split;
      const int sbo = seg2 ? bo - job.o_split : bo;
      const int s_off = seg2 ? job.gw_off2 : job.gw_off, s_ld = seg2 ? job.gw_ld2 : job.gw_ld;
      const int s_gb = seg2 ? job.gb_off2 : job.gb_off;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = 32 * sbo + (r & 3) + 8 * (r >> 2) + 4 * hi;
        slab[s_off + o * s_ld + 32 * bi + li] = acc_rr[k][r];
      }
      if (job.gb_off >= 0 && bi == 0) {
        const float tot = bsum_rr[k] + __shfl_xor(bsum_rr[k], 32, 64);
        if (hi == 0) slab[gw_floats(net) + s_gb + 32 * sbo + li] = tot;
      }
    }
    return;
  }
#pragma unroll
  for (int bo = 0; bo < 4; ++bo) {
    if (bo >= nbo) continue;
    const int ob = 32 * (4 * wo + bo);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
      if (bi >= nbi) continue;
      const int ib = 32 * (2 * wi + bi);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = ob + (r & 3) + 8 * (r >> 2) + 4 * hi;
        slab[job.gw_off + o * job.gw_ld + ib + li] = acc[bo][bi][r];
      }
    }
    if (do_bias && bo == wi) {                       // lane (li, hi) holds feature ob + li, half of the samples
      const float tot = bsum[bo] + __shfl_xor(bsum[bo], 32, 64);
      if (hi == 0) slab[gw_floats(net) + job.gb_off + ob + li] = tot;
    }
  }
}

}  // namespace nerfpp

using namespace nerfpp;

namespace {
// slices per job of one launch, in that launch's table order; the plan is indexed in build_all_jobs order
DwSched make_sched(const DwPlan& plan, bool full) {
  const JobTable all = build_all_jobs();
  DwSched sc{};
  int n = 0, wg = 0;
  for (int net = 0; net < N_NET; ++net) {
    if (net == 1) sc.njobs0 = n;
    for (int j = 0; j < all.count[net]; ++j) {
      if (dw_job_is_full(all.jobs[net][j]) != full) continue;
      sc.k[n] = plan.k[net][j];
      wg += plan.k[net][j];
      sc.wg_end[n] = wg;
      ++n;
    }
  }
  sc.njobs = n;
  return sc;
}
}  // namespace

void launch_dw(hipStream_t st, int P, const DwArgs& a) {
  const DwSched sf = make_sched(a.plan, true), sn = make_sched(a.plan, false);
  dim3 gfull(sf.wg_end[sf.njobs - 1]), gnarrow(sn.wg_end[sn.njobs - 1]);
  dim3 block(512);
  const size_t lds = (size_t)(P == 1 ? 4 : 2) * 2 * P * OPER_BYTES;
  static const int dbg = PROBE_GETENV("NERFPP_DW_DEBUG") ? atoi(PROBE_GETENV("NERFPP_DW_DEBUG")) : 0;   // 1: DMA only, 2: MFMA only
  if (P == 1) {
    hipLaunchKernelGGL((dw_kernel<1, true>), gfull, block, lds, st, a, sf, dbg, (uint32_t)lds);
    hipLaunchKernelGGL((dw_kernel<1, false>), gnarrow, block, lds, st, a, sn, dbg, (uint32_t)lds);
  } else {
    hipLaunchKernelGGL((dw_kernel<2, true>), gfull, block, lds, st, a, sf, dbg, (uint32_t)lds);
    hipLaunchKernelGGL((dw_kernel<2, false>), gnarrow, block, lds, st, a, sn, dbg, (uint32_t)lds);
  }
}
