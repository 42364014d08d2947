// Weight-gradient GEMMs:  GW[stage] = dZ^T [O x rows] * IN [rows x I]  (+ bias = column sums of dZ).
//
// The contraction runs over the SAMPLE axis (rows = n_rays*S, 65k..1.5M) and the output is at most
// 256x256 per job, so this is a split-K problem: a workgroup owns the whole output of one job over one
// slice of the rows and writes its partial result to that slice's slab; unpack_grads_kernel sums the
// slabs in a fixed order (deterministic) and maps the internal feature order back to the reference's
// parameter layout.  Every operand tensor is read exactly once per job.
//
// Data path (dw_kernel):
//   * operands are the FRAGMENT-MAJOR saved tensors of the fused MLP kernels (nerfpp_mlp.hip): per 32-row tile and
//     16-column chunk one 1 KiB block holding, at byte (2 j + hi) * 16, the 8 bf16 lane (j, hi) of the producing wave
//     had for row j.  A 32-row chunk of an operand is ld/16 consecutive blocks: the DMA (global_load_lds_dwordx4) is a
//     linear copy, one whole 1-KiB-contiguous block per wave-instruction -- also for the tensors that are only 32 or 64
//     columns wide (row-major, their instructions carried 8-32 live lanes).  In LDS the blocks sit 1152 B apart.
//   * MFMA wants the sample axis on the k-slots: fragments come from ds_read_b64_tr_b16 -- a 16-lane group reads a
//     [4 samples x 16 features] patch and receives it transposed (lane = feature, 4 samples per lane).  In a block the
//     four 8-byte pieces of a sample's 16 features are at (2 j + (q & 1)) * 16 + 8 * (q >> 1), q = feature quad, i.e.
//     the 16 lanes of a group read 128 contiguous bytes, the group next to it (the same samples of the next chunk)
//     1152 B = 32 banks further: conflict-free.  Both operands use the same sample->slot map, so the contraction is
//     exact whatever that map is.
//   * 8 waves; full 256x256 jobs: wave (wo = w>>2, wi = w&3) owns out-blocks [4wo, 4wo+4) x in-blocks
//     [2wi, 2wi+2), 8 accumulators of 32x32 (128 VGPRs); narrow jobs deal their blocks round-robin.
//     32-row chunks through an LDS ring (inline-asm DMA, counted vmcnt), one raw barrier per chunk.
//   * the number of row slices is per job (dw_plan, nerfpp_common.h): every launch fills the 256 CUs
//     once with workgroups that move about the same number of bytes.
//   * bias gradients: VALU column sums of the A fragments, split over the waves that share them.
// Rows beyond `rows` up to the next multiple of 32 are zero in every saved tensor (the MLP kernels
// zero-fill their tile tails), so no masking is needed.
#include "probe_env.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
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
// The narrow kernel deals a chunk's 1 KiB blocks (DMA wave-instructions) round-robin to its 8 waves and instantiates its
// main loop per count c_w = ceil((n_total - wave) / 8) in 1 .. CMAX: every wave must own at least one block (c_w == 0 would
// fall into the largest instantiation and issue DMAs past the operand) and at most CMAX = 4 P.
constexpr bool narrow_jobs_fit_the_dma_deal() {
  const JobTable jt = build_jobs(false);
  for (int net = 0; net < N_NET; ++net)
    for (int k = 0; k < jt.count[net]; ++k) {
      const int blocks = (jt.jobs[net][k].n_o + jt.jobs[net][k].n_i) / 16;      // per plane; P planes scale both bounds
      if (blocks < 8 || blocks > 32) return false;
    }
  return true;
}
static_assert(narrow_jobs_fit_the_dma_deal(), "narrow dW job: 8 <= (n_o + n_i) / 16 <= 32 blocks per plane");
__constant__ JobTable c_full = build_jobs(true);
__constant__ JobTable c_narrow = build_jobs(false);

constexpr int BLKP = FRAG_BYTES + 128;        // LDS stride of the 1 KiB chunk blocks (odd blocks land 32 banks off the even ones)
constexpr int OPER_BYTES = 16 * BLKP;         // 32 rows x 256 columns

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
// second read: the next 4 samples of the block (4 x 32 B further)
__device__ __forceinline__ bf16x8 tr_frag(lds_addr off, uint32_t row2 = 128) {
  __attribute__((address_space(3))) char* base = (__attribute__((address_space(3))) char*)dw_smem;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + off));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + off + row2));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// Narrow jobs: an operand image is its n/16 chunk blocks per plane; smaller chunks leave room for a deeper ring.
struct NarrowGeom {
  uint32_t nblk[2];     // 1 KiB blocks (= DMA wave-instructions) per plane of A, B
  uint32_t img[2];      // nblk * BLKP
  uint32_t chunk;       // P * (img[0] + img[1])
  int nbuf;             // ring depth: min(8, LDS bytes / chunk)
};
template <int P>
__device__ __forceinline__ NarrowGeom narrow_geom(const DwJob& job, uint32_t lds_bytes) {
  NarrowGeom g;
  g.nblk[0] = (uint32_t)job.n_o / 16;
  g.nblk[1] = (uint32_t)job.n_i / 16;
  g.img[0] = g.nblk[0] * BLKP;
  g.img[1] = g.nblk[1] * BLKP;
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
        if (FULL || x < nbo) fa[x][p] = tr_frag(buf + p * OPER_BYTES + kk * 512 + 2 * (4 * wo + x) * BLKP);
#pragma unroll
      for (int x = 0; x < 2; ++x)
        if (FULL || x < nbi) fb[x][p] = tr_frag(buf + (P + p) * OPER_BYTES + kk * 512 + 2 * (2 * wi + x) * BLKP);
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
        fa[p] = tr_frag(buf_a + p * gm.img[0] + kk * 512 + 2 * bo * BLKP);
        fb[p] = tr_frag(buf_b + p * gm.img[1] + kk * 512 + 2 * bi * BLKP);
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

// Probes build only: per-workgroup timestamps of the last launch of each instantiation -- entry, end of the
// chunk loop, end of the slab write -- with the job, its chunk count and the XCD, read back by nerfpp_probe_dw_stamps()
// (tools/probes/dw_stamps_probe.py).
#ifdef NERFPP_PROBES
static __device__ unsigned long long g_dw_stamps[2][256][6];
#define DW_STAMP(full_, slot_, val_) { if (threadIdx.x == 0 && bid < 256) g_dw_stamps[(full_) ? 0 : 1][bid][slot_] = (unsigned long long)(val_); }
extern "C" int nerfpp_probe_dw_stamps(void* host_dst, int bytes) {
  if (bytes != (int)sizeof(g_dw_stamps)) return (int)sizeof(g_dw_stamps);
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_dw_stamps), sizeof(g_dw_stamps));
}
#else
#define DW_STAMP(full_, slot_, val_) {}
#endif

// workgroup -> (job of this launch, row slice): jobs in table order (net 0 then net 1), k slices each
struct DwSched {
  int wg_end[2 * DW_JOBS];       // exclusive prefix of workgroups per job
  int k[2 * DW_JOBS];
  int njobs, njobs0;
};
// bid: index of this workgroup among the launch's workgroups of its kind (full / narrow)
template <int P, bool FULL>
__device__ __forceinline__ void dw_body(const DwArgs& a, const DwSched& sc, int dbg, uint32_t lds_bytes, const int bid) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;
  int job_id = 0;
  for (int j = 0; j < sc.njobs - 1; ++j) job_id += bid >= sc.wg_end[j];
  const int split = bid - (job_id == 0 ? 0 : sc.wg_end[job_id - 1]), ksplit = sc.k[job_id];
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
  DW_STAMP(FULL, 0, __builtin_readcyclecounter());
  DW_STAMP(FULL, 3, job_id);
  DW_STAMP(FULL, 4, nchunk);
  DW_STAMP(FULL, 5, __builtin_amdgcn_s_getreg((3 << 11) | 20));      // XCC_ID

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

  // DMA: 2 operands x P planes x (columns / 16) one-KiB blocks per 32-row chunk; full jobs: 4P wave-instructions per wave
  constexpr int DMA_PER_CHUNK = 4 * P;           // wave-instructions per wave per chunk (full jobs)
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)dw_smem;
  // per-lane position inside an operand image for the transposed reads (see header comment): 16-lane group g reads
  // chunk (g & 1) of a 32-column block for k-half (g >> 1) (8 samples each); lane a16: sample a16 >> 2 of the 4 a read
  // covers, feature quad q = a16 & 3
  const int g = lane >> 4, a16 = lane & 15;
  const int lane_off = (g & 1) * BLKP + ((((g >> 1) * 8 + (a16 >> 2)) * 2 + (a16 & 1)) * 16) + ((a16 >> 1) & 1) * 8;
  // blocks per 32-row tile of each operand TENSOR (the job may use a leading part of it) and the tile of row r: r / 32
  const size_t tb_a = (size_t)(rb_a >> 5), tb_b = (size_t)(rb_b >> 5);

  // Ring pipeline: chunks c+1 .. c+NBUF-2 stay in flight while chunk c is consumed.  All VMEM ops of
  // this kernel's main loop are LDS-DMA loads (same type, in-order), so a COUNTED vmcnt is exact:
  // "at most k*DMA_PER_CHUNK outstanding" == "chunk c has landed" when k younger chunks were issued.
  // Raw s_barrier (a __syncthreads() would drain vmcnt to 0 and kill the overlap).
  if constexpr (FULL) {
    constexpr int NBUF = P == 1 ? 4 : 2;         // LDS ring depth (P=1: 4 x 36 KiB, P=2: 2 x 72 KiB)
    auto issue = [&](int c) {
      if (c >= nchunk || dbg == 2) return;
      const size_t tile = (size_t)((r_begin >> 5) + c);
      const uint32_t buf = lds_base + (c % NBUF) * (2 * P * OPER_BYTES);
#pragma unroll
      for (int x = 0; x < 4 * P; ++x) {
        const int id = x * 8 + wave;               // 0 .. 32P-1
        const int op = id / (16 * P), rem = id - op * 16 * P, pl = rem >> 4, blk = rem & 15;
        const char* src = (op == 0 ? ga + pl * plane_a + (tile * tb_a + blk) * FRAG_BYTES
                                   : gb + pl * plane_b + (tile * tb_b + blk) * FRAG_BYTES) + lane * 16;
        glds16(src, buf + (op * P + pl) * OPER_BYTES + blk * BLKP);
      }
    };
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
    // the chunk's blocks in the order [A plane 0 .. P-1 | B plane 0 .. P-1] are dealt round-robin to the 8 waves: wave w
    // issues ids w, w + 8, ... (c_w of them; every instruction is one whole block, all 64 lanes live) and waits for ITS OWN
    // instructions before the barrier
    const int n_a = (int)gm.nblk[0], n_b = (int)gm.nblk[1], n_total = P * (n_a + n_b);
    const int c_w = (n_total - wave + 7) >> 3;
    constexpr int CMAX = (P * 32 + 7) / 8;
    const lds_addr off_a = lane_off, off_b = P * gm.img[0] + lane_off;
    // The main loop is instantiated per value of c_w (1 .. CMAX; wave-uniform, the waves of a workgroup differ by at most
    // one): the counted waits need immediates, and a 61-way switch on younger * c_w in every chunk cost 5-11 % of the
    // launch (jump table + refetch).  Every instantiation runs the same number of barriers.
    auto run = [&](auto cw_c) __attribute__((always_inline)) {
      constexpr int CW = decltype(cw_c)::value;
      int slot_i = 0, next_i = 0;
      auto issue = [&]() __attribute__((always_inline)) {          // next chunk of the slice -> next ring slot
        const int c = next_i, slot = slot_i;
        ++next_i;
        slot_i = slot_i + 1 == NB ? 0 : slot_i + 1;
        if (c >= nchunk || dbg == 2) return;
        const size_t tile = (size_t)((r_begin >> 5) + c);
        const uint32_t buf = lds_base + slot * gm.chunk;
        const char* ca = ga + tile * tb_a * FRAG_BYTES + lane * 16;
        const char* cb = gb + tile * tb_b * FRAG_BYTES + lane * 16;
#pragma unroll
        for (int k = 0; k < CW; ++k) {
          const int id = wave + 8 * k;                 // < n_total by the definition of c_w
          const bool is_b = id >= P * n_a;
          const int idl = is_b ? id - P * n_a : id, n_op = is_b ? n_b : n_a;
          const int pl = idl >= n_op ? 1 : 0, blk = idl - pl * n_op;   // P <= 2
          const char* src = (is_b ? cb + pl * plane_b : ca + pl * plane_a) + (size_t)blk * FRAG_BYTES;
          glds16(src, buf + (is_b ? P * gm.img[0] : 0u) + (uint32_t)pl * (is_b ? gm.img[1] : gm.img[0]) + (uint32_t)blk * BLKP);
        }
      };
      for (int c = 0; c < NB - 1; ++c) issue();
      int slot_c = 0;
      for (int c = 0; c < nchunk; ++c) {
        const int younger = nchunk - 1 - c < NB - 2 ? nchunk - 1 - c : NB - 2;
        switch (dbg == 2 ? 0 : younger) {            // wave-uniform; the count must be an immediate
          case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
          case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * CW) : "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CW) : "memory"); break;
          case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * CW) : "memory"); break;
          case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * CW) : "memory"); break;
          case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * CW) : "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * CW) : "memory"); break;
        }
        __builtin_amdgcn_s_barrier();
        issue();
        const lds_addr buf = slot_c * gm.chunk;
        slot_c = slot_c + 1 == NB ? 0 : slot_c + 1;
        if (dbg == 1) continue;
        compute_chunk_rr<P>(buf + off_a, buf + off_b, gm, wave, n_ib, nblk, job.gb_off >= 0, acc_rr, bsum_rr);
      }
    };
    static_assert(CMAX <= 8 && 6 * CMAX < 63, "counted waits fit the vmcnt field");
    switch (c_w) {
      case 1: run(std::integral_constant<int, 1>{}); break;
      case 2: run(std::integral_constant<int, 2>{}); break;
      case 3: run(std::integral_constant<int, 3>{}); break;
      case 4: run(std::integral_constant<int, 4>{}); break;
      default:
        if constexpr (P == 2) {
          switch (c_w) {
            case 5: run(std::integral_constant<int, 5>{}); break;
            case 6: run(std::integral_constant<int, 6>{}); break;
            case 7: run(std::integral_constant<int, 7>{}); break;
            default: run(std::integral_constant<int, 8>{}); break;
          }
        } else {
          run(std::integral_constant<int, 4>{});
        }
    }
  }

  DW_STAMP(FULL, 1, __builtin_readcyclecounter());
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
    DW_STAMP(FULL, 2, __builtin_readcyclecounter());
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
  DW_STAMP(FULL, 2, __builtin_readcyclecounter());
}

// (Round 4 also ran both kinds in ONE launch -- full-job workgroups first, the narrow ones as CUs free up, the way the fg and
// bg MLP kernels of a level share a launch since then: 2.426 vs 2.411 ms per step in alternating runs on one box, i.e. 0.6 %
// SLOWER, the weight-gradient group 0.634 vs 0.620 ms; profiles/r04_pair_launch.md.  Two launches it stays.)
template <int P, bool FULL>
__global__ __launch_bounds__(512) void dw_kernel(DwArgs a, DwSched sc, int dbg, uint32_t lds_bytes) {
  dw_body<P, FULL>(a, sc, dbg, lds_bytes, (int)blockIdx.x);
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
  const size_t lds = (size_t)(P == 1 ? 4 : 2) * 2 * P * OPER_BYTES;     // 144 KiB
  static const int dbg = PROBE_GETENV("NERFPP_DW_DEBUG") ? atoi(PROBE_GETENV("NERFPP_DW_DEBUG")) : 0;   // 1: DMA only, 2: MFMA only
  if (P == 1) {
    hipLaunchKernelGGL((dw_kernel<1, true>), gfull, block, lds, st, a, sf, dbg, (uint32_t)lds);
    hipLaunchKernelGGL((dw_kernel<1, false>), gnarrow, block, lds, st, a, sn, dbg, (uint32_t)lds);
  } else {
    hipLaunchKernelGGL((dw_kernel<2, true>), gfull, block, lds, st, a, sf, dbg, (uint32_t)lds);
    hipLaunchKernelGGL((dw_kernel<2, false>), gnarrow, block, lds, st, a, sn, dbg, (uint32_t)lds);
  }
}
