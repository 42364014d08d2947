// Weight-gradient GEMMs:  GW[stage] = dZ^T [O x rows] * IN [rows x I]  (+ bias = column sums of dZ).
//
// The contraction runs over the SAMPLE axis (rows = n_rays*S, 65k..1.5M) and the output is at most
// 256x256 per job, so this is a split-K problem: a workgroup owns the whole output of one job over one
// slice of the rows and writes its partial result to that slice's slab (non-temporal stores: ~117 MB per level that nothing
// re-reads from the L2; level 0's launches run next to level 1's forward, whose weight streams live there); unpack_grads_kernel sums the
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
//     [2wi, 2wi+2), 8 accumulators of 32x32 (128 VGPRs); every other ("narrow") job shape has its own instantiation
//     (NarrowShape / narrow_pass below): wave = block index on the longer axis of the output, two input tensors side by
//     side where jobs share their dZ operand.  32-row chunks through an LDS ring (inline-asm DMA, counted vmcnt), one raw
//     barrier per chunk.
//   * the number of row slices is per job (dw_plan, nerfpp_common.h): every launch fills the 256 CUs
//     once with workgroups that finish at about the same time.
//   * bias gradients: VALU column sums of the A fragments, split over the waves that share them.
//   * a bf16 backward (round 5): two operands are NOT saved tensors but recomputed inside their job, one 32-row chunk ahead
//     into a double-buffered tile -- H0 = relu(W0 X + b0) from the encoded point in job L1 (rc_job), dZ7 = mask7 *
//     (Wc^T dG + wsigma dsigma) from [dS | dG] and the sign words of H7 in job L7 (rc7_job); the MLP kernels then write 512 B
//     per row less each.  Those jobs are matrix-bound and get more slices (dw_plan).  The narrow launch takes the CU's whole
//     160 KiB of LDS (deeper rings), the full launch 144 KiB (160 with the recomputing jobs).
// Rows beyond `rows` up to rows_padded (a multiple of 256) are zero in every saved tensor (the MLP kernels
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
// The narrow kernel is instantiated per job SHAPE (n_o x n_i); every narrow job of the table must have one.
constexpr bool narrow_shape_known(int n_o, int n_i, int n_i1) {
  return (n_o == 256 && (n_i == 64 || n_i == 96) && n_i1 == n_i) ||                  // L0: dZ0^T X
         (n_o == 256 && (n_i == 320 || n_i == 352) && n_i1 == n_i - 256) ||          // L5: dZ5^T [X | H4]
         (n_o == DSG_LD && n_i == 288 && n_i1 == 256) ||                             // [dS | dG]^T [H7 | DIRX]
         (n_o == 32 && n_i == 128 && n_i1 == 128);                                   // rgb1: dP^T G
}
constexpr bool narrow_jobs_have_shapes() {
  const JobTable jt = build_jobs(false);
  for (int net = 0; net < N_NET; ++net)
    for (int k = 0; k < jt.count[net]; ++k)
      if (!narrow_shape_known(jt.jobs[net][k].n_o, jt.jobs[net][k].n_i, jt.jobs[net][k].n_i1)) return false;
  return true;
}
static_assert(narrow_jobs_have_shapes(), "narrow dW job without a NarrowShape instantiation (dw_body)");
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
constexpr int DW_LDS_BYTES = 8 * OPER_BYTES;   // 144 KiB: the full jobs' ring (4 x 2 operand images)
constexpr int NARROW_LDS_BYTES = 160 * 1024;    // the narrow launch takes the whole LDS of the CU: its jobs are bound by bytes in flight
// Narrow jobs, per shape.  An operand image is its n / 16 chunk blocks per plane per 32-row tile; a ring slot holds T tiles
// (two for the 160-column jobs, whose 10 KiB tiles are too little work per barrier: the serial chain wait -> barrier -> DMA
// issue -> LDS read -> MFMA took 870-980 cycles per tile against ~700 of HBM time).  The wave index runs along the LONGER
// block axis of the output, so that a wave's fragment on that axis is read once per 16 samples for all KB MFMAs that use it
// (until round 4 the blocks were dealt round-robin with both fragments read per MFMA and the reads not hoisted: the
// [dS | dG]^T H7 job ran a serial read -> MFMA chain ten times per tile, 2860 cycles against ~1870 of HBM time).
// A pass covers the in-blocks [IB0, IB0 + IBN) of the job's output (all of them, unless a slot of the whole job would not
// fit the LDS twice: the split-bf16 L5 job runs two passes over its slice).
template <int P, int N_O, int N_I1, int IB0, int IBN>
struct NarrowShape {
  static constexpr int N_OB = N_O / 32, N_IB = IBN;
  // wave = in-block (else out-block).  Nine in-blocks: eight of them one per wave, and wave w < N_OB also owns block (w, 8)
  // -- the [dS | dG]^T [H7 | DIRX] job: 5 x 9 blocks as 6 + 6 + 6 + 6 + 6 + 5 + 5 + 5
  static constexpr bool BI_WAVE = N_IB > N_OB && N_IB <= 9;
  static constexpr bool XTRA = BI_WAVE && N_IB == 9;
  static constexpr int NW = BI_WAVE ? (XTRA ? 8 : N_IB) : N_OB;    // waves with MFMA work; the others only move data
  static constexpr int KB = BI_WAVE ? N_OB : N_IB;             // blocks per wave (+ 1 accumulator for the extra block)
  static constexpr int NACC = KB + (XTRA ? 1 : 0);
  static constexpr int NBLK_A = N_O / 16, NBLK_B = 2 * IBN;    // 1 KiB blocks (= DMA wave-instructions) per plane per tile
  static constexpr int BLK_B0 = 2 * IB0;                       // first block of the pass in the job's input row
  static constexpr int NBLK_B1 = N_I1 / 16;                    // blocks of that row the first input tensor supplies
  static constexpr int T = NBLK_A + NBLK_B <= 10 ? 2 : 1;      // 32-row tiles per ring slot
  static constexpr int IMG_A = NBLK_A * BLKP, IMG_B = NBLK_B * BLKP;
  static constexpr int SUB = P * (IMG_A + IMG_B);              // one tile in LDS: [A planes | B planes]
  static constexpr int CHUNK = T * SUB;
  static constexpr int NT = P * (NBLK_A + NBLK_B), NTOT = T * NT;       // DMA wave-instructions per tile / per slot
  static constexpr int CW_HI = (NTOT + 7) / 8;                 // waves < NTOT % 8 issue CW_HI of them, the others one fewer
  static constexpr int NB = NARROW_LDS_BYTES / CHUNK > 8 ? 8 : NARROW_LDS_BYTES / CHUNK;
  static constexpr int KG = P == 1 ? 6 : 3;                    // short-axis fragments read ahead of their MFMAs (registers)
  static_assert(NW <= 8 && NTOT >= 8 && NB >= 2 && (NB - 2) * CW_HI < 63, "narrow dW shape");
};
__device__ __forceinline__ float bf16_sum8(const bf16x8& v) {
  const uint4 w = *(const uint4*)&v;
  const uint32_t u[4] = {w.x, w.y, w.z, w.w};
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) s += __uint_as_float(u[k] << 16) + __uint_as_float(u[k] & 0xffff0000u);
  return s;
}

// one 32-row chunk: 2 k16-steps x (<=4 out-blocks x <=2 in-blocks) MFMAs for this wave
// bufa / bufb: LDS offsets (incl. the lane's tr-read offset) of the A and B operand images (planes OPER_BYTES apart)
template <int P, bool FULL>
__device__ __forceinline__ void compute_chunk(lds_addr bufa, lds_addr bufb, int wo, int wi, int nbo, int nbi, bool do_bias,
                                              f32x16 (&acc)[4][2], float (&bsum)[4]) {
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    bf16x8 fa[4][P], fb[2][P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
#pragma unroll
      for (int x = 0; x < 4; ++x)
        if (FULL || x < nbo) fa[x][p] = tr_frag(bufa + p * OPER_BYTES + kk * 512 + 2 * (4 * wo + x) * BLKP);
#pragma unroll
      for (int x = 0; x < 2; ++x)
        if (FULL || x < nbi) fb[x][p] = tr_frag(bufb + p * OPER_BYTES + kk * 512 + 2 * (2 * wi + x) * BLKP);
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

// one ring slot of a narrow job: T tiles x 2 k16-steps x KB MFMAs; all LDS reads of a k16-step precede its MFMAs
template <int P, typename S>
__device__ __forceinline__ void compute_narrow(lds_addr buf, int wave, bool has_bias, f32x16 (&acc)[S::NACC], float& bsum) {
#pragma unroll
  for (int sub = 0; sub < S::T; ++sub) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const lds_addr base = buf + sub * S::SUB + kk * 512;
      bf16x8 fw[P], fk[S::KB][P];            // the wave's own fragment (long axis), the KB fragments of the short axis
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if constexpr (S::BI_WAVE) fw[p] = tr_frag(base + P * S::IMG_A + p * S::IMG_B + 2 * wave * BLKP);
        else fw[p] = tr_frag(base + p * S::IMG_A + 2 * wave * BLKP);
      }
#pragma unroll
      for (int k0 = 0; k0 < S::KB; k0 += S::KG) {
#pragma unroll
        for (int k = k0; k < k0 + S::KG && k < S::KB; ++k)
#pragma unroll
          for (int p = 0; p < P; ++p) {
            if constexpr (S::BI_WAVE) fk[k][p] = tr_frag(base + p * S::IMG_A + 2 * k * BLKP);
            else fk[k][p] = tr_frag(base + P * S::IMG_A + p * S::IMG_B + 2 * k * BLKP);
          }
#pragma unroll
        for (int k = k0; k < k0 + S::KG && k < S::KB; ++k) {
          const bf16x8* fa = S::BI_WAVE ? fk[k] : fw;
          const bf16x8* fb = S::BI_WAVE ? fw : fk[k];
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc[k], 0, 0, 0);
          if constexpr (P == 2) {
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc[k], 0, 0, 0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], acc[k], 0, 0, 0);
          }
        }
      }
      if constexpr (S::XTRA) {               // block (bo = wave, bi = 8): the wave's out-block fragment is fk[wave]
        bf16x8 fx[P];
#pragma unroll
        for (int p = 0; p < P; ++p) fx[p] = tr_frag(base + P * S::IMG_A + p * S::IMG_B + 2 * 8 * BLKP);
#pragma unroll
        for (int k = 0; k < S::KB; ++k)
          if (k == wave) {
            acc[S::KB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[k][0], fx[0], acc[S::KB], 0, 0, 0);
            if constexpr (P == 2) {
              acc[S::KB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[k][0], fx[1], acc[S::KB], 0, 0, 0);
              acc[S::KB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[k][1], fx[0], acc[S::KB], 0, 0, 0);
            }
          }
      }
      if (has_bias) {                        // column sums of dZ: out-block `wave` (BI_WAVE: wave k < KB sums out-block k)
        if constexpr (S::BI_WAVE) {
#pragma unroll
          for (int k = 0; k < S::KB; ++k)
            if (k == wave) {
              bsum += bf16_sum8(fk[k][0]);
              if constexpr (P == 2) bsum += bf16_sum8(fk[k][1]);
            }
        } else {
          bsum += bf16_sum8(fw[0]);
          if constexpr (P == 2) bsum += bf16_sum8(fw[1]);
        }
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

// A narrow job over its row slice.  The slot's NTOT blocks, in the order [tile 0: A planes | B planes][tile 1: ...], are dealt
// round-robin to the 8 waves: wave w issues ids w, w + 8, ... (CW of them; every instruction is one whole block, all 64
// lanes live) and waits for ITS OWN instructions before the barrier.  The main loop is instantiated per CW (the counted
// waits need immediates; the waves of a workgroup differ by at most one).
template <int P, int N_O, int N_I1, int IB0, int IBN>
__device__ __forceinline__ void narrow_pass(const DwArgs& a, const DwJob& job, int net, int split, int ksplit, int dbg, int bid) {
  using S = NarrowShape<P, N_O, N_I1, IB0, IBN>;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;
  const int rb_a = tensor_ld(net, job.a_tensor) * 2;     // row bytes
  const int rb_b = tensor_ld(net, job.b_tensor) * 2, rb_b2 = tensor_ld(net, job.b_tensor2) * 2;
  const char* ga = (const char*)a.ws[net].t[job.a_tensor];
  const char* gb = (const char*)a.ws[net].t[job.b_tensor];
  const char* gb2 = (const char*)a.ws[net].t[job.b_tensor2];
  const size_t plane_a = (size_t)a.rows_padded * rb_a, plane_b = (size_t)a.rows_padded * rb_b, plane_b2 = (size_t)a.rows_padded * rb_b2;
  constexpr int RT = 32 * S::T;                          // rows per ring slot (rows_padded is a multiple of 256, zero-filled)
  const int64_t rows_t = (a.rows + RT - 1) / RT * RT;
  int64_t rps = (rows_t + ksplit - 1) / ksplit;
  rps = (rps + RT - 1) / RT * RT;
  const int64_t r_begin = split * rps;
  const int64_t r_end = r_begin + rps < rows_t ? r_begin + rps : rows_t;
  const int nchunk = r_end > r_begin ? (int)((r_end - r_begin) / RT) : 0;
  DW_STAMP(false, 0, __builtin_readcyclecounter());
  DW_STAMP(false, 4, nchunk * S::T);
  DW_STAMP(false, 5, __builtin_amdgcn_s_getreg((3 << 11) | 20));      // XCC_ID

  f32x16 acc[S::NACC];
#pragma unroll
  for (int x = 0; x < S::NACC; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
  float bsum = 0.f;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)dw_smem;
  const int g = lane >> 4, a16 = lane & 15;              // per-lane position for the transposed reads (see dw_body)
  const int lane_off = (g & 1) * BLKP + ((((g >> 1) * 8 + (a16 >> 2)) * 2 + (a16 & 1)) * 16) + ((a16 >> 1) & 1) * 8;
  const size_t tb_a = (size_t)(rb_a >> 5), tb_b = (size_t)(rb_b >> 5), tb_b2 = (size_t)(rb_b2 >> 5);   // blocks per 32-row tile of each operand TENSOR
  const bool has_bias = job.gb_off >= 0 && IB0 == 0;

  auto run = [&](auto cw_c) __attribute__((always_inline)) {
    constexpr int CW = decltype(cw_c)::value;
    int slot_i = 0, next_i = 0;
    auto issue = [&]() __attribute__((always_inline)) {          // next chunk of the slice -> next ring slot
      const int c = next_i, slot = slot_i;
      ++next_i;
      slot_i = slot_i + 1 == S::NB ? 0 : slot_i + 1;
      if (c >= nchunk || dbg == 2) return;
      const size_t tile = (size_t)(r_begin >> 5) + (size_t)c * S::T;
      const uint32_t buf = lds_base + slot * S::CHUNK;
      const char* ca = ga + tile * tb_a * FRAG_BYTES + lane * 16;
      const char* cb = gb + tile * tb_b * FRAG_BYTES + lane * 16;
      const char* cb2 = gb2 + tile * tb_b2 * FRAG_BYTES + lane * 16;
#pragma unroll
      for (int k = 0; k < CW; ++k) {
        const int id = wave + 8 * k;                   // < NTOT by the choice of CW
        const int sub = S::T > 1 ? id / S::NT : 0, r = id - sub * S::NT;
        const bool is_b = r >= P * S::NBLK_A;
        const int idl = is_b ? r - P * S::NBLK_A : r, n_op = is_b ? S::NBLK_B : S::NBLK_A;
        const int pl = idl >= n_op ? 1 : 0, blk = idl - pl * n_op;   // P <= 2
        const int bblk = S::BLK_B0 + blk;                            // block of the job's input row
        const bool second = is_b && bblk >= S::NBLK_B1;              // the input's second tensor (its own row width)
        const char* src = !is_b ? ca + pl * plane_a + (size_t)sub * tb_a * FRAG_BYTES + (size_t)blk * FRAG_BYTES
                        : !second ? cb + pl * plane_b + (size_t)sub * tb_b * FRAG_BYTES + (size_t)bblk * FRAG_BYTES
                                  : cb2 + pl * plane_b2 + (size_t)sub * tb_b2 * FRAG_BYTES + (size_t)(bblk - S::NBLK_B1) * FRAG_BYTES;
        glds16(src, buf + (uint32_t)sub * S::SUB + (is_b ? (uint32_t)(P * S::IMG_A) : 0u) +
                        (uint32_t)pl * (is_b ? S::IMG_B : S::IMG_A) + (uint32_t)blk * BLKP);
      }
    };
#pragma unroll
    for (int c = 0; c < S::NB - 1; ++c) issue();
    int slot_c = 0;
    for (int c = 0; c < nchunk; ++c) {
      const int younger = nchunk - 1 - c < S::NB - 2 ? nchunk - 1 - c : S::NB - 2;
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
      const lds_addr buf = slot_c * S::CHUNK + lane_off;
      slot_c = slot_c + 1 == S::NB ? 0 : slot_c + 1;
      if (dbg == 1) continue;
      if (wave < S::NW) compute_narrow<P, S>(buf, wave, has_bias, acc, bsum);
    }
  };
  if constexpr (S::NTOT % 8 == 0) run(std::integral_constant<int, S::CW_HI>{});
  else if (wave < S::NTOT % 8) run(std::integral_constant<int, S::CW_HI>{});
  else run(std::integral_constant<int, S::CW_HI - 1>{});

  DW_STAMP(false, 1, __builtin_readcyclecounter());
  float* slab = a.slabs[net] + (size_t)split * gslab_floats(net);
  if (wave < S::NW) {
#pragma unroll
    for (int k = 0; k < S::NACC; ++k) {
      if (k == S::KB && wave >= S::N_OB) continue;       // (the extra block exists for wave < N_OB)
      const int bo = k == S::KB ? wave : (S::BI_WAVE ? k : wave), bi = IB0 + (k == S::KB ? 8 : (S::BI_WAVE ? wave : k));
      // output segment of this out-block (a job may feed two stages, see DwJob::o_split)
      const bool seg2 = job.o_split > 0 && bo >= job.o_split;
      const int sbo = seg2 ? bo - job.o_split : bo;
      const int s_off = seg2 ? job.gw_off2 : job.gw_off, s_ld = seg2 ? job.gw_ld2 : job.gw_ld;
      if (32 * bi + 32 > s_ld) continue;                 // (the dS^T DIRX block of the merged sigma / rgb0 job: not a parameter)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = 32 * sbo + (r & 3) + 8 * (r >> 2) + 4 * hi;
        __builtin_nontemporal_store(acc[k][r], slab + s_off + o * s_ld + 32 * bi + li);
      }
    }
    if (has_bias && wave < S::N_OB) {                     // wave w holds the column sums of out-block w (both mappings)
      const int bo = wave;
      const bool seg2 = job.o_split > 0 && bo >= job.o_split;
      const int sbo = seg2 ? bo - job.o_split : bo;
      const int s_gb = seg2 ? job.gb_off2 : job.gb_off;
      const float tot = bsum + __shfl_xor(bsum, 32, 64);
      if (hi == 0) __builtin_nontemporal_store(tot, slab + gw_floats(net) + s_gb + 32 * sbo + li);
    }
  }
  DW_STAMP(false, 2, __builtin_readcyclecounter());
}

template <int P, int N_O, int N_I, int N_I1>
__device__ __forceinline__ void narrow_job(const DwArgs& a, const DwJob& job, int net, int split, int ksplit, int dbg, int bid) {
  constexpr int N_IB = N_I / 32, T = (N_O + N_I) / 16 <= 10 ? 2 : 1;
  if constexpr (NARROW_LDS_BYTES / (T * P * ((N_O + N_I) / 16) * BLKP) >= 2) {
    narrow_pass<P, N_O, N_I1, 0, N_IB>(a, job, net, split, ksplit, dbg, bid);
  } else {
    constexpr int H = (N_IB + 1) / 2;
    narrow_pass<P, N_O, N_I1, 0, H>(a, job, net, split, ksplit, dbg, bid);
    // the slab stores of the first pass share vmcnt with the second pass' DMA and may retire out of order: drain them; the
    // barrier keeps the second pass' first DMA out of slots other waves still read
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    narrow_pass<P, N_O, N_I1, H, N_IB - H>(a, job, net, split, ksplit, dbg, bid);
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// Job L1 of a net when H0 is not a saved tensor (round 5; single-plane workspaces): dW1 = dZ1^T H0 with
//   H0 = relu(W0 X + b0)
// recomputed per 32-row chunk from the saved encoded point X (KX = 4 / 6 chunk blocks instead of H0's 16: the job streams
// 20 / 22 KiB per chunk instead of 32, and the training forward does not write H0 at all).  The bounding experiment of
// profiles/r05_recompute_probe.md showed what NOT to do -- a second barrier per chunk and the recompute as a serial phase
// in front of the chunk's MFMAs cost more than the bytes save.  Here the recompute runs ONE CHUNK AHEAD into a double-
// buffered tile, so that its short dependent MFMA chain, the conversion and the tile write of chunk c + 1 sit between the 16
// independent MFMAs of chunk c, with one barrier per chunk as before:
//   iteration c:  wait chunk c + 1 | barrier | DMA chunk c + NB - 1 | H0(c + 1) -> tile[(c + 1) & 1] | dW += dZ1(c)^T tile[c & 1]
// Wave w owns out-block w of H0: its W0 fragments (KX x 16 B per lane) and its bias slice stay in registers; the X blocks
// are the forward's B-operand register images (16 B of lane (j, hi) at (2 j + hi) * 16), the MFMA orientation, k order and
// bias-initialised accumulator are the forward's, so the recomputed tile is bit-identical to what the forward would have
// saved; it is written in the saved tensors' block form, and the transposed reads of the weight-gradient MFMAs run on it
// unchanged.  Rows past the end of the batch have X = 0, i.e. H0 = relu(b0) != 0 -- but dZ1 = 0 there, so they add nothing.
template <int KX>
struct RcShape {
  static constexpr int NT = 16 + KX;                        // 1 KiB blocks (= DMA wave-instructions) per chunk: dZ1, then X
  static constexpr int SLOT = NT * BLKP;
  static constexpr int TILE0 = 160 * 1024 - 2 * OPER_BYTES; // the two recomputed tiles sit at the top of the CU's LDS
  static constexpr int NB = TILE0 / SLOT;                   // ring depth: 5 for both nets
  static constexpr int CW_HI = (NT + 7) / 8, N_HI = NT % 8; // waves < N_HI issue CW_HI instructions per chunk, the others one fewer
  static_assert(NB >= 4 && NB <= 6 && N_HI != 0, "rc_job ring");
};
constexpr int RC_LDS_BYTES = 160 * 1024;

template <int KX>
__device__ __forceinline__ void rc_job(const DwArgs& a, const DwJob& job, int net, int split, int ksplit, int bid) {
  using S = RcShape<KX>;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;
  const char* ga = (const char*)a.ws[net].t[job.a_tensor];
  const char* gx = (const char*)a.ws[net].t[T_X];
  const int64_t rows32 = (a.rows + 31) / 32 * 32;
  int64_t rps = (rows32 + ksplit - 1) / ksplit;
  rps = (rps + 31) / 32 * 32;
  const int64_t r_begin = split * rps;
  const int64_t r_end = r_begin + rps < rows32 ? r_begin + rps : rows32;
  const int nchunk = r_end > r_begin ? (int)((r_end - r_begin) / 32) : 0;
  DW_STAMP(true, 0, __builtin_readcyclecounter());
  DW_STAMP(true, 4, nchunk);
  DW_STAMP(true, 5, __builtin_amdgcn_s_getreg((3 << 11) | 20));

  const int wo = wave >> 2, wi = wave & 3;
  const bool do_bias = job.gb_off >= 0;
  f32x16 acc[4][2];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[x][0][r] = 0.f; acc[x][1][r] = 0.f; }
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  // resident operands of the recompute: W0 fragments (k-chunk kc, out-block wave) of the packed forward stream, bias slice in the
  // accumulator (C / D) layout of the forward kernel's init_bias_lds
  bf16x8 w0[KX];
#pragma unroll
  for (int kc = 0; kc < KX; ++kc) w0[kc] = *(const bf16x8*)((const char*)a.fwd_w[net] + (size_t)(kc * 8 + wave) * FRAG_BYTES + lane * 16);
  f32x16 b0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = *(const float4*)(a.fwd_bias[net] + wave * 32 + hi * 16 + 4 * q);
    b0[4 * q] = v.x; b0[4 * q + 1] = v.y; b0[4 * q + 2] = v.z; b0[4 * q + 3] = v.w;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the counted waits below see DMA loads only
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)dw_smem;
  const int g = lane >> 4, a16 = lane & 15;
  const int lane_off = (g & 1) * BLKP + ((((g >> 1) * 8 + (a16 >> 2)) * 2 + (a16 & 1)) * 16) + ((a16 >> 1) & 1) * 8;
  if (nchunk == 0) goto write_out;
  {
    auto run = [&](auto cw_c) __attribute__((always_inline)) {
      constexpr int CW = decltype(cw_c)::value;
      int issued = 0;
      auto issue = [&]() __attribute__((always_inline)) {       // next chunk of the slice -> its ring slot
        const int c = issued++;
        if (c >= nchunk) return;
        const size_t tile = (size_t)(r_begin >> 5) + (size_t)c;
        const uint32_t slot = lds_base + (uint32_t)(c % S::NB) * S::SLOT;
#pragma unroll
        for (int k = 0; k < CW; ++k) {
          const int id = wave + 8 * k;                            // < NT by the choice of CW
          const char* src = id < 16 ? ga + (tile * 16 + id) * FRAG_BYTES : gx + (tile * KX + (id - 16)) * FRAG_BYTES;
          glds16(src + lane * 16, slot + (uint32_t)id * BLKP);
        }
      };
      auto wait_chunk = [&](int k) __attribute__((always_inline)) {     // chunk k has landed (wave-uniform; counts need immediates)
        const int have = issued < nchunk ? issued : nchunk;
        const int younger = have - 1 - k;
        switch (younger <= 0 ? 0 : younger) {
          case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
          case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * CW) : "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CW) : "memory"); break;
          case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * CW) : "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * CW) : "memory"); break;
        }
      };
      auto recompute = [&](int k) __attribute__((always_inline)) {      // H0 of chunk k -> tile[k & 1], out-block `wave`
        const char* xs = dw_smem + (k % S::NB) * S::SLOT + 16 * BLKP + ((2 * li + hi) << 4);
        f32x16 rc = b0;
#pragma unroll
        for (int kc = 0; kc < KX; ++kc)
          rc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0[kc], *(const bf16x8*)(xs + kc * BLKP), rc, 0, 0, 0);
        char* dst = dw_smem + S::TILE0 + (k & 1) * OPER_BYTES + 2 * wave * BLKP + ((2 * li + hi) << 4);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          uint4 d;
          uint32_t* dp = (uint32_t*)&d;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            typedef float f32x2_ __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
            typedef short s16x2_ __attribute__((ext_vector_type(2)));
            const f32x2_ v = {rc[8 * hh + 2 * w], rc[8 * hh + 2 * w + 1]};
            const s16x2_ zero = {0, 0};
            dp[w] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_, __builtin_convertvector(v, bf16x2_)), zero));
          }
          *(uint4*)(dst + hh * BLKP) = d;
        }
      };
#pragma unroll
      for (int c = 0; c < S::NB - 1; ++c) issue();
      wait_chunk(0);
      __builtin_amdgcn_s_barrier();
      recompute(0);
      for (int c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk) wait_chunk(c + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // this wave's tile writes of the previous iteration
        __builtin_amdgcn_s_barrier();
        issue();
        if (c + 1 < nchunk) recompute(c + 1);
        compute_chunk<1, true>((c % S::NB) * S::SLOT + lane_off, S::TILE0 + (c & 1) * OPER_BYTES + lane_off, wo, wi, 4, 2, do_bias, acc, bsum);
      }
    };
    if (wave < S::N_HI) run(std::integral_constant<int, S::CW_HI>{});
    else run(std::integral_constant<int, S::CW_HI - 1>{});
  }
write_out:
  DW_STAMP(true, 1, __builtin_readcyclecounter());
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
        __builtin_nontemporal_store(acc[bo][bi][r], slab + job.gw_off + o * job.gw_ld + ib + li);
      }
    }
    if (do_bias && bo == wi) {
      const float tot = bsum[bo] + __shfl_xor(bsum[bo], 32, 64);
      if (hi == 0) __builtin_nontemporal_store(tot, slab + gw_floats(net) + job.gb_off + ob + li);
    }
  }
  DW_STAMP(true, 2, __builtin_readcyclecounter());
}


// Job L7 of a net in a bf16 backward (round 5): dW7 = dZ7^T H6 with
//   dZ7 = mask7 * (Wc^T dG + wsigma dsigma)
// recomputed per 32-row chunk from the saved [dS | dG] tensor (10 chunk blocks instead of dZ7's 16) and the ReLU sign words of
// H7 (one 1 KiB block per tile), so that the dX kernel does not write dZ7 at all.  Same pipeline as rc_job (one chunk ahead,
// double-buffered tile, one barrier per chunk); here the recomputed tile is the A operand.  Wave w owns out-block w of dZ7:
// the 9 live fragments of stage BS_DH7 of the packed backward stream stay in registers; operand chunk kc < 8 is block 2 + kc of
// [dS | dG] (the dG chunks), chunk 8 is block 0 (dsigma in slot 0) -- the dX kernel's own operand order, zero-initialised
// accumulator and mask_to_frags, so the tile is bit-identical to the dZ7 that kernel would have written.
struct Rc7Shape {
  static constexpr int NT = 16 + DSG_LD / 16 + 1;            // H6 blocks, [dS | dG] blocks, the sign-word block
  static constexpr int SLOT = NT * BLKP;
  static constexpr int TILE0 = 160 * 1024 - 2 * OPER_BYTES;
  static constexpr int NB = TILE0 / SLOT;                   // 4
  static constexpr int CW_HI = (NT + 7) / 8, N_HI = NT % 8;
  static constexpr int KL = 9;                               // live k-chunks of stage BS_DH7
  static_assert(NB >= 4 && N_HI != 0 && NT == 27, "rc7_job ring");
};

__device__ __forceinline__ void rc7_job(const DwArgs& a, const DwJob& job, int net, int split, int ksplit, int bid) {
  using S = Rc7Shape;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;
  const char* gb = (const char*)a.ws[net].t[job.b_tensor];             // H6
  const char* gd = (const char*)a.ws[net].t[T_DS];                     // [dS | dG], 10 blocks per tile
  const char* gm = (const char*)(a.masks[net] + (size_t)7 * (a.rows_padded / 32) * 64);     // sign words of H7
  const int64_t rows32 = (a.rows + 31) / 32 * 32;
  int64_t rps = (rows32 + ksplit - 1) / ksplit;
  rps = (rps + 31) / 32 * 32;
  const int64_t r_begin = split * rps;
  const int64_t r_end = r_begin + rps < rows32 ? r_begin + rps : rows32;
  const int nchunk = r_end > r_begin ? (int)((r_end - r_begin) / 32) : 0;
  DW_STAMP(true, 0, __builtin_readcyclecounter());
  DW_STAMP(true, 4, nchunk);
  DW_STAMP(true, 5, __builtin_amdgcn_s_getreg((3 << 11) | 20));
  const int wo = wave >> 2, wi = wave & 3;
  const bool do_bias = job.gb_off >= 0;
  f32x16 acc[4][2];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[x][0][r] = 0.f; acc[x][1][r] = 0.f; }
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  bf16x8 wt[S::KL];
#pragma unroll
  for (int kc = 0; kc < S::KL; ++kc)
    wt[kc] = *(const bf16x8*)((const char*)a.bwd_w[net] + (size_t)(bs_frag_off(BS_DH7) + kc * 8 + wave) * FRAG_BYTES + lane * 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)dw_smem;
  const int g = lane >> 4, a16 = lane & 15;
  const int lane_off = (g & 1) * BLKP + ((((g >> 1) * 8 + (a16 >> 2)) * 2 + (a16 & 1)) * 16) + ((a16 >> 1) & 1) * 8;
  if (nchunk > 0) {
    auto run = [&](auto cw_c) __attribute__((always_inline)) {
      constexpr int CW = decltype(cw_c)::value;
      int issued = 0;
      auto issue = [&]() __attribute__((always_inline)) {
        const int c = issued++;
        if (c >= nchunk) return;
        const size_t tile = (size_t)(r_begin >> 5) + (size_t)c;
        const uint32_t slot = lds_base + (uint32_t)(c % S::NB) * S::SLOT;
#pragma unroll
        for (int k = 0; k < CW; ++k) {
          const int id = wave + 8 * k;
          const char* src = id < 16 ? gb + (tile * 16 + id) * FRAG_BYTES
                          : id < 26 ? gd + (tile * (DSG_LD / 16) + (id - 16)) * FRAG_BYTES : gm + tile * FRAG_BYTES;
          glds16(src + lane * 16, slot + (uint32_t)id * BLKP);
        }
      };
      auto wait_chunk = [&](int k) __attribute__((always_inline)) {
        const int have = issued < nchunk ? issued : nchunk;
        const int younger = have - 1 - k;
        switch (younger <= 0 ? 0 : younger) {
          case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
          case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * CW) : "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CW) : "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * CW) : "memory"); break;
        }
      };
      auto recompute = [&](int k) __attribute__((always_inline)) {      // dZ7 of chunk k -> tile[k & 1], out-block `wave`
        const char* sl = dw_smem + (k % S::NB) * S::SLOT;
        const char* ds = sl + 16 * BLKP + ((2 * li + hi) << 4);
        f32x16 rc;
#pragma unroll
        for (int r = 0; r < 16; ++r) rc[r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < S::KL; ++kc)
          rc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt[kc], *(const bf16x8*)(ds + (kc < 8 ? 2 + kc : 0) * BLKP), rc, 0, 0, 0);
        const uint4 bits = *(const uint4*)(sl + 26 * BLKP + lane * 16);
        const uint32_t wsel[4] = {bits.x, bits.y, bits.z, bits.w};
        uint32_t act = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) act = (wave >> 1) == q ? ~wsel[q] : act;          // bit set = unit active
        char* dst = dw_smem + S::TILE0 + (k & 1) * OPER_BYTES + 2 * wave * BLKP + ((2 * li + hi) << 4);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          uint4 d;
          uint32_t* dp = (uint32_t*)&d;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            typedef float f32x2_ __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
            typedef short s16x2_ __attribute__((ext_vector_type(2)));
            const f32x2_ v = {rc[8 * hh + 2 * w], rc[8 * hh + 2 * w + 1]};
            const int j = (wave & 1) * 8 + hh * 4 + w;
            const uint32_t keep = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2_, act << j) >> (s16x2_){15, 15});
            dp[w] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_)) & keep;
          }
          *(uint4*)(dst + hh * BLKP) = d;
        }
      };
#pragma unroll
      for (int c = 0; c < S::NB - 1; ++c) issue();
      wait_chunk(0);
      __builtin_amdgcn_s_barrier();
      recompute(0);
      for (int c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk) wait_chunk(c + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue();
        if (c + 1 < nchunk) recompute(c + 1);
        compute_chunk<1, true>(S::TILE0 + (c & 1) * OPER_BYTES + lane_off, (c % S::NB) * S::SLOT + lane_off, wo, wi, 4, 2, do_bias, acc, bsum);
      }
    };
    if (wave < S::N_HI) run(std::integral_constant<int, S::CW_HI>{});
    else run(std::integral_constant<int, S::CW_HI - 1>{});
  }
  DW_STAMP(true, 1, __builtin_readcyclecounter());
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
        __builtin_nontemporal_store(acc[bo][bi][r], slab + job.gw_off + o * job.gw_ld + ib + li);
      }
    }
    if (do_bias && bo == wi) {
      const float tot = bsum[bo] + __shfl_xor(bsum[bo], 32, 64);
      if (hi == 0) __builtin_nontemporal_store(tot, slab + gw_floats(net) + job.gb_off + ob + li);
    }
  }
  DW_STAMP(true, 2, __builtin_readcyclecounter());
}

// workgroup -> (job of this launch, row slice): jobs in table order (net 0 then net 1), k slices each
struct DwSched {
  int wg_end[2 * DW_JOBS];       // exclusive prefix of workgroups per job
  int k[2 * DW_JOBS];
  int njobs, njobs0;
};
// bid: index of this workgroup among the launch's workgroups of its kind (full / narrow)
template <int P, bool FULL>
__device__ __forceinline__ void dw_body(const DwArgs& a, const DwSched& sc, int dbg, const int bid) {
  (void)0;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;
  int job_id = 0;
  for (int j = 0; j < sc.njobs - 1; ++j) job_id += bid >= sc.wg_end[j];
  const int split = bid - (job_id == 0 ? 0 : sc.wg_end[job_id - 1]), ksplit = sc.k[job_id];
  const int njobs0 = sc.njobs0;
  const int net = job_id < njobs0 ? 0 : 1;
  const DwJob job = (FULL ? c_full : c_narrow).jobs[net][net == 0 ? job_id : job_id - njobs0];
  DW_STAMP(FULL, 3, job_id);
  if constexpr (!FULL) {
    if (job.n_o == 256 && job.n_i == 64) narrow_job<P, 256, 64, 64>(a, job, net, split, ksplit, dbg, bid);
    else if (job.n_o == 256 && job.n_i == 96) narrow_job<P, 256, 96, 96>(a, job, net, split, ksplit, dbg, bid);
    else if (job.n_o == 256 && job.n_i == 320) narrow_job<P, 256, 320, 64>(a, job, net, split, ksplit, dbg, bid);
    else if (job.n_o == 256) narrow_job<P, 256, 352, 96>(a, job, net, split, ksplit, dbg, bid);
    else if (job.n_o == DSG_LD) narrow_job<P, DSG_LD, 288, 256>(a, job, net, split, ksplit, dbg, bid);
    else narrow_job<P, 32, 128, 128>(a, job, net, split, ksplit, dbg, bid);
    return;
  }
  if constexpr (P == 1) {
    if (a.h0_from_x && job.b_tensor == T_H0) {               // (wave-uniform: the whole workgroup runs one job)
      if (net == 0) rc_job<kpe(0)>(a, job, net, split, ksplit, bid);
      else rc_job<kpe(1)>(a, job, net, split, ksplit, bid);
      return;
    }
    if (a.h0_from_x && job.a_tensor == T_DZ0 + 7) {
      rc7_job(a, job, net, split, ksplit, bid);
      return;
    }
  }
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

#ifdef NERFPP_PROBES
  // dbg & 4: emulate the one-layer recompute in the jobs whose input would be unsaved (H0, H2, H6; dbg & 8: every full job);
  // dbg >> 4 = k-chunks of the recompute for the H0 job (default 16).  Garbage results.
  // dbg & 512: the H0 job alone, and its input DMA carries only the rc_k blocks of the encoded point it would read instead.
  const bool rc_h0only = (dbg & 512) != 0;
  const bool rc_emul = P == 1 && (dbg & 4) != 0 &&
                       (rc_h0only ? job.b_tensor == T_H0
                                  : ((dbg & 8) != 0 || job.b_tensor == T_H0 || job.b_tensor == T_H0 + 2 || job.b_tensor == T_H0 + 6));
  const int rc_k = (job.b_tensor == T_H0 && ((dbg >> 4) & 31) > 0) ? ((dbg >> 4) & 31) : 16;
  const bool rc_short = rc_emul && rc_h0only && rc_k <= 8;       // B operand: blocks [0, rc_k) only (waves < rc_k issue one)
  bf16x8 rc_w[16];
  if (rc_emul) {
#pragma unroll
    for (int kc = 0; kc < 16; ++kc) rc_w[kc] = *(const bf16x8*)(ga + ((size_t)kc * 8 + wave) * FRAG_BYTES + lane * 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  dbg &= 3;
#else
  constexpr bool rc_short = false;
  constexpr int rc_k = 16;
  (void)rc_short; (void)rc_k;
#endif
  // Ring pipeline: chunks c+1 .. c+NBUF-2 stay in flight while chunk c is consumed.  All VMEM ops of
  // this kernel's main loop are LDS-DMA loads (same type, in-order), so a COUNTED vmcnt is exact:
  // "at most k*DMA_PER_CHUNK outstanding" == "chunk c has landed" when k younger chunks were issued.
  // Raw s_barrier (a __syncthreads() would drain vmcnt to 0 and kill the overlap).
  {
    constexpr int NBUF = P == 1 ? 4 : 2;         // LDS ring depth (P=1: 4 x 36 KiB, P=2: 2 x 72 KiB)
    static_assert(NBUF * 2 * P * OPER_BYTES == DW_LDS_BYTES, "the full jobs' ring fills the launch's LDS");
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
#ifdef NERFPP_PROBES
        if (rc_short && op == 1 && blk >= rc_k) continue;
#endif
        glds16(src, buf + (op * P + pl) * OPER_BYTES + blk * BLKP);
      }
    };
#pragma unroll
    for (int c = 0; c < NBUF - 1; ++c) issue(c);
    for (int c = 0; c < nchunk; ++c) {
      const int younger = nchunk - 1 - c < NBUF - 2 ? nchunk - 1 - c : NBUF - 2;
#ifdef NERFPP_PROBES
      if (rc_short) {                               // 2 (+ 1 for waves < rc_k) DMA instructions per wave and chunk
        const int per = 2 + (wave < rc_k ? 1 : 0);
        if (younger >= 2) { if (per == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else if (younger == 1) { if (per == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else
#endif
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_CHUNK) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_CHUNK) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue(c + NBUF - 1);
      if (dbg == 1) continue;
#ifdef NERFPP_PROBES
      if constexpr (P == 1) {
        if (rc_emul) {      // (timing only: wave w recomputes out-block w of relu(W H_prev + b) for the chunk)
          f32x16 rc;
#pragma unroll
          for (int r = 0; r < 16; ++r) rc[r] = 0.25f;
          const int kmax = rc_k;
#pragma unroll
          for (int kc = 0; kc < 16; ++kc) {
            if (kc < kmax) {
              const bf16x8 bfrag = *(const bf16x8*)(dw_smem + (c % NBUF) * (2 * P * OPER_BYTES) + OPER_BYTES + kc * BLKP + lane * 16);
              rc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rc_w[kc], bfrag, rc, 0, 0, 0);
            }
          }
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            bf16x8 q;
#pragma unroll
            for (int t = 0; t < 8; ++t) q[t] = (__bf16)fmaxf(rc[8 * hh + t], 0.f);
            *(bf16x8*)(dw_smem + DW_LDS_BYTES + (2 * wave + hh) * 1024 + ((2 * li + hi) << 4)) = q;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
      }
#endif
      const lds_addr buf = (c % NBUF) * (2 * P * OPER_BYTES) + lane_off;
      compute_chunk<P, true>(buf, buf + P * OPER_BYTES, wo, wi, nbo, nbi, do_bias, acc, bsum);
    }
  }

  DW_STAMP(FULL, 1, __builtin_readcyclecounter());
  float* slab = a.slabs[net] + (size_t)split * gslab_floats(net);
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
        __builtin_nontemporal_store(acc[bo][bi][r], slab + job.gw_off + o * job.gw_ld + ib + li);
      }
    }
    if (do_bias && bo == wi) {                       // lane (li, hi) holds feature ob + li, half of the samples
      const float tot = bsum[bo] + __shfl_xor(bsum[bo], 32, 64);
      if (hi == 0) __builtin_nontemporal_store(tot, slab + gw_floats(net) + job.gb_off + ob + li);
    }
  }
  DW_STAMP(FULL, 2, __builtin_readcyclecounter());
}

// (Round 4 also ran both kinds in ONE launch -- full-job workgroups first, the narrow ones as CUs free up, the way the fg and
// bg MLP kernels of a level share a launch since then: 2.426 vs 2.411 ms per step in alternating runs on one box, i.e. 0.6 %
// SLOWER, the weight-gradient group 0.634 vs 0.620 ms; profiles/r04_pair_launch.md.  Two launches it stays.)
template <int P, bool FULL>
__global__ __launch_bounds__(512) void dw_kernel(DwArgs a, DwSched sc, int dbg) {
#ifndef NERFPP_PROBES
  dbg = 0;                 // (1: DMA only, 2: MFMA only -- diagnostic builds)
#endif
  dw_body<P, FULL>(a, sc, dbg, (int)blockIdx.x);
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
#ifdef NERFPP_PROBES
  const size_t lds = DW_LDS_BYTES + 16 * 1024;                          // + the recomputed tile of the emulation (dbg & 4)
#else
  const size_t lds = DW_LDS_BYTES;                                      // 144 KiB
#endif
  const int dbg = PROBE_GETENV("NERFPP_DW_DEBUG") ? atoi(PROBE_GETENV("NERFPP_DW_DEBUG")) : 0;   // 1: DMA only, 2: MFMA only, 4..: recompute emulation
  if (P == 1) {
    hipLaunchKernelGGL((dw_kernel<1, true>), gfull, block, a.h0_from_x ? (size_t)RC_LDS_BYTES : lds, st, a, sf, dbg);
    hipLaunchKernelGGL((dw_kernel<1, false>), gnarrow, block, (size_t)NARROW_LDS_BYTES, st, a, sn, dbg);
  } else {
    hipLaunchKernelGGL((dw_kernel<2, true>), gfull, block, lds, st, a, sf, dbg);
    hipLaunchKernelGGL((dw_kernel<2, false>), gnarrow, block, (size_t)NARROW_LDS_BYTES, st, a, sn, dbg);
  }
}
