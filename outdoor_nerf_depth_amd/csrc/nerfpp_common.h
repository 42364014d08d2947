// Internal layouts shared by the HIP kernels and the host-side table builders.
//
// Network: the NeRF++ MLPNet of nerf-methods/nerfplusplus/nerf_network.py:70-142 with
// D=8, W=256, skip at layer 4 (the only configuration the reference's configs use).
//
// Execution model of the fused MLP kernels ("samples on lanes"):
//   every layer is computed as  H_out^T [features x samples] = W [out x in] * H_in^T  with
//   v_mfma_f32_32x32x16_bf16.  A wave owns 32 samples; MFMA lane l = (j = l & 31, hi = l >> 5)
//   holds sample j.  The accumulator (C/D) layout of the 32x32 MFMA gives lane (j, hi), register
//   r of out-block ob the feature  ob*32 + (r&3) + 8*(r>>2) + 4*hi  of sample j.  Registers
//   8h..8h+7 of an out-block, converted to bf16, ARE the B operand of the next layer's k-chunk
//   2*ob+h -- with slot t of lane-half hi meaning input feature
//        f = 16*c + 8*(t>>2) + 4*hi + (t&3)                                   (kslot map)
//   instead of the natural 16*c + 8*hi + t.  The contraction does not care as long as the A
//   operand (the weights) uses the same map, so the weights are pre-packed that way and
//   activations chain from layer to layer in registers with no LDS/HBM round trip.
//
// Packed weight stream (per net, per direction): a flat sequence of 1-KiB "fragments", one per
// (k-chunk kc, out-block ob) in order [stage][kc][ob]; fragment byte 16*l.. holds lane l's 8 bf16
//   W_eff[ob*32 + (l&31)][kslot(kc, l>>5, t)],  t = 0..7.
// With P = 2 (split-bf16 "parity" precision) every fragment is followed by its `lo` twin
// (w - bf16(w) rounded to bf16).  Stages are padded so each has a multiple of BLK_FRAGS fragments;
// kernels stream the fragments through a double-buffered LDS ring in blocks of BLK_FRAGS.
#pragma once
#include <stdint.h>
#ifdef NERFPP_PROBES
#include <stdlib.h>
#endif

namespace nerfpp {

constexpr int WIDTH = 256;
constexpr int FRAG_BYTES = 1024;
constexpr int BLK_FRAGS = 16;
constexpr int DIR_REF = 27;        // view-direction encoding width (3 + 3*2*4)
constexpr int DIRW = 32;           // padded internal width (2 k-chunks)
constexpr int N_NET = 2;           // 0 = foreground (3-D points), 1 = background (x,y,z,1/r)

__host__ __device__ constexpr int pe_dim(int net) { return net == 0 ? 3 : 4; }
__host__ __device__ constexpr int pe_ref_ch(int net) { return net == 0 ? 63 : 84; }
__host__ __device__ constexpr int kpe(int net) { return net == 0 ? 4 : 6; }       // k-chunks
__host__ __device__ constexpr int kpew(int net) { return kpe(net) * 16; }        // 64 / 96

// ---- reference parameter layout of one MLPNet (state_dict / parameters() order) --------------
// base_layers.{0..7}.0.{weight,bias}, sigma_layers.0.*, base_remap_layers.0.*, rgb_layers.0.*,
// rgb_layers.2.*
enum RefTensor { RT_L0 = 0, RT_SIGMA = 8, RT_REMAP = 9, RT_RGB0 = 10, RT_RGB1 = 11, RT_COUNT = 12 };
__host__ __device__ constexpr int ref_out(int t) {
  return t < 8 ? 256 : t == RT_SIGMA ? 1 : t == RT_REMAP ? 256 : t == RT_RGB0 ? 128 : 3;
}
__host__ __device__ constexpr int ref_in(int net, int t) {
  return t == 0 ? pe_ref_ch(net) : t == 5 ? pe_ref_ch(net) + 256 : t < 8 ? 256
       : t == RT_SIGMA ? 256 : t == RT_REMAP ? 256 : t == RT_RGB0 ? 256 + DIR_REF : 128;
}
__host__ __device__ constexpr int ref_w_off(int net, int t) {       // offset of tensor t's weight
  int off = 0;
  for (int i = 0; i < t; ++i) off += ref_out(i) * ref_in(net, i) + ref_out(i);
  return off;
}
__host__ __device__ constexpr int ref_b_off(int net, int t) {
  return ref_w_off(net, t) + ref_out(t) * ref_in(net, t);
}
__host__ __device__ constexpr int net_params(int net) { return ref_w_off(net, RT_COUNT); }
constexpr int FG_PARAMS = net_params(0);      // 595 844
constexpr int BG_PARAMS = net_params(1);      // 606 596
constexpr int LEVEL_PARAMS = FG_PARAMS + BG_PARAMS;   // 1 202 440
static_assert(FG_PARAMS == 595844 && BG_PARAMS == 606596, "reference parameter count");

// ---- forward stages ---------------------------------------------------------------------------
// The remap layer (nerf_network.py:131, no activation) is FOLDED into the colour head in the MLP kernels: with
//   Wc = Wrgb0[:, :256] * Wremap  (128 x 256),   bc = brgb0 + Wrgb0[:, :256] * bremap
// G_pre = Wc h7 + Wrgb0[:, 256:] dirs + bc, and dH7 = Wc^T dG + wsigma dsigma -- one 256 x 256 GEMM per sample less in the
// forward and one in the dX chain.  Wc / bc are derived parameters, recomputed from the float32 masters at every re-pack
// (fold_remap_kernel); their gradients dWc = dG^T H7 = M and dbc = sum dG are what the weight-gradient GEMMs produce, and
// remap_fixup_kernel turns them into the gradients of the reference's parameters (exact chain rule: nerfpp_optim.hip).
// FS_REMAP keeps its place in the enumerations (bias slots, gradient slab) with zero weight fragments.
enum FwdStage { FS_L0 = 0, FS_L5 = 5, FS_REMAP = 8, FS_SIG = 9, FS_RGB0 = 10, FS_RGB1 = 11, FS_COUNT = 12 };
__host__ __device__ constexpr int fs_nob(int s) { return s <= FS_REMAP ? 8 : s == FS_RGB0 ? 4 : 1; }
__host__ __device__ constexpr int fs_nkc(int net, int s) {
  return s == FS_L0 ? kpe(net) : s == FS_L5 ? kpe(net) + 16 : s == FS_REMAP ? 0 : s < FS_REMAP ? 16
       : s == FS_SIG ? 16 : s == FS_RGB0 ? 20 : 16;
}
constexpr int DERIVED_WC = 128 * 256;               // derived parameters of one net: [Wc | bc], addressed in the pack tables
constexpr int DERIVED_FLOATS = DERIVED_WC + 128;    // as net_params(net) + index
__host__ __device__ constexpr int fs_frags(int net, int s) { return fs_nob(s) * fs_nkc(net, s); }
__host__ __device__ constexpr int fs_frag_off(int net, int s) {
  int off = 0;
  for (int i = 0; i < s; ++i) off += fs_frags(net, i);
  return off;
}
__host__ __device__ constexpr int fs_bias_off(int s) {              // in floats, D-layout order
  int off = 0;
  for (int i = 0; i < s; ++i) off += fs_nob(i) * 32;
  return off;
}
__host__ __device__ constexpr int fwd_frags(int net) { return fs_frag_off(net, FS_COUNT); }
constexpr int FWD_BIAS_FLOATS = fs_bias_off(FS_COUNT);
static_assert(fwd_frags(0) % BLK_FRAGS == 0 && fwd_frags(1) % BLK_FRAGS == 0, "block aligned");

// ---- backward (dX chain) stages -----------------------------------------------------------------
// B0: dG = Wrgb1^T dP | B1: (dR: folded away, no fragments) | B2: dH7 = Wc^T dG + wsig dsig (8 dG chunks + 2 for dsig) |
// B3..B9: dH_{l-1} = W_l^T dZ_l for l = 7..1 (l = 5 uses only the hidden-input columns)
enum BwdStage { BS_DG = 0, BS_DR = 1, BS_DH7 = 2, BS_COUNT = 10 };
__host__ __device__ constexpr int bs_nob(int s) { return s == BS_DG ? 4 : 8; }
__host__ __device__ constexpr int bs_nkc(int s) { return s == BS_DG ? 4 : s == BS_DR ? 0 : s == BS_DH7 ? 10 : 16; }
__host__ __device__ constexpr int bs_frags(int s) { return bs_nob(s) * bs_nkc(s); }
__host__ __device__ constexpr int bs_frag_off(int s) {
  int off = 0;
  for (int i = 0; i < s; ++i) off += bs_frags(i);
  return off;
}
constexpr int BWD_FRAGS = bs_frag_off(BS_COUNT);
static_assert(BWD_FRAGS % BLK_FRAGS == 0, "block aligned");
__host__ __device__ constexpr int bs_layer(int s) { return 10 - s; }     // s = 3..9 -> l = 7..1

// Precision ids of the MLP kernels (include/nerfpp_hip.h: NERFPP_PREC_*):
//   1 bf16     both operands rounded to bf16 once, one v_mfma_f32_32x32x16_bf16 per product
//   2 split    both operands hi + lo in bf16, three passes hi*hi + hi*lo + lo*hi
//   3 fp16x2w  (forward kernels only) weights hi + lo in fp16, activations rounded to fp16 ONCE, two passes
//              Wh*A + Wl*A of v_mfma_f32_32x32x16_f16 (same rate as bf16): the weight side is exact to ~2^-21, the
//              activation side carries one 2^-12 rounding per layer.  AN INTERMEDIATE PRECISION: at initialisation that
//              rounding averages out over the 256-term sums and every returned tensor is inside north_star's 1e-4 with 2x
//              margin (tools/operand_format_study.py); on trained weights it does not (rendered rgb 3-4e-4 from float32,
//              single per-sample weights up to 1e-2: profiles/r05_fp16x2w_trained_weights.json) -- a factor 10 tighter
//              than precision 1, a factor 50 looser than precision 2, at 54 % of precision 2's forward time.  The 1e-4
//              clause stays with precision 2.  Saved tensors are written as bf16 (one plane), i.e. in precision 1's
//              workspace layout.
// w_planes: 1 KiB fragments per (k-chunk, out-block) in the packed weight streams; a_planes: 16-byte register images
// per activation chunk (and planes of the saved tensors).
__host__ __device__ constexpr int w_planes(int P) { return P == 1 ? 1 : 2; }
__host__ __device__ constexpr int a_planes(int P) { return P == 2 ? 2 : 1; }
// packed stream sizes in bytes for precision P
__host__ __device__ constexpr size_t fwd_stream_bytes(int net, int P) { return (size_t)fwd_frags(net) * FRAG_BYTES * w_planes(P); }
__host__ __device__ constexpr size_t bwd_stream_bytes(int P) { return (size_t)BWD_FRAGS * FRAG_BYTES * w_planes(P); }

// the kslot map: feature index of slot t of lane-half hi in k-chunk c
__host__ __device__ constexpr int kslot(int c, int hi, int t) { return 16 * c + 8 * (t >> 2) + 4 * hi + (t & 3); }
// feature index of accumulator register r of lane-half hi in out-block ob (32x32 MFMA C/D layout)
__host__ __device__ constexpr int dfeat(int ob, int hi, int r) { return ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ---- positional-encoding internal order --------------------------------------------------------
// Lane-half hi of a sample computes frequencies [5*hi, 5*hi+5) of every input dimension plus two
// identity terms; its m-th value (m = 8*c + t over its k-chunks) is, for m < 10*D,
//   freq k = 5*hi + m/(2D), dim d = (m % 2D)/2, sin if m even else cos,
// then (m = 10D, 10D+1) the raw inputs 2*hi, 2*hi+1 (if < D), then zero padding.
// Returns the index in the reference Embedder output (nerf_network.py:42-60) or -1.
__host__ __device__ constexpr int pe_ref_of_lane_slot(int D, int hi, int m) {
  const int nf = 10 * D;
  if (m < nf) {
    const int k = 5 * hi + m / (2 * D), d = (m % (2 * D)) / 2, s = m & 1;
    return D + k * 2 * D + s * D + d;
  }
  const int id = 2 * hi + (m - nf);
  return (m - nf) < 2 && id < D ? id : -1;
}
// view-direction encoding: lane-half hi computes frequencies 2hi, 2hi+1 (12 values), then raw
// inputs 2hi, 2hi+1 (if < 3), then padding; 16 values per lane-half.
__host__ __device__ constexpr int dir_ref_of_lane_slot(int hi, int m) {
  if (m < 12) {
    const int k = 2 * hi + m / 6, d = (m % 6) / 2, s = m & 1;
    return 3 + k * 6 + s * 3 + d;
  }
  const int id = 2 * hi + (m - 12);
  return (m - 12) < 2 && id < 3 ? id : -1;
}
// internal feature f (column of the saved X / DIRX tensors, kslot order) -> (hi, m)
__host__ __device__ constexpr int feat_hi(int f) { return ((f % 16) / 4) % 2; }
__host__ __device__ constexpr int feat_m(int f) { return 8 * (f / 16) + 4 * ((f % 16) / 8) + (f % 4); }
__host__ __device__ constexpr int pe_ref_of_feat(int net, int f) { return pe_ref_of_lane_slot(pe_dim(net), feat_hi(f), feat_m(f)); }
__host__ __device__ constexpr int dir_ref_of_feat(int f) { return dir_ref_of_lane_slot(feat_hi(f), feat_m(f)); }

// ---- saved tensors (training): FRAGMENT-MAJOR bf16, one plane per precision part ------------------------------
// A tensor of `ld` columns (a multiple of 16) over rows_padded rows (a multiple of 256) is stored as
// [rows_padded / 32 tiles][ld / 16 chunks][1 KiB block]; the block of (tile, chunk c) holds at byte (2 j + hi) * 16 the
// 8 bf16 that lane (j, hi) of the producing wave holds for row 32 * tile + j: features 16 c + kslot(0, hi, t), t = 0..7,
// i.e. the 8-byte piece q = 2 (t >> 2) + hi carries features 16 c + 4 q .. 4 q + 3 in natural order.  It is exactly the
// accumulator-layout register image of the fused MLP kernels (one contiguous 1 KiB wave-store per chunk, no transposition)
// and the weight-gradient kernel reads its sample-major MFMA fragments from it with ds_read_b64_tr_b16.  A "column offset"
// of 16 k columns is k blocks (T_DG = T_DS + 2 blocks).  Same bytes as a row-major [rows_padded][ld] tensor.
enum Tensor {
  T_X = 0, T_H0 = 1, /* .. T_H7 = 8 */ T_R = 9, T_G = 10, T_DIRX = 11,
  T_DZ0 = 12, /* .. T_DZ7 = 19 */ T_DR = 20, T_DS = 21, T_DG = 22, T_DP = 23, T_COUNT = 24
};
// dS (32 columns) and dG (128 columns) share one tensor of 160 columns: T_DS points at column 0 (chunks 0, 1),
// T_DG at column 32 (chunk 2) of it, so the weight-gradient job that contracts both with H7 reads H7 once.
constexpr int DSG_LD = 160, DG_COL0 = 32;
__host__ __device__ constexpr int tensor_ld(int net, int t) {
  return t == T_X ? kpew(net) : (t >= T_H0 && t <= T_R) ? 256 : t == T_G ? 128 : t == T_DIRX ? 32
       : (t >= T_DZ0 && t <= T_DR) ? 256 : (t == T_DS || t == T_DG) ? DSG_LD : 32;
}
__host__ __device__ constexpr int tensor_off_cols(int net, int t) {   // column offset in a row "super-struct"
  int off = 0;
  for (int i = 0; i < t; ++i) off += tensor_ld(net, i);
  return off;
}
__host__ __device__ constexpr int ws_cols(int net) { return tensor_off_cols(net, T_COUNT); }

// ---- weight-gradient GEMMs: GW[stage] = dZ^T * input --------------------------------------------
// one entry per forward stage with parameters; I spans 1 or 2 input tensors
__host__ __device__ constexpr int gw_O(int s) { return s <= FS_REMAP ? 256 : s == FS_SIG ? 32 : s == FS_RGB0 ? 128 : 32; }
__host__ __device__ constexpr int gw_I(int net, int s) {
  return s == FS_L0 ? kpew(net) : s == FS_L5 ? kpew(net) + 256 : s <= FS_SIG ? 256 : s == FS_RGB0 ? 256 + DIRW : 128;
}
__host__ __device__ constexpr int gw_off(int net, int s) {            // floats
  int off = 0;
  for (int i = 0; i < s; ++i) off += gw_O(i) * gw_I(net, i);
  return off;
}
__host__ __device__ constexpr int gw_floats(int net) { return gw_off(net, FS_COUNT); }
__host__ __device__ constexpr int gb_off(int s) {
  int off = 0;
  for (int i = 0; i < s; ++i) off += gw_O(i);
  return off;
}
constexpr int GB_FLOATS = gb_off(FS_COUNT);
__host__ __device__ constexpr int gslab_floats(int net) { return gw_floats(net) + GB_FLOATS; }
__host__ __device__ constexpr int gw_dz_tensor(int s) {
  return s < 8 ? T_DZ0 + s : s == FS_REMAP ? T_DR : s == FS_SIG ? T_DS : s == FS_RGB0 ? T_DG : T_DP;
}

struct DwJob {       // one weight-gradient GEMM: the whole (<= 256 x 256) output, contracted over a row slice
  int16_t a_tensor, b_tensor;     // dZ tensor, input tensor
  int16_t b_tensor2, n_i1;        // the input may span two tensors: columns [0, n_i1) from b_tensor, [n_i1, n_i) from b_tensor2
  int16_t o0, i0;                 // first column in each tensor
  int16_t n_o, n_i;               // valid columns (multiples of 32; 256 x 256 = "full", everything else "narrow")
  int32_t gw_off;                 // offset of element (o0, i_global0) in the slab
  int16_t gw_ld;                  // row stride of this stage's GW block
  int16_t gb_off;                 // offset of bias o0 in the slab's bias part, or -1
  // optional second output segment: out-blocks >= o_split go to (gw_off2, gw_ld2, gb_off2), rows re-based
  int16_t o_split;                // in 32-row blocks; 0 = single segment
  int16_t gw_ld2, gb_off2;
  int32_t gw_off2;
};
constexpr int DW_JOBS = 10;                  // per net (build_all_jobs order)
constexpr int DW_KMAX = 64;                  // slabs allocated per net; a job uses the first k_job of them

struct JobTable {
  DwJob jobs[N_NET][DW_JOBS];
  int count[N_NET];
};
constexpr void add_job(JobTable& jt, int net, int s, int b_tensor, int icol, bool bias) {
  DwJob j{};
  j.a_tensor = (int16_t)gw_dz_tensor(s);
  j.b_tensor = (int16_t)b_tensor;
  j.b_tensor2 = (int16_t)b_tensor;
  j.o0 = 0;
  j.i0 = 0;
  j.n_o = (int16_t)gw_O(s);
  j.n_i = (int16_t)tensor_ld(net, b_tensor);
  j.n_i1 = j.n_i;
  j.gw_off = gw_off(net, s) + icol;
  j.gw_ld = (int16_t)gw_I(net, s);
  j.gb_off = (int16_t)(bias ? gb_off(s) : -1);
  jt.jobs[net][jt.count[net]++] = j;
}
// job index per net: 0 L0 | 1-4 L1-L4 | 5 L5 = dZ5^T [X | H4] | 6,7 L6,L7 | 8 [dS | dG]^T [H7 | DIRX] = sigma weights + M
// (M = dG^T H7, see remap_fixup_kernel) + the view-direction columns of rgb0 | 9 rgb1.
// Jobs that share an operand are ONE job since round 4 (dZ5 was read by an encoded-point job and an h4 job, dG by the M job and
// a view-direction job: 768 of 9.98 KB per row and net).
constexpr JobTable build_all_jobs() {
  JobTable jt{};
  for (int net = 0; net < N_NET; ++net) {
    jt.count[net] = 0;
    for (int s = 0; s < FS_COUNT; ++s) {
      if (s == FS_L0) add_job(jt, net, s, T_X, 0, true);
      else if (s == FS_L5) {
        add_job(jt, net, s, T_X, 0, true);           // columns [0, kpew) of the stage's input are the encoded point ...
        DwJob& m = jt.jobs[net][jt.count[net] - 1];
        m.b_tensor2 = (int16_t)(T_H0 + 4);           // ... the other 256 are h4
        m.n_i = (int16_t)(kpew(net) + 256);
      } else if (s < 8) add_job(jt, net, s, T_H0 + s - 1, 0, true);
      else if (s == FS_REMAP) continue;            // dW_remap = Wrgb0r^T * M, derived in remap_fixup_kernel
      else if (s == FS_SIG) continue;              // merged into the next job
      else if (s == FS_RGB0) {
        // [dS | dG]^T * [H7 | DIRX]: out-rows 0..31 -> sigma stage (its 256 input columns; the dS^T DIRX block is not
        // written), rows 32..159 -> rgb0: columns 0..255 = M (NOT dG^T * R; fixed up after the slab sum), 256..287 =
        // the view-direction columns
        add_job(jt, net, FS_SIG, T_H0 + 7, 0, true);
        DwJob& m = jt.jobs[net][jt.count[net] - 1];
        m.n_o = DSG_LD;
        m.b_tensor2 = (int16_t)T_DIRX;
        m.n_i = (int16_t)(256 + DIRW);
        m.o_split = 1;
        m.gw_off2 = gw_off(net, FS_RGB0);
        m.gw_ld2 = (int16_t)gw_I(net, FS_RGB0);
        m.gb_off2 = (int16_t)gb_off(FS_RGB0);
      } else add_job(jt, net, s, T_G, 0, true);
    }
  }
  return jt;
}
constexpr bool dw_job_is_full(const DwJob& j) { return j.n_o == 256 && j.n_i == 256; }
// which job wrote element `src` of a gradient slab (-1: nobody -- the remap stage, derived later)
struct SlabMap { int gw[N_NET][FS_COUNT + 1]; int gb[FS_COUNT + 1]; int gi[N_NET][FS_COUNT]; };
constexpr SlabMap make_slab_map() {
  SlabMap m{};
  for (int s = 0; s <= FS_COUNT; ++s) {
    m.gb[s] = gb_off(s);
    for (int net = 0; net < N_NET; ++net) {
      m.gw[net][s] = gw_off(net, s);
      if (s < FS_COUNT) m.gi[net][s] = gw_I(net, s);
    }
  }
  return m;
}
__host__ __device__ constexpr int slab_job_index(const SlabMap& m, int net, int src) {
  int s = 0;                                         // stage of the element: weights first, then the biases
  if (src >= m.gw[net][FS_COUNT]) {
    const int b = src - m.gw[net][FS_COUNT];
    for (int t = 1; t < FS_COUNT; ++t) s += m.gb[t] <= b;
  } else {
    for (int t = 1; t < FS_COUNT; ++t) s += m.gw[net][t] <= src;
  }
  if (s < 8) return s;
  if (s == FS_REMAP) return -1;
  if (s == FS_SIG || s == FS_RGB0) return 8;
  return 9;
}

// How many row slices (= workgroups = slabs) each job gets.  The full 256x256 jobs all stream the
// same bytes per row; the narrow ones get slices in proportion to their measured cost per 32-row chunk, so that
// every workgroup of a launch finishes at about the same time and each launch fills the 256 CUs once.
// (Until round 4 the narrow weights were the jobs' bytes per row; per-workgroup timestamps -- profiles/r04_dw_stamps.md --
// showed the [dS | dG]^T H7 job, 40 output blocks dealt round-robin, at 6.9 cycles per column-chunk against 5.0-5.5 for the
// others: its 38 slices finished 30 % after everybody else.  The per-shape loops of nerfpp_dw.hip: narrow_job brought that job
// to 4.9 and the table below is their measurement.)
struct DwPlan { int k[N_NET][DW_JOBS]; };
constexpr int dw_narrow_cost(const DwJob& j) {        // shader cycles per 32-row tile, bf16 (dw_kernel<1, false>, per-shape loops)
  return j.n_o == DSG_LD ? 2258 : j.n_o == 256 ? (j.n_i == 64 ? 1426 : j.n_i == 96 ? 1534 : j.n_i == 320 ? 2927 : 3133) : 775;
}
// rc: the two L1 jobs (input H0) recompute their input from the encoded point (nerfpp_dw.hip: rc_job).  They read 20 / 22 instead
// of 32 KiB per 32-row chunk but issue 20 / 22 instead of 16 MFMAs per wave for it, which makes them MATRIX-bound where the plain
// jobs are HBM-bound (a chunk takes ~1.2x as long): they get DW_RC_SLICES slices each, the ten plain full jobs share the rest of
// the 256 workgroups.  Measured on one box, step in ms: 16 slices 2.18, 23: 2.052, 26: 2.026, 30: 2.022, 34: 2.033; no recompute
// (H0 saved): 2.047.
constexpr int DW_RC_SLICES = 28;
// ... and the two L7 jobs recompute dZ7 from [dS | dG] (rc7_job: 27 instead of 32 KiB per chunk, 25 instead of 16 MFMAs per wave).
// Measured on one box, step in ms: 24 slices 2.045, 30: 1.998, 36: 2.006, 42: 2.038; dZ7 saved: 2.006 -- the dX kernel gains what the
// matrix-bound job costs, within 0.4 %; kept for the 512 B per row it takes out of the HBM traffic.
constexpr int DW_RC7_SLICES = 30;
inline DwPlan dw_plan(int64_t rows, bool rc = false) {
  const JobTable jt = build_all_jobs();
  int64_t cap = rows / 512;
  cap = cap < 1 ? 1 : (cap > DW_KMAX ? DW_KMAX : cap);
  int n_full = 0, w_narrow = 0;
  for (int net = 0; net < N_NET; ++net)
    for (int j = 0; j < jt.count[net]; ++j) {
      if (dw_job_is_full(jt.jobs[net][j])) ++n_full;
      else w_narrow += dw_narrow_cost(jt.jobs[net][j]);
    }
  DwPlan pl{};
  // full jobs: equal shares of 256 workgroups.  Narrow jobs: largest-remainder apportionment of exactly
  // <= 256 workgroups in proportion to their cost per chunk (a 257th workgroup would wait for a whole
  // second round of the launch).
  int64_t rem[N_NET][DW_JOBS] = {};
  int used = 0;
  for (int net = 0; net < N_NET; ++net)
    for (int j = 0; j < jt.count[net]; ++j) {
      const DwJob& job = jt.jobs[net][j];
      int64_t k;
      if (dw_job_is_full(job)) {
        k = 256 / n_full;
        if (rc) k = job.b_tensor == T_H0 ? DW_RC_SLICES : job.a_tensor == T_DZ0 + 7 ? DW_RC7_SLICES
                                                        : (256 - 2 * DW_RC_SLICES - 2 * DW_RC7_SLICES) / (n_full - 4);
#ifdef NERFPP_PROBES
        // (timing experiment: NERFPP_DW_RC_K = slices of the full jobs that would recompute their input, the others share the rest)
        if (const char* e = getenv("NERFPP_DW_RC_K")) {
          const int kr = atoi(e);
          const bool rcj = job.b_tensor == T_H0 || job.b_tensor == T_H0 + 2 || job.b_tensor == T_H0 + 6;
          if (const char* e0 = getenv("NERFPP_DW_RC_K0")) {            // the H0 job separately (kr = 0: it alone differs)
            const int k0 = atoi(e0);
            if (kr == 0) k = job.b_tensor == T_H0 ? k0 : (256 - 2 * k0) / 10;
            else k = job.b_tensor == T_H0 ? k0 : rcj ? kr : (256 - 2 * k0 - 4 * kr) / 6;
          } else k = rcj ? kr : (256 - 6 * kr) / 6;
        }
#endif
      } else {
        const int64_t num = (int64_t)256 * dw_narrow_cost(job);
        k = num / w_narrow;
        rem[net][j] = num - k * w_narrow;
      }
      k = k < 1 ? 1 : (k > cap ? cap : k);
      pl.k[net][j] = (int)k;
      if (!dw_job_is_full(job)) used += (int)k;
    }
  for (int left = 256 - used; left > 0; --left) {            // hand the leftover workgroups to the largest remainders
    int bn = -1, bj = -1;
    for (int net = 0; net < N_NET; ++net)
      for (int j = 0; j < jt.count[net]; ++j)
        if (!dw_job_is_full(jt.jobs[net][j]) && pl.k[net][j] < cap && rem[net][j] >= 0 &&
            (bn < 0 || rem[net][j] > rem[bn][bj])) { bn = net; bj = j; }
    if (bn < 0) break;
    ++pl.k[bn][bj];
    rem[bn][bj] = -1;
  }
  return pl;
}

// ---- counter-based RNG for the sampling uniforms ---------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11; the generator family torch's CUDA/HIP backend uses), keyed by the
// caller's 64-bit seed; counter = (element index lo, hi, stream id, step).  Stream ids follow the order
// the reference consumes torch's RNG per step: 0 = rand_like(fg_z), 1 = rand_like(bg_z)
// (perturb_samples, ddp_train_nerf.py:444,449), 2 = fg sample_pdf u, 3 = bg sample_pdf u (:455,463); 4 = the pixel draw of
// the ray-batch sampler (nerf_sample_ray_split.py:178 np.random.choice, here nerfpp_sample_pixels).
// uniform = (word0 >> 8) * 2^-24 in [0, 1) like torch.rand's float32 mapping.  The numpy oracle carries
// the same function (oracle/nerfpp_oracle.py: philox_uniform), checked bit-for-bit in the GPU tests.
struct RngKey { uint32_t k0, k1, step_lo, enabled; };
__host__ __device__ inline RngKey make_rng_key(uint64_t seed, uint64_t step, bool enabled) {
  RngKey k;
  k.k0 = (uint32_t)seed; k.k1 = (uint32_t)(seed >> 32); k.step_lo = (uint32_t)step; k.enabled = enabled ? 1u : 0u;
  return k;
}
__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
__host__ __device__ inline uint32_t philox_word(const RngKey& key, uint32_t stream_id, uint64_t idx) {
  uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = stream_id, c3 = key.step_lo;
  uint32_t k0 = key.k0, k1 = key.k1;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = mulhi32(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const uint32_t h1 = mulhi32(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c0;
}
__host__ __device__ inline float philox_uniform(const RngKey& key, uint32_t stream_id, uint64_t idx) {
  uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = stream_id, c3 = key.step_lo;
  uint32_t k0 = key.k0, k1 = key.k1;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = mulhi32(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const uint32_t h1 = mulhi32(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return (float)(c0 >> 8) * 5.9604644775390625e-8f;
}

}  // namespace nerfpp
