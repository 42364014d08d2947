// extern "C" entry points of libnerfpp_hip.so (declared in include/nerfpp_hip.h).
#include "probe_env.h"
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/nerfpp_hip.h"
#include "nerfpp_common.h"
#include "nerfpp_kernels.h"

using namespace nerfpp;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(NERFPP_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return NERFPP_OK;
}

#define REQUIRE(cond, what) \
  do { if (!(cond)) return fail(NERFPP_ERR_ARG, "%s: requirement failed: %s", __func__, what); } while (0)

constexpr size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// single-plane workspaces (forward precisions 1 and 3): H0 is not a saved tensor, its weight-gradient job recomputes it from X
bool h0_recomputed(int WP) { return a_planes(WP) == 1; }

// ---- level tables layout (int32 elements) -------------------------------------------------------
struct TblLayout { int64_t fwd[2], bias[2], bwd[2], unpack[2], total; };
TblLayout tbl_layout() {
  TblLayout L{};
  int64_t off = 0;
  for (int net = 0; net < N_NET; ++net) {
    L.fwd[net] = off; off += (int64_t)fwd_frags(net) * 512;
    L.bias[net] = off; off += FWD_BIAS_FLOATS;
    L.bwd[net] = off; off += (int64_t)BWD_FRAGS * 512;
    L.unpack[net] = off; off += net_params(net);
  }
  L.total = off;
  return L;
}

// ---- packed buffer layout (bytes) ---------------------------------------------------------------
struct PackLayout { size_t fwd[2], bwd[2], bias[2], derived[2], total; };
PackLayout pack_layout(int P) {
  PackLayout L{};
  size_t off = 0;
  for (int net = 0; net < N_NET; ++net) {
    L.fwd[net] = off; off = align_up(off + fwd_stream_bytes(net, P), 256);
    L.bwd[net] = off; off = align_up(off + bwd_stream_bytes(P), 256);
    L.bias[net] = off; off = align_up(off + FWD_BIAS_FLOATS * sizeof(float), 256);
    L.derived[net] = off; off = align_up(off + DERIVED_FLOATS * sizeof(float), 256);     // [Wc | bc] in float32 (scratch of the pack)
  }
  L.total = off;
  return L;
}

// ---- workspace layout (bytes) -------------------------------------------------------------------
struct WsLayout {
  size_t tensor[2][T_COUNT], out_raw[2], depth_real, d_out[2], slabs[2], masks[2], fix_m[2], total;
  int64_t rows, rows_padded;
  int ksplit;
};
int choose_ksplit(int64_t rows) {              // slabs to allocate: the most any job of the plan uses
  int k = 1;
  for (int rc = 0; rc < 2; ++rc) {
    const DwPlan pl = dw_plan(rows, rc != 0);
    for (int net = 0; net < N_NET; ++net)
      for (int j = 0; j < DW_JOBS; ++j) k = pl.k[net][j] > k ? pl.k[net][j] : k;
  }
  return k;
}
WsLayout ws_layout(int n_rays, int S, int P, bool training) {
  WsLayout L{};
  L.rows = (int64_t)n_rays * S;
  L.rows_padded = (int64_t)align_up((size_t)L.rows, 256);
  L.ksplit = choose_ksplit(L.rows);
  size_t off = 0;
  for (int net = 0; net < N_NET; ++net) {
    L.out_raw[net] = off; off = align_up(off + (size_t)L.rows_padded * 16, 256);
    if (training) {
      L.d_out[net] = off; off = align_up(off + (size_t)L.rows_padded * 16, 256);
      for (int t = 0; t < T_COUNT; ++t) {
        L.tensor[net][t] = off;
        if (t == T_R || t == T_DR) continue;     // never materialised (remap_fixup_kernel derives their gradients)
        if (t == T_DG) continue;                 // columns 32..159 of the [dS | dG] tensor allocated as T_DS
        if (t == T_H0 && h0_recomputed(P)) continue;   // recomputed from X inside its weight-gradient job (nerfpp_dw.hip: rc_job)
        if (t == T_DZ0 + 7 && h0_recomputed(P)) continue;   // single-plane workspace = bf16 backward: rc7_job recomputes dZ7 from [dS | dG]
        off = align_up(off + (size_t)L.rows_padded * tensor_ld(net, t) * 2 * a_planes(P), 256);
      }
      L.slabs[net] = off; off = align_up(off + (size_t)L.ksplit * gslab_floats(net) * 4, 256);
      L.masks[net] = off; off = align_up(off + (size_t)9 * (L.rows_padded / 32) * 64 * 16, 256);
      L.fix_m[net] = off; off = align_up(off + (size_t)128 * 256 * 4, 256);
    }
  }
  L.depth_real = off; off = align_up(off + (size_t)L.rows_padded * 4, 256);
  L.total = off;
  return L;
}

NetWs make_netws(char* ws, const WsLayout& L, int net) {
  NetWs w;
  for (int t = 0; t < T_COUNT; ++t) w.t[t] = (__bf16*)(ws + L.tensor[net][t]);
  w.t[T_DG] = w.t[T_DS] + (DG_COL0 / 16) * (FRAG_BYTES / 2);      // fragment-major: column 32 = chunk block 2 of every tile
  return w;
}

bool prec_ok(int P) { return P == NERFPP_PREC_BF16 || P == NERFPP_PREC_SPLIT_BF16; }               // backward kernels
bool prec_ok_fwd(int P) { return prec_ok(P) || P == NERFPP_PREC_FP16X2W; }                          // forward kernels, packs, workspaces

// The foreground and background nets of a level are independent until the compositing kernel: their MLP kernels run as ONE
// launch (fg tiles first, bg tiles as CUs free up), one start-up and one tail instead of two.  Probes build only:
// NERFPP_MLP_SPLIT=1 launches the same kernel once per net (the two-launch form of rounds 1-3) for A/B runs.
bool split_nets() {
  static const bool on = PROBE_GETENV("NERFPP_MLP_SPLIT") != nullptr;
  return on;
}

}  // namespace

extern "C" {

const char* nerfpp_last_error(void) { return g_err; }
int nerfpp_abi_version(void) { return NERFPP_ABI_VERSION; }

int nerfpp_intersect_sphere(void* stream, int n_rays, const float* ray_o, const float* ray_d, float* fg_far,
                            int* bad_count) {
  REQUIRE(n_rays > 0 && ray_o && ray_d && fg_far, "non-null inputs, n_rays > 0");
  launch_intersect_sphere((hipStream_t)stream, n_rays, ray_o, ray_d, fg_far, bad_count);
  return check_launch("intersect_sphere");
}

int nerfpp_sample_coarse(void* stream, int n_rays, int n_samples, const float* ray_o, const float* ray_d,
                         const float* min_depth, const float* t_rand_fg, const float* t_rand_bg, float* fg_far,
                         float* fg_z, float* bg_z, int* bad_count) {
  REQUIRE(n_rays > 0 && n_samples >= 2 && n_samples <= NERFPP_MAX_SAMPLES, "2 <= n_samples <= 256");
  REQUIRE(ray_o && ray_d && min_depth && fg_far && fg_z && bg_z, "non-null pointers");
  launch_sample_coarse((hipStream_t)stream, n_rays, n_samples, ray_o, ray_d, min_depth, t_rand_fg, t_rand_bg,
                       fg_far, fg_z, bg_z, bad_count, make_rng_key(0, 0, false));
  return check_launch("sample_coarse");
}

int nerfpp_sample_coarse_rng(void* stream, int n_rays, int n_samples, const float* ray_o, const float* ray_d,
                             const float* min_depth, uint64_t seed, uint64_t step, float* fg_far, float* fg_z,
                             float* bg_z, int* bad_count) {
  REQUIRE(n_rays > 0 && n_samples >= 2 && n_samples <= NERFPP_MAX_SAMPLES, "2 <= n_samples <= 256");
  REQUIRE(ray_o && ray_d && min_depth && fg_far && fg_z && bg_z, "non-null pointers");
  launch_sample_coarse((hipStream_t)stream, n_rays, n_samples, ray_o, ray_d, min_depth, nullptr, nullptr, fg_far,
                       fg_z, bg_z, bad_count, make_rng_key(seed, step, true));
  return check_launch("sample_coarse_rng");
}

int nerfpp_rng_uniform(void* stream, uint64_t seed, uint64_t step, int stream_id, int64_t n, float* out) {
  REQUIRE(n > 0 && out && stream_id >= 0 && stream_id <= 3, "stream_id in 0..3, non-null output");
  launch_rng_uniform((hipStream_t)stream, make_rng_key(seed, step, true), (uint32_t)stream_id, n, out);
  return check_launch("rng_uniform");
}

int nerfpp_perturb_samples(void* stream, int n_rays, int n_samples, const float* z_vals, const float* t_rand,
                           float* out) {
  REQUIRE(n_rays > 0 && n_samples >= 2 && z_vals && t_rand && out, "non-null pointers");
  launch_perturb((hipStream_t)stream, n_rays, n_samples, z_vals, t_rand, out);
  return check_launch("perturb_samples");
}

int nerfpp_sample_pdf(void* stream, int n_rays, int n_bins_m, int n_new, const float* bins, const float* weights,
                      const float* u, float* samples, int64_t* above_inds) {
  REQUIRE(n_rays > 0 && n_bins_m >= 1 && n_bins_m + 1 <= 512 && n_new >= 1 && n_new <= 512, "sizes");
  REQUIRE(bins && weights && samples, "non-null pointers");
  launch_sample_pdf((hipStream_t)stream, false, n_rays, n_bins_m, n_new, bins, weights, u, samples, above_inds,
                    nullptr);
  return check_launch("sample_pdf");
}

int nerfpp_sample_fine(void* stream, int n_rays, int s_old, int n_new, const float* z_old, const float* weights,
                       const float* u, float* z_merged, float* samples, int64_t* above_inds) {
  REQUIRE(n_rays > 0 && s_old >= 3 && n_new >= 1 && s_old + n_new <= 512, "3 <= s_old, s_old + n_new <= 512");
  REQUIRE(z_old && weights && z_merged, "non-null pointers");
  launch_sample_pdf((hipStream_t)stream, true, n_rays, s_old - 2, n_new, z_old, weights, u, samples, above_inds,
                    z_merged);
  return check_launch("sample_fine");
}

int nerfpp_sample_fine_pair(void* stream, int n_rays, int s_old, int n_new, const float* fg_z_old,
                            const float* fg_weights, const float* fg_u, float* fg_z_merged, const float* bg_z_old,
                            const float* bg_weights, const float* bg_u, float* bg_z_merged) {
  REQUIRE(n_rays > 0 && s_old >= 3 && n_new >= 1 && s_old + n_new <= 512, "3 <= s_old, s_old + n_new <= 512");
  REQUIRE(fg_z_old && fg_weights && fg_z_merged && bg_z_old && bg_weights && bg_z_merged, "non-null pointers");
  const float* z[2] = {fg_z_old, bg_z_old};
  const float* w[2] = {fg_weights, bg_weights};
  const float* u[2] = {fg_u, bg_u};
  float* m[2] = {fg_z_merged, bg_z_merged};
  launch_sample_fine_pair((hipStream_t)stream, n_rays, s_old - 2, n_new, z, w, u, m, make_rng_key(0, 0, false));
  return check_launch("sample_fine_pair");
}

int nerfpp_sample_fine_pair_rng(void* stream, int n_rays, int s_old, int n_new, const float* fg_z_old,
                                const float* fg_weights, float* fg_z_merged, const float* bg_z_old,
                                const float* bg_weights, float* bg_z_merged, uint64_t seed, uint64_t step) {
  REQUIRE(n_rays > 0 && s_old >= 3 && n_new >= 1 && s_old + n_new <= 512, "3 <= s_old, s_old + n_new <= 512");
  REQUIRE(fg_z_old && fg_weights && fg_z_merged && bg_z_old && bg_weights && bg_z_merged, "non-null pointers");
  const float* z[2] = {fg_z_old, bg_z_old};
  const float* w[2] = {fg_weights, bg_weights};
  const float* u[2] = {nullptr, nullptr};
  float* m[2] = {fg_z_merged, bg_z_merged};
  launch_sample_fine_pair((hipStream_t)stream, n_rays, s_old - 2, n_new, z, w, u, m, make_rng_key(seed, step, true));
  return check_launch("sample_fine_pair_rng");
}

int nerfpp_sample_pixels(void* stream, uint64_t seed, uint64_t step, int64_t n_pixels, int n_rays, int64_t* pix) {
  REQUIRE(pix && n_rays >= 1 && n_rays <= 8192, "1 <= n_rays <= 8192, non-null output");
  REQUIRE(n_pixels >= n_rays && n_pixels < ((int64_t)1 << 31), "n_rays <= n_pixels < 2^31");
  launch_sample_pixels((hipStream_t)stream, make_rng_key(seed, step, true), n_pixels, n_rays, pix);
  return check_launch("sample_pixels");
}

int nerfpp_gather_rays(void* stream, int n_rays, int width, const float* cam, const int64_t* pix,
                       const float* rgb_img, const float* depth_img, float* ray_o, float* ray_d, float* rgb,
                       float* depth_sup, float* min_depth) {
  REQUIRE(n_rays > 0 && width > 0 && cam && pix && ray_o && ray_d && min_depth, "non-null pointers");
  REQUIRE((rgb == nullptr) || rgb_img, "rgb output needs rgb_img");
  REQUIRE((depth_sup == nullptr) || depth_img, "depth_sup output needs depth_img");
  launch_gather_rays((hipStream_t)stream, n_rays, width, cam, pix, rgb_img, depth_img, ray_o, ray_d, rgb, depth_sup,
                     min_depth);
  return check_launch("gather_rays");
}

int64_t nerfpp_level_tables_elems(void) { return tbl_layout().total; }

int nerfpp_build_level_tables(int32_t* host_tables) {
  if (!host_tables) return fail(NERFPP_ERR_ARG, "nerfpp_build_level_tables: null buffer");
  const TblLayout L = tbl_layout();
  for (int net = 0; net < N_NET; ++net) {
    const int rc = nerfpp_build_tables(net, host_tables + L.fwd[net], host_tables + L.bias[net],
                                       host_tables + L.bwd[net], host_tables + L.unpack[net]);
    if (rc != NERFPP_OK) return fail(rc, "nerfpp_build_tables(%d) failed", net);
  }
  return NERFPP_OK;
}

int nerfpp_dw_plan(int64_t rows, int backward_precision, int32_t* k_out, int32_t* is_full_out) {
  if (!k_out || rows <= 0 || !prec_ok(backward_precision)) return -1;
  const DwPlan pl = dw_plan(rows, backward_precision == 1);
  const JobTable jt = build_all_jobs();
  for (int net = 0; net < N_NET; ++net)
    for (int j = 0; j < DW_JOBS; ++j) {
      k_out[net * DW_JOBS + j] = pl.k[net][j];
      if (is_full_out) is_full_out[net * DW_JOBS + j] = dw_job_is_full(jt.jobs[net][j]) ? 1 : 0;
    }
  return DW_JOBS;
}

int64_t nerfpp_packed_bytes(int precision) { return prec_ok_fwd(precision) ? (int64_t)pack_layout(precision).total : -1; }

int nerfpp_pack_level(void* stream, int precision, const float* params, const int32_t* tables, void* packed) {
  REQUIRE(prec_ok_fwd(precision), "precision must be 1, 2 or 3");
  REQUIRE(params && tables && packed, "non-null pointers");
  const TblLayout T = tbl_layout();
  const PackLayout L = pack_layout(precision);
  hipStream_t st = (hipStream_t)stream;
  char* out = (char*)packed;
  const int32_t* tbl[3 * N_NET];
  void* outs[3 * N_NET];
  int64_t n[3 * N_NET];
  for (int net = 0; net < N_NET; ++net) {
    tbl[3 * net] = tables + T.fwd[net];      outs[3 * net] = out + L.fwd[net];      n[3 * net] = (int64_t)fwd_frags(net) * 512;
    tbl[3 * net + 1] = tables + T.bwd[net];  outs[3 * net + 1] = out + L.bwd[net];  n[3 * net + 1] = (int64_t)BWD_FRAGS * 512;
    tbl[3 * net + 2] = tables + T.bias[net]; outs[3 * net + 2] = out + L.bias[net]; n[3 * net + 2] = FWD_BIAS_FLOATS;
  }
  float* derived[N_NET] = {(float*)(out + L.derived[0]), (float*)(out + L.derived[1])};
  launch_pack_level(st, params, precision, tbl, outs, n, derived);
  return check_launch("pack_level");
}

int64_t nerfpp_workspace_bytes(int n_rays, int n_samples, int precision, int training) {
  if (n_rays <= 0 || n_samples < 2 || n_samples > NERFPP_MAX_SAMPLES || !prec_ok_fwd(precision)) return -1;
  return (int64_t)ws_layout(n_rays, n_samples, precision, training != 0).total;
}

int nerfpp_workspace_tensor(int n_rays, int n_samples, int precision, int net, int tensor, int64_t* byte_offset,
                            int32_t* ld, int64_t* plane_bytes) {
  REQUIRE(n_rays > 0 && n_samples >= 2 && n_samples <= NERFPP_MAX_SAMPLES && prec_ok_fwd(precision), "sizes / precision");
  REQUIRE(net >= 0 && net < N_NET && tensor >= 0 && tensor < T_COUNT && tensor != T_R && tensor != T_DR, "net / tensor id");
  REQUIRE(byte_offset && ld && plane_bytes, "non-null outputs");
  if (tensor == T_H0 && h0_recomputed(precision))
    return fail(NERFPP_ERR_UNSUPPORTED, "nerfpp_workspace_tensor: H0 is not materialised at precision %d (its weight-gradient job recomputes it from X)", precision);
  if (tensor == T_DZ0 + 7 && h0_recomputed(precision))
    return fail(NERFPP_ERR_UNSUPPORTED, "nerfpp_workspace_tensor: dZ7 is not materialised at precision %d (its weight-gradient job recomputes it from [dS | dG])", precision);
  const WsLayout L = ws_layout(n_rays, n_samples, precision, true);
  const int l = tensor_ld(net, tensor);
  *byte_offset = tensor == T_DG ? (int64_t)L.tensor[net][T_DS] + (DG_COL0 / 16) * FRAG_BYTES : (int64_t)L.tensor[net][tensor];
  *ld = l;
  *plane_bytes = (int64_t)L.rows_padded * l * 2;
  return NERFPP_OK;
}

int nerfpp_level_forward(void* stream, const nerfpp_forward_args* a) {
  REQUIRE(a, "args");
  REQUIRE(a->n_rays > 0 && a->n_samples >= 2 && a->n_samples <= NERFPP_MAX_SAMPLES, "sizes");
  REQUIRE(prec_ok_fwd(a->precision), "precision must be 1, 2 or 3");
  REQUIRE(a->ray_o && a->ray_d && a->fg_far && a->fg_z && a->bg_z && a->packed && a->workspace, "inputs");
  REQUIRE(a->rgb && a->depth && a->fg_weights && a->bg_weights && a->fg_dists && a->fg_rgb && a->fg_depth &&
          a->bg_rgb && a->bg_depth && a->bg_lambda, "outputs");
  hipStream_t st = (hipStream_t)stream;
  const int P = a->precision;
  const bool train = a->training != 0;
  const WsLayout L = ws_layout(a->n_rays, a->n_samples, P, train);
  const PackLayout PL = pack_layout(P);
  char* ws = (char*)a->workspace;
  const char* pk = (const char*)a->packed;
  MlpFwdArgs mm[N_NET];
  for (int net = 0; net < N_NET; ++net) {
    MlpFwdArgs& m = mm[net];
    m = MlpFwdArgs{};
    m.geom.ray_o = a->ray_o;
    m.geom.ray_d = a->ray_d;
    m.geom.z = net == 0 ? a->fg_z : a->bg_z;
    m.rows = L.rows;
    m.rows_padded = L.rows_padded;
    m.S = a->n_samples;
    m.w_stream = pk + PL.fwd[net];
    m.bias = (const float*)(pk + PL.bias[net]);
    m.out_raw = (float*)(ws + L.out_raw[net]);
    m.depth_real = (float*)(ws + L.depth_real);
    if (train) { m.ws = make_netws(ws, L, net); m.masks = (uint4*)(ws + L.masks[net]); }
    m.save_lo = a->training == 2 ? 0 : 1;          // training == 2: the backward will be single-pass bf16 (hi planes only)
    m.skip_h0 = (train && (h0_recomputed(P) || a->training == 2)) ? 1 : 0;   // (training == 2: the bf16 backward recomputes it too)
    if (const char* e = PROBE_GETENV("NERFPP_SKIP_H_RT")) m.save_lo |= atoi(e) << 8;   // (probes: tools/probes/recompute_probe.py)
  }
  if (a->ev_mlp_begin) (void)hipEventRecord((hipEvent_t)a->ev_mlp_begin, st);
  if (split_nets()) {
    launch_mlp_fwd_pair(st, P, train, mm[0], mm[1], 1);
    launch_mlp_fwd_pair(st, P, train, mm[0], mm[1], 2);
  } else {
    launch_mlp_fwd_pair(st, P, train, mm[0], mm[1], 0);
  }
  if (a->ev_mlp_end) (void)hipEventRecord((hipEvent_t)a->ev_mlp_end, st);
  launch_composite_fwd(st, a->n_rays, a->n_samples, (const float*)(ws + L.out_raw[0]),
                       (const float*)(ws + L.out_raw[1]), (const float*)(ws + L.depth_real), a->ray_d, a->fg_far,
                       a->fg_z, a->bg_z, a->rgb, a->depth, a->fg_weights, a->bg_weights, a->fg_dists, a->fg_rgb,
                       a->fg_depth, a->bg_rgb, a->bg_depth, a->bg_lambda);
  return check_launch("level_forward");
}

int nerfpp_loss(void* stream, int n_rays, int n_samples, int loss_type, float lambda_depth, float kl_sigma,
                const float* rgb, const float* rgb_gt, const float* depth, const float* depth_sup,
                const float* fg_weights, const float* fg_z, const float* fg_dists, const float* fg_far,
                float* scalars, float* g_rgb, float* g_depth, float* g_fg_weights) {
  REQUIRE(n_rays > 0 && loss_type >= 0 && loss_type <= 3, "loss_type in 0..3");
  REQUIRE(rgb && rgb_gt && scalars && g_rgb && g_depth, "non-null pointers");
  if (loss_type != NERFPP_LOSS_RGB_ONLY) REQUIRE(depth && depth_sup, "depth loss needs depth and depth_sup");
  if (loss_type == NERFPP_LOSS_KL)
    REQUIRE(fg_weights && fg_z && fg_dists && fg_far && g_fg_weights && kl_sigma > 0.f, "KL inputs");
  launch_loss((hipStream_t)stream, n_rays, n_samples, loss_type, lambda_depth, kl_sigma, rgb, rgb_gt, depth,
              depth_sup, fg_weights, fg_z, fg_dists, fg_far, scalars, g_rgb, g_depth, g_fg_weights);
  return check_launch("loss");
}

}  // extern "C"

// weight-gradient GEMMs of both nets (two launches) over the tensors the forward / dX kernels saved
static void weight_grads(hipStream_t st, const nerfpp_backward_args* a, const WsLayout& L) {
  char* ws = (char*)a->workspace;
  DwArgs dw{};
  for (int net = 0; net < N_NET; ++net) {
    dw.ws[net] = make_netws(ws, L, net);
    dw.slabs[net] = (float*)(ws + L.slabs[net]);
  }
  dw.rows = L.rows;
  dw.rows_padded = L.rows_padded;
  dw.h0_from_x = a->precision == 1 ? 1 : 0;      // every bf16 backward: X (hi plane) is in every workspace, W0 in its own pack
  const PackLayout PL = pack_layout(a->precision);
  for (int net = 0; net < N_NET; ++net) {
    dw.fwd_w[net] = (const char*)a->packed + PL.fwd[net];
    dw.fwd_bias[net] = (const float*)((const char*)a->packed + PL.bias[net]);
    dw.bwd_w[net] = (const char*)a->packed + PL.bwd[net];
    dw.masks[net] = (const uint4*)(ws + L.masks[net]);
  }
  dw.plan = dw_plan(L.rows, dw.h0_from_x != 0);
  if (a->ev_dw_begin) (void)hipEventRecord((hipEvent_t)a->ev_dw_begin, st);
  launch_dw(st, a->precision, dw);
  if (a->ev_dw_end) (void)hipEventRecord((hipEvent_t)a->ev_dw_end, st);
}
// experiment (NERFPP_DEFER_DW=1): with defer_reduce the weight-gradient GEMMs move to the deferred half as well
static bool defer_dw() {
  static const bool on = PROBE_GETENV("NERFPP_DEFER_DW") != nullptr;
  return on;
}

// split-K slabs -> flat gradient (fixed summation order, x grad_scale), then the derived remap / colour-head gradients
static void reduce_grads(hipStream_t st, const nerfpp_backward_args* a, const WsLayout& L, const TblLayout& T) {
  char* ws = (char*)a->workspace;
  const DwPlan plan = dw_plan(L.rows, a->precision == 1);
  const float* slabs[N_NET];
  int64_t slab_floats[N_NET];
  const int32_t* utbl[N_NET];
  float* m_out[N_NET];
  for (int net = 0; net < N_NET; ++net) {
    slabs[net] = (const float*)(ws + L.slabs[net]);
    slab_floats[net] = gslab_floats(net);
    utbl[net] = a->tables + T.unpack[net];
    m_out[net] = (float*)(ws + L.fix_m[net]);
  }
  // (diagnostic builds: NERFPP_REDUCE_SKIP = 1 drops the slab sum, 2 the fix-up, 3 both -- what does each cost the step?)
  static const int skip = PROBE_GETENV("NERFPP_REDUCE_SKIP") ? atoi(PROBE_GETENV("NERFPP_REDUCE_SKIP")) : 0;
  if (!(skip & 1)) launch_unpack_grads(st, slabs, slab_floats, plan, utbl, m_out, a->grad_scale, a->grads, a->bad_count);
  if (!(skip & 2)) launch_remap_fixup(st, a->grads, a->params, m_out[0], m_out[1]);
}

extern "C" {

int nerfpp_level_reduce_grads(void* stream, const nerfpp_backward_args* a) {
  REQUIRE(a, "args");
  REQUIRE(a->n_rays > 0 && a->n_samples >= 2 && a->n_samples <= NERFPP_MAX_SAMPLES, "sizes");
  REQUIRE(prec_ok(a->precision), "precision must be 1 or 2");
  REQUIRE(a->workspace && a->tables && a->grads && a->params, "workspace / tables / grads / params");
  const int WP = a->workspace_precision ? a->workspace_precision : a->precision;
  REQUIRE(prec_ok_fwd(WP) && a_planes(WP) >= a_planes(a->precision), "workspace_precision: 0, or a forward precision that saved the planes this backward reads");
  const WsLayout L = ws_layout(a->n_rays, a->n_samples, WP, true);
  if (defer_dw()) weight_grads((hipStream_t)stream, a, L);
  reduce_grads((hipStream_t)stream, a, L, tbl_layout());
  return check_launch("level_reduce_grads");
}

int nerfpp_level_backward(void* stream, const nerfpp_backward_args* a) {
  REQUIRE(a, "args");
  REQUIRE(a->n_rays > 0 && a->n_samples >= 2 && a->n_samples <= NERFPP_MAX_SAMPLES, "sizes");
  REQUIRE(prec_ok(a->precision), "precision must be 1 or 2");
  REQUIRE(a->ray_d && a->fg_far && a->fg_z && a->bg_z && a->packed && a->workspace && a->tables, "inputs");
  REQUIRE(a->grads && a->params, "gradients / params");
  LossFuse lf{};
  if (a->fused_loss) {
    REQUIRE(a->loss_type >= NERFPP_LOSS_RGB_ONLY && a->loss_type <= NERFPP_LOSS_KL, "loss_type in 0..3");
    REQUIRE(a->rgb && a->rgb_gt, "fused loss head needs rgb and rgb_gt");
    if (a->loss_type != NERFPP_LOSS_RGB_ONLY) REQUIRE(a->depth && a->depth_sup, "fused depth loss needs depth and depth_sup");
    if (a->loss_type == NERFPP_LOSS_KL) REQUIRE(a->kl_sigma > 0.f, "kl_sigma > 0");
    lf.type = a->loss_type; lf.lambda_depth = a->lambda_depth; lf.kl_sigma = a->kl_sigma;
    lf.rgb = a->rgb; lf.depth = a->depth; lf.rgb_gt = a->rgb_gt; lf.depth_sup = a->depth_sup;
  } else {
    REQUIRE(a->g_rgb && a->g_depth, "dL/d rgb and dL/d depth (or fused_loss)");
  }
  hipStream_t st = (hipStream_t)stream;
  const int P = a->precision;
  const int WP = a->workspace_precision ? a->workspace_precision : P;     // precision of the forward's saves
  REQUIRE(prec_ok_fwd(WP) && a_planes(WP) >= a_planes(P), "workspace_precision: 0, or a forward precision that saved the planes this backward reads");
  const WsLayout L = ws_layout(a->n_rays, a->n_samples, WP, true);
  const PackLayout PL = pack_layout(P);
  const TblLayout T = tbl_layout();
  char* ws = (char*)a->workspace;
  const char* pk = (const char*)a->packed;
  launch_composite_bwd(st, a->n_rays, a->n_samples, (const float*)(ws + L.out_raw[0]),
                       (const float*)(ws + L.out_raw[1]), (const float*)(ws + L.depth_real), a->ray_d, a->fg_far,
                       a->fg_z, a->bg_z, a->g_rgb, a->g_depth, a->g_fg_weights, (float*)(ws + L.d_out[0]),
                       (float*)(ws + L.d_out[1]), a->fused_loss ? &lf : nullptr);
  MlpBwdArgs mb[N_NET];
  for (int net = 0; net < N_NET; ++net) {
    MlpBwdArgs& m = mb[net];
    m = MlpBwdArgs{};
    m.rows = L.rows;
    m.rows_padded = L.rows_padded;
    m.w_stream = pk + PL.bwd[net];
    m.d_out = (const float*)(ws + L.d_out[net]);
    m.ws = make_netws(ws, L, net);
    m.masks = (const uint4*)(ws + L.masks[net]);
    m.skip_dz7 = P == 1 ? 1 : 0;
  }
  if (a->ev_bwd_begin) (void)hipEventRecord((hipEvent_t)a->ev_bwd_begin, st);
  if (split_nets()) {
    launch_mlp_bwd_pair(st, P, mb[0], mb[1], 1);
    launch_mlp_bwd_pair(st, P, mb[0], mb[1], 2);
  } else {
    launch_mlp_bwd_pair(st, P, mb[0], mb[1], 0);
  }
  if (a->ev_bwd_end) (void)hipEventRecord((hipEvent_t)a->ev_bwd_end, st);
  if (!(a->defer_reduce && defer_dw())) weight_grads(st, a, L);
  if (!a->defer_reduce) reduce_grads(st, a, L, T);
  return check_launch("level_backward");
}

int nerfpp_adam_step(void* stream, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                     int64_t n, int step, double lr, double beta1, double beta2, double eps,
                     const float* skip_if_nonzero) {
  REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "non-null pointers, step >= 1");
  launch_adam((hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, skip_if_nonzero);
  return check_launch("adam_step");
}

}  // extern "C"
