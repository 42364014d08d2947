// Kernel argument structs and host launcher prototypes (internal to the shared library).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nerfpp_common.h"

namespace nerfpp {

struct MlpGeom {                 // per-ray inputs + per-sample depths of ONE volume (fg or bg)
  const float* ray_o;            // [n_rays, 3]
  const float* ray_d;            // [n_rays, 3]
  const float* z;                // [n_rays * S]
};

struct NetWs {                   // saved tensors of one net (nerfpp_common.h: enum Tensor)
  __bf16* t[T_COUNT];            // hi plane; lo plane (P = 2) follows at +rows_padded*ld elements
};

struct MlpFwdArgs {
  MlpGeom geom;
  int64_t rows, rows_padded;     // rows = n_rays * S
  int S;
  const void* w_stream;          // packed forward weight stream of this net
  const float* bias;             // packed forward bias stream of this net
  float* out_raw;                // [rows, 4] (r, g, b, sigma_raw)
  float* depth_real;             // [rows] (background only)
  NetWs ws;
  uint4* masks;                  // ReLU sign bits [9 stages][rows_padded/32][64 lanes] (training)
  int save_lo;                   // split-bf16 training: 0 = write the hi planes of the saved tensors only (bf16 backward)
  int skip_h0;                   // training, precisions 1 / 3: H0 is not written -- its weight-gradient job recomputes it (nerfpp_dw.hip: rc_job)
};

struct MlpBwdArgs {
  int64_t rows, rows_padded;
  const void* w_stream;          // packed backward (transposed) weight stream
  const float* d_out;            // [rows, 4] (d rgb_pre[3], d sigma_raw)
  NetWs ws;
  const uint4* masks;
  int skip_dz7;                  // bf16 backward: dZ7 is not written -- its weight-gradient job recomputes it from [dS | dG] (rc7_job)
};

struct LossFuse {                // loss head folded into the compositing backward (nerfpp_backward_args::fused_loss)
  int type;                      // NERFPP_LOSS_*
  float lambda_depth, kl_sigma;
  const float* rgb;              // [n,3] forward output
  const float* depth;            // [n]   forward output
  const float* rgb_gt;           // [n,3]
  const float* depth_sup;        // [n] (NULL for rgb-only)
};

struct DwArgs {
  NetWs ws[N_NET];
  int64_t rows, rows_padded;
  float* slabs[N_NET];           // [DW_KMAX][gslab_floats(net)]; job j fills the first plan.k[net][j] slabs
  DwPlan plan;
  // H0 = relu(W0 X + b0) is recomputed inside the L1 job from the saved encoded point (single-plane workspaces): the packed
  // forward weight stream (its first kpe x 8 fragments are W0) and the forward bias stream of each net
  int h0_from_x;
  const void* fwd_w[N_NET];
  const float* fwd_bias[N_NET];
  // ... and dZ7 = mask7 * (Wc^T dG + wsigma dsigma) inside the L7 job from the saved [dS | dG] tensor and the sign words of H7:
  // the packed backward (transposed) weight stream (stage BS_DH7) and the ReLU sign words of each net
  const void* bwd_w[N_NET];
  const uint4* masks[N_NET];
};

}  // namespace nerfpp

// nerfpp_render.hip
void launch_intersect_sphere(hipStream_t st, int n, const float* o, const float* d, float* far, int* bad);
void launch_sample_coarse(hipStream_t st, int n, int S, const float* o, const float* d, const float* min_depth,
                          const float* t_fg, const float* t_bg, float* far, float* fg_z, float* bg_z, int* bad,
                          const nerfpp::RngKey& rng);
void launch_rng_uniform(hipStream_t st, const nerfpp::RngKey& rng, uint32_t stream_id, int64_t n, float* out);
void launch_perturb(hipStream_t st, int n, int S, const float* z, const float* t, float* out);
void launch_sample_pdf(hipStream_t st, bool fused, int n, int M, int S_new, const float* bins_or_zold,
                       const float* weights, const float* u, float* samples, int64_t* above, float* merged);
void launch_sample_fine_pair(hipStream_t st, int n, int M, int S_new, const float* const* z_old,
                             const float* const* weights, const float* const* u, float* const* merged,
                             const nerfpp::RngKey& rng);
void launch_composite_fwd(hipStream_t st, int n, int S, const float* raw_fg, const float* raw_bg,
                          const float* depth_real_bg, const float* ray_d, const float* fg_far,
                          const float* fg_z, const float* bg_z, float* rgb, float* depth, float* fg_weights,
                          float* bg_weights, float* fg_dists, float* fg_rgb, float* fg_depth, float* bg_rgb,
                          float* bg_depth, float* bg_lambda);
void launch_composite_bwd(hipStream_t st, int n, int S, const float* raw_fg, const float* raw_bg,
                          const float* depth_real_bg, const float* ray_d, const float* fg_far,
                          const float* fg_z, const float* bg_z, const float* g_rgb, const float* g_depth,
                          const float* g_fg_weights, float* dout_fg, float* dout_bg, const nerfpp::LossFuse* lf);
void launch_loss(hipStream_t st, int n, int S, int type, float lambda_depth, float kl_sigma, const float* rgb,
                 const float* rgb_gt, const float* depth, const float* depth_sup, const float* fg_weights,
                 const float* fg_z, const float* fg_dists, const float* fg_far, float* scalars, float* g_rgb,
                 float* g_depth, float* g_fg_weights);
int launch_sample_pixels(hipStream_t st, const nerfpp::RngKey& rng, int64_t n_pixels, int k, int64_t* pix);
void launch_gather_rays(hipStream_t st, int n, int W, const float* cam, const int64_t* pix, const float* rgb_img,
                        const float* depth_img, float* ray_o, float* ray_d, float* rgb, float* depth_sup,
                        float* min_depth);
// nerfpp_mlp.hip
// both nets of a level in one launch (which = 0), or one of them alone (1 = fg, 2 = bg)
void launch_mlp_fwd_pair(hipStream_t st, int P, bool train, const nerfpp::MlpFwdArgs& a_fg, const nerfpp::MlpFwdArgs& a_bg, int which);
void launch_mlp_bwd_pair(hipStream_t st, int P, const nerfpp::MlpBwdArgs& a_fg, const nerfpp::MlpBwdArgs& a_bg, int which);
// nerfpp_dw.hip
void launch_dw(hipStream_t st, int P, const nerfpp::DwArgs& a);
// nerfpp_optim.hip
// derived[net]: DERIVED_FLOATS floats of scratch for the net's folded colour-head parameters (filled here, then packed)
void launch_pack_level(hipStream_t st, const float* params, int P, const int32_t* const* tbl, void* const* out,
                       const int64_t* n, float* const* derived);
void launch_unpack_grads(hipStream_t st, const float* const* slabs, const int64_t* slab_floats,
                         const nerfpp::DwPlan& plan, const int32_t* const* tbl, float* const* m_out, float scale,
                         float* grads_lvl, const int32_t* bad_count);
void launch_remap_fixup(hipStream_t st, float* grads_lvl, const float* params_lvl, const float* m0, const float* m1);
void launch_adam(hipStream_t st, float* p, const float* g, float* m, float* v, int64_t n, int step, double lr,
                 double beta1, double beta2, double eps, const float* skip);
