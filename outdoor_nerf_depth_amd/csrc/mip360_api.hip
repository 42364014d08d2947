// extern "C" entry points of libmip360_hip.so (declared in include/mip360_hip.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/mip360_hip.h"

void mip360_launch_resample(hipStream_t st, int n, int m_in, const float* sd, const float* w, float dil, float anneal,
                            float pad, int ns, const float* jit, float s_near, float s_far, const float* tn,
                            const float* tf, float* sd_out, float* td_out);
void mip360_launch_cast_encode(hipStream_t st, int n, int S, const float* td, const float* o, const float* d,
                               const float* radii, const float* basis_t, void* enc, int bf16, int ld);
void mip360_launch_render(hipStream_t st, int n, int S, const float* density, const float* rgbs, const float* td,
                          const float* dirs, int opaque, float bg, float* w, float* rgb, float* acc, float* dm, float* depth);
void mip360_launch_render_bwd(hipStream_t st, int n, int S, const float* density, const float* rgbs, const float* td,
                              const float* dirs, int opaque, float bg, const float* g_w, const float* g_rgb,
                              const float* g_dm, float* g_density, float* g_rgbs);
void mip360_launch_losses(hipStream_t st, int n, int s_nerf, int s_prop, int n_prop, const float* rgb, const float* rgb_gt,
                          const float* dm, const float* sup, const float* sd_nerf, const float* w_nerf,
                          const float* const* sd_prop, const float* const* w_prop, int charb, float charb_pad,
                          float data_mult, int depth_type, float lambda_depth, float depth_weight, float inter_mult,
                          float dist_mult, float* scalars, float* g_rgb, float* g_dm, float* g_w_nerf,
                          float* const* g_w_prop, float* ws, float prop_depth_weight, const float* const* dm_prop,
                          float* const* g_dm_prop);
void mip360_launch_depth_klurf(hipStream_t st, int type, int n, int S, const float* w, const float* td, const float* sup,
                               const float* dm, const float* dirs, float sigma, float scale, float* out, float* g_w,
                               float* g_dm, float* accum);
int mip360_launch_linear_fm(hipStream_t st, int M, int N, int K, const void* A, int lda, const void* W, int ldw, const float* bias,
                            int act, void* C, int ldc, void* mask);
int mip360_launch_grad_weight_fm(hipStream_t st, int M, int I, int O, const void* H, int ldh, const void* dZ, int lddz, int ksplit,
                                 float* slabs, int ldc, float* bias_slabs);
int mip360_launch_grad_weight_fm_multi(hipStream_t st, int n, int M, int ksplit, const int* I, const int* O, const void* const* H, const int* ldh,
                                       const void* const* dZ, const int* lddz, float* const* slabs);
int mip360_launch_rowdot_fm(hipStream_t st, int M, int K, const void* A, int lda, const void* w, const float* bias, int act, float act_param,
                            float* out, int ldo);
int mip360_launch_grad_weight_col_fm(hipStream_t st, int M, int I, const void* H, int ldh, const void* dZ, int lddz, int zcol, int ksplit,
                                     float* slabs, int ldc, float* bias_slabs);
int mip360_launch_to_fm(hipStream_t st, int rows, int cols, const void* src, int ld_src, void* dst, int ld_dst, int col0_dst);
int mip360_launch_from_fm(hipStream_t st, int rows, int cols, const void* src, int ld_src, int col0_src, void* dst, int ld_dst);
void mip360_launch_linear(hipStream_t st, int M, int N, int K, const void* A, int lda, const void* W, int ldw, const float* bias,
                          int act, float act_param, void* C16, int ldc, float* C32, int ldc32, const void* aux, int ldaux,
                          void* mask, int ldmask);
bool mip360_grad_weight_is_wide(int M, int I, int O, int ldh, int lddz);
void mip360_launch_grad_weight_reduce(hipStream_t st, int rows, int I_slab, int O, int ksplit, const float* slabs, float* out, int ldc,
                                      float scale, float* bias_out);
void mip360_launch_grad_weight(hipStream_t st, int M, int I, int O, const void* H, int ldh, const void* dZ, int lddz, int ksplit,
                               float* slabs, float* out, int ldc, float scale, float* bias_out);
void mip360_launch_col_sum(hipStream_t st, int M, int O, const void* dZ, int ld, int nslice, float* partial, float* out,
                           float scale);
void mip360_launch_head_backward(hipStream_t st, int64_t rows, const float* density, const float* g_density, const float* rgb,
                                 const float* g_rgb, float pad, void* d_raw, int ld_raw, int raw_col, int raw_zero_to,
                                 void* d_pre);
void mip360_launch_sumsq(hipStream_t st, int64_t n, const float* g, float* partial, int nblocks);
void mip360_launch_clip_mult(hipStream_t st, int n_partial, const float* partial, float max_norm, float* out);
void mip360_launch_adam(hipStream_t st, int64_t n, float* p, const float* g, float* m, float* v, const float* gmult, float lr,
                        float b1, float b2, float eps, float bc1, float bc2);
void mip360_launch_pack_weight_batch(hipStream_t st, int n, const mip360_pack_desc* descs);
void mip360_launch_pack_weight(hipStream_t st, int n_in, int n_out, const float* k, void* fwd, int ld_fwd, void* bwd, int ld_bwd,
                               void* fwd_fm, int ld_fwd_fm, void* bwd_fm, int ld_bwd_fm, int bwd_rows, int bwd_col0);
int mip360_launch_outer_masked_fm(hipStream_t st, int M, int N, const void* z, const void* w, const void* mask, void* out, int ldc);
int mip360_launch_prop_mlp_fm(hipStream_t st, int rows, const void* x_fm, int ldx, int x_col0, const void* const* w_fm, const int* ldw,
                              const float* const* bias, void* const* h_fm, void* const* masks, const void* wd, const float* bd,
                              float act_param, float* density);
int mip360_launch_prop_mlp_bwd_fm(hipStream_t st, int rows, const void* z, const void* wd, const void* const* masks,
                                  const void* const* wb_fm, const int* ldwb, void* const* dz_fm);
int mip360_launch_view_branch_fm(hipStream_t st, int rows, int n_samples, const void* bott_fm, const void* dir_table, const void* w1_fm,
                                 int ldw1, const float* b1, const void* w2_fm, int ldw2, const float* b2, float rgb_padding,
                                 void* view_in, int ld_view, void* h, int ld_h, float* rgb);
int mip360_launch_view_branch_bwd_fm(hipStream_t st, int rows, const float* density, const float* g_density, const float* rgb,
                                     const float* g_rgb, float rgb_padding, const void* h, int ld_h, const void* wb3_fm, int ldwb3,
                                     const void* wb2_fm, int ldwb2, void* d_pre, void* d_hz, int ld_dhz, void* heads_fm);
void mip360_launch_dir_encode(hipStream_t st, int n, int S, const float* viewdirs, void* out, int ld, int col0, int width);

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
  if (e != hipSuccess) return fail(MIP360_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return MIP360_OK;
}
#define REQUIRE(cond, what) \
  do { if (!(cond)) return fail(MIP360_ERR_ARG, "%s: requirement failed: %s", __func__, what); } while (0)
}  // namespace

extern "C" {

const char* mip360_last_error(void) { return g_err; }
int mip360_abi_version(void) { return MIP360_ABI_VERSION; }

int mip360_resample(void* stream, int n_rays, int m_in, const float* sdist_in, const float* weights_in, float dilation,
                    float anneal, float resample_padding, int num_samples, const float* jitter01, float s_near,
                    float s_far, const float* t_near, const float* t_far, float* sdist_out, float* tdist_out) {
  REQUIRE(n_rays > 0 && m_in >= 1 && m_in <= MIP360_MAX_BINS, "1 <= m_in <= 128");
  REQUIRE(num_samples >= 2 && num_samples <= MIP360_MAX_SAMPLES, "2 <= num_samples <= 64");
  REQUIRE(sdist_in && weights_in && t_near && t_far && sdist_out && tdist_out, "non-null pointers");
  mip360_launch_resample((hipStream_t)stream, n_rays, m_in, sdist_in, weights_in, dilation, anneal, resample_padding,
                         num_samples, jitter01, s_near, s_far, t_near, t_far, sdist_out, tdist_out);
  return check_launch("resample");
}

int mip360_cast_encode(void* stream, int n_rays, int n_samples, const float* tdist, const float* origins,
                       const float* directions, const float* radii, const float* basis_t, void* enc, int out_bf16, int ld) {
  REQUIRE(n_rays > 0 && n_samples >= 1 && ld >= MIP360_IPE_DIM, "sizes, ld >= 504");
  REQUIRE(tdist && origins && directions && radii && basis_t && enc, "non-null pointers");
  REQUIRE(out_bf16 != 2 || (ld % 16 == 0 && ld >= 512 && ((int64_t)n_rays * n_samples) % 32 == 0), "fm output: ld % 16 == 0, rows % 32 == 0");
  mip360_launch_cast_encode((hipStream_t)stream, n_rays, n_samples, tdist, origins, directions, radii, basis_t, enc,
                            out_bf16, ld);
  return check_launch("cast_encode");
}

int mip360_render_level(void* stream, int n_rays, int n_samples, const float* density, const float* rgb_samples,
                        const float* tdist, const float* directions, int opaque_background, float bg_rgb, float* weights,
                        float* rgb, float* acc, float* distance_mean, float* depth) {
  REQUIRE(n_rays > 0 && n_samples >= 1 && n_samples <= MIP360_MAX_SAMPLES, "1 <= n_samples <= 64");
  REQUIRE(density && tdist && directions && weights, "non-null pointers");
  mip360_launch_render((hipStream_t)stream, n_rays, n_samples, density, rgb_samples, tdist, directions, opaque_background,
                       bg_rgb, weights, rgb, acc, distance_mean, depth);
  return check_launch("render_level");
}

int mip360_render_level_backward(void* stream, int n_rays, int n_samples, const float* density, const float* rgb_samples,
                                 const float* tdist, const float* directions, int opaque_background, float bg_rgb,
                                 const float* g_weights, const float* g_rgb, const float* g_distance_mean,
                                 float* g_density, float* g_rgb_samples) {
  REQUIRE(n_rays > 0 && n_samples >= 1 && n_samples <= MIP360_MAX_SAMPLES, "1 <= n_samples <= 64");
  REQUIRE(density && tdist && directions && g_density, "non-null pointers");
  mip360_launch_render_bwd((hipStream_t)stream, n_rays, n_samples, density, rgb_samples, tdist, directions,
                           opaque_background, bg_rgb, g_weights, g_rgb, g_distance_mean, g_density, g_rgb_samples);
  return check_launch("render_level_backward");
}

int mip360_losses(void* stream, int n_rays, int s_nerf, int s_prop, int n_prop, const float* rgb, const float* rgb_gt,
                  const float* distance_mean, const float* depth_sup, const float* sdist_nerf, const float* w_nerf,
                  const float* const* sdist_prop, const float* const* w_prop, int charb, float charb_padding,
                  float data_loss_mult, int depth_loss_type, float lambda_depth, float depth_weight, float interlevel_mult,
                  float distortion_mult, float* scalars, float* g_rgb, float* g_distance_mean, float* g_w_nerf,
                  float* const* g_w_prop, float* workspace, float prop_depth_weight, const float* const* dm_prop,
                  float* const* g_dm_prop) {
  REQUIRE(n_rays > 0 && s_nerf >= 1 && s_nerf <= MIP360_MAX_SAMPLES && s_prop >= 1 && s_prop <= MIP360_MAX_SAMPLES, "sizes");
  REQUIRE(n_prop >= 0 && n_prop <= 4, "0 <= n_prop <= 4");
  REQUIRE(rgb && rgb_gt && sdist_nerf && w_nerf && scalars && g_rgb && g_distance_mean && g_w_nerf && workspace, "pointers");
  REQUIRE(depth_loss_type >= 0 && depth_loss_type <= 2, "depth_loss_type in 0..2");
  if (depth_loss_type) REQUIRE(distance_mean && depth_sup, "depth term needs distance_mean and depth_sup");
  for (int k = 0; k < n_prop; ++k) REQUIRE(sdist_prop && w_prop && g_w_prop && sdist_prop[k] && w_prop[k] && g_w_prop[k], "proposal arrays");
  if (dm_prop) for (int k = 0; k < n_prop; ++k) REQUIRE(!dm_prop[k] || (g_dm_prop && g_dm_prop[k]), "g_dm_prop for every dm_prop");
  mip360_launch_losses((hipStream_t)stream, n_rays, s_nerf, s_prop, n_prop, rgb, rgb_gt, distance_mean, depth_sup, sdist_nerf,
                       w_nerf, sdist_prop, w_prop, charb, charb_padding, data_loss_mult, depth_loss_type, lambda_depth,
                       depth_weight, interlevel_mult, distortion_mult, scalars, g_rgb, g_distance_mean, g_w_nerf, g_w_prop,
                       workspace, prop_depth_weight, dm_prop, g_dm_prop);
  return check_launch("losses");
}

int mip360_depth_loss_klurf(void* stream, int depth_loss_type, int n_rays, int n_samples, const float* weights,
                            const float* tdist, const float* depth_sup, const float* distance_mean, const float* directions,
                            float sigma, float scale, float* loss_out, float* g_weights, float* g_distance_mean,
                            float* total_accum) {
  REQUIRE(depth_loss_type == MIP360_DEPTH_KL || depth_loss_type == MIP360_DEPTH_URF, "depth_loss_type must be 3 (kl) or 4 (urf)");
  REQUIRE(n_rays > 0 && n_samples >= 1 && n_samples <= MIP360_MAX_SAMPLES, "1 <= n_samples <= 64");
  REQUIRE(weights && tdist && depth_sup && loss_out && sigma > 0.f, "non-null pointers, sigma > 0");
  if (depth_loss_type == MIP360_DEPTH_KL) REQUIRE(directions, "kl needs the ray directions");
  if (depth_loss_type == MIP360_DEPTH_URF) REQUIRE(distance_mean, "urf needs distance_mean");
  if (n_rays != n_samples && n_rays != 1)
    return fail(MIP360_ERR_ARG,
                "mip360_depth_loss_klurf: operands could not be broadcast together with shapes (%d,) (%d,) -- upstream's "
                "loss.sum(-2) * depth_mask (internal/depth_loss.py:27,64) needs n_rays == n_samples or n_rays == 1",
                n_samples, n_rays);
  mip360_launch_depth_klurf((hipStream_t)stream, depth_loss_type, n_rays, n_samples, weights, tdist, depth_sup, distance_mean,
                            directions, sigma, scale, loss_out, g_weights, g_distance_mean, total_accum);
  return check_launch("depth_loss_klurf");
}

int mip360_dir_encode(void* stream, int n_rays, int n_samples, const float* viewdirs, void* out_bf16, int ld, int col0,
                      int width) {
  REQUIRE(n_rays > 0 && n_samples >= 1 && viewdirs && out_bf16, "non-null pointers");
  REQUIRE(width >= 27 && col0 >= 0 && col0 + width <= ld, "27 <= width, col0 + width <= ld");
  mip360_launch_dir_encode((hipStream_t)stream, n_rays, n_samples, viewdirs, out_bf16, ld, col0, width);
  return check_launch("dir_encode");
}

int mip360_linear_bf16(void* stream, int m, int n, int k, const void* a, int lda, const void* w, int ldw, const float* bias,
                       int act, float act_param, void* c_bf16, int ldc, float* c_f32, int ldc32, const void* aux, int ldaux) {
  REQUIRE(m > 0 && n > 0 && k > 0 && k % 32 == 0, "k must be a positive multiple of 32");
  REQUIRE(a && w && (c_bf16 || c_f32), "non-null operands, at least one output");
  REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= k && ldw >= k, "leading dimensions: multiples of 8, >= k");
  REQUIRE(act >= 0 && act <= 4, "act in 0..4");
  REQUIRE(act != 4 || (aux && ldaux >= n), "act 4 needs the mask tensor");
  mip360_launch_linear((hipStream_t)stream, m, n, k, a, lda, w, ldw, bias, act, act_param, c_bf16, ldc, c_f32, ldc32, aux, ldaux,
                       nullptr, 0);
  return check_launch("linear_bf16");
}

static bool mask_args_ok(int m, int n, const void* mask, int ldmask) {
  return mask && ldmask % 16 == 0 && ldmask >= (m + 255) / 256 * 256 && n > 0;
}

int64_t mip360_relu_mask_bytes(int m, int n, int* ldmask) {
  if (m <= 0 || n <= 0) return 0;
  const int ld = (m + 255) / 256 * 256;
  if (ldmask) *ldmask = ld;
  return (int64_t)((n + 255) / 256 * 32) * ld;
}

int mip360_linear_relu_mask_bf16(void* stream, int m, int n, int k, const void* a, int lda, const void* w, int ldw,
                                 const float* bias, void* c_bf16, int ldc, void* mask, int ldmask) {
  REQUIRE(m > 0 && n > 0 && k > 0 && k % 32 == 0, "k must be a positive multiple of 32");
  REQUIRE(a && w && c_bf16, "non-null operands and output");
  REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= k && ldw >= k && ldc >= n, "leading dimensions: multiples of 8, >= k / n");
  REQUIRE(mask_args_ok(m, n, mask, ldmask), "mask buffer: ldmask = roundup(m, 256) (mip360_relu_mask_bytes)");
  mip360_launch_linear((hipStream_t)stream, m, n, k, a, lda, w, ldw, bias, 5, 0.f, c_bf16, ldc, nullptr, 0, nullptr, 0, mask, ldmask);
  return check_launch("linear_relu_mask_bf16");
}

int mip360_linear_masked_bf16(void* stream, int m, int n, int k, const void* a, int lda, const void* w, int ldw, void* c_bf16,
                              int ldc, const void* mask, int ldmask) {
  REQUIRE(m > 0 && n > 0 && k > 0 && k % 32 == 0, "k must be a positive multiple of 32");
  REQUIRE(a && w && c_bf16, "non-null operands and output");
  REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= k && ldw >= k && ldc >= n, "leading dimensions: multiples of 8, >= k / n");
  REQUIRE(mask_args_ok(m, n, mask, ldmask), "mask buffer: ldmask = roundup(m, 256) (mip360_relu_mask_bytes)");
  mip360_launch_linear((hipStream_t)stream, m, n, k, a, lda, w, ldw, nullptr, 6, 0.f, c_bf16, ldc, nullptr, 0, nullptr, 0,
                       (void*)mask, ldmask);
  return check_launch("linear_masked_bf16");
}

// ---- fragment-major dense layers (mip360_fm.hip)

int64_t mip360_fm_mask_bytes(int m, int n) { return m > 0 && n > 0 ? (int64_t)(m / 256) * (n / 256) * 8 * 64 * 16 : 0; }

int mip360_to_fm(void* stream, int rows, int cols, const void* src_bf16, int ld_src, void* dst_fm, int ld_dst, int col0_dst) {
  REQUIRE(src_bf16 && dst_fm && rows > 0 && cols > 0, "non-null pointers");
  REQUIRE(mip360_launch_to_fm((hipStream_t)stream, rows, cols, src_bf16, ld_src, dst_fm, ld_dst, col0_dst) == 0,
          "rows % 32 == 0, cols / ld_dst / col0_dst % 16 == 0, ld_src % 4 == 0");
  return check_launch("to_fm");
}

int mip360_from_fm(void* stream, int rows, int cols, const void* src_fm, int ld_src, int col0_src, void* dst_bf16, int ld_dst) {
  REQUIRE(src_fm && dst_bf16 && rows > 0 && cols > 0, "non-null pointers");
  REQUIRE(mip360_launch_from_fm((hipStream_t)stream, rows, cols, src_fm, ld_src, col0_src, dst_bf16, ld_dst) == 0,
          "rows % 32 == 0, cols / ld_src / col0_src % 16 == 0, ld_dst % 4 == 0");
  return check_launch("from_fm");
}

int mip360_linear_fm(void* stream, int m, int n, int k, const void* a_fm, int lda, const void* w_fm, int ldw, const float* bias,
                     int act, void* c_fm, int ldc, void* mask) {
  REQUIRE(a_fm && w_fm && c_fm, "non-null operands and output");
  REQUIRE(lda >= k && ldw >= k && ldc >= n, "leading dimensions >= k / n");
  REQUIRE(mip360_launch_linear_fm((hipStream_t)stream, m, n, k, a_fm, lda, w_fm, ldw, bias, act, c_fm, ldc, mask) == 0,
          "m, n multiples of 256; k a multiple of 32, >= 160; leading dimensions multiples of 16; act 0 / 1 need bias, 1 / 2 the mask");
  return check_launch("linear_fm");
}

int mip360_grad_weight_bf16(void* stream, int m, int n_in, int n_out, const void* h, int ldh, const void* dz, int lddz,
                            int ksplit, float* slabs, float* grad_kernel, int ldg, float scale, float* grad_bias) {
  REQUIRE(m > 0 && n_in > 0 && n_out > 0 && n_in % 8 == 0 && ksplit >= 1 && ksplit <= 256, "sizes (n_in multiple of 8, 1 <= ksplit <= 256)");
  REQUIRE(h && dz && slabs && ldh >= n_in && lddz >= n_out && ldg >= n_out && ldh % 8 == 0 && lddz % 8 == 0, "pointers / leading dimensions");
  mip360_launch_grad_weight((hipStream_t)stream, m, n_in, n_out, h, ldh, dz, lddz, ksplit, slabs, grad_kernel, ldg, scale, grad_bias);
  return check_launch("grad_weight_bf16");
}

int mip360_grad_weight_fm(void* stream, int m, int n_in, int n_out, const void* h_fm, int ldh, const void* dz_fm, int lddz,
                          int ksplit, float* slabs, float* grad_kernel, int ldg, float scale, float* grad_bias) {
  REQUIRE(h_fm && dz_fm && slabs && ksplit >= 1 && ksplit <= 256 && ldh >= n_in && lddz >= n_out && ldg >= n_out, "pointers / leading dimensions");
  float* bias_slabs = grad_bias ? slabs + (size_t)ksplit * n_in * ldg : nullptr;
  REQUIRE(mip360_launch_grad_weight_fm((hipStream_t)stream, m, n_in, n_out, h_fm, ldh, dz_fm, lddz, ksplit, slabs, ldg, bias_slabs) == 0,
          "m a multiple of 32, n_in / n_out multiples of 256, leading dimensions multiples of 16");
  if (grad_kernel) mip360_launch_grad_weight_reduce((hipStream_t)stream, n_in, n_in, n_out, ksplit, slabs, grad_kernel, ldg, scale, grad_bias);
  return check_launch("grad_weight_fm");
}

int mip360_grad_weight_fm_multi(void* stream, int n, int m, int ksplit, const int* n_in, const int* n_out, const void* const* h_fm,
                                const int* ldh, const void* const* dz_fm, const int* lddz, float* const* slabs) {
  REQUIRE(n_in && n_out && h_fm && ldh && dz_fm && lddz && slabs, "pointers");
  REQUIRE(mip360_launch_grad_weight_fm_multi((hipStream_t)stream, n, m, ksplit, n_in, n_out, h_fm, ldh, dz_fm, lddz, slabs) == 0,
          "1 <= n <= 8 problems, m a multiple of 32, 1 <= ksplit <= 256, n_in / n_out multiples of 256, leading dimensions multiples of 16");
  return check_launch("grad_weight_fm_multi");
}

int mip360_rowdot_fm(void* stream, int m, int k, const void* a_fm, int lda, const void* w_bf16, const float* bias, int act,
                     float act_param, float* out, int ldo) {
  REQUIRE(a_fm && w_bf16 && out && lda >= k && ldo >= 1, "pointers / leading dimensions");
  REQUIRE(mip360_launch_rowdot_fm((hipStream_t)stream, m, k, a_fm, lda, w_bf16, bias, act, act_param, out, ldo) == 0,
          "m a multiple of 32, k / lda multiples of 16, act in 0..2");
  return check_launch("rowdot_fm");
}

int mip360_prop_mlp_fm(void* stream, int rows, const void* x_fm, int ldx, int x_col0, const void* const* w_fm, const int* ldw,
                       const float* const* bias, void* const* h_fm, void* const* masks, const void* wd_bf16, const float* bd,
                       float act_param, float* density) {
  const int rc = mip360_launch_prop_mlp_fm((hipStream_t)stream, rows, x_fm, ldx, x_col0, w_fm, ldw, bias, h_fm, masks, wd_bf16, bd,
                                           act_param, density);
  REQUIRE(rc != 1, "rows a multiple of 256, ldx / x_col0 / ldw multiples of 16, 512 operand columns, h_fm and masks together, "
                   "density or training outputs");
  REQUIRE(rc == 0, "hipFuncSetAttribute");
  return check_launch("prop_mlp_fm");
}

int mip360_prop_mlp_bwd_fm(void* stream, int rows, const void* z_bf16, const void* wd_bf16, const void* const* masks,
                           const void* const* wb_fm, const int* ldwb, void* const* dz_fm) {
  const int rc = mip360_launch_prop_mlp_bwd_fm((hipStream_t)stream, rows, z_bf16, wd_bf16, masks, wb_fm, ldwb, dz_fm);
  REQUIRE(rc != 1, "rows a multiple of 256, four masks / outputs, wb_fm[1..3] with ldwb multiples of 16 >= 256");
  REQUIRE(rc == 0, "hipFuncSetAttribute");
  return check_launch("prop_mlp_bwd_fm");
}

int mip360_view_branch_fm(void* stream, int rows, int n_samples, const void* bott_fm, const void* dir_table_bf16, const void* w1_fm, int ldw1,
                          const float* b1, const void* w2_fm, int ldw2, const float* b2, float rgb_padding, void* view_in_bf16,
                          int ld_view, void* h_bf16, int ld_h, float* rgb) {
  const int rc = mip360_launch_view_branch_fm((hipStream_t)stream, rows, n_samples, bott_fm, dir_table_bf16, w1_fm, ldw1, b1, w2_fm, ldw2, b2,
                                              rgb_padding, view_in_bf16, ld_view, h_bf16, ld_h, rgb);
  REQUIRE(rc != 1, "rows a multiple of 256 and of n_samples, ldw1 >= 288 / ldw2 >= 128 multiples of 16, ld_view >= 288 / ld_h >= 128 "
                   "multiples of 4, non-null operands");
  REQUIRE(rc == 0, "hipFuncSetAttribute");
  return check_launch("view_branch_fm");
}

int mip360_view_branch_bwd_fm(void* stream, int rows, const float* density, const float* g_density, const float* rgb, const float* g_rgb,
                              float rgb_padding, const void* h_bf16, int ld_h, const void* wb_rgb_fm, int ld_wb_rgb, const void* wb_view_fm,
                              int ld_wb_view, void* d_pre_bf16, void* d_hz_bf16, int ld_dhz, void* heads_fm) {
  const int rc = mip360_launch_view_branch_bwd_fm((hipStream_t)stream, rows, density, g_density, rgb, g_rgb, rgb_padding, h_bf16, ld_h,
                                                  wb_rgb_fm, ld_wb_rgb, wb_view_fm, ld_wb_view, d_pre_bf16, d_hz_bf16, ld_dhz, heads_fm);
  REQUIRE(rc != 1, "rows a multiple of 256, non-null operands, ld_h / ld_dhz >= 128 multiples of 4, ld_wb_rgb >= 32 / ld_wb_view >= 128 "
                   "multiples of 16");
  REQUIRE(rc == 0, "hipFuncSetAttribute");
  return check_launch("view_branch_bwd_fm");
}

int mip360_grad_weight_col_fm(void* stream, int m, int n_in, const void* h_fm, int ldh, const void* dz_fm, int lddz, int zcol,
                              int ksplit, float* slabs, float* grad_kernel, float scale, float* grad_bias) {
  REQUIRE(h_fm && dz_fm && slabs && ksplit >= 1 && ksplit <= 256 && ldh >= n_in, "pointers / leading dimensions");
  float* bias_slabs = grad_bias ? slabs + (size_t)ksplit * n_in : nullptr;
  REQUIRE(mip360_launch_grad_weight_col_fm((hipStream_t)stream, m, n_in, h_fm, ldh, dz_fm, lddz, zcol, ksplit, slabs, 1, bias_slabs) == 0,
          "m a multiple of 32, n_in and leading dimensions multiples of 16, 0 <= zcol < lddz");
  if (grad_kernel) mip360_launch_grad_weight_reduce((hipStream_t)stream, n_in, n_in, 1, ksplit, slabs, grad_kernel, 1, scale, grad_bias);
  return check_launch("grad_weight_col_fm");
}

int mip360_grad_weight_reduce(void* stream, int rows, int n_in, int n_out, int ksplit, const float* slabs, float* grad_kernel, int ldg,
                              float scale, float* grad_bias) {
  REQUIRE(rows > 0 && rows <= n_in && n_out > 0 && ksplit >= 1 && ksplit <= 256 && slabs && grad_kernel && ldg >= n_out, "arguments");
  mip360_launch_grad_weight_reduce((hipStream_t)stream, rows, n_in, n_out, ksplit, slabs, grad_kernel, ldg, scale, grad_bias);
  return check_launch("grad_weight_reduce");
}

int mip360_grad_weight_tile(int m, int n_in, int n_out, int ldh, int lddz) {
  return mip360_grad_weight_is_wide(m, n_in, n_out, ldh, lddz) ? 256 : 128;
}

int mip360_grad_bias_bf16(void* stream, int m, int n_out, const void* dz, int lddz, int nslice, float* partial, float* grad_bias,
                          float scale) {
  REQUIRE(m > 0 && n_out > 0 && dz && partial && grad_bias && nslice >= 1 && nslice <= 1024 && lddz >= n_out, "arguments");
  mip360_launch_col_sum((hipStream_t)stream, m, n_out, dz, lddz, nslice, partial, grad_bias, scale);
  return check_launch("grad_bias_bf16");
}

int mip360_head_backward(void* stream, int64_t rows, const float* density, const float* g_density, const float* rgb,
                         const float* g_rgb, float rgb_padding, void* d_raw_bf16, int ld_raw, int raw_col, int raw_zero_to,
                         void* d_pre_bf16) {
  REQUIRE(rows > 0 && density && g_density && d_raw_bf16 && raw_col >= 0 && raw_zero_to <= ld_raw && raw_col < ld_raw, "arguments");
  REQUIRE(!d_pre_bf16 || (rgb && g_rgb), "colour head needs rgb and g_rgb");
  mip360_launch_head_backward((hipStream_t)stream, rows, density, g_density, rgb, g_rgb, rgb_padding, d_raw_bf16, ld_raw, raw_col,
                              raw_zero_to, d_pre_bf16);
  return check_launch("head_backward");
}

int mip360_sum_squares(void* stream, int64_t n, const float* g, float* partial, int n_blocks) {
  REQUIRE(n > 0 && g && partial && n_blocks >= 1 && n_blocks <= 1024, "arguments");
  mip360_launch_sumsq((hipStream_t)stream, n, g, partial, n_blocks);
  return check_launch("sum_squares");
}

int mip360_clip_multiplier(void* stream, int n_partial, const float* partial, float grad_max_norm, float* mult_and_norm) {
  REQUIRE(n_partial >= 1 && partial && mult_and_norm, "arguments");
  mip360_launch_clip_mult((hipStream_t)stream, n_partial, partial, grad_max_norm, mult_and_norm);
  return check_launch("clip_multiplier");
}

int mip360_adam_step(void* stream, int64_t n, float* params, const float* grads, float* mu, float* nu, const float* grad_mult,
                     int step, double lr, double beta1, double beta2, double eps) {
  REQUIRE(n > 0 && params && grads && mu && nu && step >= 1, "arguments");
  const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
  mip360_launch_adam((hipStream_t)stream, n, params, grads, mu, nu, grad_mult, (float)lr, (float)beta1, (float)beta2, (float)eps,
                     (float)bc1, (float)bc2);
  return check_launch("adam_step");
}

int mip360_pack_weight(void* stream, int n_in, int n_out, const float* kernel, void* fwd_bf16, int ld_fwd, void* bwd_bf16,
                       int ld_bwd) {
  REQUIRE(n_in > 0 && n_out > 0 && kernel && (fwd_bf16 || bwd_bf16), "arguments");
  REQUIRE((!fwd_bf16 || ld_fwd >= n_in) && (!bwd_bf16 || ld_bwd >= n_out), "leading dimensions");
  mip360_launch_pack_weight((hipStream_t)stream, n_in, n_out, kernel, fwd_bf16, ld_fwd, bwd_bf16, ld_bwd, nullptr, 0, nullptr, 0, 0, 0);
  return check_launch("pack_weight");
}

int mip360_pack_weight_fm(void* stream, int n_in, int n_out, const float* kernel, void* fwd_bf16, int ld_fwd, void* bwd_bf16, int ld_bwd,
                          void* fwd_fm, int ld_fwd_fm, void* bwd_fm, int ld_bwd_fm, int bwd_rows, int bwd_col0) {
  REQUIRE(n_in > 0 && n_out > 0 && kernel, "arguments");
  REQUIRE((!fwd_bf16 || ld_fwd >= n_in) && (!bwd_bf16 || ld_bwd >= n_out), "leading dimensions");
  REQUIRE((!fwd_fm || (ld_fwd_fm >= n_in && ld_fwd_fm % 16 == 0)) && (!bwd_fm || (ld_bwd_fm >= bwd_col0 + n_out && ld_bwd_fm % 16 == 0 && bwd_col0 >= 0)),
          "fm leading dimensions (multiples of 16)");
  mip360_launch_pack_weight((hipStream_t)stream, n_in, n_out, kernel, fwd_bf16, ld_fwd, bwd_bf16, ld_bwd, fwd_fm, ld_fwd_fm, bwd_fm, ld_bwd_fm,
                            bwd_rows, bwd_col0);
  return check_launch("pack_weight_fm");
}

int mip360_pack_weights_fm_batch(void* stream, int n, const mip360_pack_desc* descs) {
  REQUIRE(n >= 1 && n <= MIP360_PACK_BATCH_MAX && descs, "1 <= n <= 16 descriptors");
  for (int t = 0; t < n; ++t) {
    const mip360_pack_desc& d = descs[t];
    REQUIRE(d.n_in > 0 && d.n_out > 0 && d.kernel, "arguments");
    REQUIRE((!d.fwd_bf16 || d.ld_fwd >= d.n_in) && (!d.bwd_bf16 || d.ld_bwd >= d.n_out), "leading dimensions");
    REQUIRE((!d.fwd_fm || (d.ld_fwd_fm >= d.n_in && d.ld_fwd_fm % 16 == 0)) &&
            (!d.bwd_fm || (d.ld_bwd_fm >= d.bwd_col0 + d.n_out && d.ld_bwd_fm % 16 == 0 && d.bwd_col0 >= 0)),
            "fm leading dimensions (multiples of 16)");
  }
  mip360_launch_pack_weight_batch((hipStream_t)stream, n, descs);
  return check_launch("pack_weights_fm_batch");
}

int mip360_outer_masked_fm(void* stream, int m, int n, const void* z_bf16, const void* w_bf16, const void* mask, void* c_fm, int ldc) {
  REQUIRE(z_bf16 && w_bf16 && mask && c_fm && ldc >= n, "non-null pointers, ldc >= n");
  REQUIRE(mip360_launch_outer_masked_fm((hipStream_t)stream, m, n, z_bf16, w_bf16, mask, c_fm, ldc) == 0, "m, n multiples of 256, ldc of 16");
  return check_launch("outer_masked_fm");
}

}  // extern "C"
