/* mip360_hip.h -- C ABI of libmip360_hip.so: the MI355X (gfx950) kernels of the MipNeRF-360 depth-supervised path
 * of cwchenwang/outdoor-nerf-depth (BASELINE config 5; nerf-methods/mipnerf360, JAX upstream).
 *
 * Upstream has no FFI: the path is jitted JAX inside internal/models.py (Model.__call__ :76-303, MLP.__call__
 * :436-606), internal/{stepfun,coord,render}.py and internal/train_utils.py (:72-169).  Each entry point names the
 * upstream lines it replaces.  Conventions as in nerfpp_hip.h: plain C, raw DEVICE pointers (float32 unless stated),
 * a `void* stream` (hipStream_t), return MIP360_OK or an error code with mip360_last_error(); the library is
 * stateless and the caller owns every buffer.  Network / sampler shape = configs/360.gin (the configuration
 * scripts/train_kitti.sh uses): 2 proposal levels x 64 samples (PropMLP 4 x 256) + 1 NeRF level x 32 samples
 * (NerfMLP 8 x 1024, bottleneck 256, view branch 128), warp = contract, raydist = reciprocal, opaque background,
 * icosahedron-2 basis (21 directions), IPE degrees [0, 12).
 */
#ifndef MIP360_HIP_H
#define MIP360_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIP360_ABI_VERSION 8
#define MIP360_OK 0
#define MIP360_ERR_ARG 1
#define MIP360_ERR_HIP 2

#define MIP360_N_BASIS 21          /* geopoly.generate_basis('icosahedron', 2) */
#define MIP360_IPE_DIM 504         /* 21 * 2 * 12 */
#define MIP360_IPE_LD 512          /* row stride of the encoded-sample tensor (zero padded) */
#define MIP360_MAX_BINS 128        /* bins of a step function handed to mip360_resample */
#define MIP360_MAX_SAMPLES 64      /* samples per ray per level */
#define MIP360_DEPTH_NONE 0
#define MIP360_DEPTH_MSE 1
#define MIP360_DEPTH_L1 2
#define MIP360_DEPTH_KL 3          /* ds_nerf_depth_loss            internal/depth_loss.py:5-29 */
#define MIP360_DEPTH_URF 4         /* urban_radiance_field_depth_loss                     :31-65 */

const char* mip360_last_error(void);
int mip360_abi_version(void);

/* One sampling level of Model.__call__ (models.py:158-208):
 *   max_dilate_weights(sdist, weights, dilation, domain, renormalize=True)[1:-1]      stepfun.py:100-130
 *   logits = where(s[1:] > s[:-1], anneal * log(w + resample_padding), -inf)          models.py:179-183
 *   sdist' = sample_intervals(rng, sdist, logits, num_samples, single_jitter=True, domain)   stepfun.py:166-270
 *   tdist' = s_to_t(sdist')  with raydist_fn = reciprocal                             coord.py:63-100
 * sdist_in [n, m_in+1], weights_in [n, m_in]; dilation <= 0 skips the dilation (level 0, where m_in = 1).
 * jitter01 [n] in [0,1) replaces jax.random.uniform (one value per ray: single_jitter); NULL = deterministic.
 * Outputs sdist_out, tdist_out [n, num_samples+1]. */
int mip360_resample(void* stream, int n_rays, int m_in, const float* sdist_in, const float* weights_in,
                    float dilation, float anneal, float resample_padding, int num_samples,
                    const float* jitter01, float s_near, float s_far, const float* t_near,
                    const float* t_far, float* sdist_out, float* tdist_out);

/* Featurise the conical frustums of one level (models.py:213-226 + MLP.predict_density :449-456):
 *   cast_rays(tdist, o, d, radii, 'cone', diag=False)        render.py:45-82,108-133
 *   track_linearize(contract, means, covs)                    coord.py:21-27,39-60
 *   lift_and_diagonalize(., ., basis)                         coord.py:131-135
 *   integrated_pos_enc(., ., 0, 12)                           coord.py:103-128
 * tdist [n, S+1]; origins, directions [n,3]; radii [n]; basis_t [3, 21] row-major.
 * enc [n*S, ld]: float32 (out_bf16 = 0) or bfloat16 (out_bf16 = 1); columns 504..min(ld, 512)-1 are zero-filled (K
 * padding of the first dense layer); ld may be larger when enc is a column window of a wider row (the skip buffer).
 * out_bf16 = 2: bfloat16 in the fragment-major layout below -- enc = the fm tensor's base + 1024 * (first column / 16)
 * bytes, ld = the tensor's columns (a multiple of 16), n * S a multiple of 32; bit-identical values. */
int mip360_cast_encode(void* stream, int n_rays, int n_samples, const float* tdist,
                       const float* origins, const float* directions, const float* radii,
                       const float* basis_t, void* enc, int out_bf16, int ld);

/* compute_alpha_weights (render.py:136-158) + volumetric_rendering (render.py:161-216: rgb, acc, distance_mean,
 * depth; the percentile outputs are not produced).  density [n,S], rgb_samples [n,S,3] or NULL (proposal levels),
 * tdist [n,S+1], directions [n,3].  Outputs may be NULL except weights. */
int mip360_render_level(void* stream, int n_rays, int n_samples, const float* density,
                        const float* rgb_samples, const float* tdist, const float* directions,
                        int opaque_background, float bg_rgb, float* weights, float* rgb, float* acc,
                        float* distance_mean, float* depth);

/* backward of the above (upstream: jax autograd): given dL/d weights [n,S] (interlevel / distortion / kl terms,
 * may be NULL), dL/d rgb [n,3] (may be NULL) and dL/d distance_mean [n] (may be NULL), writes dL/d density [n,S]
 * and dL/d rgb_samples [n,S,3] (NULL for proposal levels). */
int mip360_render_level_backward(void* stream, int n_rays, int n_samples, const float* density,
                                 const float* rgb_samples, const float* tdist, const float* directions,
                                 int opaque_background, float bg_rgb, const float* g_weights,
                                 const float* g_rgb, const float* g_distance_mean, float* g_density,
                                 float* g_rgb_samples);

/* Loss terms of one training step with their gradients (train_utils.py:72-169, loss_fn :258-300):
 *   data term of the NeRF level: charb (sqrt(resid^2 + charb_padding^2)) or mse, mean over [n,3]   :82-107
 *   depth term on `distance_mean`: 'mse' ((m dm - m sup)^2).mean() or 'l1', m = sup > 0, averaged over ALL rays :108-119.
 *     Upstream counts it twice: inside the data loss, data_loss_mult * lambda_depth * depth[nerf level] (:136-139, the
 *     coarse levels carry data_coarse_loss_mult = 0), and again as stats['loss_disp_mse'] = lambda_depth * sum over
 *     ALL levels (:143, added to the total at :268-269).  depth_weight multiplies lambda_depth for the NeRF level
 *     (reference: 2 = 1 + 1), prop_depth_weight for the proposal levels' distance_mean (reference: 1; dm_prop NULL
 *     or weight 0 drops them).
 *   interlevel loss: mean lossfun_outer(c, w, c_prop, w_prop) per proposal level (gradient to w_prop only)  :149-160
 *   distortion loss: distortion_mult * mean lossfun_distortion(c, w)                                   :163-169
 * sdist_nerf [n,Sn+1], w_nerf [n,Sn]; per proposal level k < n_prop: sdist_prop[k] [n,Sp+1], w_prop[k] [n,Sp].
 * scalars[6] = {total, data, depth of the NeRF level (unweighted), interlevel, distortion, sum of the proposal
 * levels' depth terms (unweighted)}.  Gradients: g_rgb [n,3], g_distance_mean [n], g_w_nerf [n,Sn],
 * g_w_prop[k] [n,Sp], g_dm_prop[k] [n].  depth_loss_type: 0 none, 1 mse, 2 l1.  workspace >= (4 + n_prop) * n floats. */
int mip360_losses(void* stream, int n_rays, int s_nerf, int s_prop, int n_prop, const float* rgb,
                  const float* rgb_gt, const float* distance_mean, const float* depth_sup,
                  const float* sdist_nerf, const float* w_nerf, const float* const* sdist_prop,
                  const float* const* w_prop, int charb, float charb_padding, float data_loss_mult,
                  int depth_loss_type, float lambda_depth, float depth_weight, float interlevel_mult,
                  float distortion_mult, float* scalars, float* g_rgb, float* g_distance_mean,
                  float* g_w_nerf, float* const* g_w_prop, float* workspace, float prop_depth_weight,
                  const float* const* dm_prop, float* const* g_dm_prop);

/* depth_loss.depth_loss(weights, tdist, termination_depth, predicted_depth, sigma, dirs, type) for type 'kl' / 'urf' of ONE
 * sampling level (internal/depth_loss.py:67-102, called per level at train_utils.py:121-128), value and gradients:
 *   steps = 0.5 (tdist[:-1] + tdist[1:]); kl: lengths = (tdist[1:] - tdist[:-1]) * |directions|,
 *   loss = -log(w + 1e-7) * exp(-(steps - gt)^2 / (2 sigma)) * lengths (:24); urf: expected-depth + line-of-sight terms
 *   with N(0, sigma / 3) (:44-60).
 * Upstream reduces with loss.sum(-2) -- over the RAY axis of [n_rays, n_samples] -- and multiplies the [n_samples]
 * result by the [n_rays] mask (:25-27, :57-64): that broadcasts only for n_rays == n_samples or n_rays == 1.  This entry
 * point reproduces exactly that (column s meets the mask / expected term of ray s, or of ray 0) and returns
 * MIP360_ERR_ARG with the broadcasting message for any other shape, like JAX raises.
 * weights [n,S], tdist [n,S+1], depth_sup [n], distance_mean [n] (urf, else NULL), directions [n,3] (kl, else NULL).
 * loss_out[0] = the value.  g_weights [n,S] / g_distance_mean [n] (NULL to skip) are ACCUMULATED: += scale * d value / d x
 * (scale = the term's weight in the total: lambda_depth for a proposal level, (data_loss_mult + 1) * lambda_depth for the
 * NeRF level, train_utils.py:136-143).  total_accum (NULL or float[2]): [0] += scale * value, [1] += value. */
int mip360_depth_loss_klurf(void* stream, int depth_loss_type, int n_rays, int n_samples, const float* weights,
                            const float* tdist, const float* depth_sup, const float* distance_mean,
                            const float* directions, float sigma, float scale, float* loss_out, float* g_weights,
                            float* g_distance_mean, float* total_accum);

/* One dense layer on the matrix cores: C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]), bf16 operands (row-major, K
 * contiguous, leading dimensions lda / ldw in elements, multiples of 8), float32 accumulation
 * (v_mfma_f32_32x32x16_bf16).  Replaces flax nn.Dense + nn.relu in MLP.__call__ (models.py:436-606).
 * K must be a multiple of 32 (pad with zeros); M, N arbitrary.  act: 0 none, 1 relu, 2 softplus(v + act_param) (the
 * density head: density_activation(raw + density_bias), models.py:497), 3 sigmoid(v) * (1 + 2 act_param) - act_param
 * (the colour head with rgb_padding, models.py:573-594), 4 v * (aux[m][n] > 0) (backward through a ReLU: aux = the
 * layer's saved bf16 output [M, ldaux]; with a = dZ of the next layer and w = its kernel [in, out] this is the dX
 * chain dZ_l = (dZ_{l+1} K_{l+1}^T) * relu'(H_l)).  Either output may be NULL: c_bf16 [M, ldc] bfloat16,
 * c_f32 [M, ldc32] float32. */
int mip360_linear_bf16(void* stream, int m, int n, int k, const void* a, int lda, const void* w, int ldw,
                       const float* bias, int act, float act_param, void* c_bf16, int ldc, float* c_f32,
                       int ldc32, const void* aux, int ldaux);

/* The same layer with its ReLU pattern kept as one BIT per element instead of being re-derived from the saved bf16
 * output in the backward pass (jax.grad of nn.relu: the cotangent is passed where the output is > 0).
 *   mip360_linear_relu_mask_bf16 : C = relu(A W^T + b) as bf16, and mask byte [(n >> 3) * ldmask + m], bit (n & 7) =
 *                                  (C[m][n] > 0) -- column-byte-major, so that the 16 consecutive rows a thread of the
 *                                  GEMM epilogue owns are 16 consecutive bytes.
 *   mip360_linear_masked_bf16    : C = (A W^T) with the elements whose mask bit is clear set to zero: the dX chain
 *                                  dZ_l = (dZ_{l+1} K_{l+1}^T) * relu'(H_l) reading 1/16 of the bytes act 4 reads
 *                                  (the 16 mask bytes of a thread are fetched ahead of its K loop).
 * The mask buffer holds mip360_relu_mask_bytes(m, n, &ldmask) bytes; ldmask = m rounded up to 256. */
int64_t mip360_relu_mask_bytes(int m, int n, int* ldmask);
int mip360_linear_relu_mask_bf16(void* stream, int m, int n, int k, const void* a, int lda, const void* w, int ldw,
                                 const float* bias, void* c_bf16, int ldc, void* mask, int ldmask);
int mip360_linear_masked_bf16(void* stream, int m, int n, int k, const void* a, int lda, const void* w, int ldw,
                              void* c_bf16, int ldc, const void* mask, int ldmask);

/* ---- fragment-major ("fm") dense layers: the wide layers' activations in the order the matrix cores consume and
 * produce them (csrc/mip360_fm.hip).  A [rows, ld] bf16 tensor (rows % 32 == 0, ld % 16 == 0) is 1 KiB blocks of
 * 32 rows x 16 columns, block (r / 32, c / 16) at ((r / 32) * (ld / 16) + c / 16) * 1024 bytes; in a block, element
 * (row, f) -- hi = (f / 4) % 2, t = 4 (f / 8) + f % 4 -- is at byte 16 * (8 (row >> 2) + 4 (hi ^ (row >> 4)) + (row & 3))
 * + 2 t.  Replaces nn.Dense + nn.relu of MLP.__call__ (internal/models.py:436-606) for the 256- / 1024-wide layers.
 *   mip360_to_fm / mip360_from_fm : row-major bf16 [rows, cols] (row stride ld_src / ld_dst elements) <-> columns
 *       [col0, col0 + cols) of an fm tensor with ld columns.
 *   mip360_linear_fm : C = act(A W^T + b); A [m, k], W [n, k], C [m, n] all fm; m, n multiples of 256, k a multiple of
 *       32 (>= 160).  act 0: bias only; 1: ReLU, and one bit per output (non-zero) to `mask` (mip360_fm_mask_bytes
 *       bytes); 2: no bias, outputs whose bit in `mask` (written by an act-1 call with the same m, n) is clear are
 *       zeroed -- the dX chain.  The bias is added by the matrix cores as bf16 hi + lo (2^-17 relative). */
int mip360_to_fm(void* stream, int rows, int cols, const void* src_bf16, int ld_src, void* dst_fm, int ld_dst, int col0_dst);
int mip360_from_fm(void* stream, int rows, int cols, const void* src_fm, int ld_src, int col0_src, void* dst_bf16, int ld_dst);
int64_t mip360_fm_mask_bytes(int m, int n);
int mip360_linear_fm(void* stream, int m, int n, int k, const void* a_fm, int lda, const void* w_fm, int ldw,
                     const float* bias, int act, void* c_fm, int ldc, void* mask);
/* n <= 8 weight-gradient problems over the same m rows in ONE launch (the PropMLP's four layers): slabs[p] receives what
 * mip360_grad_weight_fm(grad_kernel = NULL) writes for problem p with this ksplit (ksplit x n_in[p] x n_out[p] floats, then
 * ksplit x n_out[p] bias slabs); sum them with mip360_grad_weight_reduce.  With the workgroups of all
 * problems on the chip together, ksplit ~ 256 / (total 256 x 256 tiles) fills it: a fraction of the split-K slab traffic. */
int mip360_grad_weight_fm_multi(void* stream, int n, int m, int ksplit, const int* n_in, const int* n_out, const void* const* h_fm,
                                const int* ldh, const void* const* dz_fm, const int* lddz, float* const* slabs);
/* One output column (the density head): out[m * ldo] = act(A[m, :k] . w + bias[0]); A fm, w bf16 [k]; act 0 / 1 (ReLU) /
 * 2 (softplus(x + act_param)).  mip360_grad_weight_col_fm: d kernel[i] = scale * sum_m H[m][i] z[m], z = column zcol of
 * the fm tensor dz; slabs >= ksplit * (n_in + 1) floats; grad_bias = scale * sum_m z[m]. */
int mip360_rowdot_fm(void* stream, int m, int k, const void* a_fm, int lda, const void* w_bf16, const float* bias, int act,
                     float act_param, float* out, int ldo);
int mip360_grad_weight_col_fm(void* stream, int m, int n_in, const void* h_fm, int ldh, const void* dz_fm, int lddz, int zcol,
                              int ksplit, float* slabs, float* grad_kernel, float scale, float* grad_bias);
/* The PropMLP forward (models.py:436-606 with configs/360.gin:12-13: four 256-wide ReLU layers on the 512 padded IPE columns,
 * density head) as ONE launch: = four mip360_linear_fm (act 1) + mip360_rowdot_fm (act 2) with the activations in registers
 * between the layers (csrc/mip360_prop.hip).  x_fm [rows, ldx] fm, columns [x_col0, x_col0 + 512); w_fm[l] fm [256, ldw[l]]
 * (the fwd_fm copies of mip360_pack_weight_fm), bias[l] [256]; rows a multiple of 256.  h_fm / masks (4 pointers each, both
 * or neither): H_l [rows, 256] fm and the ReLU mask of layer l in mip360_linear_fm's format (mip360_fm_mask_bytes(rows, 256)
 * each) -- what the backward pass reads; NULL: inference, nothing but the density leaves the chip.  density (may be NULL
 * when training outputs are requested) [rows] = softplus(H_3 . wd + bd[0] + act_param), bit-identical to mip360_rowdot_fm
 * on the H_3 this call writes.  Layer outputs agree with mip360_linear_fm up to the summation order (float bias first here,
 * bf16 hi + lo bias last there). */
int mip360_prop_mlp_fm(void* stream, int rows, const void* x_fm, int ldx, int x_col0, const void* const* w_fm, const int* ldw,
                       const float* const* bias, void* const* h_fm, void* const* masks, const void* wd_bf16, const float* bd,
                       float act_param, float* density);
/* The dX chain of the same MLP as ONE launch: dZ_3 = mask_3 . bf16(z (x) wd) (= mip360_outer_masked_fm), dZ_{l-1} = mask_{l-1} .
 * bf16(dZ_l W_l) (= mip360_linear_fm act 2) for l = 3, 2, 1, dZ_l in registers between the layers; all four dZ_l [rows, 256] fm
 * are written for mip360_grad_weight_fm.  z bf16 [rows] (d loss / d raw density), wd bf16 [256], masks[l] as written by
 * mip360_prop_mlp_fm / mip360_linear_fm act 1, wb_fm[l] (l = 1..3; entry 0 ignored) the bwd_fm copies of
 * mip360_pack_weight_fm [256, ldwb[l]].  Bit-identical to the four launches it replaces. */
int mip360_prop_mlp_bwd_fm(void* stream, int rows, const void* z_bf16, const void* wd_bf16, const void* const* masks,
                           const void* const* wb_fm, const int* ldwb, void* const* dz_fm);
/* The view branch of the NerfMLP, forward, as ONE launch (models.py:560-606): [bottleneck (256) | pos_enc(viewdirs, 0, 4) (27 -> 32)]
 * -> Dense(128) + ReLU -> Dense(3) -> sigmoid * (1 + 2 rgb_padding) - rgb_padding; = mip360_from_fm + mip360_dir_encode + two
 * mip360_linear_bf16 (act 1 / 3).  bott_fm [rows, 256] fm (mip360_linear_fm act 0); dir_table_bf16 [rows / n_samples, 32]: the rays'
 * direction features, written by mip360_dir_encode(n_rays, S = 1, viewdirs, table, ld 32, col0 0, width 32); w1_fm fm
 * [128, ldw1 >= 288], w2_fm fm [32, ldw2 >= 128] with rows 3..31 zero (fwd_fm copies of mip360_pack_weight_fm), b1 [128], b2 [3].
 * view_in_bf16 [rows, ld_view >= 288] / h_bf16 [rows, ld_h >= 128]: the row-major operands the backward entry points read
 * (either may be NULL: not written); rgb [rows, 3].  rows a multiple of 256. */
int mip360_view_branch_fm(void* stream, int rows, int n_samples, const void* bott_fm, const void* dir_table_bf16, const void* w1_fm, int ldw1,
                          const float* b1, const void* w2_fm, int ldw2, const float* b2, float rgb_padding, void* view_in_bf16,
                          int ld_view, void* h_bf16, int ld_h, float* rgb);
/* The heads' backward up to the bottleneck as ONE launch: = mip360_head_backward + mip360_linear_bf16 act 4 (d_hz = relu'(h) .
 * (d_pre W_rgb)) + mip360_linear_bf16 (d_bott = d_hz W_view[:, :256]) + mip360_to_fm.  density / g_density [rows], rgb / g_rgb
 * [rows, 3]; h_bf16 [rows, ld_h] the saved ReLU output; wb_rgb_fm fm [128, ld_wb_rgb >= 32] and wb_view_fm fm [>= 256, ld_wb_view >=
 * 128]: the bwd_fm copies of mip360_pack_weight_fm of the two view-branch kernels.  Written: d_pre_bf16 [rows, 32] and d_hz_bf16
 * [rows, ld_dhz] row-major (operands of the two mip360_grad_weight_bf16 launches of the branch), heads_fm [rows, 320] fm =
 * [d_bott (256) | d raw density | 0 ...] (operand of the heads' weight gradients and of the trunk's first masked dX). */
int mip360_view_branch_bwd_fm(void* stream, int rows, const float* density, const float* g_density, const float* rgb, const float* g_rgb,
                              float rgb_padding, const void* h_bf16, int ld_h, const void* wb_rgb_fm, int ld_wb_rgb, const void* wb_view_fm,
                              int ld_wb_view, void* d_pre_bf16, void* d_hz_bf16, int ld_dhz, void* heads_fm);
/* c_fm[m][n] = bf16(z[m] * w[n]) where bit (m, n) of `mask` is set (z bf16 [m], w bf16 [n]): mip360_linear_fm act 2 for a
 * one-column operand -- the PropMLP's dZ of the last trunk layer (its only head is the density column).
 * mip360_grad_weight_col_fm with lddz == 1 reads z from such a plain vector. */
int mip360_outer_masked_fm(void* stream, int m, int n, const void* z_bf16, const void* w_bf16, const void* mask, void* c_fm, int ldc);
/* mip360_pack_weight (below) that also writes the fm operand copies: fwd_fm [n_out, ld_fwd_fm] = kernel^T, bwd_fm element
 * (i, bwd_col0 + o) for i < bwd_rows; either may be NULL; padding is never written (zero the buffers once). */
int mip360_pack_weight_fm(void* stream, int n_in, int n_out, const float* kernel, void* fwd_bf16, int ld_fwd, void* bwd_bf16, int ld_bwd,
                          void* fwd_fm, int ld_fwd_fm, void* bwd_fm, int ld_bwd_fm, int bwd_rows, int bwd_col0);
/* n calls of mip360_pack_weight_fm as ONE launch (all parameter tensors of an MLP after an Adam step: twelve 5-10 us launches on
 * the update stream otherwise run next to the following step's first kernels).  descs: host array, 1 <= n <= 16. */
typedef struct mip360_pack_desc {
  const float* kernel; int32_t n_in, n_out;
  void* fwd_bf16; void* bwd_bf16; void* fwd_fm; void* bwd_fm;
  int32_t ld_fwd, ld_bwd, ld_fwd_fm, ld_bwd_fm, bwd_rows, bwd_col0;
} mip360_pack_desc;
#define MIP360_PACK_BATCH_MAX 16
int mip360_pack_weights_fm_batch(void* stream, int n, const mip360_pack_desc* descs);
/* mip360_grad_weight_bf16 (below) with both operands in fm layout; n_in, n_out multiples of 256, m of 32.  Same slab
 * contract: grad_kernel == NULL leaves the sums to mip360_grad_weight_reduce. */
int mip360_grad_weight_fm(void* stream, int m, int n_in, int n_out, const void* h_fm, int ldh, const void* dz_fm, int lddz,
                          int ksplit, float* slabs, float* grad_kernel, int ldg, float scale, float* grad_bias);

/* ---- training side (upstream: jax.value_and_grad + optax, train_utils.py:215-236, 303-370) ---------------------- */

/* d kernel [n_in, n_out] (flax layout, float32, row stride ldg) = scale * H[m, n_in]^T dZ[m, n_out]: bf16 operands,
 * MFMA with ds_read_b64_tr_b16 transposed fragments, split over `ksplit` row slices into `slabs`
 * (>= ksplit * (n_in * ldg + n_out) floats) that are summed in a fixed order.  grad_bias [n_out] (may be NULL) =
 * scale * column sums of dZ, accumulated from the fragments the first input tile already holds.
 * grad_kernel == NULL: only the slabs are produced (grad_bias != NULL still asks for the bias slabs); the caller sums
 * them later with mip360_grad_weight_reduce, e.g. on another stream under the next layer's GEMMs. */
int mip360_grad_weight_bf16(void* stream, int m, int n_in, int n_out, const void* h, int ldh, const void* dz,
                            int lddz, int ksplit, float* slabs, float* grad_kernel, int ldg, float scale,
                            float* grad_bias);
/* The fixed-order sum of the slabs written by mip360_grad_weight_bf16(m, n_in, n_out, ..., ksplit, slabs, ldg): kernel
 * gradient rows [0, rows) (rows <= n_in: the zero-padded input columns of a layer are dropped here) and, if grad_bias is
 * not NULL, the bias gradient, in one launch. */
int mip360_grad_weight_reduce(void* stream, int rows, int n_in, int n_out, int ksplit, const float* slabs,
                              float* grad_kernel, int ldg, float scale, float* grad_bias);
/* Output tile edge (256 or 128) the call above will use for these sizes: the caller picks ksplit (1..256) so that
 * ceil(n_in / tile) * ceil(n_out / tile) * ksplit fills the 256 CUs (multiples of 8 keep a row slice on one XCD). */
int mip360_grad_weight_tile(int m, int n_in, int n_out, int ldh, int lddz);

/* d bias [n_out] = scale * column sums of dZ [m, n_out] (bf16); partial >= nslice * n_out floats. */
int mip360_grad_bias_bf16(void* stream, int m, int n_out, const void* dz, int lddz, int nslice, float* partial,
                          float* grad_bias, float scale);

/* Backward through the heads (models.py:497, 573-594): d raw_density = g_density * (1 - exp(-density)) written to
 * column raw_col of a bf16 [rows, ld_raw] tensor (columns raw_col+1 .. raw_zero_to-1 zero-filled: K padding);
 * d rgb_pre = g_rgb (1 + 2p) s (1 - s), s = (rgb + p) / (1 + 2p), to columns 0..2 of a bf16 [rows, 32] tensor
 * (3..31 zero-filled); d_pre_bf16 NULL for the proposal MLP. */
int mip360_head_backward(void* stream, int64_t rows, const float* density, const float* g_density,
                         const float* rgb, const float* g_rgb, float rgb_padding, void* d_raw_bf16, int ld_raw,
                         int raw_col, int raw_zero_to, void* d_pre_bf16);

/* clip_gradients (train_utils.py:215-236): per-tensor partial sums of squares (n_blocks floats each, deterministic),
 * then mult = min(1, grad_max_norm / (eps + sqrt(sum of all partials))) over ONE MLP's tensors;
 * mult_and_norm[2] = {mult, norm} stays on the device and feeds mip360_adam_step. */
int mip360_sum_squares(void* stream, int64_t n, const float* g, float* partial, int n_blocks);
int mip360_clip_multiplier(void* stream, int n_partial, const float* partial, float grad_max_norm,
                           float* mult_and_norm);

/* optax.adam step on one tensor (train_utils.py:303-331; defaults lr 2e-3 -> 2e-5 log-decay with 512 warm-up steps,
 * b1 0.9, b2 0.999, eps 1e-6, configs.py:118-124); grad_mult: device scalar from mip360_clip_multiplier or NULL. */
int mip360_adam_step(void* stream, int64_t n, float* params, const float* grads, float* mu, float* nu,
                     const float* grad_mult, int step, double lr, double beta1, double beta2, double eps);

/* float32 flax kernel [n_in, n_out] -> the two bf16 operand copies of the dense-layer kernel: fwd [n_out, ld_fwd]
 * (transposed; the caller zero-fills the K padding once) and bwd [n_in, ld_bwd] (as stored). */
int mip360_pack_weight(void* stream, int n_in, int n_out, const float* kernel, void* fwd_bf16, int ld_fwd,
                       void* bwd_bf16, int ld_bwd);

/* View-direction encoding of the NerfMLP's second stage (models.py:395-399,548-553): pos_enc(viewdirs, 0, 4,
 * append_identity=True) = 27 values per ray, broadcast over the ray's samples into columns [col0, col0 + 27) of a
 * bfloat16 [n*S, ld] tensor (the bottleneck GEMM writes columns [0, 256) of the same rows); columns up to
 * col0 + width are zero-filled (K padding of the next layer). */
int mip360_dir_encode(void* stream, int n_rays, int n_samples, const float* viewdirs, void* out_bf16, int ld,
                      int col0, int width);

#ifdef __cplusplus
}
#endif
#endif /* MIP360_HIP_H */
