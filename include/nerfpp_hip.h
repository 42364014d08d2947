/* nerfpp_hip.h -- C ABI of libnerfpp_hip.so, the MI355X (gfx950) implementation of the NeRF++
 * depth-supervised render/train hot path of cwchenwang/outdoor-nerf-depth.
 *
 * The reference has no FFI: its hot path is PyTorch calls inside
 * nerf-methods/nerfplusplus/ddp_train_nerf.py (per-cascade-level loop :432-498 training,
 * :156-221 inference) and ddp_model.py:74-147 (NerfNet.forward).  Each entry point below names
 * the reference call it replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers (float32 unless stated), sizes, and a `void* stream`
 *     (a hipStream_t; NULL = default stream).  No torch types, no exceptions across the ABI.
 *   - every function returns NERFPP_OK (0) or an error code; nerfpp_last_error() returns a
 *     thread-local description.  Kernels are enqueued asynchronously on `stream`.
 *   - the library is stateless: the caller owns every buffer (parameters, packed weights, index
 *     tables, workspace, outputs).  Sizes come from the *_bytes / *_sizes queries.
 *   - all arrays are dense row-major; "rows" means n_rays * n_samples in (ray, sample) order.
 *   - network shape is fixed to the reference's only configuration: netdepth 8, netwidth 256,
 *     skip at layer 4, max_freq_log2 10 / 4 (configs/kitti.txt:38-44).
 */
#ifndef NERFPP_HIP_H
#define NERFPP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERFPP_ABI_VERSION 8

#define NERFPP_OK 0
#define NERFPP_ERR_ARG 1          /* bad argument (null pointer, size out of range) */
#define NERFPP_ERR_HIP 2          /* a HIP runtime call / kernel launch failed */
#define NERFPP_ERR_INTERNAL 3
#define NERFPP_ERR_UNSUPPORTED 4
#define NERFPP_ERR_COMM 5         /* RCCL missing or an RCCL call failed: nerfpp_comm_last_error() */
#define NERFPP_ERR_LAUNCH NERFPP_ERR_HIP

/* precision of the MLP kernels */
#define NERFPP_PREC_BF16 1        /* single-pass bf16 MFMA, f32 accumulate ("speed") */
#define NERFPP_PREC_SPLIT_BF16 2  /* hi+lo split bf16, 3 MFMA passes, ~1e-5 rel. of float32 ("parity") */
#define NERFPP_PREC_FP16X2W 3     /* FORWARD only (nerfpp_level_forward, nerfpp_pack_level, workspace queries): weights hi+lo in
                                     fp16, activations rounded to fp16 once, 2 passes of v_mfma_f32_32x32x16_f16.  An
                                     INTERMEDIATE precision: within 1e-4 of float32 at initialisation, 3-4e-4 (rendered rgb) on
                                     trained weights (tests/test_gpu_round5.py) -- 10x tighter than precision 1 at 1.5x its
                                     forward time; the 1e-4 clause is precision 2's.  Its training-mode forward saves
                                     single-plane bf16 tensors: run the backward with precision 1, workspace_precision 3.
                                     Activations beyond +-65504 overflow (the reference's trained nets stay far below). */

/* depth loss types (ddp_train_nerf.py:20-26; `los` / `nll` are dead code in the reference) */
#define NERFPP_LOSS_RGB_ONLY 0
#define NERFPP_LOSS_MSE 1
#define NERFPP_LOSS_L1 2
#define NERFPP_LOSS_KL 3

#define NERFPP_FG_PARAMS 595844     /* MLPNet(input_ch=63, input_ch_viewdirs=27).parameters() */
#define NERFPP_BG_PARAMS 606596     /* MLPNet(input_ch=84, ...) */
#define NERFPP_LEVEL_PARAMS 1202440 /* NerfNet.parameters(): fg_net then bg_net, state_dict order */
#define NERFPP_MAX_SAMPLES 256      /* samples per ray per level (reference: 64 and 64+128) */

const char* nerfpp_last_error(void);
int nerfpp_abi_version(void);

/* ---------------------------------------------------------------- sampling (float32, bit-exact bins) */

/* intersect_sphere(ray_o, ray_d)                                   ddp_train_nerf.py:51-66
 * *bad_count (device int, caller zeroes it) is incremented for every ray whose closest point to
 * the origin lies outside the unit sphere (the reference raises an Exception). */
int nerfpp_intersect_sphere(void* stream, int n_rays, const float* ray_o, const float* ray_d,
                            float* fg_far, int* bad_count);

/* level-0 depths: fg[i] = near + i*step, bg = linspace(0,1,S), each followed by
 * perturb_samples when t_rand_* is non-NULL                        ddp_train_nerf.py:438-449, :166-175
 * t_rand_* [n_rays, S] replace torch.rand_like so results are reproducible from outside. */
int nerfpp_sample_coarse(void* stream, int n_rays, int n_samples, const float* ray_o,
                         const float* ray_d, const float* min_depth, const float* t_rand_fg,
                         const float* t_rand_bg, float* fg_far, float* fg_z, float* bg_z,
                         int* bad_count);

/* The same with the stratified jitter drawn inside the kernel: t_rand = Philox4x32-10 keyed by `seed`,
 * counter (element index, stream id, step) -- what torch.rand_like does in the reference
 * (ddp_train_nerf.py:71), without the two extra launches and [n,S] buffers.  Stream ids: 0 fg, 1 bg. */
int nerfpp_sample_coarse_rng(void* stream, int n_rays, int n_samples, const float* ray_o,
                             const float* ray_d, const float* min_depth, uint64_t seed, uint64_t step,
                             float* fg_far, float* fg_z, float* bg_z, int* bad_count);

/* n uniforms of stream `stream_id` (0 fg jitter, 1 bg jitter, 2 fg sample_pdf u, 3 bg sample_pdf u) of
 * (seed, step): exactly the values the *_rng entry points consume, element i = flat index [ray, sample].
 * Exposed so a caller (and the parity tests) can replay a step through the explicit-uniform calls. */
int nerfpp_rng_uniform(void* stream, uint64_t seed, uint64_t step, int stream_id, int64_t n, float* out);

/* perturb_samples(z_vals) with explicit uniforms                   ddp_train_nerf.py:69-78 */
int nerfpp_perturb_samples(void* stream, int n_rays, int n_samples, const float* z_vals,
                           const float* t_rand, float* out);

/* sample_pdf(bins [n,M+1], weights [n,M], N_samples, det)          ddp_train_nerf.py:81-130
 * u [n, n_new] = the uniforms, or NULL for det=True (linspace(0,1,n_new)).
 * samples [n, n_new]; above_inds [n, n_new] int64 (may be NULL). */
int nerfpp_sample_pdf(void* stream, int n_rays, int n_bins_m, int n_new, const float* bins,
                      const float* weights, const float* u, float* samples, int64_t* above_inds);

/* fine depths of one volume: bins = mid-points of z_old, weights = w[:, 1:-1], sample_pdf, then
 * sort(cat(z_old, samples))                                        ddp_train_nerf.py:450-465
 * z_old, weights [n, S_old]; z_merged [n, S_old + n_new]; samples / above_inds may be NULL. */
int nerfpp_sample_fine(void* stream, int n_rays, int s_old, int n_new, const float* z_old,
                       const float* weights, const float* u, float* z_merged, float* samples,
                       int64_t* above_inds);

/* the same for the foreground and the background volume of a level in one launch (the two calls at
 * ddp_train_nerf.py:450-457 and :458-465); u may be NULL for both (det=True). */
int nerfpp_sample_fine_pair(void* stream, int n_rays, int s_old, int n_new, const float* fg_z_old,
                            const float* fg_weights, const float* fg_u, float* fg_z_merged,
                            const float* bg_z_old, const float* bg_weights, const float* bg_u,
                            float* bg_z_merged);

/* the same with u = torch.rand(N_rays, N_samples) (ddp_train_nerf.py:104) drawn inside the kernel
 * (Philox streams 2 = fg, 3 = bg of (seed, step)). */
int nerfpp_sample_fine_pair_rng(void* stream, int n_rays, int s_old, int n_new, const float* fg_z_old,
                                const float* fg_weights, float* fg_z_merged, const float* bg_z_old,
                                const float* bg_weights, float* bg_z_merged, uint64_t seed, uint64_t step);

/* ray batch from a GPU-resident frame (nerf_sample_ray_split.py:10-34 get_rays_single_image and
 * :178-221 random_sample): for every flat pixel index pix[i] (int64, row-major H x W)
 *   ray_d = c2w[:3,:3] * K^-1 * [u+.5, v+.5, 1]^T (un-normalised), ray_o = c2w[:3,3],
 *   rgb / depth_sup gathered from the frame's images, min_depth = 1e-4.
 * cam: 21 floats on the device = K^-1 (3x3 row-major) then c2w[:3,:4] (row-major).
 * rgb_img [H*W,3] / depth_img [H*W] and their outputs may be NULL. */
/* The pixel draw of RaySamplerSingleImage.random_sample (nerf_sample_ray_split.py:178: np.random.choice(H * W, size=(N_rand,),
 * replace=False)): n_rays DISTINCT flat pixel indices in [0, n_pixels), uniform over ordered tuples of distinct values, drawn
 * inside one kernel from the counter-based generator (stream 4 of (seed, step)) instead of permuting all H * W pixels.
 * pix: int64 [n_rays] on the device.  1 <= n_rays <= 8192 <= ... <= n_pixels < 2^31. */
int nerfpp_sample_pixels(void* stream, uint64_t seed, uint64_t step, int64_t n_pixels, int n_rays, int64_t* pix);

int nerfpp_gather_rays(void* stream, int n_rays, int width, const float* cam, const int64_t* pix,
                       const float* rgb_img, const float* depth_img, float* ray_o, float* ray_d,
                       float* rgb, float* depth_sup, float* min_depth);

/* ---------------------------------------------------------------- parameters */

/* Index tables tying the reference's flat parameter order to the packed MFMA weight streams.
 * Host-side, no GPU needed.  net: 0 = fg_net, 1 = bg_net. */
int nerfpp_table_sizes(int net, int64_t* fwd_elems, int64_t* fwd_bias_elems, int64_t* bwd_elems,
                       int64_t* slab_floats, int64_t* n_params);
int nerfpp_build_tables(int net, int32_t* fwd_tbl, int32_t* bias_tbl, int32_t* bwd_tbl,
                        int32_t* unpack_tbl);
/* all tables of one level in one int32 buffer (what the device-side calls take) */
int64_t nerfpp_level_tables_elems(void);
int nerfpp_build_level_tables(int32_t* host_tables);

/* Host-side introspection of the weight-gradient launch plan for a level with `rows` = n_rays * n_samples and a backward at
 * `backward_precision` (1 or 2): k_out[net * 10 + job] = row slices (= workgroups = gradient slabs) of job `job` of net `net`
 * (job order: L0, L1..L4, L5 (encoded-point and h4 columns), L6, L7, [sigma | rgb0] (M and view-dir columns),
 * rgb1), is_full_out[...] = 1 for the 256 x 256 jobs.  In a bf16 backward job L1 recomputes its input H0 from the encoded point
 * and gets more slices than the other full jobs.  No GPU needed.  Returns the number of jobs per net (10). */
int nerfpp_dw_plan(int64_t rows, int backward_precision, int32_t* k_out, int32_t* is_full_out);

/* packed weights of one level (both nets, forward + backward streams + biases) */
int64_t nerfpp_packed_bytes(int precision);
/* params: the level's NERFPP_LEVEL_PARAMS float32 values in NerfNet.parameters() order */
int nerfpp_pack_level(void* stream, int precision, const float* params, const int32_t* tables,
                      void* packed);

/* ---------------------------------------------------------------- one cascade level */

int64_t nerfpp_workspace_bytes(int n_rays, int n_samples, int precision, int training);

/* Where a saved tensor of a training-mode forward / backward lives inside `workspace` (for inspection and tests; the
 * product path never needs it).  tensor: 0 X (encoded point, internal column order), 1..8 H0..H7 (trunk activations, 256
 * columns in the reference's neuron order), 10 G (128), 11 DIRX (32), 12..19 dZ0..dZ7 (256), 21 [dS | dG] (160; 22 = its
 * dG part), 23 dP (32).  Layout: FRAGMENT-MAJOR bf16 -- element (row r, column f) of a tensor with `ld` columns is at
 *   byte_offset + ((r / 32) * (ld / 16) + f / 16) * 1024 + (2 * (r % 32) + ((f % 16) / 4) % 2) * 16
 *               + 2 * (4 * ((f % 16) / 8) + f % 4)
 * (per 32-row tile and 16-column chunk one 1 KiB block = the register image of the wave that produced it); in split-bf16
 * precision the `lo` plane follows at + plane_bytes.  Rows = n_rays * n_samples in (ray, sample) order.
 * Precisions 1 and 3 do not materialise H0 (tensor 1) and dZ7 (tensor 19): the weight-gradient jobs that need them recompute
 * them per 32-row chunk, H0 from X (ABI 7) and dZ7 from [dS | dG] (ABI 8: no longer allocated either); the call returns
 * NERFPP_ERR_UNSUPPORTED for them.  (A precision-2 forward with training == 2 leaves its H0 planes unwritten, and a precision-1
 * backward over a precision-2 workspace its dZ7 planes, for the same reason: every bf16 backward recomputes both -- their
 * contents are then undefined.) */
int nerfpp_workspace_tensor(int n_rays, int n_samples, int precision, int net, int tensor,
                            int64_t* byte_offset, int32_t* ld, int64_t* plane_bytes);

typedef struct {
  int32_t n_rays, n_samples;       /* S = 64 (level 0) or 192 (level 1) in the reference */
  int32_t precision, training;     /* training != 0 keeps what nerfpp_level_backward needs; 2 (precision 2 only): the
                                      backward will run at precision 1 (workspace_precision 2 there), so only the hi planes
                                      of the saved tensors are written -- half the store traffic of the forward */
  const float* ray_o;              /* [n,3] */
  const float* ray_d;              /* [n,3] un-normalised */
  const float* fg_far;             /* [n]   fg_z_max */
  const float* fg_z;               /* [n,S] */
  const float* bg_z;               /* [n,S] ascending inverse distance */
  const void* packed;              /* nerfpp_pack_level output */
  void* workspace;                 /* nerfpp_workspace_bytes */
  /* outputs: the reference's `ret` dict (ddp_model.py:136-146) */
  float* rgb;                      /* [n,3] */
  float* depth;                    /* [n]   */
  float* fg_weights;               /* [n,S] */
  float* bg_weights;               /* [n,S] (flipped order, as the reference returns it) */
  float* fg_dists;                 /* [n,S] */
  float* fg_rgb;                   /* [n,3] */
  float* fg_depth;                 /* [n]   */
  float* bg_rgb;                   /* [n,3] */
  float* bg_depth;                 /* [n]   */
  float* bg_lambda;                /* [n]   */
  /* optional profiling taps: hipEvent_t handles (or NULL) recorded on `stream` immediately before the
   * first and after the last MLP kernel of this call (foreground + background net), so a caller can
   * time the level's MLP forward live */
  void* ev_mlp_begin;
  void* ev_mlp_end;
} nerfpp_forward_args;

/* ret = net(ray_o, ray_d, fg_z_max, fg_z_vals, bg_z_vals)          ddp_model.py:74-147 */
int nerfpp_level_forward(void* stream, const nerfpp_forward_args* args);

/* loss head + its gradient                                          ddp_train_nerf.py:481-493
 * scalars[4] = {loss, rgb_loss, depth_loss, #valid rays}; g_* are dL/d(rgb, depth, fg_weights).
 * depth_sup may be NULL for NERFPP_LOSS_RGB_ONLY; g_fg_weights may be NULL unless KL. */
int nerfpp_loss(void* stream, int n_rays, int n_samples, int loss_type, float lambda_depth,
                float kl_sigma, const float* rgb, const float* rgb_gt, const float* depth,
                const float* depth_sup, const float* fg_weights, const float* fg_z,
                const float* fg_dists, const float* fg_far, float* scalars, float* g_rgb,
                float* g_depth, float* g_fg_weights);

typedef struct {
  int32_t n_rays, n_samples;
  int32_t precision;               /* of the backward kernels and of `packed` (1 or 2) */
  int32_t workspace_precision;     /* precision of the forward that filled `workspace`; 0 = same as
                                      `precision`.  2 or 3 with precision 1: split-bf16 (1e-4 outputs and loss) / fp16x2w
                                      forward, single-pass bf16 backward over the (hi) planes it saved */
  const float* ray_d;
  const float* fg_far;
  const float* fg_z;
  const float* bg_z;
  const void* packed;
  void* workspace;                 /* the SAME workspace the training-mode forward filled */
  const int32_t* tables;           /* device copy of nerfpp_build_level_tables */
  const float* g_rgb;              /* [n,3] dL/d rgb */
  const float* g_depth;            /* [n]   dL/d depth */
  const float* g_fg_weights;       /* [n,S] dL/d fg_weights or NULL */
  float grad_scale;                /* multiplies every gradient (1/world_size pre-scaling) */
  float* grads;                    /* [NERFPP_LEVEL_PARAMS] dL/d params, parameters() order */
  /* optional profiling taps (hipEvent_t or NULL): around the dX-chain kernels (both nets) and around
   * the weight-gradient GEMM kernels (both launches) */
  void* ev_bwd_begin;
  void* ev_bwd_end;
  void* ev_dw_begin;
  void* ev_dw_end;
  const float* params;             /* [NERFPP_LEVEL_PARAMS] the level's float32 parameters (the remap /
                                      colour-head weight gradients are derived through them) */
  int32_t defer_reduce;            /* != 0: stop after the weight-gradient GEMMs (their split-K slabs stay in
                                      `workspace`); the caller finishes with nerfpp_level_reduce_grads, e.g. on
                                      another stream so that it runs under the next level's forward */
  /* Fused loss head (ABI 4).  fused_loss != 0: dL/d rgb, dL/d depth and (KL) dL/d fg_weights are formed INSIDE the
   * compositing backward from the fields below -- the arithmetic of nerfpp_loss (ddp_train_nerf.py:481-493), so the
   * loss launch leaves the critical path between forward and backward; g_rgb / g_depth / g_fg_weights are ignored and
   * may be NULL.  nerfpp_loss stays the source of the logged scalars (any stream, any time after the forward). */
  int32_t fused_loss;
  int32_t loss_type;               /* NERFPP_LOSS_* */
  float lambda_depth;
  float kl_sigma;
  const float* rgb;                /* [n,3] forward output */
  const float* depth;              /* [n]   forward output */
  const float* rgb_gt;             /* [n,3] */
  const float* depth_sup;          /* [n]; NULL for NERFPP_LOSS_RGB_ONLY */
  /* Bad-camera count for the data-parallel step (ABI 8).  NULL, or a device int32 (the counter nerfpp_sample_coarse*
   * accumulates in `bad`): `grads` then has NERFPP_LEVEL_PARAMS + 1 elements and the slab-sum launch of this call (or of
   * nerfpp_level_reduce_grads) writes (float)*bad_count -- unscaled -- into grads[NERFPP_LEVEL_PARAMS], where it rides the
   * gradient all-reduce and serves as nerfpp_adam_step's skip_if_nonzero on every rank (ddp_train_nerf.py:62-63 raises on
   * the spot; a caller that reads the count later must keep poisoned gradients out of the parameters on ALL ranks). */
  const int32_t* bad_count;
} nerfpp_backward_args;

/* loss.backward() for one level (autograd in the reference)         ddp_train_nerf.py:497 */
int nerfpp_level_backward(void* stream, const nerfpp_backward_args* args);

/* Second half of nerfpp_level_backward when it was called with defer_reduce: sums the split-K slabs in a fixed order
 * into `grads` (x grad_scale) and derives the remap / colour-head gradients.  Same args struct; the caller orders it
 * after the backward call (stream order or an event) and before anything that overwrites `workspace`. */
int nerfpp_level_reduce_grads(void* stream, const nerfpp_backward_args* args);

/* torch.optim.Adam single step (ddp_train_nerf.py:324,498); step is the 1-based step count.
 * skip_if_nonzero: NULL, or a device float; when it is != 0 at execution time the call changes nothing.  The
 * reference raises BEFORE the optimiser step when a camera lies outside the unit sphere (:62-63); a caller that
 * reads the bad-camera count later (asynchronously) passes it here -- summed over the ranks with the gradient
 * all-reduce -- so that a poisoned gradient never reaches the parameters or the Adam moments. */
int nerfpp_adam_step(void* stream, float* params, const float* grads, float* exp_avg,
                     float* exp_avg_sq, int64_t n, int step, double lr, double beta1, double beta2,
                     double eps, const float* skip_if_nonzero);

/* ---------------------------------------------------------------- data-parallel gradient averaging (RCCL over xGMI)
 * DistributedDataParallel's gradient average of ddp_train_nerf.py:323 (process group :298) behind the C ABI, for hosts that do
 * not go through torch.distributed.  RCCL is bound with dlopen("librccl.so.1") at the first call (NERFPP_ERR_COMM when it
 * cannot be loaded; there is no host-side fallback).  One process per GPU: rank 0 creates the 128-byte id, every rank gets it
 * over a channel of the host's choosing and builds its communicator; `comm` is an opaque ncclComm_t.
 * nerfpp_allreduce_mean: grads[count] (device float32, in place) <- mean over the ranks, on `stream`.  prescaled != 0: the
 * gradients already carry 1 / world_size (nerfpp_backward_args.grad_scale), only the SUM all-reduce is issued. */
const char* nerfpp_comm_last_error(void);
int nerfpp_rccl_unique_id(char out_id[128]);
int nerfpp_rccl_comm_init(void** comm, int world_size, const char id[128], int rank);
int nerfpp_rccl_comm_destroy(void* comm);
int nerfpp_allreduce_mean(void* stream, void* rccl_comm, float* grads, int64_t count, int world_size, int prescaled);

#ifdef __cplusplus
}
#endif
#endif /* NERFPP_HIP_H */
