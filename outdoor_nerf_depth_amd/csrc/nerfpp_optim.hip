// Parameter-side kernels: weight packing into MFMA fragment streams, gradient un-packing
// (deterministic sum over the split-K slabs + internal -> reference order), Adam.
// Adam follows torch.optim.Adam's single-tensor update exactly (lr 5e-4, betas (0.9, 0.999),
// eps 1e-8 in the reference: ddp_train_nerf.py:324).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nerfpp_common.h"
#include "nerfpp_kernels.h"

namespace nerfpp {

// One launch packs a whole level: both nets' forward and backward fragment streams (bf16 hi[/lo]
// planes) and the forward bias vectors (f32), each a gather through its index table.
struct PackSegs {
  static constexpr int N = 3 * N_NET;
  int blk_end[N];                 // exclusive prefix of 256-thread blocks per segment
  const int32_t* tbl[N];
  void* out[N];
  int64_t n[N];
  int pbase[N];                   // offset of the net's parameters in the level's flat buffer
  int np[N];                      // parameters of the net: table entries >= np address its derived parameters
  const float* derived[N];        // [Wc | bc] of the net (fold_remap_kernel)
  int is_f32[N];
};
// Wc = Wrgb0[:, :256] * Wremap, bc = brgb0 + Wrgb0[:, :256] * bremap (nerfpp_common.h, forward stages): float32, fixed order.
// thread = (net, o, f): Wrgb0[o][j] is a broadcast, Wremap[j][f] coalesced over f.
__global__ void fold_remap_kernel(const float* __restrict__ params_lvl, float* __restrict__ d0, float* __restrict__ d1) {
  const int net = blockIdx.y;
  const float* params = params_lvl + (net ? FG_PARAMS : 0);
  float* out = net ? d1 : d0;
  const int w_g = ref_w_off(net, RT_RGB0), b_g = ref_b_off(net, RT_RGB0), ldg = ref_in(net, RT_RGB0);
  const int w_r = ref_w_off(net, RT_REMAP), b_r = ref_b_off(net, RT_REMAP);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < DERIVED_WC) {
    const int o = t >> 8, f = t & 255;
    float acc = 0.f;
    for (int j = 0; j < 256; ++j) acc += params[w_g + o * ldg + j] * params[w_r + j * 256 + f];
    out[t] = acc;
  } else if (t < DERIVED_FLOATS) {
    const int o = t - DERIVED_WC;
    float acc = params[b_g + o];
    for (int j = 0; j < 256; ++j) acc += params[w_g + o * ldg + j] * params[b_r + j];
    out[t] = acc;
  }
}
template <int P>
__global__ void pack_level_kernel(const float* __restrict__ params, PackSegs sg) {
  int seg = 0;
#pragma unroll
  for (int k = 0; k < PackSegs::N - 1; ++k) seg += (int)blockIdx.x >= sg.blk_end[k];
  const int blk0 = seg == 0 ? 0 : sg.blk_end[seg - 1];
  const int64_t e = (int64_t)((int)blockIdx.x - blk0) * blockDim.x + threadIdx.x;
  if (e >= sg.n[seg]) return;
  const int32_t src = __builtin_nontemporal_load(sg.tbl[seg] + e);       // (the gather tables: 10 MB per level, read once per pack)
  const float v = src < 0 ? 0.f : src < sg.np[seg] ? params[sg.pbase[seg] + src] : sg.derived[seg][src - sg.np[seg]];
  if (sg.is_f32[seg]) {
    ((float*)sg.out[seg])[e] = v;
    return;
  }
  const int64_t F = e >> 9, within = e & 511;
  if constexpr (P == 3) {                      // fp16 hi + lo (nerfpp_common.h: precision ids)
    _Float16* out = (_Float16*)sg.out[seg];
    const _Float16 h = (_Float16)v;
    out[(F * 2) * 512 + within] = h;
    out[(F * 2 + 1) * 512 + within] = (_Float16)(v - (float)h);
  } else {
    __bf16* out = (__bf16*)sg.out[seg];
    const __bf16 h = (__bf16)v;
    out[(F * P) * 512 + within] = h;
    if (P == 2) out[(F * P + 1) * 512 + within] = (__bf16)(v - (float)h);
  }
}

// Both nets in one launch: deterministic sum over the split-K slabs, internal -> reference order,
// DDP pre-scale.  The first 256 columns of rgb_layers.0.weight.grad hold M = dG^T H7 at this point
// (see remap_fixup_kernel); they are also copied to m_out[net] so the fix-up can overwrite them.
__constant__ SlabMap c_slab_map = make_slab_map();
struct UnpackArgs {
  const float* slabs[N_NET];
  int64_t slab_floats[N_NET];
  const int32_t* tbl[N_NET];
  float* m_out[N_NET];
  DwPlan plan;                    // slabs filled per job
};
__global__ void unpack_grads_kernel(UnpackArgs a, float scale, float* __restrict__ grads, const int32_t* __restrict__ bad_count) {
  const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= LEVEL_PARAMS) return;
  // the bad-camera count of the step goes with the gradients (nerfpp_backward_args::bad_count): one element behind them
  if (gi == 0 && bad_count != nullptr) grads[LEVEL_PARAMS] = (float)*bad_count;
  const int net = gi >= FG_PARAMS;
  const int i = (int)(gi - (net ? FG_PARAMS : 0));
  const int32_t src = __builtin_nontemporal_load(a.tbl[net] + i);
  const float* sl = a.slabs[net];
  const int64_t sf = a.slab_floats[net];
  const int job = slab_job_index(c_slab_map, net, src);
  const int ksplit = job >= 0 ? a.plan.k[net][job] : 0;        // remap stage: filled in by remap_fixup_kernel
  float acc = 0.f;                                               // fixed summation order: deterministic
  int s = 0;
  // (non-temporal loads: the ~117 MB of slabs of a level are read once; streamed through the L2 with the default policy they
  // evict the packed weight streams the MLP kernels running next to this launch re-read for every tile)
  for (; s + 4 <= ksplit; s += 4) {                              // 4 independent loads in flight
    const float v0 = __builtin_nontemporal_load(sl + (size_t)s * sf + src), v1 = __builtin_nontemporal_load(sl + (size_t)(s + 1) * sf + src);
    const float v2 = __builtin_nontemporal_load(sl + (size_t)(s + 2) * sf + src), v3 = __builtin_nontemporal_load(sl + (size_t)(s + 3) * sf + src);
    acc += (v0 + v1) + (v2 + v3);
  }
  for (; s < ksplit; ++s) acc += __builtin_nontemporal_load(sl + (size_t)s * sf + src);
  acc *= scale;
  grads[gi] = acc;
  const int w_g = ref_w_off(net, RT_RGB0), ldg = ref_in(net, RT_RGB0);
  const int r = i - w_g;
  if (r >= 0 && r < 128 * ldg) {
    const int o = r / ldg, c = r - o * ldg;
    if (c < 256) a.m_out[net][o * 256 + c] = acc;
  }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float one_minus_b1, float b2, float one_minus_b2,
                            float step_size, float sqrt_bias2, float eps, const float* __restrict__ skip) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // device-side predicate (nerfpp_adam_step: skip_if_nonzero): the reference raises BEFORE the step when a camera is
  // outside the unit sphere (ddp_train_nerf.py:62-63); here that count is read later, so the update must not happen
  if (skip != nullptr && *skip != 0.f) return;
  // (streamed once per step, on a side stream next to the MLP kernels: non-temporal, so that 24 MB per level do not pass
  // through the L2 lines the packed weight streams live in.  The parameters themselves are re-read by the fold / re-pack.)
  const float gi = __builtin_nontemporal_load(g + i);
  float mi = __builtin_nontemporal_load(m + i), vi = __builtin_nontemporal_load(v + i);
  mi = mi + (gi - mi) * one_minus_b1;                   // exp_avg.lerp_(grad, 1 - beta1)
  vi = vi * b2 + one_minus_b2 * gi * gi;                // exp_avg_sq.mul_(beta2).addcmul_(...)
  const float denom = sqrtf(vi) / sqrt_bias2 + eps;      // (sqrt(v) / sqrt(bias2)).add_(eps)
  p[i] = p[i] - step_size * (mi / denom);
  __builtin_nontemporal_store(mi, m + i);
  __builtin_nontemporal_store(vi, v + i);
}

// The remap layer has no activation, so with M = dG^T * H7 (what the weight-gradient GEMM leaves in
// the first 256 columns of rgb_layers.0.weight.grad) and db_g = rgb_layers.0.bias.grad:
//   d rgb_layers.0.weight[:, :256] = M * Wr^T + db_g (x) b_r        (R = H7 Wr^T + b_r)
//   d base_remap_layers.0.weight   = Wg[:, :256]^T * M              (dR = dG Wg[:, :256])
//   d base_remap_layers.0.bias     = Wg[:, :256]^T * db_g
// which removes both 256-wide saves (R, dR) and one 256x256 GEMM over all samples.  float32, fixed
// summation order.
// One wave per output row-segment so that every global read is coalesced:
//   part A (128 x 256 outputs): out[o][j] = sum_i M[o][i] * Wr[j][i] + db_g[o] * b_r[j]
//           wave = (o, 4 consecutive j): lanes stride over i, 4 dot products, wave reduction
//   part B (256 x 256): dWr[j][i] = sum_o Wg[o][j] * M[o][i]      thread = (j, i), lanes over i
//   part C (256):       db_r[j]   = sum_o Wg[o][j] * db_g[o]
// blockIdx.y = net; grads / params are the level's flat buffers; m0 / m1 = the nets' M copies.
__global__ void remap_fixup_kernel(float* __restrict__ grads_lvl, const float* __restrict__ params_lvl,
                                   const float* __restrict__ m0, const float* __restrict__ m1) {
  const int net = blockIdx.y;
  float* grads = grads_lvl + (net ? FG_PARAMS : 0);
  const float* params = params_lvl + (net ? FG_PARAMS : 0);
  const float* m = net ? m1 : m0;
  const int w_g = ref_w_off(net, RT_RGB0), b_g = ref_b_off(net, RT_RGB0), ldg = ref_in(net, RT_RGB0);
  const int w_r = ref_w_off(net, RT_REMAP), b_r = ref_b_off(net, RT_REMAP);
  constexpr int A_WAVES = 128 * 64;                       // (o, j-quad)
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (gw < A_WAVES) {
    const int o = gw >> 6, j0 = (gw & 63) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = lane; i < 256; i += 64) {
      const float mv = m[o * 256 + i];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += mv * params[w_r + (j0 + q) * 256 + i];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) acc[q] += __shfl_xor(acc[q], d, 64);
    }
    const float sel = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
    if (lane < 4) grads[w_g + o * ldg + j0 + lane] = sel + grads[b_g + o] * params[b_r + j0 + lane];
    return;
  }
  const int t = (gw - A_WAVES) * 64 + lane;
  if (t < 256 * 256) {                                    // d Wr[j][i]
    const int j = t >> 8, i = t & 255;
    float acc = 0.f;
    for (int o = 0; o < 128; ++o) acc += params[w_g + o * ldg + j] * m[o * 256 + i];
    grads[w_r + j * 256 + i] = acc;
  } else if (t < 256 * 256 + 256) {                       // d b_r[j]
    const int j = t - 256 * 256;
    float acc = 0.f;
    for (int o = 0; o < 128; ++o) acc += params[w_g + o * ldg + j] * grads[b_g + o];
    grads[b_r + j] = acc;
  }
}

}  // namespace nerfpp

using namespace nerfpp;

void launch_remap_fixup(hipStream_t st, float* grads_lvl, const float* params_lvl, const float* m0, const float* m1) {
  const int total = 128 * 64 * 64 + 256 * 256 + 256;          // part A waves * 64 + part B + part C threads
  hipLaunchKernelGGL(remap_fixup_kernel, dim3((total + 255) / 256, N_NET), dim3(256), 0, st, grads_lvl, params_lvl, m0, m1);
}

// segs: per net {fwd stream, bwd stream, bias}; tables / outs / sizes in that order
void launch_pack_level(hipStream_t st, const float* params, int P, const int32_t* const* tbl, void* const* out,
                       const int64_t* n, float* const* derived) {
  hipLaunchKernelGGL(fold_remap_kernel, dim3((DERIVED_FLOATS + 255) / 256, N_NET), dim3(256), 0, st, params, derived[0], derived[1]);
  PackSegs sg{};
  int blk = 0;
  for (int k = 0; k < PackSegs::N; ++k) {
    blk += (int)((n[k] + 255) / 256);
    sg.blk_end[k] = blk;
    sg.tbl[k] = tbl[k];
    sg.out[k] = out[k];
    sg.n[k] = n[k];
    sg.pbase[k] = (k / 3) == 0 ? 0 : FG_PARAMS;
    sg.np[k] = net_params(k / 3);
    sg.derived[k] = derived[k / 3];
    sg.is_f32[k] = (k % 3) == 2;
  }
  if (P == 1) hipLaunchKernelGGL(pack_level_kernel<1>, dim3(blk), dim3(256), 0, st, params, sg);
  else if (P == 2) hipLaunchKernelGGL(pack_level_kernel<2>, dim3(blk), dim3(256), 0, st, params, sg);
  else hipLaunchKernelGGL(pack_level_kernel<3>, dim3(blk), dim3(256), 0, st, params, sg);
}
void launch_unpack_grads(hipStream_t st, const float* const* slabs, const int64_t* slab_floats, const DwPlan& plan,
                         const int32_t* const* tbl, float* const* m_out, float scale, float* grads_lvl, const int32_t* bad_count) {
  UnpackArgs a{};
  a.plan = plan;
  for (int net = 0; net < N_NET; ++net) {
    a.slabs[net] = slabs[net]; a.slab_floats[net] = slab_floats[net]; a.tbl[net] = tbl[net]; a.m_out[net] = m_out[net];
  }
  hipLaunchKernelGGL(unpack_grads_kernel, dim3((LEVEL_PARAMS + 255) / 256), dim3(256), 0, st, a, scale, grads_lvl, bad_count);
}
void launch_adam(hipStream_t st, float* p, const float* g, float* m, float* v, int64_t n, int step, double lr,
                 double beta1, double beta2, double eps, const float* skip) {
  const double bias1 = 1.0 - pow(beta1, step), bias2 = 1.0 - pow(beta2, step);
  const float step_size = (float)(lr / bias1);
  const float sqrt_bias2 = (float)sqrt(bias2);
  // 1 - beta is formed in double like Python does, then rounded once (1.f - 0.9f != (float)0.1)
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v, n,
                     (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size, sqrt_bias2, (float)eps, skip);
}
