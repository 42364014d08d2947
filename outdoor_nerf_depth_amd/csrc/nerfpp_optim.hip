// Parameter-side kernels: weight packing into MFMA fragment streams, gradient un-packing
// (deterministic sum over the split-K slabs + internal -> reference order), Adam.
// Adam follows torch.optim.Adam's single-tensor update exactly (lr 5e-4, betas (0.9, 0.999),
// eps 1e-8 in the reference: ddp_train_nerf.py:324).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nerfpp_common.h"
#include "nerfpp_kernels.h"

namespace nerfpp {

template <int P>
__global__ void pack_kernel(const float* __restrict__ params, const int32_t* __restrict__ tbl, int64_t n,
                            __bf16* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int32_t src = tbl[e];
  const float v = src >= 0 ? params[src] : 0.f;
  const int64_t F = e >> 9, within = e & 511;
  const __bf16 h = (__bf16)v;
  out[(F * P) * 512 + within] = h;
  if (P == 2) out[(F * P + 1) * 512 + within] = (__bf16)(v - (float)h);
}

__global__ void gather_f32_kernel(const float* __restrict__ params, const int32_t* __restrict__ tbl, int64_t n,
                                  float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int32_t src = tbl[e];
  out[e] = src >= 0 ? params[src] : 0.f;
}

__global__ void unpack_grads_kernel(const float* __restrict__ slabs, int ksplit, int64_t slab_floats,
                                    const int32_t* __restrict__ tbl, int64_t n, float scale,
                                    float* __restrict__ grads) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t src = tbl[i];
  float acc = 0.f;
  for (int s = 0; s < ksplit; ++s) acc += slabs[(size_t)s * slab_floats + src];
  grads[i] = acc * scale;
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float one_minus_b1, float b2, float one_minus_b2,
                            float step_size, float sqrt_bias2, float eps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  float mi = m[i], vi = v[i];
  mi = mi + (gi - mi) * one_minus_b1;                   // exp_avg.lerp_(grad, 1 - beta1)
  vi = vi * b2 + one_minus_b2 * gi * gi;                // exp_avg_sq.mul_(beta2).addcmul_(...)
  const float denom = sqrtf(vi) / sqrt_bias2 + eps;      // (sqrt(v) / sqrt(bias2)).add_(eps)
  p[i] = p[i] - step_size * (mi / denom);
  m[i] = mi;
  v[i] = vi;
}

// The remap layer has no activation, so with M = dG^T * H7 (what the weight-gradient GEMM leaves in
// the first 256 columns of rgb_layers.0.weight.grad) and db_g = rgb_layers.0.bias.grad:
//   d rgb_layers.0.weight[:, :256] = M * Wr^T + db_g (x) b_r        (R = H7 Wr^T + b_r)
//   d base_remap_layers.0.weight   = Wg[:, :256]^T * M              (dR = dG Wg[:, :256])
//   d base_remap_layers.0.bias     = Wg[:, :256]^T * db_g
// which removes both 256-wide saves (R, dR) and one 256x256 GEMM over all samples.  float32, fixed
// summation order.
__global__ void remap_copy_m_kernel(const float* __restrict__ grads, int w_g, int ldg, float* __restrict__ m) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 128 * 256) m[i] = grads[w_g + (i >> 8) * ldg + (i & 255)];
}
// One wave per output row-segment so that every global read is coalesced:
//   part A (128 x 256 outputs): out[o][j] = sum_i M[o][i] * Wr[j][i] + db_g[o] * b_r[j]
//           wave = (o, 4 consecutive j): lanes stride over i, 4 dot products, wave reduction
//   part B (256 x 256): dWr[j][i] = sum_o Wg[o][j] * M[o][i]      thread = (j, i), lanes over i
//   part C (256):       db_r[j]   = sum_o Wg[o][j] * db_g[o]
__global__ void remap_fixup_kernel(float* __restrict__ grads, const float* __restrict__ params,
                                   const float* __restrict__ m, int w_g, int b_g, int ldg, int w_r, int b_r) {
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

void launch_remap_fixup(hipStream_t st, int net, float* grads, const float* params, float* tmp_m) {
  const int w_g = ref_w_off(net, RT_RGB0), b_g = ref_b_off(net, RT_RGB0), ldg = ref_in(net, RT_RGB0);
  const int w_r = ref_w_off(net, RT_REMAP), b_r = ref_b_off(net, RT_REMAP);
  hipLaunchKernelGGL(remap_copy_m_kernel, dim3(128), dim3(256), 0, st, grads, w_g, ldg, tmp_m);
  const int total = 128 * 64 * 64 + 256 * 256 + 256;          // part A waves * 64 + part B + part C threads
  hipLaunchKernelGGL(remap_fixup_kernel, dim3((total + 255) / 256), dim3(256), 0, st, grads, params, tmp_m, w_g, b_g,
                     ldg, w_r, b_r);
}

void launch_pack(hipStream_t st, const float* params, const int32_t* tbl, int64_t n, int P, void* out) {
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (P == 1) hipLaunchKernelGGL(pack_kernel<1>, grid, block, 0, st, params, tbl, n, (__bf16*)out);
  else hipLaunchKernelGGL(pack_kernel<2>, grid, block, 0, st, params, tbl, n, (__bf16*)out);
}
void launch_gather_f32(hipStream_t st, const float* params, const int32_t* tbl, int64_t n, float* out) {
  hipLaunchKernelGGL(gather_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, params, tbl, n, out);
}
void launch_unpack_grads(hipStream_t st, const float* slabs, int ksplit, int64_t slab_floats, const int32_t* tbl,
                         int64_t n_params, float scale, float* grads) {
  hipLaunchKernelGGL(unpack_grads_kernel, dim3((unsigned)((n_params + 255) / 256)), dim3(256), 0, st, slabs,
                     ksplit, slab_floats, tbl, n_params, scale, grads);
}
void launch_adam(hipStream_t st, float* p, const float* g, float* m, float* v, int64_t n, int step, double lr,
                 double beta1, double beta2, double eps) {
  const double bias1 = 1.0 - pow(beta1, step), bias2 = 1.0 - pow(beta2, step);
  const float step_size = (float)(lr / bias1);
  const float sqrt_bias2 = (float)sqrt(bias2);
  // 1 - beta is formed in double like Python does, then rounded once (1.f - 0.9f != (float)0.1)
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v, n,
                     (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size, sqrt_bias2, (float)eps);
}
