// Sampling, alpha-compositing and loss kernels of the NeRF++ path (all HBM-bound, float32).
// Compiled with -ffp-contract=off: the sample-bin arithmetic must round exactly like the
// oracle / the reference (no implicit FMA contraction); the one fused multiply-add the
// reference does use (torch.linspace) is written as an explicit fmaf.
//
// Reference lines (nerf-methods/nerfplusplus/):
//   intersect_sphere ddp_train_nerf.py:51-66 | coarse depths :438-449 | perturb_samples :69-78
//   sample_pdf :81-130 | merge :457,465 | compositing ddp_model.py:96-134
//   losses depth_loss.py:4-44, utils.py:12-16, loss head ddp_train_nerf.py:481-493
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "nerfpp_common.h"
#include "nerfpp_kernels.h"

namespace nerfpp {

constexpr float TINY = 1e-6f;
constexpr float HUGE_NUM = 1e10f;

__device__ __forceinline__ float sum3(float a, float b, float c) { return (a + b) + c; }

__device__ __forceinline__ float sphere_far(const float* __restrict__ ray_o,
                                            const float* __restrict__ ray_d, int ray, int* bad) {
  const float ox = ray_o[ray * 3], oy = ray_o[ray * 3 + 1], oz = ray_o[ray * 3 + 2];
  const float dx = ray_d[ray * 3], dy = ray_d[ray * 3 + 1], dz = ray_d[ray * 3 + 2];
  const float dd = sum3(dx * dx, dy * dy, dz * dz);
  const float d1 = -sum3(dx * ox, dy * oy, dz * oz) / dd;
  const float px = ox + d1 * dx, py = oy + d1 * dy, pz = oz + d1 * dz;
  const float ray_d_cos = 1.f / sqrtf(dd);
  const float pn = sum3(px * px, py * py, pz * pz);
  if (pn >= 1.f && bad) atomicAdd(bad, 1);
  const float d2 = sqrtf(1.f - pn) * ray_d_cos;
  return d1 + d2;
}

__global__ void intersect_sphere_kernel(int n, const float* __restrict__ ray_o,
                                        const float* __restrict__ ray_d, float* __restrict__ far,
                                        int* bad) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray < n) far[ray] = sphere_far(ray_o, ray_d, ray, bad);
}

__device__ __forceinline__ float torch_linspace01(int i, int steps) {
  const float step = 1.f / (float)(steps - 1);
  return i < steps / 2 ? fmaf(step, (float)i, 0.f) : fmaf(-step, (float)(steps - 1 - i), 1.f);
}

__device__ __forceinline__ float perturb(float zm, float z0, float zp, bool first, bool last, float t) {
  const float upper = last ? z0 : .5f * (zp + z0);
  const float lower = first ? z0 : .5f * (z0 + zm);
  return lower + (upper - lower) * t;
}

// one thread per (ray, sample)
__global__ void sample_coarse_kernel(int n, int S, const float* __restrict__ ray_o,
                                     const float* __restrict__ ray_d,
                                     const float* __restrict__ min_depth,
                                     const float* __restrict__ t_fg, const float* __restrict__ t_bg,
                                     float* __restrict__ fg_far, float* __restrict__ fg_z,
                                     float* __restrict__ bg_z, int* bad, RngKey rng) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * S) return;
  const int ray = idx / S, i = idx - ray * S;
  const float far = sphere_far(ray_o, ray_d, ray, i == 0 ? bad : nullptr);
  if (i == 0) fg_far[ray] = far;
  const float near = min_depth[ray];
  const float step = (far - near) / (float)(S - 1);
  const float z0 = near + (float)i * step;
  float out = z0;
  if (t_fg || rng.enabled) {
    const float zm = near + (float)(i - 1) * step, zp = near + (float)(i + 1) * step;
    out = perturb(zm, z0, zp, i == 0, i == S - 1, rng.enabled ? philox_uniform(rng, 0, (uint64_t)idx) : t_fg[idx]);
  }
  fg_z[idx] = out;
  const float b0 = torch_linspace01(i, S);
  out = b0;
  if (t_bg || rng.enabled) {
    const float bm = i > 0 ? torch_linspace01(i - 1, S) : 0.f;
    const float bp = i < S - 1 ? torch_linspace01(i + 1, S) : 0.f;
    out = perturb(bm, b0, bp, i == 0, i == S - 1, rng.enabled ? philox_uniform(rng, 1, (uint64_t)idx) : t_bg[idx]);
  }
  bg_z[idx] = out;
}

__global__ void rng_uniform_kernel(RngKey rng, uint32_t stream_id, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = philox_uniform(rng, stream_id, (uint64_t)i);
}

__global__ void perturb_kernel(int n, int S, const float* __restrict__ z, const float* __restrict__ t,
                               float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * S) return;
  const int i = idx % S;
  const float z0 = z[idx];
  const float zm = i > 0 ? z[idx - 1] : 0.f, zp = i < S - 1 ? z[idx + 1] : 0.f;
  out[idx] = perturb(zm, z0, zp, i == 0, i == S - 1, t[idx]);
}

// ------------------------------------------------------------------------------------------------
// inverse-CDF sampling (+ optional merge with the old depths).  One wave per ray, 4 rays/block.
//   FUSED: bins are the mid-points of z_old [n, S_old], weights = w_full[n, S_old][1:-1]
//   else : bins [n, M+1], weights [n, M] given explicitly (the reference's sample_pdf signature)
// Arithmetic order = oracle: wsum and cdf by sequential float64 accumulation rounded to float32.
// ------------------------------------------------------------------------------------------------
constexpr int SP_MAX = 512;     // max S_old + S_new and max M+1
// LDS hand-off between the lanes of ONE wave (its LDS operations retire in order; the compiler must not reorder across)
__device__ __forceinline__ void lds_wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// blockIdx.y selects one of up to two independent problems (the foreground and background volumes of
// a cascade level are re-sampled in one launch: with one wave per ray a single volume of 1024 rays
// occupies 4 waves per CU and is latency-bound, so the second one is free).
struct SamplePdfProblem {
  const float* bins_or_zold; const float* weights; const float* u;
  float* samples; int64_t* above; float* merged;
};
struct SamplePdfArgs { SamplePdfProblem p[2]; RngKey rng; };   // rng.enabled: u = philox stream 2 + blockIdx.y
template <bool FUSED>
__global__ __launch_bounds__(256) void sample_pdf_kernel(int n, int M, int S_new, SamplePdfArgs args) {
  const SamplePdfProblem& pr = args.p[blockIdx.y];
  const float* __restrict__ bins_or_zold = pr.bins_or_zold;
  const float* __restrict__ weights = pr.weights;
  const float* __restrict__ u_in = pr.u;
  float* __restrict__ samples_out = pr.samples;
  int64_t* __restrict__ above_out = pr.above;
  float* __restrict__ merged_out = pr.merged;
  __shared__ float s_cdf[4][SP_MAX];
  __shared__ float s_w[4][SP_MAX];
  __shared__ float s_bins[4][SP_MAX];
  __shared__ float s_z[4][SP_MAX];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  const bool active = ray < n;
  const int S_old = FUSED ? M + 2 : 0;
  float* cdf = s_cdf[wave];
  float* wq = s_w[wave];
  float* bins = s_bins[wave];
  float* zs = s_z[wave];
  if (active) {
    if (FUSED) {
      const float* zo = bins_or_zold + (size_t)ray * S_old;
      for (int i = lane; i < S_old; i += 64) zs[i] = zo[i];
      for (int k = lane; k < M; k += 64) wq[k] = weights[(size_t)ray * S_old + k + 1] + TINY;
    } else {
      for (int i = lane; i <= M; i += 64) bins[i] = bins_or_zold[(size_t)ray * (M + 1) + i];
      for (int k = lane; k < M; k += 64) wq[k] = weights[(size_t)ray * M + k] + TINY;
    }
  }
  __syncthreads();
  if (active) {
    if (FUSED)
      for (int i = lane; i <= M; i += 64) bins[i] = .5f * (zs[i + 1] + zs[i]);
    if (lane == 0) {
      double acc = 0.0;
      for (int k = 0; k < M; ++k) acc += (double)wq[k];
      cdf[0] = (float)acc;                          // wsum, handed to the other lanes through LDS
    }
  }
  __syncthreads();
  if (active) {                                     // the M divisions in parallel (they were a dependent chain on lane 0) ...
    const float wsum = cdf[0];
    for (int k = lane; k < M; k += 64) wq[k] = wq[k] / wsum;
  }
  __syncthreads();
  if (active && lane == 0) {                        // ... the running sum stays sequential float64 (= torch's CPU cumsum order)
    double acc = 0.0;
    cdf[0] = 0.f;
    for (int k = 0; k < M; ++k) {
      acc += (double)wq[k];
      cdf[k + 1] = (float)acc;
    }
  }
  __syncthreads();
  if (active) {
    for (int j = lane; j < S_new; j += 64) {
      const float u = args.rng.enabled ? philox_uniform(args.rng, 2u + blockIdx.y, (uint64_t)ray * S_new + j)
                      : u_in ? u_in[(size_t)ray * S_new + j] : torch_linspace01(j, S_new);
      // above = #{k < M : u >= cdf[k]}; the cdf is non-decreasing (a running sum of non-negative terms, rounded monotonically),
      // so the count is the position of the first entry above u: binary search instead of M comparisons
      int above = 0, hi_ = M;
      while (above < hi_) {
        const int mid = (above + hi_) >> 1;
        const bool ge = u >= cdf[mid];
        above = ge ? mid + 1 : above;
        hi_ = ge ? hi_ : mid;
      }
      const int below = above - 1 > 0 ? above - 1 : 0;
      const float cdf_lo = cdf[below], cdf_hi = cdf[above];
      const float bin_lo = bins[below], bin_hi = bins[above];
      float denom = cdf_hi - cdf_lo;
      denom = denom < TINY ? 1.f : denom;
      const float t = (u - cdf_lo) / denom;
      const float s = bin_lo + t * (bin_hi - bin_lo + TINY);
      if (samples_out) samples_out[(size_t)ray * S_new + j] = s;
      if (above_out) above_out[(size_t)ray * S_new + j] = above;
      if (FUSED) zs[S_old + j] = s;
    }
  }
  if (!FUSED) return;
  __syncthreads();
  if (active && merged_out) {
    const int S_tot = S_old + S_new;
    // sort(cat(z_old, samples)) (ddp_train_nerf.py:457,465) as a MERGE: z_old is ascending in every caller of the
    // reference (stratified depths, or the previous level's sorted depths), so only the S_new new depths need sorting --
    // a bitonic network in LDS (28 compare-exchange passes for 128 values instead of ranking all 192 against all 192) --
    // and every element's place in the output is its own index plus a binary-search count in the other list
    // (old before new on ties, like the stable sort of the concatenation; the output is values only, so any correct
    // order of equal values is bit-identical).  A z_old that is not ascending falls through to the rank sort below.
    bool sorted_old = true;
    for (int i = lane; i + 1 < S_old; i += 64) sorted_old = sorted_old && zs[i] <= zs[i + 1];
    if (__all(sorted_old) && S_new <= SP_MAX) {
      int P2 = 1;
      while (P2 < S_new) P2 <<= 1;
      float* nb = wq;                                   // the pdf is no longer needed: its array takes the new depths
      for (int i = lane; i < P2; i += 64) nb[i] = i < S_new ? zs[S_old + i] : __builtin_inff();
      lds_wave_sync();
      for (int k = 2; k <= P2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int t = lane; t < (P2 >> 1); t += 64) {
            const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi_i = lo | j;
            const bool up = (lo & k) == 0;
            const float x = nb[lo], y = nb[hi_i];
            if ((x > y) == up) { nb[lo] = y; nb[hi_i] = x; }
          }
          lds_wave_sync();
        }
      }
      float* mo = merged_out + (size_t)ray * S_tot;
      for (int i = lane; i < S_old; i += 64) {          // old depth i: + #{new < it}
        const float v = zs[i];
        int lo = 0, hi_i = S_new;
        while (lo < hi_i) {
          const int mid = (lo + hi_i) >> 1;
          const bool lt = nb[mid] < v;
          lo = lt ? mid + 1 : lo;
          hi_i = lt ? hi_i : mid;
        }
        mo[i + lo] = v;
      }
      for (int p = lane; p < S_new; p += 64) {          // p-th smallest new depth: + #{old <= it}
        const float v = nb[p];
        int lo = 0, hi_i = S_old;
        while (lo < hi_i) {
          const int mid = (lo + hi_i) >> 1;
          const bool le = zs[mid] <= v;
          lo = le ? mid + 1 : lo;
          hi_i = le ? hi_i : mid;
        }
        mo[p + lo] = v;
      }
      return;
    }
    if (S_tot <= 192) {
      // up to three elements per lane ranked in ONE pass over the list (a third of the LDS reads of the per-element loop
      // below, three independent compare chains); same comparison, same result
      const int e0 = lane, e1 = lane + 64, e2 = lane + 128;
      const float v0 = e0 < S_tot ? zs[e0] : 0.f, v1 = e1 < S_tot ? zs[e1] : 0.f, v2 = e2 < S_tot ? zs[e2] : 0.f;
      int r0 = 0, r1 = 0, r2 = 0;
#pragma unroll 4
      for (int k = 0; k < S_tot; ++k) {
        const float o = zs[k];
        r0 += (o < v0 || (o == v0 && k < e0)) ? 1 : 0;
        r1 += (o < v1 || (o == v1 && k < e1)) ? 1 : 0;
        r2 += (o < v2 || (o == v2 && k < e2)) ? 1 : 0;
      }
      float* mo = merged_out + (size_t)ray * S_tot;
      if (e0 < S_tot) mo[r0] = v0;
      if (e1 < S_tot) mo[r1] = v1;
      if (e2 < S_tot) mo[r2] = v2;
      return;
    }
    for (int e = lane; e < S_tot; e += 64) {          // rank sort: values only, stable
      const float v = zs[e];
      int rank = 0;
      for (int k = 0; k < S_tot; ++k) {
        const float o = zs[k];
        rank += (o < v || (o == v && k < e)) ? 1 : 0;
      }
      merged_out[(size_t)ray * S_tot + rank] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// alpha compositing.  One wave per ray; lane L owns the CPL consecutive samples [L*CPL, L*CPL+CPL).
// raw_*: [n*S, 4] = (r, g, b, sigma_raw) per sample in natural sample order (bg NOT flipped; the
// flip of ddp_model.py:116-117 is done here by indexing).
// ------------------------------------------------------------------------------------------------
constexpr int CPL_MAX = 4;       // supports S <= 256

__device__ __forceinline__ float wave_excl_prod(float x, int lane, float* total) {
  float v = x;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float t = __shfl_up(v, d, 64);
    if (lane >= d) v *= t;
  }
  *total = __shfl(v, 63, 64);
  const float e = __shfl_up(v, 1, 64);
  return lane == 0 ? 1.f : e;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
// exclusive suffix sum: out(lane) = sum_{l > lane} x(l)
__device__ __forceinline__ float wave_excl_suffix_sum(float x, int lane) {
  float v = x;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float t = __shfl_down(v, d, 64);
    if (lane + d < 64) v += t;
  }
  const float e = __shfl_down(v, 1, 64);
  return lane == 63 ? 0.f : e;
}

struct RaySamples {          // per-lane state of one volume (fg or bg) of one ray
  float dist[CPL_MAX], e[CPL_MAX], alpha[CPL_MAX], q[CPL_MAX], T[CPL_MAX], w[CPL_MAX];
  float c[CPL_MAX][3], zval[CPL_MAX], sraw[CPL_MAX];
  float lambda;              // product of all q (fg: bg_lambda)
};

template <bool BG>
__device__ __forceinline__ void load_volume(RaySamples& v, int S, int cpl, int lane, size_t row0,
                                            const float4* __restrict__ raw, const float* __restrict__ z,
                                            const float* __restrict__ depth_real, float dnorm, float far) {
  float run = 1.f;
#pragma unroll
  for (int k = 0; k < CPL_MAX; ++k) {
    const int i = lane * cpl + k;
    const bool ok = k < cpl && i < S;
    float dist = 0.f, sraw = 0.f, zv = 0.f;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (ok) {
      const int s = BG ? S - 1 - i : i;            // sample index in natural order
      const float4 r = raw[row0 + s];
      c0 = r.x; c1 = r.y; c2 = r.z; sraw = r.w;
      if (BG) {
        const float zi = z[s];
        dist = i < S - 1 ? zi - z[s - 1] : HUGE_NUM;
        zv = depth_real[row0 + s];
      } else {
        const float zi = z[i];
        dist = dnorm * ((i < S - 1 ? z[i + 1] : far) - zi);
        zv = zi;
      }
    }
    const float sigma = fabsf(sraw);
    const float e = ok ? expf(-sigma * dist) : 1.f;
    const float alpha = 1.f - e;
    const float q = ok ? 1.f - alpha + TINY : 1.f;
    v.dist[k] = dist; v.e[k] = e; v.alpha[k] = alpha; v.q[k] = q;
    v.c[k][0] = c0; v.c[k][1] = c1; v.c[k][2] = c2; v.zval[k] = zv; v.sraw[k] = sraw;
    v.T[k] = run;                       // lane-local exclusive product
    run *= q;
  }
  float total;
  const float pre = wave_excl_prod(run, lane, &total);
  v.lambda = total;
#pragma unroll
  for (int k = 0; k < CPL_MAX; ++k) {
    v.T[k] *= pre;
    v.w[k] = v.alpha[k] * v.T[k];
  }
}

__global__ __launch_bounds__(256) void composite_fwd_kernel(
    int n, int S, const float4* __restrict__ raw_fg, const float4* __restrict__ raw_bg,
    const float* __restrict__ depth_real_bg, const float* __restrict__ ray_d,
    const float* __restrict__ fg_far, const float* __restrict__ fg_z, const float* __restrict__ bg_z,
    float* __restrict__ rgb, float* __restrict__ depth, float* __restrict__ fg_weights,
    float* __restrict__ bg_weights, float* __restrict__ fg_dists, float* __restrict__ fg_rgb,
    float* __restrict__ fg_depth, float* __restrict__ bg_rgb, float* __restrict__ bg_depth,
    float* __restrict__ bg_lambda) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  if (ray >= n) return;
  const int cpl = (S + 63) / 64;
  const size_t row0 = (size_t)ray * S;
  const float dx = ray_d[ray * 3], dy = ray_d[ray * 3 + 1], dz = ray_d[ray * 3 + 2];
  const float dnorm = sqrtf(sum3(dx * dx, dy * dy, dz * dz));
  RaySamples v;
  load_volume<false>(v, S, cpl, lane, row0, raw_fg, fg_z + row0, nullptr, dnorm, fg_far[ray]);
  float a0 = 0, a1 = 0, a2 = 0, ad = 0;
#pragma unroll
  for (int k = 0; k < CPL_MAX; ++k) {
    const int i = lane * cpl + k;
    if (k < cpl && i < S) {
      fg_weights[row0 + i] = v.w[k];
      fg_dists[row0 + i] = v.dist[k];
      a0 += v.w[k] * v.c[k][0]; a1 += v.w[k] * v.c[k][1]; a2 += v.w[k] * v.c[k][2];
      ad += v.w[k] * v.zval[k];
    }
  }
  const float f0 = wave_sum(a0), f1 = wave_sum(a1), f2 = wave_sum(a2), fd = wave_sum(ad);
  const float lam = v.lambda;
  load_volume<true>(v, S, cpl, lane, row0, raw_bg, bg_z + row0, depth_real_bg, dnorm, 0.f);
  a0 = a1 = a2 = ad = 0;
#pragma unroll
  for (int k = 0; k < CPL_MAX; ++k) {
    const int i = lane * cpl + k;
    if (k < cpl && i < S) {
      bg_weights[row0 + i] = v.w[k];
      a0 += v.w[k] * v.c[k][0]; a1 += v.w[k] * v.c[k][1]; a2 += v.w[k] * v.c[k][2];
      ad += v.w[k] * v.zval[k];
    }
  }
  const float b0 = lam * wave_sum(a0), b1 = lam * wave_sum(a1), b2 = lam * wave_sum(a2);
  const float bd = lam * wave_sum(ad);
  if (lane == 0) {
    fg_rgb[ray * 3] = f0; fg_rgb[ray * 3 + 1] = f1; fg_rgb[ray * 3 + 2] = f2;
    bg_rgb[ray * 3] = b0; bg_rgb[ray * 3 + 1] = b1; bg_rgb[ray * 3 + 2] = b2;
    rgb[ray * 3] = f0 + b0; rgb[ray * 3 + 1] = f1 + b1; rgb[ray * 3 + 2] = f2 + b2;
    fg_depth[ray] = fd; bg_depth[ray] = bd; depth[ray] = fd + bd; bg_lambda[ray] = lam;
  }
}

// Backward of the compositing: given dL/d rgb [n,3], dL/d depth [n] and
// (KL only) dL/d fg_weights [n,S], writes per sample (d rgb_pre-sigmoid[3], d sigma_raw) for the
// fg and bg MLPs, natural sample order.
// KL term formed in place (fused loss head): dL/d fg_weights of depth_kl, the arithmetic of kl_terms_kernel
struct KlFuse { bool on; float gt, inv2s, lambda_depth; };
template <bool BG>
__device__ __forceinline__ void volume_backward(const RaySamples& v, int S, int cpl, int lane, size_t row0,
                                                const float gC[3], float gD, float g_lam_lam,
                                                const float* __restrict__ g_w_extra,
                                                float4* __restrict__ dout, const KlFuse kl = KlFuse{false, 0.f, 0.f, 0.f}) {
  float gw[CPL_MAX], term = 0.f;
#pragma unroll
  for (int k = 0; k < CPL_MAX; ++k) {
    const int i = lane * cpl + k;
    const bool ok = k < cpl && i < S;
    float g = gC[0] * v.c[k][0] + gC[1] * v.c[k][1] + gC[2] * v.c[k][2] + gD * v.zval[k];
    if (ok && g_w_extra) g += g_w_extra[row0 + i];
    if (!BG && ok && kl.on) {
      const float dz = v.zval[k] - kl.gt;
      const float w = v.w[k] + 1e-5f;
      const float e = expf(-(dz * dz) * kl.inv2s);
      g += -kl.lambda_depth * e * v.dist[k] / (w * (float)S);
    }
    gw[k] = ok ? g : 0.f;
    term += gw[k] * v.w[k];
  }
  float suffix = wave_excl_suffix_sum(term, lane);       // sum over later lanes
#pragma unroll
  for (int k = CPL_MAX - 1; k >= 0; --k) {
    const int i = lane * cpl + k;
    if (k < cpl && i < S) {
      const float d_a = gw[k] * v.T[k] - (suffix + g_lam_lam) / v.q[k];
      const float d_sigma = d_a * v.dist[k] * v.e[k];
      const float sgn = v.sraw[k] > 0.f ? 1.f : (v.sraw[k] < 0.f ? -1.f : 0.f);
      float4 o;
      o.x = v.w[k] * gC[0] * v.c[k][0] * (1.f - v.c[k][0]);
      o.y = v.w[k] * gC[1] * v.c[k][1] * (1.f - v.c[k][1]);
      o.z = v.w[k] * gC[2] * v.c[k][2] * (1.f - v.c[k][2]);
      o.w = d_sigma * sgn;
      const int s = BG ? S - 1 - i : i;
      dout[row0 + s] = o;
    }
    suffix += gw[k] * v.w[k];
  }
}

// LF (fused loss head, nerfpp_backward_args::fused_loss): dL/d rgb, dL/d depth and (KL) dL/d fg_weights are formed here
// with the arithmetic of loss_kernel / kl_terms_kernel instead of being read back from a loss launch that sits between
// the compositing forward and this kernel on the critical path.  The mse / l1 normaliser (#rays with a prior) depends
// on the batch only: every workgroup recounts it from depth_sup (n floats, L2-resident; exact in float up to 2^24).
template <bool LF>
__global__ __launch_bounds__(256) void composite_bwd_kernel(
    int n, int S, const float4* __restrict__ raw_fg, const float4* __restrict__ raw_bg,
    const float* __restrict__ depth_real_bg, const float* __restrict__ ray_d,
    const float* __restrict__ fg_far, const float* __restrict__ fg_z, const float* __restrict__ bg_z,
    const float* __restrict__ g_rgb, const float* __restrict__ g_depth,
    const float* __restrict__ g_fg_weights, float4* __restrict__ dout_fg, float4* __restrict__ dout_bg, LossFuse lf) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  float cnt = 0.f;
  if (LF && (lf.type == 1 || lf.type == 2)) {
    __shared__ float s_cnt[4];
    float c = 0.f;
    for (int r = threadIdx.x; r < n; r += 256) c += lf.depth_sup[r] > 0.f ? 1.f : 0.f;
    c = wave_sum(c);
    if (lane == 0) s_cnt[wave] = c;
    __syncthreads();
    cnt = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
  }
  if (ray >= n) return;
  const int cpl = (S + 63) / 64;
  const size_t row0 = (size_t)ray * S;
  const float dx = ray_d[ray * 3], dy = ray_d[ray * 3 + 1], dz = ray_d[ray * 3 + 2];
  const float dnorm = sqrtf(sum3(dx * dx, dy * dy, dz * dz));
  float gC[3], gD;
  KlFuse kl{false, 0.f, 0.f, 0.f};
  if (LF) {
#pragma unroll
    for (int c = 0; c < 3; ++c) gC[c] = 2.f * (lf.rgb[ray * 3 + c] - lf.rgb_gt[ray * 3 + c]) / (float)(3 * n);
    gD = 0.f;
    if (lf.type == 1 || lf.type == 2) {
      const float gt = lf.depth_sup[ray];
      if (gt > 0.f && cnt > 0.f) {
        const float d = lf.depth[ray] - gt;
        gD = lf.type == 1 ? lf.lambda_depth * 2.f * d / cnt
                          : lf.lambda_depth * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / cnt;
      }
    } else if (lf.type == 3) {
      const float gt = lf.depth_sup[ray];
      kl.on = gt > 0.f && gt < fg_far[ray];
      kl.gt = gt;
      kl.inv2s = 1.f / (2.f * lf.kl_sigma);
      kl.lambda_depth = lf.lambda_depth;
    }
  } else {
    gC[0] = g_rgb[ray * 3]; gC[1] = g_rgb[ray * 3 + 1]; gC[2] = g_rgb[ray * 3 + 2];
    gD = g_depth[ray];
  }
  RaySamples vb;
  load_volume<true>(vb, S, cpl, lane, row0, raw_bg, bg_z + row0, depth_real_bg, dnorm, 0.f);
  float a0 = 0, a1 = 0, a2 = 0, ad = 0;
#pragma unroll
  for (int k = 0; k < CPL_MAX; ++k) {
    a0 += vb.w[k] * vb.c[k][0]; a1 += vb.w[k] * vb.c[k][1]; a2 += vb.w[k] * vb.c[k][2];
    ad += vb.w[k] * vb.zval[k];
  }
  const float g_lam = gC[0] * wave_sum(a0) + gC[1] * wave_sum(a1) + gC[2] * wave_sum(a2) + gD * wave_sum(ad);
  RaySamples vf;
  load_volume<false>(vf, S, cpl, lane, row0, raw_fg, fg_z + row0, nullptr, dnorm, fg_far[ray]);
  const float lam = vf.lambda;
  volume_backward<false>(vf, S, cpl, lane, row0, gC, gD, g_lam * lam, LF ? nullptr : g_fg_weights, dout_fg, kl);
  const float gCb[3] = {lam * gC[0], lam * gC[1], lam * gC[2]};
  volume_backward<true>(vb, S, cpl, lane, row0, gCb, lam * gD, 0.f, nullptr, dout_bg);
}

// ------------------------------------------------------------------------------------------------
// loss head + its gradient w.r.t. (rgb, depth, fg_weights).  ONE workgroup (deterministic
// reduction order).  type: 0 rgb only, 1 mse, 2 l1, 3 kl.
// scalars[0..3] = loss, rgb_loss, depth_loss, #valid rays
// ------------------------------------------------------------------------------------------------
__device__ double block_sum(double v, double* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int d = blockDim.x >> 1; d > 0; d >>= 1) {
    if (t < d) sh[t] += sh[t + d];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// depth_kl (depth_loss.py:20-44) per-sample work, one wave per ray (coalesced over the samples):
//   term[n,s] = -log(w + 1e-5) * exp(-(z - gt)^2 / (2 sigma)) * dists,   masked by 0 < gt < fg_far
//   g_w[n,s]  = d(lambda * sum(term) / S) / d w   -- needs no global normaliser (the reference divides by S)
// The per-ray sums go to ray_sum (the caller passes g_depth, which is all zeros for this loss type and is
// overwritten by loss_kernel afterwards); loss_kernel adds them up in a fixed order.
__global__ __launch_bounds__(256) void kl_terms_kernel(
    int n, int S, float lambda_depth, float kl_sigma, const float* __restrict__ depth_sup,
    const float* __restrict__ fg_weights, const float* __restrict__ fg_z, const float* __restrict__ fg_dists,
    const float* __restrict__ fg_far, float* __restrict__ ray_sum, float* __restrict__ g_fg_weights) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  if (ray >= n) return;
  const float gt = depth_sup[ray];
  const bool m = gt > 0.f && gt < fg_far[ray];
  const float inv2s = 1.f / (2.f * kl_sigma);
  double acc = 0.0;
  for (int s = lane; s < S; s += 64) {
    const size_t i = (size_t)ray * S + s;
    float g = 0.f;
    if (m) {
      const float dz = fg_z[i] - gt;
      const float w = fg_weights[i] + 1e-5f;
      const float e = expf(-(dz * dz) * inv2s);
      acc += (double)(-logf(w) * e * fg_dists[i]);
      g = -lambda_depth * e * fg_dists[i] / (w * (float)S);
    }
    g_fg_weights[i] = g;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if (lane == 0) ray_sum[ray] = (float)acc;
}

__global__ __launch_bounds__(1024) void loss_kernel(
    int n, int S, int type, float lambda_depth, float kl_sigma, const float* __restrict__ rgb,
    const float* __restrict__ rgb_gt, const float* __restrict__ depth, const float* __restrict__ depth_sup,
    const float* __restrict__ fg_weights, const float* __restrict__ fg_z, const float* __restrict__ fg_dists,
    const float* __restrict__ fg_far, float* __restrict__ scalars, float* __restrict__ g_rgb,
    float* __restrict__ g_depth, float* __restrict__ g_fg_weights) {
  __shared__ double sh[1024];
  double s_rgb = 0, s_dep = 0, s_cnt = 0;
  for (int r = threadIdx.x; r < n; r += blockDim.x) {
    for (int c = 0; c < 3; ++c) {
      const float d = rgb[r * 3 + c] - rgb_gt[r * 3 + c];
      s_rgb += (double)(d * d);
    }
    if (type == 0) continue;
    const float gt = depth_sup[r];
    if (type == 3) {
      const bool m = gt > 0.f && gt < fg_far[r];
      if (m) {
        s_cnt += 1;
        s_dep += (double)g_depth[r];             // per-ray sum left there by kl_terms_kernel
      }
    } else if (gt > 0.f) {
      const float d = gt - depth[r];
      s_cnt += 1;
      s_dep += type == 1 ? (double)(d * d) : (double)fabsf(d);
    }
  }
  const double t_rgb = block_sum(s_rgb, sh), t_dep = block_sum(s_dep, sh), t_cnt = block_sum(s_cnt, sh);
  const float rgb_loss = (float)(t_rgb / (3.0 * n));
  float depth_loss = 0.f;
  if (type == 3) depth_loss = (float)(t_dep / S);
  else if (type != 0) depth_loss = (float)(t_dep / t_cnt);          // 0/0 = NaN like the reference
  if (threadIdx.x == 0) {
    scalars[0] = type == 0 ? rgb_loss : rgb_loss + lambda_depth * depth_loss;
    scalars[1] = rgb_loss;
    scalars[2] = depth_loss;
    scalars[3] = (float)t_cnt;
  }
  const float cnt = (float)t_cnt;
  for (int r = threadIdx.x; r < n; r += blockDim.x) {
    for (int c = 0; c < 3; ++c)
      g_rgb[r * 3 + c] = 2.f * (rgb[r * 3 + c] - rgb_gt[r * 3 + c]) / (float)(3 * n);
    float gd = 0.f;
    if (type == 1 || type == 2) {
      const float gt = depth_sup[r];
      if (gt > 0.f && cnt > 0.f) {
        const float d = depth[r] - gt;
        gd = type == 1 ? lambda_depth * 2.f * d / cnt
                       : lambda_depth * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / cnt;
      }
    }
    g_depth[r] = gd;
    if (g_fg_weights && type != 3)               // (KL: written by kl_terms_kernel)
      for (int s = 0; s < S; ++s) g_fg_weights[(size_t)r * S + s] = 0.f;
  }
}

}  // namespace nerfpp

// ------------------------------------------------------------------------------------------------
// host launchers (called from the C ABI in nerfpp_api.hip)
// ------------------------------------------------------------------------------------------------
using namespace nerfpp;

void launch_intersect_sphere(hipStream_t st, int n, const float* o, const float* d, float* far, int* bad) {
  hipLaunchKernelGGL(intersect_sphere_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, o, d, far, bad);
}
void launch_sample_coarse(hipStream_t st, int n, int S, const float* o, const float* d, const float* min_depth,
                          const float* t_fg, const float* t_bg, float* far, float* fg_z, float* bg_z, int* bad,
                          const nerfpp::RngKey& rng) {
  const int tot = n * S;
  hipLaunchKernelGGL(sample_coarse_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, n, S, o, d, min_depth,
                     t_fg, t_bg, far, fg_z, bg_z, bad, rng);
}
void launch_rng_uniform(hipStream_t st, const nerfpp::RngKey& rng, uint32_t stream_id, int64_t n, float* out) {
  hipLaunchKernelGGL(rng_uniform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rng, stream_id, n, out);
}
void launch_perturb(hipStream_t st, int n, int S, const float* z, const float* t, float* out) {
  const int tot = n * S;
  hipLaunchKernelGGL(perturb_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, n, S, z, t, out);
}
void launch_sample_pdf(hipStream_t st, bool fused, int n, int M, int S_new, const float* bins_or_zold,
                       const float* weights, const float* u, float* samples, int64_t* above, float* merged) {
  dim3 grid((n + 3) / 4), block(256);
  SamplePdfArgs a{};
  a.p[0] = SamplePdfProblem{bins_or_zold, weights, u, samples, above, merged};
  if (fused) hipLaunchKernelGGL(sample_pdf_kernel<true>, grid, block, 0, st, n, M, S_new, a);
  else hipLaunchKernelGGL(sample_pdf_kernel<false>, grid, block, 0, st, n, M, S_new, a);
}
void launch_sample_fine_pair(hipStream_t st, int n, int M, int S_new, const float* const* z_old,
                             const float* const* weights, const float* const* u, float* const* merged,
                             const nerfpp::RngKey& rng) {
  dim3 grid((n + 3) / 4, 2), block(256);
  SamplePdfArgs a{};
  a.rng = rng;
  for (int k = 0; k < 2; ++k) a.p[k] = SamplePdfProblem{z_old[k], weights[k], u[k], nullptr, nullptr, merged[k]};
  hipLaunchKernelGGL(sample_pdf_kernel<true>, grid, block, 0, st, n, M, S_new, a);
}
void launch_composite_fwd(hipStream_t st, int n, int S, const float* raw_fg, const float* raw_bg,
                          const float* depth_real_bg, const float* ray_d, const float* fg_far,
                          const float* fg_z, const float* bg_z, float* rgb, float* depth, float* fg_weights,
                          float* bg_weights, float* fg_dists, float* fg_rgb, float* fg_depth, float* bg_rgb,
                          float* bg_depth, float* bg_lambda) {
  hipLaunchKernelGGL(composite_fwd_kernel, dim3((n + 3) / 4), dim3(256), 0, st, n, S, (const float4*)raw_fg,
                     (const float4*)raw_bg, depth_real_bg, ray_d, fg_far, fg_z, bg_z, rgb, depth, fg_weights,
                     bg_weights, fg_dists, fg_rgb, fg_depth, bg_rgb, bg_depth, bg_lambda);
}
void launch_composite_bwd(hipStream_t st, int n, int S, const float* raw_fg, const float* raw_bg,
                          const float* depth_real_bg, const float* ray_d, const float* fg_far,
                          const float* fg_z, const float* bg_z, const float* g_rgb, const float* g_depth,
                          const float* g_fg_weights, float* dout_fg, float* dout_bg, const nerfpp::LossFuse* lf) {
  if (lf)
    hipLaunchKernelGGL(composite_bwd_kernel<true>, dim3((n + 3) / 4), dim3(256), 0, st, n, S, (const float4*)raw_fg,
                       (const float4*)raw_bg, depth_real_bg, ray_d, fg_far, fg_z, bg_z, g_rgb, g_depth,
                       g_fg_weights, (float4*)dout_fg, (float4*)dout_bg, *lf);
  else
    hipLaunchKernelGGL(composite_bwd_kernel<false>, dim3((n + 3) / 4), dim3(256), 0, st, n, S, (const float4*)raw_fg,
                       (const float4*)raw_bg, depth_real_bg, ray_d, fg_far, fg_z, bg_z, g_rgb, g_depth,
                       g_fg_weights, (float4*)dout_fg, (float4*)dout_bg, nerfpp::LossFuse{});
}
void launch_loss(hipStream_t st, int n, int S, int type, float lambda_depth, float kl_sigma, const float* rgb,
                 const float* rgb_gt, const float* depth, const float* depth_sup, const float* fg_weights,
                 const float* fg_z, const float* fg_dists, const float* fg_far, float* scalars, float* g_rgb,
                 float* g_depth, float* g_fg_weights) {
  if (type == 3)
    hipLaunchKernelGGL(kl_terms_kernel, dim3((n + 3) / 4), dim3(256), 0, st, n, S, lambda_depth, kl_sigma, depth_sup,
                       fg_weights, fg_z, fg_dists, fg_far, g_depth, g_fg_weights);
  hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(1024), 0, st, n, S, type, lambda_depth, kl_sigma, rgb, rgb_gt,
                     depth, depth_sup, fg_weights, fg_z, fg_dists, fg_far, scalars, g_rgb, g_depth,
                     g_fg_weights);
}

// ------------------------------------------------------------------------------------------------
// f-1: ray batch from a GPU-resident frame.  nerf_sample_ray_split.py:10-34 (get_rays_single_image:
// half-pixel centres, d = R * K^-1 * [u+.5, v+.5, 1]^T un-normalised, o = t) and :178-221
// (random_sample: gather rgb / depth at the selected pixels, min_depth = 1e-4).
// cam[21] = K^-1 (3x3 row-major) then c2w[:3, :4] (row-major), float32, as the host computed them.
// ------------------------------------------------------------------------------------------------
namespace nerfpp {
__global__ void gather_rays_kernel(int n, int W, const float* __restrict__ cam, const int64_t* __restrict__ pix,
                                   const float* __restrict__ rgb_img, const float* __restrict__ depth_img,
                                   float* __restrict__ ray_o, float* __restrict__ ray_d, float* __restrict__ rgb,
                                   float* __restrict__ depth_sup, float* __restrict__ min_depth) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t p = pix[i];
  const float u = (float)(p % W) + 0.5f, v = (float)(p / W) + 0.5f;
  const float cx = (cam[0] * u + cam[1] * v) + cam[2];
  const float cy = (cam[3] * u + cam[4] * v) + cam[5];
  const float cz = (cam[6] * u + cam[7] * v) + cam[8];
  const float* c2w = cam + 9;
  ray_d[i * 3 + 0] = (c2w[0] * cx + c2w[1] * cy) + c2w[2] * cz;
  ray_d[i * 3 + 1] = (c2w[4] * cx + c2w[5] * cy) + c2w[6] * cz;
  ray_d[i * 3 + 2] = (c2w[8] * cx + c2w[9] * cy) + c2w[10] * cz;
  ray_o[i * 3 + 0] = c2w[3]; ray_o[i * 3 + 1] = c2w[7]; ray_o[i * 3 + 2] = c2w[11];
  if (rgb) { rgb[i * 3] = rgb_img[p * 3]; rgb[i * 3 + 1] = rgb_img[p * 3 + 1]; rgb[i * 3 + 2] = rgb_img[p * 3 + 2]; }
  if (depth_sup) depth_sup[i] = depth_img[p];
  min_depth[i] = 1e-4f;
}
}  // namespace nerfpp

// ------------------------------------------------------------------------------------------------
// k distinct pixels out of n, uniform over ordered k-tuples of distinct values: the distribution of
// np.random.choice(H * W, size=(N_rand,), replace=False) (nerf_sample_ray_split.py:178), without permuting all H * W
// pixels (torch.randperm of 465 750 elements cost 0.3 ms per step in the training loop, bench.py cli_loop).
// Definition (oracle/nerfpp_oracle.py: sample_pixels): element i owns the draws d = 0, 1, ... of Philox stream 4 at counter
// i + 2^20 d, mapped to [0, n) by multiply-shift with Lemire's rejection (exactly uniform); x_i = its first draw that differs
// from x_j for every j < i.  Parallel form, one workgroup: every element draws, the (value, index) pairs are sorted by a bitonic
// network in LDS, an element whose predecessor in sorted order carries the same value (that one has the smaller index)
// draws again, until nobody has to -- the rejected draws are exactly those of the sequential definition (a draw that is
// rejected against a value which is itself rejected later equals the earlier final value that one collided with).
// ------------------------------------------------------------------------------------------------
namespace nerfpp {
constexpr int PIX_THREADS = 1024;
__global__ __launch_bounds__(PIX_THREADS) void sample_pixels_kernel(RngKey rng, uint32_t n, int k, int kp, int64_t* __restrict__ pix) {
  extern __shared__ __attribute__((aligned(16))) char pix_smem[];
  unsigned long long* keys = (unsigned long long*)pix_smem;            // [kp]  (value << 32) | index, padding = ~0
  uint32_t* x = (uint32_t*)(pix_smem + (size_t)kp * 8);                // [k]   current values
  uint32_t* ndraw = x + k;                                             // [k]   draws consumed
  __shared__ int again;
  const uint32_t thresh = (0u - n) % n;                                // Lemire: reject the low 2^32 mod n products
  auto draw = [&](int i) {
    uint32_t d = ndraw[i], v;
    for (;;) {
      const uint32_t w = philox_word(rng, 4u, (uint64_t)i + ((uint64_t)d << 20));
      ++d;
      const uint64_t m = (uint64_t)w * n;
      if ((uint32_t)m >= thresh) { v = (uint32_t)(m >> 32); break; }
    }
    ndraw[i] = d;
    x[i] = v;
  };
  for (int i = threadIdx.x; i < k; i += PIX_THREADS) { ndraw[i] = 0; draw(i); }
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) again = 0;
    for (int i = threadIdx.x; i < kp; i += PIX_THREADS)
      keys[i] = i < k ? ((unsigned long long)x[i] << 32) | (unsigned)i : ~0ull;
    __syncthreads();
    for (int size = 2; size <= kp; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = threadIdx.x; t < kp / 2; t += PIX_THREADS) {
          const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
          const bool up = (lo & size) == 0;
          const unsigned long long a = keys[lo], b = keys[hi];
          if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
        }
        __syncthreads();
      }
    for (int p = threadIdx.x + 1; p < k; p += PIX_THREADS)
      if ((uint32_t)(keys[p] >> 32) == (uint32_t)(keys[p - 1] >> 32)) { draw((int)(uint32_t)keys[p]); again = 1; }
    __syncthreads();
    if (!again) break;
  }
  for (int i = threadIdx.x; i < k; i += PIX_THREADS) pix[i] = (int64_t)x[i];
}
}  // namespace nerfpp

int launch_sample_pixels(hipStream_t st, const nerfpp::RngKey& rng, int64_t n_pixels, int k, int64_t* pix) {
  int kp = 2;
  while (kp < k) kp <<= 1;
  const size_t lds = (size_t)kp * 8 + (size_t)k * 8;
  hipLaunchKernelGGL(nerfpp::sample_pixels_kernel, dim3(1), dim3(nerfpp::PIX_THREADS), lds, st, rng, (uint32_t)n_pixels, k, kp, pix);
  return 0;
}

void launch_gather_rays(hipStream_t st, int n, int W, const float* cam, const int64_t* pix, const float* rgb_img,
                        const float* depth_img, float* ray_o, float* ray_d, float* rgb, float* depth_sup,
                        float* min_depth) {
  hipLaunchKernelGGL(nerfpp::gather_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, W, cam, pix, rgb_img,
                     depth_img, ray_o, ray_d, rgb, depth_sup, min_depth);
}
