// MipNeRF-360 ray-side kernels for gfx950: proposal resampling, conical-frustum featurisation
// (cast -> contract -> lift -> IPE), alpha compositing forward / backward, and the loss terms with their
// gradients.  All float32, HBM / latency bound, one wave per ray (the NeRF++ compositing design re-used).
// Compiled with -ffp-contract=off so the arithmetic order is the oracle's (oracle/mip360_oracle.py).
//
// Upstream lines (nerf-methods/mipnerf360/internal/): stepfun.py:30-283, coord.py:21-135, render.py:21-216,
// models.py:158-226, train_utils.py:72-169.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/mip360_hip.h"

namespace mip360 {

constexpr float EPS = 1.1920928955078125e-07f;        // jnp.finfo(float32).eps
constexpr float EPS2 = EPS * EPS;
constexpr int MAXE = 3 * MIP360_MAX_BINS + 1;          // edges after dilation
constexpr int RPB = 4;                                  // rays per 256-thread block (one wave each)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}
// inclusive prefix sum across the wave
__device__ __forceinline__ float wave_incl_sum(float x, int lane) {
  float v = x;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}
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

// ------------------------------------------------------------------------------------------------------------
// One sampling level: dilate -> anneal -> softmax -> cdf -> sample_intervals -> s_to_t.   models.py:158-208
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resample_kernel(
    int n, int m_in, const float* __restrict__ sdist_in, const float* __restrict__ w_in, float dilation, float anneal,
    float padding, int ns, const float* __restrict__ jitter01, float s_near, float s_far,
    const float* __restrict__ t_near, const float* __restrict__ t_far, float* __restrict__ sdist_out,
    float* __restrict__ tdist_out) {
  __shared__ float s_t[RPB][MAXE + 3];        // edges of the (dilated) step function
  __shared__ float s_w[RPB][MAXE + 3];        // bin values: pdf, then weights, then softmax weights
  __shared__ float s_a[RPB][MAXE + 3];        // scratch: unsorted edges / cdf
  __shared__ float s_c[RPB][MIP360_MAX_SAMPLES + 2];
  __shared__ float s_p[RPB][MIP360_MAX_BINS];   // pdf of the input intervals
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * RPB + wave;
  if (ray >= n) return;                         // whole wave exits together; no block-level barrier below
  float* t = s_t[wave];
  float* w = s_w[wave];
  float* a = s_a[wave];
  float* c = s_c[wave];
  const float* ti = sdist_in + (size_t)ray * (m_in + 1);
  const float* wi = w_in + (size_t)ray * m_in;
  int nb;                                       // bins of the step function that is sampled
  if (dilation > 0.f) {
    // max_dilate_weights (stepfun.py:100-130): p = w / max(eps^2, dt); t0 = t[:-1] - d, t1 = t[1:] + d
    const int ne = 3 * m_in + 1;
    for (int i = lane; i <= m_in; i += 64) a[i] = ti[i];
    for (int i = lane; i < m_in; i += 64) {
      a[m_in + 1 + i] = ti[i] - dilation;
      a[2 * m_in + 1 + i] = ti[i + 1] + dilation;
    }
    __builtin_amdgcn_wave_barrier();
    // sort of the 3 m + 1 edges (values only).  The three lists are each non-decreasing, so the rank of an element is its
    // index in its own list plus, by binary search, the number of elements of the lists BEFORE it (in concatenation order)
    // that are <= it and of the lists AFTER it that are < it -- exactly the rank the all-pairs comparison
    // (o < v || (o == v && k < e)) gives, in 14 LDS reads instead of 193.
    for (int e = lane; e < ne; e += 64) {
      const float v = a[e];
      const int list = e <= m_in ? 0 : (e <= 2 * m_in ? 1 : 2);
      const int start[3] = {0, m_in + 1, 2 * m_in + 1}, len[3] = {m_in + 1, m_in, m_in};
      int rank = e - start[list];
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        if (o == list) continue;
        const float* base = a + start[o];
        int lo_ = 0, hi_ = len[o];                 // first index whose element is > v (o before) or >= v (o after)
        while (lo_ < hi_) {
          const int mid = (lo_ + hi_) >> 1;
          const float x = base[mid];
          const bool go_right = o < list ? (x <= v) : (x < v);
          lo_ = go_right ? mid + 1 : lo_;
          hi_ = go_right ? hi_ : mid;
        }
        rank += lo_;
      }
      t[rank] = fminf(fmaxf(v, s_near), s_far);
    }
    __builtin_amdgcn_wave_barrier();
    // max-pool of the pdf over the dilated supports.  The pdf of every input interval is computed once (it used to be
    // recomputed, division included, for every output bin); supports [t_i - d, t_{i+1} + d) have non-decreasing ends, so
    // the intervals covering an edge form one contiguous range [first i with hi_i > tk, first i with lo_i > tk).
    float* pdf = s_p[wave];
    for (int i = lane; i < m_in; i += 64) pdf[i] = wi[i] / fmaxf(EPS2, ti[i + 1] - ti[i]);
    __builtin_amdgcn_wave_barrier();
    const float* lo_e = a + m_in + 1;             // t_i - d       (still in the unsorted edge array)
    const float* hi_e = a + 2 * m_in + 1;         // t_{i+1} + d
    float part = 0.f;
    for (int k = lane; k < ne - 1; k += 64) {
      const float tk = t[k];
      int b0 = 0, b1 = m_in;                      // first i with hi_i > tk
      while (b0 < b1) { const int mid = (b0 + b1) >> 1; const bool r = hi_e[mid] <= tk; b0 = r ? mid + 1 : b0; b1 = r ? b1 : mid; }
      int a0 = 0, a1 = m_in;                      // first i with lo_i > tk
      while (a0 < a1) { const int mid = (a0 + a1) >> 1; const bool r = lo_e[mid] <= tk; a0 = r ? mid + 1 : a0; a1 = r ? a1 : mid; }
      float p = 0.f;
      for (int i = b0; i < a0; ++i) p = fmaxf(p, pdf[i]);
      const float wd = p * (t[k + 1] - tk);
      w[k] = wd;
      part += wd;
    }
    const float tot = fmaxf(EPS2, wave_sum(part));
    __builtin_amdgcn_wave_barrier();
    // renormalise and trim [1:-1] (models.py:172-173): edges 1..ne-2, bins 1..ne-3
    nb = ne - 3;
    for (int k = lane; k < nb; k += 64) a[k] = w[k + 1] / tot;
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < nb; k += 64) w[k] = a[k];
    for (int k = lane; k <= nb; k += 64) a[k] = t[k + 1];
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k <= nb; k += 64) t[k] = a[k];
    __builtin_amdgcn_wave_barrier();
  } else {
    nb = m_in;
    for (int i = lane; i <= m_in; i += 64) t[i] = ti[i];
    for (int i = lane; i < m_in; i += 64) w[i] = wi[i];
    __builtin_amdgcn_wave_barrier();
  }
  // logits (models.py:179-183) and softmax (stepfun.py:158)
  float mx = -INFINITY;
  for (int k = lane; k < nb; k += 64) {
    const float l = t[k + 1] > t[k] ? anneal * logf(w[k] + padding) : -INFINITY;
    a[k] = l;
    mx = fmaxf(mx, l);
  }
  mx = wave_max(mx);
  float part = 0.f;
  for (int k = lane; k < nb; k += 64) {
    const float e = expf(a[k] - mx);
    a[k] = e;
    part += e;
  }
  const float tot = wave_sum(part);
  __builtin_amdgcn_wave_barrier();
  // integrate_weights (stepfun.py:133-152): cw0 = [0, min(1, cumsum(w[:-1])), 1]; sequential float32 like numpy
  for (int k = lane; k < nb; k += 64) a[k] = a[k] / tot;     // (the divisions in parallel; the running sum stays sequential)
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    float run = 0.f;
    w[0] = 0.f;
    for (int k = 0; k < nb - 1; ++k) {
      run += a[k];
      w[k + 1] = fminf(1.f, run);
    }
    w[nb] = 1.f;
  }
  __builtin_amdgcn_wave_barrier();
  // uniform positions (stepfun.py:195-213) in double like np.linspace, then invert_cdf by sorted_interp
  const double eps = (double)EPS;
  for (int k = lane; k < ns; k += 64) {
    double u;
    if (jitter01) {
      const double u_max = eps + (1.0 - eps) / ns;
      const double max_jitter = (1.0 - u_max) / (ns - 1) - eps;
      const double step = (1.0 - u_max) / (ns - 1);
      u = (double)k * step + (double)jitter01[ray] * max_jitter;
    } else {
      const double pad = 1.0 / (2 * ns);
      const double step = ((1.0 - pad - eps) - pad) / (ns - 1);
      u = pad + (double)k * step;
    }
    const float uf = (float)u;
    // last edge with cw0 <= u and first edge with cw0 > u (cw0 and t are non-decreasing)
    // (cw0 is non-decreasing with cw0[0] = 0: binary search for the first edge above u)
    int s0 = 0, s1 = nb + 1;
    while (s0 < s1) { const int mid = (s0 + s1) >> 1; const bool r = uf >= w[mid]; s0 = r ? mid + 1 : s0; s1 = r ? s1 : mid; }
    const int hi = s0 <= nb ? s0 : nb;
    const int lo = s0 <= nb ? (s0 > 0 ? s0 - 1 : 0) : nb;
    const float xp0 = w[lo], xp1 = w[hi], fp0 = t[lo], fp1 = t[hi];
    float off = (uf - xp0) / (xp1 - xp0);
    off = isnan(off) ? 0.f : off;
    off = fminf(fmaxf(off, 0.f), 1.f);
    c[k] = fp0 + off * (fp1 - fp0);
  }
  __builtin_amdgcn_wave_barrier();
  // sample_intervals (stepfun.py:253-270) + s_to_t for the reciprocal warp (coord.py:93-99)
  const float sn = 1.f / t_near[ray], sf = 1.f / t_far[ray];
  for (int k = lane; k <= ns; k += 64) {
    float s;
    if (k == 0) s = fmaxf(s_near, 2.f * c[0] - (c[1] + c[0]) / 2.f);
    else if (k == ns) s = fminf(s_far, 2.f * c[ns - 1] - (c[ns - 1] + c[ns - 2]) / 2.f);
    else s = (c[k] + c[k - 1]) / 2.f;
    sdist_out[(size_t)ray * (ns + 1) + k] = s;
    tdist_out[(size_t)ray * (ns + 1) + k] = 1.f / (s * sf + (1.f - s) * sn);
  }
}

// ------------------------------------------------------------------------------------------------------------
// cast_rays (cone, full covariance) -> contract (mean + Jacobian) -> lift onto the basis -> IPE.
// One thread per (sample row, basis direction): 21 threads per row, 3 rows per wave.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float safe_sin(float x) {      // math.py:26-38 with np.mod semantics
  const float t = 314.15927f;                                // float32(100 pi)
  if (fabsf(x) >= t) {
    x = x - floorf(x / t) * t;
  }
  return sinf(x);
}

// MODE 0: float32 rows, 1: bf16 rows (row-major), 2: bf16 in the fragment-major layout of mip360_fm.hip -- a workgroup of
// 704 threads owns one 32-row block (thread t: row t / 21, basis t % 21), assembles its 32 blocks of 1 KiB in LDS and writes
// them out whole (enc = the tensor's base + the block offset of its first column, ld = the tensor's columns)
template <int MODE>
__global__ __launch_bounds__(MODE == 2 ? 704 : 256) void cast_encode_kernel(
    int n, int S, const float* __restrict__ tdist, const float* __restrict__ origins,
    const float* __restrict__ directions, const float* __restrict__ radii, const float* __restrict__ basis_t,
    void* __restrict__ enc, int ld) {
  constexpr bool BF16 = MODE != 0, FM = MODE == 2;
  const int wave_g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int sub = FM ? (int)threadIdx.x / MIP360_N_BASIS : lane / MIP360_N_BASIS;
  const int j = FM ? (int)threadIdx.x - sub * MIP360_N_BASIS : lane - sub * MIP360_N_BASIS;
  const int64_t row_raw = FM ? (int64_t)blockIdx.x * 32 + sub : (int64_t)wave_g * 3 + sub;
  const int64_t rows = (int64_t)n * S;
  const bool live = (FM ? sub < 32 : sub < 3) && row_raw < rows;
  const int64_t row = live ? row_raw : rows - 1;                 // idle lanes compute a valid row and store nothing
  if (!FM && (int64_t)wave_g * 3 >= rows) return;
  // per-row quantities: contracted mean cm, Jacobian J of the contraction, lifted covariance cov
  auto row_math = [&](int64_t row, float (&cm)[3], float (&J)[3][3], float (&cov)[3][3]) {
    const int ray = (int)(row / S), smp = (int)(row - (int64_t)ray * S);
    const float t0 = tdist[(size_t)ray * (S + 1) + smp], t1 = tdist[(size_t)ray * (S + 1) + smp + 1];
    const float d[3] = {directions[ray * 3], directions[ray * 3 + 1], directions[ray * 3 + 2]};
    const float o[3] = {origins[ray * 3], origins[ray * 3 + 1], origins[ray * 3 + 2]};
    const float br = radii[ray];
    // conical_frustum_to_gaussian, stable form (render.py:64-73)
    const float mu = (t0 + t1) / 2.f, hw = (t1 - t0) / 2.f;
    const float denom = fmaxf(EPS, 3.f * mu * mu + hw * hw);
    const float t_mean = mu + (2.f * mu * hw * hw) / denom;
    const float hw4 = hw * hw * hw * hw;
    const float t_var = (hw * hw) / 3.f - (4.f / 15.f) * hw4 * (12.f * mu * mu - hw * hw) / (denom * denom);
    float r_var = (mu * mu) / 4.f + (5.f / 12.f) * hw * hw - (4.f / 15.f) * hw4 / denom;
    r_var *= br * br;
    // lift_gaussian, diag = False (render.py:21-42)
    const float dms = fmaxf(1e-10f, d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float mean[3];
  #pragma unroll
    for (int a = 0; a < 3; ++a) {
      mean[a] = d[a] * t_mean + o[a];
  #pragma unroll
      for (int b = 0; b < 3; ++b) {
        const float d_outer = d[a] * d[b];
        const float null_outer = (a == b ? 1.f : 0.f) - d[a] * (d[b] / dms);
        cov[a][b] = t_var * d_outer + r_var * null_outer;
      }
    }
    // contract + its Jacobian (coord.py:21-27, track_linearize :39-60)
    const float m2 = fmaxf(EPS, mean[0] * mean[0] + mean[1] * mean[1] + mean[2] * mean[2]);
    if (m2 <= 1.f) {
  #pragma unroll
      for (int a = 0; a < 3; ++a) {
        cm[a] = mean[a];
  #pragma unroll
        for (int b = 0; b < 3; ++b) J[a][b] = a == b ? 1.f : 0.f;
      }
    } else {
      const float r = sqrtf(m2);
      const float s = (2.f * r - 1.f) / m2;
      const float ds = (1.f - r) / (m2 * m2);
  #pragma unroll
      for (int a = 0; a < 3; ++a) {
        cm[a] = s * mean[a];
  #pragma unroll
        for (int b = 0; b < 3; ++b) J[a][b] = (a == b ? s : 0.f) + 2.f * ds * mean[a] * mean[b];
      }
    }
  };
  float cm[3], J[3][3], cov[3][3];
  if (FM) {
    // one thread per row does the row's arithmetic (its 21 basis lanes would each repeat it), the others pick it up from LDS
    __shared__ float rowq[32][21];
    if (threadIdx.x < 32) {
      const int64_t r = (int64_t)blockIdx.x * 32 + threadIdx.x;
      row_math(r < rows ? r : rows - 1, cm, J, cov);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        rowq[threadIdx.x][a] = cm[a];
#pragma unroll
        for (int b = 0; b < 3; ++b) { rowq[threadIdx.x][3 + 3 * a + b] = J[a][b]; rowq[threadIdx.x][12 + 3 * a + b] = cov[a][b]; }
      }
    }
    __syncthreads();
    const int rs = sub < 32 ? sub : 31;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      cm[a] = rowq[rs][a];
#pragma unroll
      for (int b = 0; b < 3; ++b) { J[a][b] = rowq[rs][3 + 3 * a + b]; cov[a][b] = rowq[rs][12 + 3 * a + b]; }
    }
  } else {
    row_math(row, cm, J, cov);
  }
  // lift_and_diagonalize (coord.py:131-135): mean . b_j and b_j^T (J cov J^T) b_j = (J^T b_j)^T cov (J^T b_j)
  const float bj[3] = {basis_t[j], basis_t[MIP360_N_BASIS + j], basis_t[2 * MIP360_N_BASIS + j]};
  const float lm = cm[0] * bj[0] + cm[1] * bj[1] + cm[2] * bj[2];
  float v[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) v[b] = J[0][b] * bj[0] + J[1][b] * bj[1] + J[2][b] * bj[2];
  float lv = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) lv += v[a] * (cov[a][0] * v[0] + cov[a][1] * v[1] + cov[a][2] * v[2]);
  // integrated_pos_enc, degrees [0, 12) (coord.py:108-128): column k*21 + j = sin, 252 + k*21 + j = cos
  constexpr int ND = 12, HALF = ND * MIP360_N_BASIS;
  // bf16 rows feeding a GEMM (ld >= 512 columns, 16-byte aligned): the 21-lane pieces of a row are 42-byte runs of
  // 2-byte stores, so the wave's 3 rows (1 KiB each incl. the zero padding 504..511) are assembled in LDS and leave
  // as 16 bytes per lane (3 wave-stores instead of 24 partial ones)
  // (FM: 32 rows of 512 + 4 elements -- consecutive rows start 2 banks apart, so the 8-byte reads of the copy-out, one row per
  // lane, are conflict-free; used flat)
  __shared__ __attribute__((aligned(16))) __bf16 rows_lds[FM ? 33 : 4][FM ? 1 : 3][MIP360_IPE_LD];
  const bool staged = !FM && BF16 && ld >= MIP360_IPE_LD && (ld & 7) == 0 && (((uintptr_t)enc) & 15) == 0;
  const int wave_l = threadIdx.x >> 6;
  constexpr int FM_ROW = MIP360_IPE_LD + 4;                       // staged row stride (elements)
  auto fm_at = [&](int r, int col) -> __bf16* { return &rows_lds[0][0][0] + r * FM_ROW + col; };
  // one (sin, cos) pair of degree k to its two columns
  auto emit = [&](int k, float es, float ec) {
    if (FM || staged || BF16) {
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      const bf16x2_t p = __builtin_convertvector((f32x2_t){es, ec}, bf16x2_t);      // one v_cvt_pk_bf16_f32 for the pair
      if (FM) {
        *fm_at(sub, k * MIP360_N_BASIS + j) = p[0];
        *fm_at(sub, HALF + k * MIP360_N_BASIS + j) = p[1];
      } else if (staged) {
        rows_lds[wave_l][sub][k * MIP360_N_BASIS + j] = p[0];
        rows_lds[wave_l][sub][HALF + k * MIP360_N_BASIS + j] = p[1];
      } else {
        __bf16* e = (__bf16*)enc + (size_t)row * ld;
        e[k * MIP360_N_BASIS + j] = p[0];
        e[HALF + k * MIP360_N_BASIS + j] = p[1];
      }
    } else {
      float* e = (float*)enc + (size_t)row * ld;
      e[k * MIP360_N_BASIS + j] = es;
      e[HALF + k * MIP360_N_BASIS + j] = ec;
    }
  };
  if (live) {
    if (BF16) {
      // bf16 features: direct evaluation at the degrees 0, 2, 5, 8 only (safe_sin's argument handling, Cody-Waite reduction,
      // hardware sine AND cosine of the one reduced argument, one exponential); the one to three degrees after each follow by
      // angle doubling -- sin 2x = 2 sin x cos x, cos 2x = 1 - 2 sin^2 x, exp(-v/2)^4 = exp(-4v/2) -- 7 VALU operations instead
      // of two range reductions + three transcendentals.  A doubling multiplies the absolute error by <= 2.8: 4e-6 at degree 1,
      // 1e-5 at 4 and 7, 1e-4 at 11 (tools: the numpy twin in tests/test_layout_emulation.py), inside the frequency-scaled
      // tolerance 2e-6 + 3e-6 . 2^k the float32 path is tested to and far below the bf16 grid.
      float sn = 0.f, cs = 0.f, damp = 0.f;
#pragma unroll
      for (int k = 0; k < ND; ++k) {
        if (k == 0 || k == 2 || k == 5 || k == 8) {
          const float sc = (float)(1 << k);
          float x = lm * sc;
          const float t = 314.15927f;
          if (fabsf(x) >= t) x = x - floorf(x / t) * t;
          const float n = rintf(x * 0.15915494309189535f);
          float r = fmaf(-n, 6.2831854820251465f, x);
          r = fmaf(-n, -1.7484556000744883e-7f, r) * 0.15915494309189535f;
          sn = __builtin_amdgcn_sinf(r);
          cs = __builtin_amdgcn_cosf(r);
          damp = __expf(-0.5f * (lv * sc * sc));
        } else {
          const float u = sn + sn;
          const float s2 = u * cs;
          cs = fmaf(-u, sn, 1.f);
          sn = s2;
          const float d2 = damp * damp;
          damp = d2 * d2;
        }
        emit(k, damp * sn, damp * cs);
      }
    } else {
#pragma unroll
      for (int k = 0; k < ND; ++k) {
        const float sc = (float)(1 << k);
        const float sm = lm * sc, sv = lv * sc * sc;
        const float damp = expf(-0.5f * sv);
        emit(k, damp * safe_sin(sm), damp * safe_sin(sm + 1.5707963267948966f));
      }
    }
  }
  if (FM) {
    if (live && j < MIP360_IPE_LD - 2 * HALF) *fm_at(sub, 2 * HALF + j) = (__bf16)0.f;
    __syncthreads();
    // unit v of block cb = (row, hi): columns 16 cb + {4 hi + 0..3, 8 + 4 hi + 0..3} of the staged row
    char* dst = (char*)enc + (size_t)blockIdx.x * (size_t)(ld >> 4) * 1024;
    for (int u = threadIdx.x; u < 32 * 64; u += 704) {
      const int cb = u >> 6, v = u & 63;
      const int r = ((v >> 3) << 2) | (v & 3), hi = ((v >> 2) & 1) ^ (r >> 4);
      const __bf16* src = fm_at(r, cb * 16 + 4 * hi);
      const uint2 a = *(const uint2*)src, b = *(const uint2*)(src + 8);
      *(uint4*)(dst + (size_t)u * 16) = make_uint4(a.x, a.y, b.x, b.y);
    }
    return;
  }
  if (staged) {
    if (live && j < MIP360_IPE_LD - 2 * HALF) rows_lds[wave_l][sub][2 * HALF + j] = (__bf16)0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int64_t row0 = (int64_t)wave_g * 3;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int piece = it * 64 + lane, r = piece >> 6, c8 = (piece & 63) * 8;
      if (row0 + r < rows) *(uint4*)((__bf16*)enc + (size_t)(row0 + r) * ld + c8) = *(const uint4*)&rows_lds[wave_l][r][c8];
    }
    return;
  }
  if (!live) return;
  const int pad_to = ld < MIP360_IPE_LD ? ld : MIP360_IPE_LD;               // zero padding 504..511 (K of the next layer)
  for (int col = 2 * HALF + j; col < pad_to; col += MIP360_N_BASIS) {
    if (BF16) ((__bf16*)enc)[(size_t)row * ld + col] = (__bf16)0.f;
    else ((float*)enc)[(size_t)row * ld + col] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------------------------
// compute_alpha_weights + volumetric_rendering, S <= 64: lane = sample.                       render.py:136-216
// ------------------------------------------------------------------------------------------------------------
struct LevelRay {
  float delta, dd, alpha, trans, w, tmid, acc, logexp, dm, t_first, t_last;
  bool dm_clipped;
};
__device__ __forceinline__ LevelRay level_forward(int S, int lane, float density, const float* __restrict__ td,
                                                  float dnorm, bool opaque) {
  LevelRay r;
  const bool ok = lane < S;
  const float ta = ok ? td[lane] : 0.f, tb = ok ? td[lane + 1] : 0.f;
  r.delta = (tb - ta) * dnorm;
  r.dd = ok ? density * r.delta : 0.f;
  if (opaque && lane == S - 1) r.dd = INFINITY;
  r.alpha = ok ? 1.f - expf(-r.dd) : 0.f;
  const float incl = wave_incl_sum((ok && !(opaque && lane == S - 1)) ? r.dd : 0.f, lane);
  const float excl = incl - ((ok && !(opaque && lane == S - 1)) ? r.dd : 0.f);
  r.trans = expf(-excl);
  r.w = ok ? r.alpha * r.trans : 0.f;
  r.tmid = 0.5f * (ta + tb);
  r.acc = wave_sum(r.w);
  const float lg = ok ? r.w * logf(r.tmid) : 0.f;
  r.logexp = wave_sum(lg) / fmaxf(EPS, r.acc);
  r.t_first = __shfl(ta, 0, 64);
  r.t_last = __shfl(tb, S - 1, 64);
  float e = expf(r.logexp);
  e = isnan(e) ? INFINITY : e;
  r.dm = fminf(fmaxf(e, r.t_first), r.t_last);
  r.dm_clipped = !(e >= r.t_first && e <= r.t_last);
  return r;
}

__global__ __launch_bounds__(256) void render_level_kernel(
    int n, int S, const float* __restrict__ density, const float* __restrict__ rgbs, const float* __restrict__ tdist,
    const float* __restrict__ dirs, int opaque, float bg, float* __restrict__ weights, float* __restrict__ rgb,
    float* __restrict__ acc, float* __restrict__ dmean, float* __restrict__ depth) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * RPB + wave;
  if (ray >= n) return;
  const float dx = dirs[ray * 3], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
  const float dens = lane < S ? density[(size_t)ray * S + lane] : 0.f;
  const LevelRay r = level_forward(S, lane, dens, tdist + (size_t)ray * (S + 1), dnorm, opaque != 0);
  if (lane < S) weights[(size_t)ray * S + lane] = r.w;
  const float bg_w = fmaxf(0.f, 1.f - r.acc);
  if (rgbs && rgb) {
    float c[3] = {0.f, 0.f, 0.f};
    if (lane < S) {
      const float* p = rgbs + ((size_t)ray * S + lane) * 3;
      c[0] = r.w * p[0]; c[1] = r.w * p[1]; c[2] = r.w * p[2];
    }
    const float c0 = wave_sum(c[0]), c1 = wave_sum(c[1]), c2 = wave_sum(c[2]);
    if (lane == 0) { rgb[ray * 3] = c0 + bg_w * bg; rgb[ray * 3 + 1] = c1 + bg_w * bg; rgb[ray * 3 + 2] = c2 + bg_w * bg; }
  }
  float dsum = wave_sum(lane < S ? r.w * r.tmid : 0.f);
  dsum = isnan(dsum) ? INFINITY : dsum;
  if (lane == 0) {
    if (acc) acc[ray] = r.acc;
    if (dmean) dmean[ray] = r.dm;
    if (depth) depth[ray] = fminf(fmaxf(dsum, r.t_first), r.t_last);
  }
}

__global__ __launch_bounds__(256) void render_level_bwd_kernel(
    int n, int S, const float* __restrict__ density, const float* __restrict__ rgbs, const float* __restrict__ tdist,
    const float* __restrict__ dirs, int opaque, float bg, const float* __restrict__ g_w_in,
    const float* __restrict__ g_rgb, const float* __restrict__ g_dm, float* __restrict__ g_density,
    float* __restrict__ g_rgbs) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * RPB + wave;
  if (ray >= n) return;
  const float dx = dirs[ray * 3], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
  const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
  const bool ok = lane < S;
  const float dens = ok ? density[(size_t)ray * S + lane] : 0.f;
  const LevelRay r = level_forward(S, lane, dens, tdist + (size_t)ray * (S + 1), dnorm, opaque != 0);
  float gw = (ok && g_w_in) ? g_w_in[(size_t)ray * S + lane] : 0.f;
  if (g_rgb && rgbs && ok) {
    const float* p = rgbs + ((size_t)ray * S + lane) * 3;
    const float g0 = g_rgb[ray * 3], g1 = g_rgb[ray * 3 + 1], g2 = g_rgb[ray * 3 + 2];
    // rgb = sum_i w_i c_i + max(0, 1 - acc) bg
    const float dbg = (1.f - r.acc > 0.f) ? -bg : 0.f;
    gw += g0 * (p[0] + dbg) + g1 * (p[1] + dbg) + g2 * (p[2] + dbg);
    if (g_rgbs) {
      float* q = g_rgbs + ((size_t)ray * S + lane) * 3;
      q[0] = r.w * g0; q[1] = r.w * g1; q[2] = r.w * g2;
    }
  } else if (g_rgbs && ok) {
    float* q = g_rgbs + ((size_t)ray * S + lane) * 3;
    q[0] = 0.f; q[1] = 0.f; q[2] = 0.f;
  }
  if (g_dm && ok && !r.dm_clipped && r.acc > EPS) {
    // dm = exp(sum_i w_i log tmid_i / acc):  d dm / d w_i = dm (log tmid_i - E) / acc
    gw += g_dm[ray] * r.dm * (logf(r.tmid) - r.logexp) / r.acc;
  }
  // d L / d x_i = g_i (1 - alpha_i) T_i - sum_{k>i} g_k w_k ; x = density * delta
  const float gww = ok ? gw * r.w : 0.f;
  const float suffix = wave_excl_suffix_sum(gww, lane);
  float gx = gw * (1.f - r.alpha) * r.trans - suffix;
  if (opaque && lane == S - 1) gx = 0.f;
  if (ok) g_density[(size_t)ray * S + lane] = gx * r.delta;
}

// ------------------------------------------------------------------------------------------------------------
// losses: per-ray partial sums + gradients (one wave per ray), then a fixed-order reduction by one workgroup.
// workspace [4][n]: data, depth, interlevel, distortion partial sums per ray.
// ------------------------------------------------------------------------------------------------------------
struct LossArgs {
  int n, s_nerf, s_prop, n_prop;
  const float* rgb; const float* rgb_gt; const float* dm; const float* sup;
  const float* sd_nerf; const float* w_nerf;
  const float* sd_prop[4]; const float* w_prop[4];
  int charb; float charb_pad, data_mult; int depth_type; float lambda_depth, depth_weight, inter_mult, dist_mult;
  float prop_depth_weight; const float* dm_prop[4]; float* g_dm_prop[4];
  float* g_rgb; float* g_dm; float* g_w_nerf; float* g_w_prop[4]; float* ws;
};

__global__ __launch_bounds__(256) void losses_ray_kernel(LossArgs a) {
  __shared__ float s_c[RPB][MIP360_MAX_SAMPLES + 2];     // nerf edges
  __shared__ float s_w[RPB][MIP360_MAX_SAMPLES + 2];     // nerf weights
  __shared__ float s_tp[RPB][MIP360_MAX_SAMPLES + 2];    // proposal edges
  __shared__ float s_cy[RPB][MIP360_MAX_SAMPLES + 2];    // cumulative proposal weights
  __shared__ float s_go[RPB][MIP360_MAX_SAMPLES + 2];    // d loss / d w_outer per nerf interval
  __shared__ int s_lo[RPB][MIP360_MAX_SAMPLES + 2], s_hi[RPB][MIP360_MAX_SAMPLES + 2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * RPB + wave;
  if (ray >= a.n) return;
  const int n = a.n, Sn = a.s_nerf, Sp = a.s_prop;
  float* c = s_c[wave]; float* w = s_w[wave]; float* tp = s_tp[wave]; float* cy = s_cy[wave]; float* go = s_go[wave];
  int* lo = s_lo[wave]; int* hi = s_hi[wave];
  // ---- data term (train_utils.py:82-107) and depth term (:108-119) ----
  float data = 0.f;
  if (lane < 3) {
    const float resid = a.rgb[ray * 3 + lane] - a.rgb_gt[ray * 3 + lane];
    const float denom = 3.f * (float)n;
    if (a.charb) {
      const float v = sqrtf(resid * resid + a.charb_pad * a.charb_pad);
      data = v;
      a.g_rgb[ray * 3 + lane] = a.data_mult * resid / v / denom;
    } else {
      data = resid * resid;
      a.g_rgb[ray * 3 + lane] = a.data_mult * 2.f * resid / denom;
    }
  }
  data = wave_sum(data);
  float dep = 0.f;
  if (lane == 0) {
    float g = 0.f;
    if (a.depth_type) {
      const float sup = a.sup[ray];
      const float m = sup > 0.f ? 1.f : 0.f;
      const float diff = m * a.dm[ray] - m * sup;
      // train_utils.py:139-143 + :268-269: data_loss_mult * lambda * dep[-1] (inside `data`) + lambda * dep[-1] (inside
      // stats['loss_disp_mse'], NOT scaled by data_loss_mult): depth_weight counts the two appearances, data_mult scales one
      const float k = (a.data_mult + (a.depth_weight - 1.f)) * a.lambda_depth / (float)n;
      if (a.depth_type == 1) { dep = diff * diff; g = k * 2.f * diff * m; }
      else { dep = fabsf(diff); g = k * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * m; }
    }
    a.g_dm[ray] = g;
    // the proposal levels' distance_mean enters only through stats['loss_disp_mse'] (train_utils.py:143, :268-269)
    for (int k = 0; k < a.n_prop; ++k) {
      if (!a.dm_prop[k]) continue;
      float dk = 0.f, gk = 0.f;
      if (a.depth_type) {
        const float sup = a.sup[ray];
        const float m = sup > 0.f ? 1.f : 0.f;
        const float diff = m * a.dm_prop[k][ray] - m * sup;
        const float kk = a.lambda_depth * a.prop_depth_weight / (float)n;
        if (a.depth_type == 1) { dk = diff * diff; gk = kk * 2.f * diff * m; }
        else { dk = fabsf(diff); gk = kk * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * m; }
      }
      a.g_dm_prop[k][ray] = gk;
      a.ws[(4 + k) * n + ray] = dk;
    }
  }
  // ---- distortion (stepfun.py:273-283) ----
  for (int i = lane; i <= Sn; i += 64) c[i] = a.sd_nerf[(size_t)ray * (Sn + 1) + i];
  for (int i = lane; i < Sn; i += 64) w[i] = a.w_nerf[(size_t)ray * Sn + i];
  __builtin_amdgcn_wave_barrier();
  float dist = 0.f;
  if (lane < Sn) {
    const float ut = (c[lane + 1] + c[lane]) / 2.f, wi = w[lane], dt = c[lane + 1] - c[lane];
    float inner = 0.f;
    for (int j = 0; j < Sn; ++j) inner += w[j] * fabsf(ut - (c[j + 1] + c[j]) / 2.f);
    dist = wi * inner + wi * wi * dt / 3.f;
    a.g_w_nerf[(size_t)ray * Sn + lane] = a.dist_mult * (2.f * inner + 2.f * wi * dt / 3.f) / (float)n;
  }
  dist = wave_sum(dist);
  // ---- interlevel (stepfun.py:64-87, train_utils.py:149-160): gradient to the proposal weights only ----
  float inter = 0.f;
  for (int k = 0; k < a.n_prop; ++k) {
    for (int i = lane; i <= Sp; i += 64) tp[i] = a.sd_prop[k][(size_t)ray * (Sp + 1) + i];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      float run = 0.f;
      cy[0] = 0.f;
      for (int i = 0; i < Sp; ++i) { run += a.w_prop[k][(size_t)ray * Sp + i]; cy[i + 1] = run; }
    }
    // searchsorted(t_env = tp, v = c): lo = last edge <= v (else 0), hi = first edge > v (else last)
    for (int i = lane; i <= Sn; i += 64) {
      const float v = c[i];
      int l = 0, h = Sp;
      bool found = false;
      for (int e = 0; e <= Sp; ++e) {
        const bool ge = v >= tp[e];
        l = ge ? e : l;
        if (!ge && !found) { h = e; found = true; }
      }
      lo[i] = l; hi[i] = h;
    }
    __builtin_amdgcn_wave_barrier();
    float term = 0.f;
    if (lane < Sn) {
      const float w_outer = cy[hi[lane + 1]] - cy[lo[lane]];
      const float ex = fmaxf(0.f, w[lane] - w_outer);
      term = ex * ex / (w[lane] + EPS);
      go[lane] = -2.f * ex / (w[lane] + EPS);
    }
    inter += wave_sum(term);
    __builtin_amdgcn_wave_barrier();
    const float scale = a.inter_mult / ((float)n * (float)Sn);
    for (int j = lane; j < Sp; j += 64) {
      float g = 0.f;
      for (int i = 0; i < Sn; ++i) g += (j >= lo[i] && j < hi[i + 1]) ? go[i] : 0.f;
      a.g_w_prop[k][(size_t)ray * Sp + j] = scale * g;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) {
    a.ws[ray] = data; a.ws[n + ray] = dep; a.ws[2 * n + ray] = inter; a.ws[3 * n + ray] = dist;
  }
}

__global__ __launch_bounds__(1024) void losses_reduce_kernel(int n, int s_nerf, const float* __restrict__ ws, float data_mult,
                                                             int depth_type, float lambda_depth, float depth_weight,
                                                             float inter_mult, float dist_mult, int n_prop_dm,
                                                             float prop_depth_weight, float* __restrict__ scalars) {
  __shared__ double sh[5][1024];
  double p[5] = {0, 0, 0, 0, 0};
  for (int r = threadIdx.x; r < n; r += blockDim.x) {
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] += (double)ws[q * n + r];
    for (int k = 0; k < n_prop_dm; ++k) p[4] += (double)ws[(4 + k) * n + r];
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) sh[q][threadIdx.x] = p[q];
  __syncthreads();
  for (int d = blockDim.x >> 1; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d)
#pragma unroll
      for (int q = 0; q < 5; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float data = (float)(sh[0][0] / (3.0 * n));
    const float dep = depth_type ? (float)(sh[1][0] / n) : 0.f;
    const float inter = inter_mult * (float)(sh[2][0] / ((double)n * s_nerf));
    const float dist = dist_mult * (float)(sh[3][0] / n);
    const float dep_prop = depth_type ? (float)(sh[4][0] / n) : 0.f;
    scalars[0] = data_mult * (data + lambda_depth * dep) + lambda_depth * (depth_weight - 1.f) * dep +
                 lambda_depth * prop_depth_weight * dep_prop + inter + dist;
    scalars[1] = data; scalars[2] = dep; scalars[3] = inter; scalars[4] = dist; scalars[5] = dep_prop;
  }
}

// pos_enc(viewdirs, 0, 4, append_identity=True) (coord.py:138-147, models.py:395-399) broadcast over the samples of a
// ray into columns [col0, col0 + 27) of a bf16 [n*S, ld] tensor; columns up to col0 + width are zero-filled.
__global__ void dir_encode_kernel(int n, int S, const float* __restrict__ viewdirs, __bf16* __restrict__ out, int ld,
                                  int col0, int width) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = idx / width;
  const int c = (int)(idx - row * width);
  if (row >= (int64_t)n * S) return;
  const int ray = (int)(row / S);
  float v = 0.f;
  if (c < 3) v = viewdirs[ray * 3 + c];
  else if (c < 27) {
    // four_feat = sin(concat([scaled_x, scaled_x + pi/2])): scaled_x index = k*3 + d
    const int q = c - 3, half = q / 12, r = q - half * 12, k = r / 3, d = r - k * 3;
    const float sx = viewdirs[ray * 3 + d] * (float)(1 << k);
    v = sinf(half ? sx + 1.5707963267948966f : sx);
  }
  out[(size_t)row * ld + col0 + c] = (__bf16)v;
}

// ------------------------------------------------------------------------------------------------------------
// depth_loss.depth_loss for 'kl' / 'urf' of ONE level (internal/depth_loss.py:5-102), value + gradients.
//   steps = mid-points of tdist, lengths = interval widths * |dirs|;
//   kl : l[r,s] = -log(w + 1e-7) * exp(-(steps - gt[r])^2 / (2 sigma)) * lengths
//   urf: near[r,s] = [gt - sigma <= steps <= gt + sigma] * (w - N(steps - gt; 0, sigma / 3))^2,
//        empty[r,s] = [steps < gt - sigma] * w^2, expected[r] = (gt - pred)^2
// Upstream reduces with `.sum(-2)` -- over the RAY axis -- and multiplies the [S] result by the [n] mask, which only
// broadcasts for n == S (column s meets ray s's mask / expected term) or n == 1 (every column meets ray 0's); the C ABI
// rejects other shapes like JAX does.  value = mean over the S columns.  One workgroup: thread -> column(s), a sequential
// float32 sum over the rays of a column (deterministic).  g_w [n,S] and g_dm [n] are ACCUMULATED (+= scale * gradient).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void depth_klurf_kernel(int type, int n, int S, const float* __restrict__ w,
                                                          const float* __restrict__ td, const float* __restrict__ sup,
                                                          const float* __restrict__ dm, const float* __restrict__ dirs,
                                                          float sigma, float scale, float* __restrict__ out,
                                                          float* __restrict__ g_w, float* __restrict__ g_dm,
                                                          float* __restrict__ accum) {
  __shared__ double sh[256];
  double part = 0.0;
  const float inv_S = 1.f / (float)S;
  const float usig = sigma / 3.f;                                    // URF_SIGMA_SCALE_FACTOR
  const float log_norm = logf(usig) + logf(sqrtf(2.f * 3.14159265358979323846f));
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    const int ri = n == 1 ? 0 : s;                                   // the ray whose mask / expected term meets column s
    const float mk = sup[ri] > 0.f ? 1.f : 0.f;
    float col = 0.f;
    for (int r = 0; r < n; ++r) {
      const float t0 = td[(size_t)r * (S + 1) + s], t1 = td[(size_t)r * (S + 1) + s + 1];
      const float step = 0.5f * (t0 + t1), gt = sup[r], wi = w[(size_t)r * S + s];
      float g = 0.f;
      if (type == 3) {
        const float dx = dirs[r * 3], dy = dirs[r * 3 + 1], dz = dirs[r * 3 + 2];
        const float len = (t1 - t0) * sqrtf(dx * dx + dy * dy + dz * dz);
        const float d = step - gt;
        const float e = expf(-(d * d) / (2.f * sigma)) * len;
        col += -logf(wi + 1e-7f) * e;
        g = -e / (wi + 1e-7f);
      } else {
        const float d = step - gt;
        const bool near = step <= gt + sigma && step >= gt - sigma, empty = step < gt - sigma;
        const float pdf = expf(-(d * d) / (2.f * usig * usig) - log_norm);
        if (near) { col += (wi - pdf) * (wi - pdf); g += 2.f * (wi - pdf); }
        if (empty) { col += wi * wi; g += 2.f * wi; }
      }
      if (g_w) g_w[(size_t)r * S + s] += scale * inv_S * mk * g;
    }
    if (type == 4) {
      const float diff = sup[ri] - dm[ri];
      col += diff * diff;
      if (g_dm && n != 1) g_dm[ri] += scale * inv_S * mk * (-2.f * diff);
    }
    part += (double)(col * mk);
  }
  sh[threadIdx.x] = part;
  __syncthreads();
  for (int d = blockDim.x >> 1; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float value = (float)(sh[0] / (double)S);
    out[0] = value;
    if (type == 4 && n == 1 && g_dm) g_dm[0] += scale * (sup[0] > 0.f ? 1.f : 0.f) * (-2.f * (sup[0] - dm[0]));   // S columns x 1/S
    if (accum) { accum[0] += scale * value; accum[1] += value; }
  }
}

}  // namespace mip360

using namespace mip360;

void mip360_launch_dir_encode(hipStream_t st, int n, int S, const float* viewdirs, void* out, int ld, int col0, int width) {
  const int64_t tot = (int64_t)n * S * width;
  hipLaunchKernelGGL(dir_encode_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, n, S, viewdirs, (__bf16*)out, ld,
                     col0, width);
}

void mip360_launch_resample(hipStream_t st, int n, int m_in, const float* sd, const float* w, float dil, float anneal,
                            float pad, int ns, const float* jit, float s_near, float s_far, const float* tn,
                            const float* tf, float* sd_out, float* td_out) {
  hipLaunchKernelGGL(resample_kernel, dim3((n + RPB - 1) / RPB), dim3(256), 0, st, n, m_in, sd, w, dil, anneal, pad, ns, jit,
                     s_near, s_far, tn, tf, sd_out, td_out);
}
void mip360_launch_cast_encode(hipStream_t st, int n, int S, const float* td, const float* o, const float* d,
                               const float* radii, const float* basis_t, void* enc, int bf16, int ld) {
  const int64_t rows = (int64_t)n * S;
  const int64_t waves = (rows + 2) / 3;
  const unsigned grid = (unsigned)((waves + 3) / 4);
  if (bf16 == 2) hipLaunchKernelGGL(cast_encode_kernel<2>, dim3((unsigned)(((int64_t)n * S + 31) / 32)), dim3(704), 0, st, n, S, td, o, d, radii, basis_t, enc, ld);
  else if (bf16) hipLaunchKernelGGL(cast_encode_kernel<1>, dim3(grid), dim3(256), 0, st, n, S, td, o, d, radii, basis_t, enc, ld);
  else hipLaunchKernelGGL(cast_encode_kernel<0>, dim3(grid), dim3(256), 0, st, n, S, td, o, d, radii, basis_t, enc, ld);
}
void mip360_launch_render(hipStream_t st, int n, int S, const float* density, const float* rgbs, const float* td,
                          const float* dirs, int opaque, float bg, float* w, float* rgb, float* acc, float* dm, float* depth) {
  hipLaunchKernelGGL(render_level_kernel, dim3((n + RPB - 1) / RPB), dim3(256), 0, st, n, S, density, rgbs, td, dirs, opaque,
                     bg, w, rgb, acc, dm, depth);
}
void mip360_launch_render_bwd(hipStream_t st, int n, int S, const float* density, const float* rgbs, const float* td,
                              const float* dirs, int opaque, float bg, const float* g_w, const float* g_rgb,
                              const float* g_dm, float* g_density, float* g_rgbs) {
  hipLaunchKernelGGL(render_level_bwd_kernel, dim3((n + RPB - 1) / RPB), dim3(256), 0, st, n, S, density, rgbs, td, dirs,
                     opaque, bg, g_w, g_rgb, g_dm, g_density, g_rgbs);
}
void mip360_launch_losses(hipStream_t st, int n, int s_nerf, int s_prop, int n_prop, const float* rgb, const float* rgb_gt,
                          const float* dm, const float* sup, const float* sd_nerf, const float* w_nerf,
                          const float* const* sd_prop, const float* const* w_prop, int charb, float charb_pad,
                          float data_mult, int depth_type, float lambda_depth, float depth_weight, float inter_mult,
                          float dist_mult, float* scalars, float* g_rgb, float* g_dm, float* g_w_nerf,
                          float* const* g_w_prop, float* ws, float prop_depth_weight, const float* const* dm_prop,
                          float* const* g_dm_prop) {
  LossArgs a{};
  a.prop_depth_weight = prop_depth_weight;
  int n_dm = 0;
  for (int k = 0; k < n_prop; ++k) {
    a.dm_prop[k] = dm_prop ? dm_prop[k] : nullptr;
    a.g_dm_prop[k] = g_dm_prop ? g_dm_prop[k] : nullptr;
    if (a.dm_prop[k]) n_dm = k + 1;
  }
  a.n = n; a.s_nerf = s_nerf; a.s_prop = s_prop; a.n_prop = n_prop;
  a.rgb = rgb; a.rgb_gt = rgb_gt; a.dm = dm; a.sup = sup; a.sd_nerf = sd_nerf; a.w_nerf = w_nerf;
  for (int k = 0; k < n_prop; ++k) { a.sd_prop[k] = sd_prop[k]; a.w_prop[k] = w_prop[k]; a.g_w_prop[k] = g_w_prop[k]; }
  a.charb = charb; a.charb_pad = charb_pad; a.data_mult = data_mult; a.depth_type = depth_type;
  a.lambda_depth = lambda_depth; a.depth_weight = depth_weight; a.inter_mult = inter_mult; a.dist_mult = dist_mult;
  a.g_rgb = g_rgb; a.g_dm = g_dm; a.g_w_nerf = g_w_nerf; a.ws = ws;
  hipLaunchKernelGGL(losses_ray_kernel, dim3((n + RPB - 1) / RPB), dim3(256), 0, st, a);
  hipLaunchKernelGGL(losses_reduce_kernel, dim3(1), dim3(1024), 0, st, n, s_nerf, ws, data_mult, depth_type, lambda_depth,
                     depth_weight, inter_mult, dist_mult, n_dm, prop_depth_weight, scalars);
}
void mip360_launch_depth_klurf(hipStream_t st, int type, int n, int S, const float* w, const float* td, const float* sup,
                               const float* dm, const float* dirs, float sigma, float scale, float* out, float* g_w,
                               float* g_dm, float* accum) {
  hipLaunchKernelGGL(depth_klurf_kernel, dim3(1), dim3(256), 0, st, type, n, S, w, td, sup, dm, dirs, sigma, scale, out, g_w,
                     g_dm, accum);
}
