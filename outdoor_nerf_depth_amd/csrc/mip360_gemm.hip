// Dense layer on the gfx950 matrix cores for the MipNeRF-360 MLPs (PropMLP 4 x 256, NerfMLP 8 x 1024):
//   C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]),  bf16 operands, float32 accumulation (v_mfma_f32_32x32x16_bf16).
// Replaces flax nn.Dense + nn.relu of MLP.__call__ (nerf-methods/mipnerf360/internal/models.py:436-606).
//
// The 1024-wide NerfMLP does not fit the NeRF++ "whole network in registers" design (32 samples x 1024 features of
// float32 accumulators = 512 VGPRs per lane), so its layers run as tiled GEMMs whose [rows, 1024] bf16 activations
// round-trip through L2 / Infinity Cache (268 MB per layer at 4096 rays x 32 samples).
//
// Tiling: 256 threads = 2 x 2 waves, workgroup tile 128 x 128, K step 32, double-buffered LDS; a wave owns a
// 64 x 64 sub-tile = 2 x 2 MFMA blocks.  Both operands are K-contiguous, so an MFMA fragment is one 16-byte LDS
// read per lane (row l & 31, k offset 8 * (l >> 5)); LDS rows are padded to 80 bytes.  The next K tile is fetched
// into registers while the current one is multiplied (one __syncthreads per K step).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mip360 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, LDS_ROW = 40;      // elements; 40 * 2 B = 80 B row stride

template <int ACT>
__global__ __launch_bounds__(256) void linear_bf16_kernel(int M, int N, int K, const __bf16* __restrict__ A, int lda,
                                                          const __bf16* __restrict__ W, int ldw,
                                                          const float* __restrict__ bias, __bf16* __restrict__ C16, int ldc,
                                                          float* __restrict__ C32, int ldc32, float act_param,
                                                          const __bf16* __restrict__ aux, int ldaux) {
  __shared__ __attribute__((aligned(16))) __bf16 sA[2][BM * LDS_ROW];
  __shared__ __attribute__((aligned(16))) __bf16 sB[2][BN * LDS_ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: consecutive workgroups (which land on different XCDs) take different N tiles of the
  // same M tile, so an A tile is read by all XCDs at about the same time and W (small) stays in every L2
  const int tiles_n = (N + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  // global -> LDS: 512 16-byte chunks per operand tile, 2 per thread: chunk c -> row c >> 2, k offset (c & 3) * 8
  uint4 ra[2], rb[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = tid + q * 256, row = c >> 2, kc = (c & 3) * 8;
      const int m = m0 + row, n = n0 + row;
      ra[q] = m < M ? *(const uint4*)(A + (size_t)m * lda + k0 + kc) : make_uint4(0, 0, 0, 0);
      rb[q] = n < N ? *(const uint4*)(W + (size_t)n * ldw + k0 + kc) : make_uint4(0, 0, 0, 0);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = tid + q * 256, row = c >> 2, kc = (c & 3) * 8;
      *(uint4*)(&sA[buf][row * LDS_ROW + kc]) = ra[q];
      *(uint4*)(&sB[buf][row * LDS_ROW + kc]) = rb[q];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / BK;
  fetch(0);
  stash(0);
  __syncthreads();
  const int frow = lane & 31, fk = (lane >> 5) * 8;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) fetch((kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *(const bf16x8*)(&sA[buf][(wm * 64 + i * 32 + frow) * LDS_ROW + ks * 16 + fk]);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *(const bf16x8*)(&sB[buf][(wn * 64 + j * 32 + frow) * LDS_ROW + ks * 16 + fk]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) stash(buf ^ 1);
    __syncthreads();
  }
  // epilogue: lane (j = lane & 31, hi = lane >> 5), register r -> row (r & 3) + 8 (r >> 2) + 4 hi, column j
  const int hi = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + frow;
    if (n >= N) continue;
    const float b = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (m >= M) continue;
        float v = acc[i][j][r] + b;
        if (ACT == 1) v = fmaxf(v, 0.f);
        if (ACT == 2) { const float x = v + act_param; v = fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }   // softplus(raw + density_bias)
        if (ACT == 4) v = (float)aux[(size_t)m * ldaux + n] > 0.f ? v : 0.f;                             // ReLU mask of the layer's saved output
        if (ACT == 3) v = (1.f / (1.f + expf(-v))) * (1.f + 2.f * act_param) - act_param;                // padded sigmoid
        if (C16) C16[(size_t)m * ldc + n] = (__bf16)v;
        if (C32) C32[(size_t)m * ldc32 + n] = v;
      }
    }
  }
}

}  // namespace mip360

void mip360_launch_linear(hipStream_t st, int M, int N, int K, const void* A, int lda, const void* W, int ldw, const float* bias,
                          int act, float act_param, void* C16, int ldc, float* C32, int ldc32, const void* aux, int ldaux) {
  using namespace mip360;
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
#define MIP360_LAUNCH(ACT) hipLaunchKernelGGL(linear_bf16_kernel<ACT>, dim3(tiles), dim3(256), 0, st, M, N, K, (const __bf16*)A, \
                                              lda, (const __bf16*)W, ldw, bias, (__bf16*)C16, ldc, C32, ldc32, act_param, (const __bf16*)aux, ldaux)
  if (act == 1) MIP360_LAUNCH(1);
  else if (act == 2) MIP360_LAUNCH(2);
  else if (act == 3) MIP360_LAUNCH(3);
  else if (act == 4) MIP360_LAUNCH(4);
  else MIP360_LAUNCH(0);
#undef MIP360_LAUNCH
}
