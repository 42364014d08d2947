// The view branch of the NerfMLP (models.py:560-606 with configs/360.gin: bottleneck 256 -> [bottleneck | pos_enc(viewdirs, 0, 4)]
// -> Dense(128) + ReLU -> Dense(3) -> padded sigmoid) as ONE launch per level, forward: what was from_fm + dir_encode + two
// row-major GEMMs over 131 072 x 288 / 128 operands (0.12 ms for 1 % of the step's FLOP).
//
// Same scheme as mip360_prop.hip: a wave owns 32 rows, lane (row, hi) of v_mfma_f32_32x32x16_bf16 carries the sample as the B
// operand.  The 16 bottleneck fragments of the wave's rows are 1-KiB blocks of the fm tensor mip360_linear_fm (act 0) wrote,
// loaded 16 bytes per lane; the two direction fragments come from a per-RAY table [rays, 32] that mip360_dir_encode (S = 1)
// writes once per step (computing them in the kernel -- 16 libm sines per lane -- cost registers and time for values that are
// the same for all samples of a ray).  Both weight matrices (72 + 8 KiB as fm blocks) stay in LDS for the whole launch;
// workgroups are persistent over 256-row tiles and issue the next tile's operand loads before the current tile's MFMAs.  For the
// backward pass the kernel writes what the row-major launches wrote -- view_in [rows, 288] and h [rows, 128] in bf16, 8 bytes
// per lane and half fragment (the two lanes of a row complete 16-byte pieces; L2 assembles the rows) -- or nothing but rgb
// (inference).
#include "probe_env.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <type_traits>

namespace mip360view {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int NW = 8;
constexpr int BOTT = 256, DIR_LD = 32, DIR_DIM = 27, VIEW_W = 128, K1 = BOTT + DIR_LD;      // 288
constexpr int NKC1 = K1 / 16, NOB1 = VIEW_W / 32, NKC2 = VIEW_W / 16;                        // 18, 4, 8
constexpr int LDS_W1 = 0, LDS_W2 = NOB1 * NKC1 * 1024, LDS_B = LDS_W2 + NKC2 * 1024, LDS_TOTAL = LDS_B + (VIEW_W + 4) * 4;

struct Args {
  int rows, S;                                          // rows a multiple of 256; row / S = ray
  const char* bott;                                     // fm [rows, 256]
  const uint16_t* dir_table;                            // bf16 [rows / S, 32]: pos_enc(viewdirs, 0, 4) + identity, zero padded
  const char* w1; int w1_bpr;                           // fm [128, ld >= 288]
  const char* w2; int w2_bpr;                           // fm [32 (3 live rows), ld >= 128]
  const float* b1; const float* b2;                     // [128], [3]
  float rgb_padding;
  uint16_t* view_in; int ld_view;                       // row-major bf16 [rows, ld_view >= 288] or nullptr
  uint16_t* h; int ld_h;                                // row-major bf16 [rows, ld_h >= 128] or nullptr
  float* rgb;                                           // [rows, 3]
};

extern __shared__ __attribute__((aligned(16))) char smem[];

__device__ __forceinline__ uint32_t unit_of(int row, int hi) { return 8u * (row >> 2) + 4u * (hi ^ (row >> 4)) + (row & 3); }

__device__ __forceinline__ void glds_frag(const char* sbase, uint32_t voff, uint32_t lds_abs) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_abs);
  const uint64_t b = (uint64_t)(uintptr_t)sbase;
  const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
}

// N weight-fragment reads and their N MFMAs: four reads in flight ahead of the MFMA that consumes them, nothing hoisted further
// (left alone, the scheduler issues all 72 reads of the first layer up front: 256 VGPRs and 1.1 KB of scratch)
template <int N>
__device__ __forceinline__ void shape_schedule(std::integral_constant<int, N>) {
  constexpr int D = 4;
#pragma unroll
  for (int i = 0; i < D; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
  for (int i = 0; i < N - D; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  }
#pragma unroll
  for (int i = 0; i < D; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(NW * 64, 1) void view_branch_fwd_kernel(const Args a) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = lane & 31, hi = lane >> 5;
  const uint32_t u16 = unit_of(row, hi) * 16u;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // ---- both weight matrices and the biases to LDS, once
  for (int f = wave; f < NOB1 * NKC1 + NKC2; f += NW) {
    const bool first = f < NOB1 * NKC1;
    const int g = first ? f : f - NOB1 * NKC1;
    const char* src = first ? a.w1 + ((size_t)(g / NKC1) * a.w1_bpr + (g % NKC1)) * 1024 : a.w2 + (size_t)g * 1024;
    glds_frag(src, (uint32_t)lane * 16u, lds0 + (first ? LDS_W1 : LDS_W2) + g * 1024);
  }
  if (threadIdx.x < VIEW_W) ((float*)(smem + LDS_B))[threadIdx.x] = a.b1[threadIdx.x];
  if (threadIdx.x < 3) ((float*)(smem + LDS_B))[VIEW_W + threadIdx.x] = a.b2[threadIdx.x];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int tiles = a.rows / (NW * 32);
  // B operand of a tile: 16 bottleneck fragments from global memory + the ray's 32 direction columns (the table mip360_dir_encode
  // with S = 1 wrote): this lane's two half-units of each of the 2 fragments
  auto load_tile = [&](int tile, u32x4 (&in)[NKC1]) {
    const size_t rb = (size_t)tile * NW + wave;
    const char* brow = a.bott + rb * (BOTT / 16) * 1024 + u16;
#pragma unroll
    for (int c = 0; c < BOTT / 16; ++c) in[c] = *(const u32x4*)(brow + (size_t)c * 1024);
    const uint16_t* drow = a.dir_table + ((rb * 32 + row) / (size_t)a.S) * DIR_LD + 4 * hi;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const uint2 lo = *(const uint2*)(drow + 16 * c), hi8 = *(const uint2*)(drow + 16 * c + 8);
      in[BOTT / 16 + c] = (u32x4){lo.x, lo.y, hi8.x, hi8.y};
    }
  };
  // one tile from its operand registers; the NEXT tile's operand loads are issued first, into the other register set
  auto process = [&](int tile, u32x4 (&in)[NKC1], u32x4 (&nxt)[NKC1]) {
    const size_t rb = (size_t)tile * NW + wave;
    const size_t grow = rb * 32 + row;
    if (tile + (int)gridDim.x < tiles) load_tile(tile + (int)gridDim.x, nxt);
    __builtin_amdgcn_sched_barrier(0);
    if (a.view_in) {
      uint16_t* vrow = a.view_in + grow * (size_t)a.ld_view + 4 * hi;
#pragma unroll
      for (int c = 0; c < NKC1; ++c) {
        *(uint2*)(vrow + 16 * c) = make_uint2(in[c][0], in[c][1]);
        *(uint2*)(vrow + 16 * c + 8) = make_uint2(in[c][2], in[c][3]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- Dense(128) + ReLU
    f32x16 acc[NOB1];
#pragma unroll
    for (int ob = 0; ob < NOB1; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *(const float4*)(smem + LDS_B + (ob * 32 + 8 * q + 4 * hi) * 4);
        acc[ob][4 * q] = v.x; acc[ob][4 * q + 1] = v.y; acc[ob][4 * q + 2] = v.z; acc[ob][4 * q + 3] = v.w;
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kc = 0; kc < NKC1; ++kc)
#pragma unroll
      for (int ob = 0; ob < NOB1; ++ob)
        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(smem + LDS_W1 + (ob * NKC1 + kc) * 1024 + u16),
                                                          __builtin_bit_cast(bf16x8, in[kc]), acc[ob], 0, 0, 0);
    shape_schedule(std::integral_constant<int, NKC1 * NOB1>{});
    u32x4 hf[NKC2];
#pragma unroll
    for (int ob = 0; ob < NOB1; ++ob)
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const f32x2 f = {acc[ob][2 * p], acc[ob][2 * p + 1]};
        uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
        w = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), (s16x2){0, 0}));
        hf[2 * ob + (p >> 2)][p & 3] = w;
      }
    if (a.h) {
      uint16_t* hrow = a.h + grow * (size_t)a.ld_h + 4 * hi;
#pragma unroll
      for (int c = 0; c < NKC2; ++c) {
        *(uint2*)(hrow + 16 * c) = make_uint2(hf[c][0], hf[c][1]);
        *(uint2*)(hrow + 16 * c + 8) = make_uint2(hf[c][2], hf[c][3]);
      }
    }
    // ---- Dense(3) + padded sigmoid: out-block 0, features 0 .. 2 = registers 0 .. 2 of the hi = 0 lanes
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    if (hi == 0) {
      o[0] = ((const float*)(smem + LDS_B))[VIEW_W];
      o[1] = ((const float*)(smem + LDS_B))[VIEW_W + 1];
      o[2] = ((const float*)(smem + LDS_B))[VIEW_W + 2];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kc = 0; kc < NKC2; ++kc)
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(smem + LDS_W2 + kc * 1024 + u16), __builtin_bit_cast(bf16x8, hf[kc]), o, 0, 0, 0);
    shape_schedule(std::integral_constant<int, NKC2>{});
    if (hi == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        a.rgb[grow * 3 + c] = (1.f / (1.f + expf(-o[c]))) * (1.f + 2.f * a.rgb_padding) - a.rgb_padding;
    }
  };
  u32x4 in_a[NKC1], in_b[NKC1];
  if ((int)blockIdx.x < tiles) load_tile(blockIdx.x, in_a);
  for (int tile = blockIdx.x; tile < tiles; tile += 2 * gridDim.x) {
    process(tile, in_a, in_b);
    if (tile + (int)gridDim.x < tiles) process(tile + gridDim.x, in_b, in_a);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The heads' backward up to the bottleneck, as one launch: what was mip360_head_backward + mip360_linear_bf16 (act 4: d_hz =
// relu'(h) . (d_pre W_rgb)) + mip360_linear_bf16 (d_bott = d_hz W_view[:, :256]) + mip360_to_fm.  Per row: d raw density and the
// three d pre-sigmoid values (head_backward_kernel's arithmetic), d_hz through 8 MFMAs (K = 32, the three live columns in one k
// step), masked by the saved h, d_bott through 64 MFMAs; both backward operand copies (8 + 64 KiB of fm blocks) stay in LDS.
// Written: d_pre [rows, 32] and d_hz [rows, 128] row-major (the operands of the two row-major weight-gradient launches) and
// heads [rows, 320] fm = [d_bott (256) | d_raw | 0 ...] -- the operand of the heads' weight gradients and of the trunk's first dX.
constexpr int HEAD_K = BOTT + 64;
constexpr int BW_LDS_W3 = 0, BW_LDS_W2 = NOB1 * 2 * 1024, BW_LDS_TOTAL = BW_LDS_W2 + (BOTT / 32) * NKC2 * 1024;   // 8 + 64 KiB
struct BwdArgs {
  int rows;
  const float* density; const float* g_density; const float* rgb; const float* g_rgb; float pad;
  const uint16_t* h; int ld_h;                          // saved ReLU output, row-major [rows, ld_h >= 128]
  const char* wb3; int wb3_bpr;                         // fm [128, ld >= 32]: element (j, c) = W_rgb[j][c]
  const char* wb2; int wb2_bpr;                         // fm [256, ld >= 128]: element (i, j) = W_view[i][j]
  uint16_t* d_pre;                                      // row-major [rows, 32]
  uint16_t* d_hz; int ld_dhz;                           // row-major [rows, ld >= 128]
  char* heads;                                          // fm [rows, 320]
};

__global__ __launch_bounds__(NW * 64, 1) void view_branch_bwd_kernel(const BwdArgs a) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = lane & 31, hi = lane >> 5;
  const uint32_t u16 = unit_of(row, hi) * 16u;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  for (int f = wave; f < NOB1 * 2 + (BOTT / 32) * NKC2; f += NW) {
    const bool first = f < NOB1 * 2;
    const int g = first ? f : f - NOB1 * 2;
    const char* src = first ? a.wb3 + ((size_t)(g / 2) * a.wb3_bpr + (g % 2)) * 1024 : a.wb2 + ((size_t)(g / NKC2) * a.wb2_bpr + (g % NKC2)) * 1024;
    glds_frag(src, (uint32_t)lane * 16u, lds0 + (first ? BW_LDS_W3 : BW_LDS_W2) + g * 1024);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int tiles = a.rows / (NW * 32);
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const size_t rb = (size_t)tile * NW + wave;
    const size_t grow = rb * 32 + row;
    // ---- per row: d raw density, d pre-sigmoid (head_backward_kernel)
    const float draw = a.g_density[grow] * (1.f - expf(-a.density[grow]));
    float dp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float sg = (a.rgb[grow * 3 + c] + a.pad) / (1.f + 2.f * a.pad);
      dp[c] = a.g_rgb[grow * 3 + c] * (1.f + 2.f * a.pad) * sg * (1.f - sg);
    }
    const f32x2 p01 = {dp[0], dp[1]}, p2z = {dp[2], 0.f};
    const uint32_t w01 = __builtin_bit_cast(uint32_t, __builtin_convertvector(p01, bf16x2));
    const uint32_t w2z = __builtin_bit_cast(uint32_t, __builtin_convertvector(p2z, bf16x2));
    // B operand of the first GEMM: columns 0 .. 2 of the 32 live in element pairs 0, 1 of the hi = 0 lanes' first fragment
    u32x4 dpf[2];
    dpf[0] = hi == 0 ? (u32x4){w01, w2z, 0u, 0u} : (u32x4){0u, 0u, 0u, 0u};
    dpf[1] = (u32x4){0u, 0u, 0u, 0u};
    if (hi == 0) {
      uint4* o = (uint4*)(a.d_pre + grow * 32);
      o[0] = make_uint4(w01, w2z, 0u, 0u);
      o[1] = make_uint4(0u, 0u, 0u, 0u); o[2] = o[1]; o[3] = o[1];
    }
    // saved h of this lane's half-units (for the mask)
    uint2 hm[NKC2][2];
    {
      const uint16_t* hrow = a.h + grow * (size_t)a.ld_h + 4 * hi;
#pragma unroll
      for (int c = 0; c < NKC2; ++c) { hm[c][0] = *(const uint2*)(hrow + 16 * c); hm[c][1] = *(const uint2*)(hrow + 16 * c + 8); }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- d_hz = relu'(h) . (d_pre W_rgb): 4 out-blocks x 2 k steps
    f32x16 acc[NOB1];
#pragma unroll
    for (int ob = 0; ob < NOB1; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ob][r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int ob = 0; ob < NOB1; ++ob)
        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(smem + BW_LDS_W3 + (ob * 2 + kc) * 1024 + u16),
                                                          __builtin_bit_cast(bf16x8, dpf[kc]), acc[ob], 0, 0, 0);
    shape_schedule(std::integral_constant<int, 2 * NOB1>{});
    u32x4 dz[NKC2];
#pragma unroll
    for (int ob = 0; ob < NOB1; ++ob)
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const f32x2 f = {acc[ob][2 * p], acc[ob][2 * p + 1]};
        uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
        // keep a half where the saved activation (a non-negative bf16) is > 0
        const int c = 2 * ob + (p >> 2), q = p & 3;
        const uint32_t hv = q < 2 ? (q == 0 ? hm[c][0].x : hm[c][0].y) : (q == 2 ? hm[c][1].x : hm[c][1].y);
        const uint32_t keep = ((hv & 0x7FFFu) ? 0xFFFFu : 0u) | ((hv & 0x7FFF0000u) ? 0xFFFF0000u : 0u);
        dz[c][q] = w & keep;
      }
    {
      uint16_t* zrow = a.d_hz + grow * (size_t)a.ld_dhz + 4 * hi;
#pragma unroll
      for (int c = 0; c < NKC2; ++c) {
        *(uint2*)(zrow + 16 * c) = make_uint2(dz[c][0], dz[c][1]);
        *(uint2*)(zrow + 16 * c + 8) = make_uint2(dz[c][2], dz[c][3]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- d_bott = d_hz W_view[:, :256]: 8 out-blocks x 8 k steps, stored as the first 16 fm blocks of the heads operand
    char* hrowp = a.heads + rb * (HEAD_K / 16) * 1024 + u16;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x16 o[4];
#pragma unroll
      for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ob][r] = 0.f;
#pragma unroll
      for (int kc = 0; kc < NKC2; ++kc)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
          o[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(smem + BW_LDS_W2 + ((4 * half + ob) * NKC2 + kc) * 1024 + u16),
                                                          __builtin_bit_cast(bf16x8, dz[kc]), o[ob], 0, 0, 0);
      shape_schedule(std::integral_constant<int, 4 * NKC2>{});
#pragma unroll
      for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          u32x4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x2 f = {o[ob][8 * b + 2 * q], o[ob][8 * b + 2 * q + 1]};
            v[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
          }
          __builtin_nontemporal_store(v, (u32x4*)(hrowp + (size_t)(2 * (4 * half + ob) + b) * 1024));
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- column 256 = d raw density, columns 257 .. 319 zero (blocks 16 .. 19 of the row block)
    {
      const f32x2 r0 = {draw, 0.f};
      const uint32_t wr = __builtin_bit_cast(uint32_t, __builtin_convertvector(r0, bf16x2));
      const u32x4 first = hi == 0 ? (u32x4){wr, 0u, 0u, 0u} : (u32x4){0u, 0u, 0u, 0u};
      __builtin_nontemporal_store(first, (u32x4*)(hrowp + (size_t)16 * 1024));
#pragma unroll
      for (int b = 17; b < HEAD_K / 16; ++b) __builtin_nontemporal_store((u32x4){0u, 0u, 0u, 0u}, (u32x4*)(hrowp + (size_t)b * 1024));
    }
  }
}

}  // namespace mip360view

static inline bool view_first_launch_on_this_device(std::atomic<uint64_t>& done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  return (done.fetch_or(bit) & bit) == 0;
}

int mip360_launch_view_branch_fm(hipStream_t st, int rows, int n_samples, const void* bott_fm, const void* dir_table, const void* w1_fm,
                                 int ldw1, const float* b1, const void* w2_fm, int ldw2, const float* b2, float rgb_padding,
                                 void* view_in, int ld_view, void* h, int ld_h, float* rgb) {
  using namespace mip360view;
  if (rows <= 0 || rows % 256 || n_samples <= 0 || rows % n_samples || !bott_fm || !dir_table || !w1_fm || !w2_fm || !b1 || !b2 || !rgb) return 1;
  if (ldw1 % 16 || ldw1 < K1 || ldw2 % 16 || ldw2 < VIEW_W) return 1;
  if ((view_in && (ld_view < K1 || ld_view % 4)) || (h && (ld_h < VIEW_W || ld_h % 4))) return 1;
  Args a{};
  a.rows = rows; a.S = n_samples; a.bott = (const char*)bott_fm; a.dir_table = (const uint16_t*)dir_table;
  a.w1 = (const char*)w1_fm; a.w1_bpr = ldw1 / 16; a.w2 = (const char*)w2_fm; a.w2_bpr = ldw2 / 16;
  a.b1 = b1; a.b2 = b2; a.rgb_padding = rgb_padding;
  a.view_in = (uint16_t*)view_in; a.ld_view = ld_view; a.h = (uint16_t*)h; a.ld_h = ld_h; a.rgb = rgb;
  static std::atomic<uint64_t> done{0};
  if (view_first_launch_on_this_device(done))
    if (hipFuncSetAttribute((const void*)view_branch_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess) return 3;
  const int tiles = rows / 256;
  hipLaunchKernelGGL(view_branch_fwd_kernel, dim3(tiles < 256 ? tiles : 256), dim3(NW * 64), LDS_TOTAL, st, a);
  return 0;
}

int mip360_launch_view_branch_bwd_fm(hipStream_t st, int rows, const float* density, const float* g_density, const float* rgb,
                                     const float* g_rgb, float rgb_padding, const void* h, int ld_h, const void* wb3_fm, int ldwb3,
                                     const void* wb2_fm, int ldwb2, void* d_pre, void* d_hz, int ld_dhz, void* heads_fm) {
  using namespace mip360view;
  if (rows <= 0 || rows % 256 || !density || !g_density || !rgb || !g_rgb || !h || !wb3_fm || !wb2_fm || !d_pre || !d_hz || !heads_fm) return 1;
  if (ld_h < VIEW_W || ld_h % 4 || ld_dhz < VIEW_W || ld_dhz % 4 || ldwb3 % 16 || ldwb3 < 32 || ldwb2 % 16 || ldwb2 < VIEW_W) return 1;
  BwdArgs a{};
  a.rows = rows; a.density = density; a.g_density = g_density; a.rgb = rgb; a.g_rgb = g_rgb; a.pad = rgb_padding;
  a.h = (const uint16_t*)h; a.ld_h = ld_h; a.wb3 = (const char*)wb3_fm; a.wb3_bpr = ldwb3 / 16; a.wb2 = (const char*)wb2_fm; a.wb2_bpr = ldwb2 / 16;
  a.d_pre = (uint16_t*)d_pre; a.d_hz = (uint16_t*)d_hz; a.ld_dhz = ld_dhz; a.heads = (char*)heads_fm;
  static std::atomic<uint64_t> done{0};
  if (view_first_launch_on_this_device(done))
    if (hipFuncSetAttribute((const void*)view_branch_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BW_LDS_TOTAL) != hipSuccess) return 3;
  const int tiles = rows / 256;
  hipLaunchKernelGGL(view_branch_bwd_kernel, dim3(tiles < 256 ? tiles : 256), dim3(NW * 64), BW_LDS_TOTAL, st, a);
  return 0;
}
