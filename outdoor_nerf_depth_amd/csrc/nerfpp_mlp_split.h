// Split-bf16 forward of the fused NeRF++ MLP (precision 2), "unit-pipelined" form (round 6).  Included by nerfpp_mlp.hip inside
// namespace nerfpp, after its helpers; same arithmetic as mlp_fwd_body<NET, 2, ...> -- every accumulator sees the same MFMA chain in
// the same order (k-chunks ascending, Whi Ahi, Whi Alo, Wlo Ahi) and every activation the same conversion, so the results are
// bit-identical to the stage-at-a-time kernel it replaces (tests/test_gpu_split_pipeline.py compares the two builds bit for bit).
//
// Why.  Per-block cycle stamps of the split-bf16 kernels (profiles/r06_split_stamps.md): one wave per SIMD, so nothing hides
//   (a) the bubble at every weight-block barrier -- wait, barrier, then the first LDS reads of the new block with no MFMA to
//       cover their latency: ~590 cycles per block in the inference AND the training kernel (2128 cycles per 48-MFMA block
//       against 1536 of matrix-pipe time; 1381 per 24-MFMA block against 768), 20-32 % of a tile;
//   (b) the per-stage epilogue (conversion of 128 accumulators to hi / lo operand chunks, ReLU, sign words, bias reload):
//       ~3000-4000 cycles per stage with the matrix pipe idle, 16-17 % of a tile.
// How.
//   * The weight stream is consumed in UNITS of four fragment pairs (8 KiB: 4 out-blocks of one k-chunk, or 4 k-chunks of a
//     one-out-block head = 12 MFMAs).  The fragments of unit u + 1 are read from LDS BEFORE the MFMAs of unit u are issued --
//     across block barriers and across stage boundaries -- so a barrier (and the LDS latency behind it) is followed by 12 MFMAs
//     whose operands are already in registers.
//   * LAZY epilogue.  A stage's pre-activations leave the accumulators as raw float32 copies in VGPRs (one wave per SIMD has 512
//     registers: 128 AGPRs of accumulators, 128 VGPRs of raw values) -- out-blocks 0-3 in the stage's last unit (they are final
//     one unit earlier; that unit's MFMAs go to out-blocks 4-7), out-blocks 4-7 in the NEXT stage's first unit (whose MFMAs go
//     to out-blocks 0-3) -- and each half takes the next stage's bias right behind its copy.  The hi / lo operand chunks are
//     converted from the raw copy one chunk per k-chunk, one k-chunk ahead of the MFMAs that consume them.  The VALU work of the
//     epilogue thus sits between MFMAs (<= 5 issue slots per MFMA are free) instead of in a phase where the matrix pipe idles.
//   * Training: NO wave roles.  Every wave fetches its quarter of each weight block and writes the chunks of its own tile
//     straight after converting them (two 1 KiB wave-stores per chunk, spread over the consuming stage by construction).  Stores
//     and weight DMA share vmcnt and retire out of order WITH RESPECT TO EACH OTHER, but loads retire in order among loads: with
//     the newest block's 8 DMA instructions as the youngest loads of the wave, `vmcnt(8)` leaves at most 8 operations outstanding
//     -- if they are those 8 loads everything older has retired, and if some of those 8 have retired then every OLDER load has too.
//     So the block before the newest is guaranteed either way; what the wait additionally drains is the wave's stores, which by
//     then are at least a unit old.  3-slot ring of 16-fragment blocks (two blocks ahead, DMA in pieces between the MFMAs), half
//     the barriers of the 8-fragment roles pipe, no hand-off traffic, 33 KiB less LDS, one instruction stream for all waves.
//     (The roles pipe -- loader wave, LDS hand-off region, helper waves -- existed to avoid a full drain when a stage's 32
//     wave-stores went out in one burst; with lazy conversion there is no burst.  A 2-slot ring with a full drain was the first
//     form: 3 % slower per step.)
#pragma once

namespace nerfpp {

struct WUnit { bf16x8 h[4], l[4]; };           // four fragment pairs (hi plane, lo plane) of the weight stream

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) as a fold expression: the unit loops are expanded by the front
// end (a `#pragma unroll` loop of 32-44 units with everything inlined into it is past LLVM's pragma-unroll size limit; left
// rolled it indexes the accumulator arrays at run time, i.e. through scratch memory: 6x slower)
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(const F& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(const F& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <typename Pipe, bool RING>
struct UnitFeed {
  static constexpr int UPB = Pipe::BLKF / 4;   // units per weight block
  Pipe& pipe;
  const char* blk = nullptr;                   // LDS address (this lane's 16 bytes) of the block that holds the next unit
  WUnit nxt;
  __device__ __forceinline__ explicit UnitFeed(Pipe& p) : pipe(p) {}
  // read global unit gu (a constant after unrolling) of the stream into nxt; entering a new block passes its barrier first
  __device__ __forceinline__ void fetch(int gu) {
    if (gu % UPB == 0) {
      // every read of the block whose slot this barrier frees has been ISSUED; make sure it has also returned (ring pipe: the
      // roles pipe's sync_step waits for lgkmcnt(0) itself)
      if constexpr (RING) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      blk = pipe.acquire();
    }
    const char* p = blk + (gu % UPB) * (8 * FRAG_BYTES);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      nxt.h[g] = *(const bf16x8*)(p + 2 * g * FRAG_BYTES);
      nxt.l[g] = *(const bf16x8*)(p + (2 * g + 1) * FRAG_BYTES);
    }
  }
};

// One stage of NOB out-blocks x NKC k-chunks (the first LIVE of them carry data), first global unit GU0 of GUN in the stream.
//   getb(kc)  -> the B operand chunk kc (ready by then);   ahead(i) runs in unit i, before the unit's MFMAs in program order.
// Units: NOB = 8: unit i = (k-chunk i / 2, out-blocks 4 (i & 1) ..);  NOB = 4: unit i = k-chunk i;  NOB = 1: k-chunks 4 i .. 4 i + 3.
template <int NOB, int NKC, int LIVE, int GU0, int GUN, typename Feed, typename GetB, typename Ahead>
__device__ __forceinline__ void stage_units(Feed& feed, f32x16 (&acc)[NOB], const GetB& getb, const Ahead& ahead) {
  static_assert(NOB == 8 || NOB == 4 || NOB == 1, "unit shapes");
  constexpr int NU = NOB == 8 ? 2 * NKC : NOB == 4 ? NKC : NKC / 4;
  static_for<NU>([&](auto i_c) __attribute__((always_inline)) {
    constexpr int i = decltype(i_c)::value;
    // A unit is one scheduling region: nothing is hoisted out of the unit it was placed in (the conversions would otherwise run
    // many chunks ahead and keep their results alive), and inside it the VALU / LDS / store instructions are dealt out between
    // the 12 MFMAs (<= 5 issue slots per MFMA are hidden; a clump of 14 VALU between two MFMAs is not).
    __builtin_amdgcn_sched_barrier(0);
    const WUnit w = feed.nxt;
    feed.pipe.trickle((GU0 + i) % Feed::UPB, Feed::UPB);
    if (GU0 + i + 1 < GUN) feed.fetch(GU0 + i + 1);
    ahead(i);
    if constexpr (NOB == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kc = 4 * i + j;
        if (kc >= LIVE) continue;
        const Frag<2>& b = getb(kc);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h[j], b.v[0], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h[j], b.v[1], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.l[j], b.v[0], acc[0], 0, 0, 0);
      }
    } else {
      constexpr int kc = NOB == 8 ? i >> 1 : i, ob0 = NOB == 8 ? 4 * (i & 1) : 0;
      if constexpr (kc < LIVE) {
        const Frag<2>& b = getb(kc);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[ob0 + g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h[g], b.v[0], acc[ob0 + g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[ob0 + g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h[g], b.v[1], acc[ob0 + g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[ob0 + g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.l[g], b.v[0], acc[ob0 + g], 0, 0, 0);
      }
    }
    if constexpr (probe::UNIT_VALU > 0) {
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, probe::UNIT_VALU, 0);     // its fillers: VALU ...
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    // ... one LDS read ...
        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);                    // ... one store
      }
    }
  });
  __builtin_amdgcn_sched_barrier(0);
}

// chunk c = 2 ob + hh of a stage output: ReLU + hi / lo split of accumulator registers 8 hh .. 8 hh + 7 of out-block ob, and the
// chunk's sign bits into the stage's sign words (layout and arithmetic of acc_to_frags_relu_bits<.., 2, true>)
__device__ __forceinline__ void conv_chunk_relu(const float (&raw_ob)[16], int ob, int hh, Frag<2>& out, uint32_t (&m)[4]) {
  u32x4 dh, dl;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float a = raw_ob[8 * hh + 2 * w], b = raw_ob[8 * hh + 2 * w + 1];
    const uint32_t d = pack2<1>(a, b);
    const int j = (ob & 1) * 8 + hh * 4 + w;
    m[ob >> 1] |= (d >> j) & (0x80008000u >> j);
    const uint32_t lo = pack2<1>(a - __uint_as_float(d << 16), b - __uint_as_float(d & 0xffff0000u));
    const uint32_t neg = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, d) >> (s16x2){15, 15});
    dh[w] = d & ~neg;
    dl[w] = lo & ~neg;
  }
  out.v[0] = __builtin_bit_cast(bf16x8, dh);
  out.v[1] = __builtin_bit_cast(bf16x8, dl);
}

// the accumulator registers of one out-block -> raw float32 copies in VGPRs, read HERE (the empty asm pins the v_accvgpr_read:
// left to the coalescer the read sinks to its use in the next stage, by when the accumulator holds other values, and the whole
// 16-register tuple gets copied to keep the old ones)
__device__ __forceinline__ void raw_copy_ob(const f32x16& acc_ob, float (&raw_ob)[16]) {
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    float r = acc_ob[e];
    asm volatile("" : "+v"(r));
    raw_ob[e] = r;
  }
}

// the bias of one out-block (LDS copy of the bias stream) into its accumulator registers
__device__ __forceinline__ void init_bias_ob(f32x16& acc_ob, uint32_t lds_off_bytes, int ob, int hi) {
  if constexpr ((probe::EXP & 64) != 0) {          // (probes, garbage results: zero accumulators instead of the bias reads)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_ob[r] = 0.f;
    return;
  }
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  LDS_AS char* base = (LDS_AS char*)smem;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = *(LDS_AS f32x4*)(base + lds_off_bytes + (ob * 32 + hi * 16 + 4 * q) * 4);
    acc_ob[4 * q] = v[0]; acc_ob[4 * q + 1] = v[1]; acc_ob[4 * q + 2] = v[2]; acc_ob[4 * q + 3] = v[3];
  }
}

// HI (training): only the hi planes of the saved tensors are written and H0 is not saved (the backward is single-pass bf16 and
// recomputes H0 in its weight-gradient job: nerfpp_api.hip sets save_lo = 0 and skip_h0 = 1 together)
template <int NET, int NW, bool TRAIN, bool HI>
__device__ __forceinline__ void mlp_fwd_body_split(const MlpFwdArgs& a, const int bid) {
  constexpr int P = 2;
  using LD = FwdLds<NET, P, NW, TRAIN>;
  constexpr int KPE = kpe(NET);
  static_assert(!LD::ROLES, "ring pipe");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5;
  const size_t row_raw = (size_t)bid * (NW * 32) + wave * 32 + (lane & 31);
  const bool valid = row_raw < (size_t)a.rows;
  const size_t row = valid ? row_raw : (size_t)a.rows - 1;
  const size_t plane_rows = a.rows_padded;
  const size_t wrow0 = (size_t)bid * (NW * 32) + wave * 32;                 // this wave's first tile row
  constexpr int NPS = HI ? 1 : 2;                                           // planes of the saved tensors that are written out
  probe::kernel_prologue(a.out_raw);
  // rows past the end of the batch are written as zeros (nerfpp_mlp.hip: saved tensors): a per-lane AND, no branch in a unit
  const uint32_t vmask = valid ? 0xffffffffu : 0u;
  char* pe_stash = smem + LD::STASH + wave * (KPE * a_planes(P) * 1024);
  const size_t nblk32 = a.rows_padded / 32;
  uint4* mask_out = a.masks + (wrow0 / 32) * 64 + lane;                     // + stage * nblk32 * 64

  WeightPipe<w_planes(P), NW, LD::MODE, LD::NBUF, LD::BF, 0> pipe;
  pipe.stamp_off = LD::TOTAL;
  pipe.init(a.w_stream, fwd_frags(NET) / LD::BF, wave, lane);
  for (int i = threadIdx.x; i < FWD_BIAS_FLOATS / 4; i += NW * 64)
    *(float4*)(smem + LD::BIAS + i * 16) = ((const float4*)a.bias)[i];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  UnitFeed<decltype(pipe), true> feed(pipe);
  constexpr int GUN = fwd_frags(NET) / 4;
  static_assert(LD::BF % 4 == 0, "whole units per block");

  float x[4], vd[3], depth_real;
  sample_point<NET>(a.geom, row, a.S, x, vd, &depth_real);
  Frag<P> pe[KPE], df[2];
  encode_point<NET, P>(x, hi, pe);
  encode_dir<P>(vd, hi, df);
  auto mask_frag = [&](Frag<P>& f) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      u32x4 d = __builtin_bit_cast(u32x4, f.v[p]);
#pragma unroll
      for (int w = 0; w < 4; ++w) d[w] &= vmask;
      f.v[p] = __builtin_bit_cast(bf16x8, d);
    }
  };
  if constexpr (TRAIN) {
#pragma unroll
    for (int c = 0; c < KPE; ++c) mask_frag(pe[c]);
    mask_frag(df[0]); mask_frag(df[1]);
    save_frags<KPE, P>(a.ws.t[T_X], plane_rows * kpew(NET), kpew(NET), wrow0, lane, pe, NPS);
    save_frags<2, P>(a.ws.t[T_DIRX], plane_rows * 32, 32, wrow0, lane, df, NPS);
  }
  stash_frags<KPE, P>(pe_stash, lane, pe);

  // chunk c of a saved tensor: two 1 KiB wave-stores (one with HI) straight from the registers that hold it
  auto save_chunk = [&](__bf16* base, int ld, int c, Frag<P>& f, bool skip) __attribute__((always_inline)) {
    if constexpr (TRAIN) {
      mask_frag(f);
      if (!skip) store_chunk<P>(base, plane_rows * ld, ld, wrow0, lane, c, f, NPS);
    }
  };
  auto save_bits = [&](const uint32_t (&m)[4], int mask_stage) __attribute__((always_inline)) {
    if constexpr (TRAIN) mask_out[(size_t)mask_stage * nblk32 * 64] = make_uint4(m[0], m[1], m[2], m[3]);
  };

  f32x16 acc[8];
  float raw[8][16];
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) init_bias_ob(acc[ob], LD::BIAS + fs_bias_off(FS_L0) * 4, ob, hi);
  feed.fetch(0);
  // out-blocks ob0 .. ob0 + 3 are final: raw copies out, the bias of stage `next` in (next < 0: the heads use their own accumulators)
  auto retire = [&](int ob0, int next) __attribute__((always_inline)) {
#pragma unroll
    for (int ob = ob0; ob < ob0 + 4; ++ob) {
      raw_copy_ob(acc[ob], raw[ob]);
      if (next >= 0) init_bias_ob(acc[ob], LD::BIAS + (fs_bias_off(FS_L0) + next * 256) * 4, ob, hi);
    }
  };

  // A trunk stage l (1..7): input = H_{l-1}, converted lazily from the raw copy of the previous stage's accumulators (chunk 0 is
  // ready: that stage's last unit), preceded by the K0 chunks of the encoded point in L5; its last unit retires out-blocks 0-3 and
  // converts chunk 0 of H_l.  mi / mo: sign words of H_{l-1} / H_l.
  auto trunk = [&](auto l_c, auto k0_c, Frag<P> (&hin)[16], Frag<P> (&hout)[16], const Frag<P> (&pre)[KPE], uint32_t (&mi)[4],
                   uint32_t (&mo)[4]) __attribute__((always_inline)) {
    constexpr int l = decltype(l_c)::value, K0 = decltype(k0_c)::value, NKC = K0 + 16;
    constexpr int GU0 = fs_frag_off(NET, l) / 4;
    __bf16* const base_in = TRAIN ? a.ws.t[T_H0 + l - 1] : nullptr;
    constexpr bool skip_in = TRAIN && l == 1 && HI;                         // H0: recomputed by its weight-gradient job (bf16 backward)
    stage_units<8, NKC, NKC, GU0, GUN>(feed, acc,
      [&](int kc) -> const Frag<P>& { return kc < K0 ? pre[kc < K0 ? kc : 0] : hin[kc - K0]; },
      [&](int i) __attribute__((always_inline)) {
        const int kc = i >> 1, half = i & 1;
        if (i == 0) retire(4, l);                                           // (this unit's MFMAs go to out-blocks 0-3)
        if (half == 0) {
          const int c = kc + 1 - K0;                                        // the chunk the NEXT k-chunk consumes
          if (c >= 1 && c <= 15) {
            conv_chunk_relu(raw[c >> 1], c >> 1, c & 1, hin[c], mi);
            save_chunk(base_in, 256, c, hin[c], skip_in);
            if (c == 15) save_bits(mi, l - 1);
          }
        }
        if (i == 2 * NKC - 1) {                                             // last unit (MFMAs on out-blocks 4-7): 0-3 are final
          retire(0, l < 7 ? l + 1 : -1);
          conv_chunk_relu(raw[0], 0, 0, hout[0], mo);
          save_chunk(TRAIN ? a.ws.t[T_H0 + l] : nullptr, 256, 0, hout[0], false);
        }
      });
  };

  uint32_t m0[4] = {0, 0, 0, 0}, m1[4] = {0, 0, 0, 0};
  Frag<P> h0[16], h1[16];
  // L0: the encoded point (registers)
  {
    stage_units<8, KPE, KPE, 0, GUN>(feed, acc,
      [&](int kc) -> const Frag<P>& { return pe[kc]; },
      [&](int i) __attribute__((always_inline)) {
        if (i == 2 * KPE - 1) {
          retire(0, 1);
          conv_chunk_relu(raw[0], 0, 0, h0[0], m0);
          save_chunk(TRAIN ? a.ws.t[T_H0] : nullptr, 256, 0, h0[0], TRAIN && HI);
        }
      });
  }
  const Frag<P> (&nopre)[KPE] = pe;                                           // (unused where K0 = 0)
  trunk(IC(1), IC(0), h0, h1, nopre, m0, m1);
#pragma unroll
  for (int w = 0; w < 4; ++w) m0[w] = 0;
  trunk(IC(2), IC(0), h1, h0, nopre, m1, m0);
#pragma unroll
  for (int w = 0; w < 4; ++w) m1[w] = 0;
  trunk(IC(3), IC(0), h0, h1, nopre, m0, m1);
#pragma unroll
  for (int w = 0; w < 4; ++w) m0[w] = 0;
  trunk(IC(4), IC(0), h1, h0, nopre, m1, m0);
#pragma unroll
  for (int w = 0; w < 4; ++w) m1[w] = 0;
  {
    // L5: input = cat(encoded point, h4)                                 nerf_network.py:127-129
    Frag<P> pe2[KPE];
    unstash_frags<KPE, P>(pe_stash, lane, pe2);
    trunk(IC(5), IC(KPE), h0, h1, pe2, m0, m1);
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) m0[w] = 0;
  trunk(IC(6), IC(0), h1, h0, nopre, m1, m0);
#pragma unroll
  for (int w = 0; w < 4; ++w) m1[w] = 0;
  // L7: its last unit leaves chunk 0 of h7; the sigma stage (one out-block: four k-chunks per unit) converts the rest as it goes
  Frag<P> (&h7)[16] = h1;
  uint32_t (&m7)[4] = m1;
  trunk(IC(7), IC(0), h0, h7, nopre, m0, m7);
  // sigma from h7                                                        nerf_network.py:131-136
  // (the remap layer is folded into the colour head: nerfpp_common.h, forward stages.  R is not a tensor here.)
  f32x16 acc1[1], acc4[4];
  init_bias_ob(acc1[0], LD::BIAS + fs_bias_off(FS_SIG) * 4, 0, hi);
#pragma unroll
  for (int ob = 0; ob < 4; ++ob) init_bias_ob(acc4[ob], LD::BIAS + fs_bias_off(FS_RGB0) * 4, ob, hi);
  __bf16* const base7 = TRAIN ? a.ws.t[T_H0 + 7] : nullptr;
  auto conv7 = [&](int c) __attribute__((always_inline)) {
    conv_chunk_relu(raw[c >> 1], c >> 1, c & 1, h7[c], m7);
    save_chunk(base7, 256, c, h7[c], false);
  };
  // unit i consumes chunks 4 i .. 4 i + 3: unit 0 converts 1-3 (chunk 0: L7's last unit) and, from the out-blocks it retires, 4-7
  stage_units<1, 16, 16, fs_frag_off(NET, FS_SIG) / 4, GUN>(feed, acc1,
    [&](int kc) -> const Frag<P>& { return h7[kc]; },
    [&](int i) __attribute__((always_inline)) {
      if (i == 0) {
#pragma unroll
        for (int c = 1; c < 4; ++c) conv7(c);
        retire(4, -1);
      }
      if (i < 3) {
#pragma unroll
        for (int c = 4 * i + 4; c < 4 * i + 8; ++c) conv7(c);
        if (i == 2) save_bits(m7, 7);
      }
    });
  const float sigma_raw = acc1[0][0];
  // colour head: relu(Wc h7 + Wrgb0[:, 256:] dirs + bc)                  nerf_network.py:131,137-138
  Frag<P> g[8];
  uint32_t mg[4] = {0, 0, 0, 0};
  stage_units<4, 20, 18, fs_frag_off(NET, FS_RGB0) / 4, GUN>(feed, acc4,
    [&](int kc) -> const Frag<P>& { return kc < 16 ? h7[kc < 16 ? kc : 0] : df[kc < 18 ? kc - 16 : 0]; },
    [&](int i) __attribute__((always_inline)) {
      if (i >= 18) {                                                          // (padding units: no MFMAs) the colour head's output
#pragma unroll
        for (int c = 4 * (i - 18); c < 4 * (i - 18) + 4; ++c) {
          if ((c & 1) == 0) raw_copy_ob(acc4[c >> 1], raw[c >> 1]);
          conv_chunk_relu(raw[c >> 1], c >> 1, c & 1, g[c], mg);
          save_chunk(TRAIN ? a.ws.t[T_G] : nullptr, 128, c, g[c], false);
        }
        if (i == 19) save_bits(mg, 8);
      }
    });
  init_bias_ob(acc1[0], LD::BIAS + fs_bias_off(FS_RGB1) * 4, 0, hi);
  stage_units<1, 16, 8, fs_frag_off(NET, FS_RGB1) / 4, GUN>(feed, acc1,
    [&](int kc) -> const Frag<P>& { return g[kc < 8 ? kc : 0]; },
    [&](int) __attribute__((always_inline)) {});
  if (valid && hi == 0) {
    float4 o;
    o.x = 1.f / (1.f + expf(-acc1[0][0]));
    o.y = 1.f / (1.f + expf(-acc1[0][1]));
    o.z = 1.f / (1.f + expf(-acc1[0][2]));
    o.w = sigma_raw;
    ((float4*)a.out_raw)[row] = o;
    if (NET == 1) a.depth_real[row] = depth_real;
  }
  probe::dump_stamps(LD::TOTAL, wave, lane);
}

// ------------------------------------------------------------------------------------------------------------------------------
// backward (dX chain), same construction: units, lazy epilogue (dH accumulators -> raw copy -> masked hi / lo chunks of dZ, one
// chunk per k-chunk ahead of the MFMAs that consume it), ring pipe with a 2-slot ring of 16-fragment blocks, no wave roles: every
// wave DMAs its own ReLU sign words (1 KiB per wave and stage, one stage ahead, into a 2-slot LDS area: ALL loads of a wave are
// then LDS-DMA loads, which retire in order -- what the counted vmcnt waits of the ring rely on) and stores the chunks of its own
// dZ tiles right behind their conversion.  Same MFMA chains and conversions as mlp_bwd_body<NET, 2, NW>: bit-identical dZ tensors.
// ------------------------------------------------------------------------------------------------------------------------------
template <int NW>
struct BwdLdsV2 {
  static constexpr int BF = BLK_FRAGS, NBUF = probe::V2T_NBUF_BWD;
  static constexpr int W = NBUF * BF * 2 * FRAG_BYTES;
  static constexpr int MASKS = W;                      // 2 slots x NW KiB of ReLU sign words (every wave DMAs its own 1 KiB per stage)
  static constexpr int TOTAL = MASKS + 2 * NW * 1024;
};

// chunk c = 2 ob + hh of dZ: the accumulator registers 8 hh .. 8 hh + 7 of out-block ob, masked by the forward's sign words
// (arithmetic of mask_to_frags<.., 2>)
__device__ __forceinline__ void conv_chunk_mask(const float (&raw_ob)[16], int ob, int hh, const uint4 bits, Frag<2>& out) {
  const uint32_t act[4] = {~bits.x, ~bits.y, ~bits.z, ~bits.w};       // bit set = unit active
  u32x4 dh, dl;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float a = raw_ob[8 * hh + 2 * w], b = raw_ob[8 * hh + 2 * w + 1];
    const int j = (ob & 1) * 8 + hh * 4 + w;
    const uint32_t keep = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, act[ob >> 1] << j) >> (s16x2){15, 15});
    const uint32_t hi = pack2<1>(a, b);
    const uint32_t lo = pack2<1>(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
    dh[w] = hi & keep;
    dl[w] = lo & keep;
  }
  out.v[0] = __builtin_bit_cast(bf16x8, dh);
  out.v[1] = __builtin_bit_cast(bf16x8, dl);
}

template <int NET, int NW>
__device__ __forceinline__ void mlp_bwd_body_split(const MlpBwdArgs& a, const int bid) {
  constexpr int P = 2;
  using LD = BwdLdsV2<NW>;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5;
  const size_t row_raw = (size_t)bid * (NW * 32) + wave * 32 + (lane & 31);
  const bool valid = row_raw < (size_t)a.rows;
  const size_t row = valid ? row_raw : (size_t)a.rows - 1;
  const size_t plane_rows = a.rows_padded;
  const size_t wrow0 = (size_t)bid * (NW * 32) + wave * 32;
  const size_t nblk32 = a.rows_padded / 32;
  const char* mask_g = (const char*)(a.masks + (wrow0 / 32) * 64);        // this wave's 1 KiB of sign words; + stage * nblk32 * 1024
  const uint32_t lds0 = lds_base_addr();
  // sign words of `mstage` -> LDS slot (one 1 KiB DMA); read back once the wave's counted waits have covered it
  auto mask_dma = [&](int mstage, int slot) __attribute__((always_inline)) {
    glds16xN_saddr<1>(mask_g + (size_t)mstage * nblk32 * 1024, (uint32_t)lane * 16u, lds0 + LD::MASKS + (slot * NW + wave) * 1024);
  };
  auto mask_get = [&](int slot) __attribute__((always_inline)) -> uint4 {
    return *(const uint4*)(smem + LD::MASKS + (slot * NW + wave) * 1024 + lane * 16);
  };
  mask_dma(8, 0);
  mask_dma(7, 1);

  WeightPipe<P, NW, PIPE_RING, LD::NBUF, LD::BF> pipe;
  pipe.stamp_off = LD::TOTAL;
  pipe.init(a.w_stream, BWD_FRAGS / LD::BF, wave, lane);
  UnitFeed<decltype(pipe), true> feed(pipe);
  constexpr int GUN = BWD_FRAGS / 4;

  float4 d = ((const float4*)a.d_out)[row];
  if (!valid) d = make_float4(0.f, 0.f, 0.f, 0.f);                        // rows past the end: zero gradients, zero dZ everywhere
  {
    // dP [rows, 32] and dS [rows, 32] (chunks 0, 1 of [dS | dG]) come straight from d_out
    Frag<P> t[2];
    t[0] = zero_frag<P>(); t[1] = zero_frag<P>();
    if (hi == 0) { set_slot<P>(t[0], 0, d.x); set_slot<P>(t[0], 1, d.y); set_slot<P>(t[0], 2, d.z); }
    save_frags<2, P>(a.ws.t[T_DP], plane_rows * 32, 32, wrow0, lane, t);
    t[0] = zero_frag<P>();
    if (hi == 0) set_slot<P>(t[0], 0, d.w);
    save_frags<2, P>(a.ws.t[T_DS], plane_rows * DSG_LD, DSG_LD, wrow0, lane, t);
  }
  auto store = [&](__bf16* base, int ld, int c, const Frag<P>& f) __attribute__((always_inline)) {
    store_chunk<P>(base, plane_rows * ld, ld, wrow0, lane, c, f);
  };
  auto zero_ob = [&](f32x16& acc_ob) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_ob[r] = 0.f;
  };

  feed.fetch(0);                          // (its counted wait covers the two sign-word DMAs above: they are older than block 0)
  const uint4 mk8 = mask_get(0), mk7 = mask_get(1);
  float raw[8][16];
  // B0: dG = Wrgb1^T dP (one live k-chunk), masked by G > 0 (sign words 8)
  Frag<P> dg[8];
  {
    Frag<P> in0 = zero_frag<P>();
    if (hi == 0) { set_slot<P>(in0, 0, d.x); set_slot<P>(in0, 1, d.y); set_slot<P>(in0, 2, d.z); }
    f32x16 acc4[4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) zero_ob(acc4[ob]);
    stage_units<4, 4, 1, bs_frag_off(BS_DG) / 4, GUN>(feed, acc4,
      [&](int) -> const Frag<P>& { return in0; },
      [&](int i) __attribute__((always_inline)) {
        if (i == 1) {                                                     // (units 1-3 are block padding: no MFMAs)
#pragma unroll
          for (int ob = 0; ob < 4; ++ob) raw_copy_ob(acc4[ob], raw[ob]);
          conv_chunk_mask(raw[0], 0, 0, mk8, dg[0]);
          store(a.ws.t[T_DG], DSG_LD, 0, dg[0]);
        }
      });
  }
  // B2: dH7 = Wc^T dG + wsigma dsigma (8 dG chunks, the dsigma chunk, one chunk of padding), masked by H7 > 0
  f32x16 acc[8];
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) zero_ob(acc[ob]);
  Frag<P> dsig = zero_frag<P>();
  if (hi == 0) set_slot<P>(dsig, 0, d.w);
  auto retire = [&](int ob0) __attribute__((always_inline)) {
#pragma unroll
    for (int ob = ob0; ob < ob0 + 4; ++ob) { raw_copy_ob(acc[ob], raw[ob]); zero_ob(acc[ob]); }
  };
  Frag<P> dza[16], dzb[16];
  float raw4[4][16];                                                       // (B0's raw values live through B2 while `raw` takes B2's)
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int e = 0; e < 16; ++e) raw4[ob][e] = raw[ob][e];
  stage_units<8, 10, 9, bs_frag_off(BS_DH7) / 4, GUN>(feed, acc,
    [&](int kc) -> const Frag<P>& { return kc < 8 ? dg[kc < 8 ? kc : 0] : dsig; },
    [&](int i) __attribute__((always_inline)) {
      const int kc = i >> 1, half = i & 1;
      if (half == 0 && kc + 1 <= 7) {
        const int c = kc + 1;
        conv_chunk_mask(raw4[c >> 1], c >> 1, c & 1, mk8, dg[c]);
        store(a.ws.t[T_DG], DSG_LD, c, dg[c]);
      }
      if (i == 19) {                                                       // (kc = 9 is padding: out-blocks 0-3 have long been final)
        retire(0);
        conv_chunk_mask(raw[0], 0, 0, mk7, dza[0]);
        store(a.ws.t[T_DZ0 + 7], 256, 0, dza[0]);
      }
    });
  // B3 .. B9: dH_{l-1} = W_l^T dZ_l, l = 7 .. 1; dZ_l is converted lazily from the previous stage's raw copy with sign words l
  // (sign words l - 1 go to the LDS slot that held sign words l + 1: slot l & 1)
  auto trunk = [&](auto l_c, Frag<P> (&zin)[16], Frag<P> (&zout)[16], const uint4 mk_in, uint4& mk_out) __attribute__((always_inline)) {
    constexpr int l = decltype(l_c)::value;
    constexpr int GU0 = bs_frag_off(10 - l) / 4;
    __bf16* const base_in = a.ws.t[T_DZ0 + l];
    stage_units<8, 16, 16, GU0, GUN>(feed, acc,
      [&](int kc) -> const Frag<P>& { return zin[kc]; },
      [&](int i) __attribute__((always_inline)) {
        const int kc = i >> 1, half = i & 1;
        if (i == 0) retire(4);                                             // (this unit's MFMAs go to out-blocks 0-3)
        if (i == 2) mask_dma(l - 1, l & 1);                                // (sign words l + 1 were last read in the previous stage)
        if (half == 0 && kc + 1 <= 15) {
          const int c = kc + 1;
          conv_chunk_mask(raw[c >> 1], c >> 1, c & 1, mk_in, zin[c]);
          store(base_in, 256, c, zin[c]);
        }
        if (i == 31) {
          retire(0);
          mk_out = mask_get(l & 1);                                        // (issued 29 units = 7 counted block waits ago)
          conv_chunk_mask(raw[0], 0, 0, mk_out, zout[0]);
          store(a.ws.t[T_DZ0 + l - 1], 256, 0, zout[0]);
        }
      });
  };
  uint4 mka = mk7, mkb;
  trunk(IC(7), dza, dzb, mka, mkb);
  trunk(IC(6), dzb, dza, mkb, mka);
  trunk(IC(5), dza, dzb, mka, mkb);
  trunk(IC(4), dzb, dza, mkb, mka);
  trunk(IC(3), dza, dzb, mka, mkb);
  trunk(IC(2), dzb, dza, mkb, mka);
  trunk(IC(1), dza, dzb, mka, mkb);
  // dZ0: no stage consumes it -- the rest of the tile is converted and written here (chunk 0: the last unit above)
  retire(4);
#pragma unroll
  for (int c = 1; c < 16; ++c) {
    conv_chunk_mask(raw[c >> 1], c >> 1, c & 1, mkb, dzb[c]);
    store(a.ws.t[T_DZ0], 256, c, dzb[c]);
  }
  probe::dump_stamps(LD::TOTAL, wave, lane);
}

}  // namespace nerfpp
