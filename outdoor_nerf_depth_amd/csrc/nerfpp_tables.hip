// Host-side builders for the index tables that tie the reference's parameter layout
// (nerf_network.py:88-117 state_dict order) to the packed MFMA weight streams and to the
// weight-gradient slabs.  Pure host code: runs (and is tested) without a GPU.
#include <hip/hip_runtime.h>
#include "nerfpp_common.h"
#include "../../include/nerfpp_hip.h"

using namespace nerfpp;

namespace {

// reference flat index (within one net) of W_eff[o][f] of forward stage s, or -1 for padding
int fwd_weff_ref(int net, int s, int o, int f) {
  const int pr = pe_ref_ch(net), kw = kpew(net);
  if (s == FS_L0) {
    if (f >= kw) return -1;
    const int c = pe_ref_of_feat(net, f);
    return c < 0 ? -1 : ref_w_off(net, 0) + o * ref_in(net, 0) + c;
  }
  if (s == FS_L5) {
    int c;
    if (f < kw) { c = pe_ref_of_feat(net, f); if (c < 0) return -1; }
    else { c = pr + (f - kw); }
    return ref_w_off(net, 5) + o * ref_in(net, 5) + c;
  }
  if (s < 8) return ref_w_off(net, s) + o * 256 + f;
  if (s == FS_REMAP) return ref_w_off(net, RT_REMAP) + o * 256 + f;
  if (s == FS_SIG) return o == 0 ? ref_w_off(net, RT_SIGMA) + f : -1;
  if (s == FS_RGB0) {
    if (o >= 128) return -1;
    int c;
    if (f < 256) c = f;
    else if (f < 256 + DIRW) { c = dir_ref_of_feat(f - 256); if (c < 0) return -1; c += 256; }
    else return -1;
    return ref_w_off(net, RT_RGB0) + o * ref_in(net, RT_RGB0) + c;
  }
  // FS_RGB1
  return (o < 3 && f < 128) ? ref_w_off(net, RT_RGB1) + o * 128 + f : -1;
}

// what the PACKED forward stream holds at W_eff[o][f] of stage s: the reference parameter, or -- the h7 columns of the
// colour head -- the derived Wc (index net_params(net) + o * 256 + f, nerfpp_common.h)
int fwd_wsrc(int net, int s, int o, int f) {
  if (s == FS_RGB0 && o < 128 && f < 256) return net_params(net) + o * 256 + f;
  return fwd_weff_ref(net, s, o, f);
}

int fwd_beff_ref(int net, int s, int o) {
  if (s < 8) return ref_b_off(net, s) + o;
  if (s == FS_REMAP) return ref_b_off(net, RT_REMAP) + o;
  if (s == FS_SIG) return o == 0 ? ref_b_off(net, RT_SIGMA) : -1;
  if (s == FS_RGB0) return o < 128 ? ref_b_off(net, RT_RGB0) + o : -1;
  return o < 3 ? ref_b_off(net, RT_RGB1) + o : -1;
}

int fwd_bsrc(int net, int s, int o) {
  if (s == FS_RGB0 && o < 128) return net_params(net) + DERIVED_WC + o;       // bc
  if (s == FS_REMAP) return -1;                                               // (folded into bc; the slots stay, zero)
  return fwd_beff_ref(net, s, o);
}

// backward stage s: Wt_eff[o][f] = d(stage input f) / d(stage output o) weight
int bwd_weff_ref(int net, int s, int o, int f) {
  if (s == BS_DG) return (o < 128 && f < 3) ? ref_w_off(net, RT_RGB1) + f * 128 + o : -1;
  if (s == BS_DR) return -1;                                                  // (folded: no fragments)
  if (s == BS_DH7) {
    if (f < 128) return net_params(net) + f * 256 + o;                        // Wc[f][o] (derived)
    if (f == 128) return ref_w_off(net, RT_SIGMA) + o;
    return -1;
  }
  const int l = bs_layer(s);
  if (l == 5) return ref_w_off(net, 5) + f * ref_in(net, 5) + pe_ref_ch(net) + o;
  return ref_w_off(net, l) + f * 256 + o;
}

}  // namespace

extern "C" {

int nerfpp_table_sizes(int net, int64_t* fwd_elems, int64_t* fwd_bias_elems, int64_t* bwd_elems,
                       int64_t* slab_floats, int64_t* n_params) {
  if (net < 0 || net > 1) return NERFPP_ERR_ARG;
  *fwd_elems = (int64_t)fwd_frags(net) * 512;
  *fwd_bias_elems = FWD_BIAS_FLOATS;
  *bwd_elems = (int64_t)BWD_FRAGS * 512;
  *slab_floats = gslab_floats(net);
  *n_params = net_params(net);
  return NERFPP_OK;
}

// fwd_tbl[F*512 + l*8 + t], bias_tbl[fs_bias_off(s) + ob*32 + hi*16 + r], bwd_tbl likewise,
// unpack_tbl[ref index] = index into one gradient slab ([GW stages ... | GB stages ...])
int nerfpp_build_tables(int net, int32_t* fwd_tbl, int32_t* bias_tbl, int32_t* bwd_tbl,
                        int32_t* unpack_tbl) {
  if (net < 0 || net > 1) return NERFPP_ERR_ARG;
  for (int s = 0; s < FS_COUNT; ++s) {
    const int nob = fs_nob(s), nkc = fs_nkc(net, s);
    for (int kc = 0; kc < nkc; ++kc)
      for (int ob = 0; ob < nob; ++ob) {
        const int64_t F = fs_frag_off(net, s) + kc * nob + ob;
        for (int l = 0; l < 64; ++l)
          for (int t = 0; t < 8; ++t)
            fwd_tbl[F * 512 + l * 8 + t] = fwd_wsrc(net, s, ob * 32 + (l & 31), kslot(kc, l >> 5, t));
      }
    for (int ob = 0; ob < nob; ++ob)
      for (int hi = 0; hi < 2; ++hi)
        for (int r = 0; r < 16; ++r)
          bias_tbl[fs_bias_off(s) + ob * 32 + hi * 16 + r] = fwd_bsrc(net, s, dfeat(ob, hi, r));
  }
  for (int s = 0; s < BS_COUNT; ++s) {
    const int nob = bs_nob(s), nkc = bs_nkc(s);
    for (int kc = 0; kc < nkc; ++kc)
      for (int ob = 0; ob < nob; ++ob) {
        const int64_t F = bs_frag_off(s) + kc * nob + ob;
        for (int l = 0; l < 64; ++l)
          for (int t = 0; t < 8; ++t)
            bwd_tbl[F * 512 + l * 8 + t] = bwd_weff_ref(net, s, ob * 32 + (l & 31), kslot(kc, l >> 5, t));
      }
  }
  const int np = net_params(net);
  for (int i = 0; i < np; ++i) unpack_tbl[i] = -1;
  for (int s = 0; s < FS_COUNT; ++s) {
    const int O = gw_O(s), I = gw_I(net, s);
    for (int o = 0; o < O; ++o) {
      for (int i = 0; i < I; ++i) {
        const int r = fwd_weff_ref(net, s, o, i);
        if (r >= 0) unpack_tbl[r] = gw_off(net, s) + o * I + i;
      }
      const int rb = fwd_beff_ref(net, s, o);
      if (rb >= 0) unpack_tbl[rb] = gw_floats(net) + gb_off(s) + o;
    }
  }
  for (int i = 0; i < np; ++i)
    if (unpack_tbl[i] < 0) return NERFPP_ERR_INTERNAL;     // every parameter must have a home
  return NERFPP_OK;
}

}  // extern "C"
