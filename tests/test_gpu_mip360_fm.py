"""GPU tests of the fragment-major ("fm") dense-layer path of the MipNeRF-360 MLPs (csrc/mip360_fm.hip, include/mip360_hip.h),
through the C ABI:

* the layout itself: mip360_to_fm / mip360_from_fm against the formula of the header, written out in numpy;
* mip360_linear_fm (bias / ReLU + bit mask / masked dX) against a float64 GEMM of the same bf16 operands -- whole tensors and
  column windows of wider ones (the skip layer's [hidden | encoding] buffer), several tiles per workgroup;
* mip360_grad_weight_fm bit for bit against the row-major kernel and against float64; the one-column kernels
  (mip360_rowdot_fm, mip360_grad_weight_col_fm from an fm column and from a plain vector), mip360_outer_masked_fm bit for bit
  against the GEMM it replaces, mip360_pack_weight_fm and the fm mode of mip360_cast_encode bit for bit against the
  row-major results pushed through mip360_to_fm;
* one MLP forward + backward on the fm path against the row-major path (same bf16 roundings: tight) and the bf16-rounding
  float64 reference of tests/test_gpu_mip360.py.
The end-to-end training step against the oracle (tests/test_gpu_mip360_round3.py) runs on the fm path by default."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import mip360_oracle as O                                    # noqa: E402
from oracle.nerfpp_oracle import round_bf16                               # noqa: E402
from tests.test_gpu_mip360 import T, N, dev, _rays, _mlp_bf16_fwd_bwd    # noqa: E402


@pytest.fixture(scope='module')
def M():
    dev()
    from outdoor_nerf_depth_amd import mip360
    return mip360


def fm_index(r, c, ld):
    """element index of (r, c) in an fm tensor with ld columns (include/mip360_hip.h)"""
    r, c = np.asarray(r), np.asarray(c)
    row, f = r % 32, c % 16
    hi, t = (f // 4) % 2, 4 * (f // 8) + f % 4
    unit = 8 * (row >> 2) + 4 * (hi ^ (row >> 4)) + (row & 3)
    return ((r // 32) * (ld // 16) + c // 16) * 512 + unit * 8 + t


def np_to_fm(x, ld=None, col0=0, out=None):
    rows, cols = x.shape
    ld = cols if ld is None else ld
    out = np.zeros(rows * ld, x.dtype) if out is None else out
    rr, cc = np.meshgrid(np.arange(rows), np.arange(cols), indexing='ij')
    out[fm_index(rr, cc + col0, ld)] = x
    return out


def bf(a):
    return T(a).to(torch.bfloat16)


# ------------------------------------------------------------------------------------------------------------ layout
def test_to_fm_from_fm_follow_the_documented_formula(M):
    rs = np.random.RandomState(0)
    rows, cols, ld, col0 = 96, 48, 80, 32
    x = round_bf16(rs.randn(rows, cols).astype(np.float32))
    buf = torch.zeros(rows * ld, dtype=torch.bfloat16, device=dev())
    M.to_fm(bf(x), out=buf, ld=ld, col0=col0)
    want = np_to_fm(x, ld, col0)
    np.testing.assert_array_equal(N(buf), want)
    back = M.from_fm(buf, rows, cols, ld=ld, col0=col0)
    np.testing.assert_array_equal(N(back), x)
    # a bijection onto the block range: every element index of the [rows, cols] window is hit exactly once
    rr, cc = np.meshgrid(np.arange(rows), np.arange(ld), indexing='ij')
    idx = fm_index(rr, cc, ld).ravel()
    assert np.array_equal(np.sort(idx), np.arange(rows * ld))


# ------------------------------------------------------------------------------------------------------------ dense layers
@pytest.mark.parametrize('m,n,k', [(256, 256, 160), (512, 512, 320), (1024, 256, 512), (66 * 256, 256, 256), (2048, 1024, 1536)])
def test_linear_fm_against_float64(M, m, n, k):
    """(66 * 256 rows x 1 column tile: more tiles than one round of workgroups on any grid <= 64 -- the ring runs across
    tile boundaries; 2048 x 1024 x 1536: the skip layer's shape)"""
    rs = np.random.RandomState(m + n + k)
    a = round_bf16(rs.randn(m, k).astype(np.float32))
    w = round_bf16((rs.randn(n, k) / np.sqrt(k)).astype(np.float32))
    b = rs.randn(n).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    a_fm, w_fm = M.to_fm(bf(a)), M.to_fm(bf(w))
    out = M.fm_buffer(m, n, dev())
    mask = M.fm_mask_buffer(m, n, dev())
    M.linear_fm(a_fm, w_fm, T(b), 0, m, n, k, out, None)
    np.testing.assert_allclose(N(M.from_fm(out, m, n)), ref + b, rtol=2 ** -7, atol=2e-3)
    M.linear_fm(a_fm, w_fm, T(b), 1, m, n, k, out, mask)
    relu = N(M.from_fm(out, m, n))
    np.testing.assert_allclose(relu, np.maximum(ref + b, 0), rtol=2 ** -7, atol=2e-3)
    assert (relu >= 0).all() and not np.signbit(relu).any()
    # the dX form applies exactly the pattern the ReLU call stored
    M.linear_fm(a_fm, w_fm, None, 2, m, n, k, out, mask)
    got = N(M.from_fm(out, m, n))
    np.testing.assert_allclose(got, ref * (relu != 0), rtol=2 ** -7, atol=2e-3)
    assert ((got == 0) | (relu != 0)).all()


def test_linear_fm_column_windows_of_wider_tensors(M):
    """A = columns [256, 768) of a [rows, 768] tensor, C = columns [0, 256) of another [rows, 768] one (the skip layer's
    buffer): only the window is read / written."""
    rs = np.random.RandomState(5)
    m, n, k, ld = 512, 256, 512, 768
    a = round_bf16(rs.randn(m, k).astype(np.float32))
    w = round_bf16((rs.randn(n, k) / np.sqrt(k)).astype(np.float32))
    b = rs.randn(n).astype(np.float32)
    a_buf = torch.full((m * ld,), 7.0, dtype=torch.bfloat16, device=dev())
    M.to_fm(bf(a), out=a_buf, ld=ld, col0=256)
    c_buf = torch.full((m * ld,), -3.0, dtype=torch.bfloat16, device=dev())
    M.linear_fm(a_buf, M.to_fm(bf(w)), T(b), 0, m, n, k, c_buf, None, lda=ld, ldc=ld, a_col0=256, out_col0=0)
    ref = a.astype(np.float64) @ w.astype(np.float64).T + b
    np.testing.assert_allclose(N(M.from_fm(c_buf, m, n, ld=ld, col0=0)), ref, rtol=2 ** -7, atol=2e-3)
    assert (N(M.from_fm(c_buf, m, ld - n, ld=ld, col0=n)) == -3.0).all()


def test_linear_fm_rejects_shapes_it_does_not_take(M):
    x = torch.zeros(256 * 256, dtype=torch.bfloat16, device=dev())
    b = torch.zeros(256, device=dev())
    for (m, n, k) in [(255, 256, 160), (256, 128, 160), (256, 256, 128), (256, 256, 176)]:
        with pytest.raises(M.Mip360Error):
            M.linear_fm(x, x, b, 0, m, n, k, x, None)
    with pytest.raises(M.Mip360Error):
        M.linear_fm(x, x, b, 1, 256, 256, 160, x, None)           # ReLU needs somewhere to put the mask


# ------------------------------------------------------------------------------------------------------------ gradients
@pytest.mark.parametrize('m,n_in,n_out,ksplit', [(256, 256, 256, 1), (4096, 512, 256, 8), (32 * 96 * 2, 256, 512, 2), (2048, 1536, 1024, 4)])
def test_grad_weight_fm_equals_the_row_major_kernel_and_float64(M, m, n_in, n_out, ksplit):
    """(32 * 96 * 2 rows in 2 slices: 96 chunks per slice -> the ping-pong loop; the others the lock-step one)"""
    rs = np.random.RandomState(m)
    h = round_bf16(rs.randn(m, n_in).astype(np.float32))
    dz = round_bf16(rs.randn(m, n_out).astype(np.float32))
    L = M.lib()
    slabs = torch.empty(ksplit * (n_in * n_out + n_out), device=dev())
    out, bias = torch.empty(n_in, n_out, device=dev()), torch.empty(n_out, device=dev())
    h16, dz16 = bf(h), bf(dz)
    h_fm, dz_fm = M.to_fm(h16), M.to_fm(dz16)                      # (kept alive: the calls below only see raw pointers)
    M._check(L.mip360_grad_weight_fm(M._stream(), m, n_in, n_out, M._p(h_fm), n_in, M._p(dz_fm), n_out, ksplit,
                                     M._p(slabs), M._p(out), n_out, 1.0, M._p(bias)), 'grad_weight_fm')
    out2, bias2 = torch.empty_like(out), torch.empty_like(bias)
    M._check(L.mip360_grad_weight_bf16(M._stream(), m, n_in, n_out, M._p(h16), n_in, M._p(dz16), n_out, ksplit, M._p(slabs), M._p(out2),
                                       n_out, 1.0, M._p(bias2)), 'grad_weight_bf16')
    ref = h.astype(np.float64).T @ dz.astype(np.float64)
    np.testing.assert_allclose(N(out), ref, rtol=0, atol=2e-5 * np.abs(ref).max() + 1e-4)
    np.testing.assert_allclose(N(bias), dz.astype(np.float64).sum(0), rtol=0, atol=1e-3)
    if m // 32 // ksplit < 96:                                    # same summation order as the row-major kernel: identical bits
        np.testing.assert_array_equal(N(out), N(out2))
    else:
        np.testing.assert_allclose(N(out), N(out2), rtol=0, atol=2e-5 * np.abs(ref).max())


def test_one_column_kernels(M):
    rs = np.random.RandomState(2)
    m, k, ld, zcol = 1024, 512, 640, 256
    a = round_bf16(rs.randn(m, k).astype(np.float32))
    w = round_bf16((rs.randn(k) / np.sqrt(k)).astype(np.float32))
    z = round_bf16(rs.randn(m).astype(np.float32))
    L = M.lib()
    a_buf = torch.zeros(m * ld, dtype=torch.bfloat16, device=dev())
    M.to_fm(bf(a), out=a_buf, ld=ld, col0=128)
    out = torch.empty(m, 1, device=dev())
    b = torch.tensor([0.25], device=dev())
    w16 = bf(w)
    M._check(L.mip360_rowdot_fm(M._stream(), m, k, M._fm_ptr(a_buf, 128), ld, M._p(w16), M._p(b), 2, -1.0, M._p(out), 1), 'rowdot_fm')
    ref = a.astype(np.float64) @ w.astype(np.float64) + 0.25
    np.testing.assert_allclose(N(out)[:, 0], np.logaddexp(ref - 1.0, 0), rtol=2e-5, atol=2e-5)
    # column dot product: z as column `zcol` of an fm tensor, and as a plain vector
    zt = np.zeros((m, 320), np.float32)
    zt[:, zcol] = z
    z_fm = M.to_fm(bf(zt))
    want = a.astype(np.float64).T @ z.astype(np.float64)
    for (zp, ldz, col) in ((z_fm, 320, zcol), (bf(z), 1, 0)):
        ks = 16
        slabs = torch.empty(ks * (k + 1), device=dev())
        gk, gb = torch.empty(k, 1, device=dev()), torch.empty(1, device=dev())
        M._check(L.mip360_grad_weight_col_fm(M._stream(), m, k, M._fm_ptr(a_buf, 128), ld, M._p(zp), ldz, col, ks, M._p(slabs), M._p(gk),
                                             1.0, M._p(gb)), 'grad_weight_col_fm')
        np.testing.assert_allclose(N(gk)[:, 0], want, rtol=0, atol=2e-5 * np.abs(want).max())
        np.testing.assert_allclose(N(gb)[0], z.astype(np.float64).sum(), rtol=0, atol=1e-4)


def test_outer_masked_fm_equals_the_gemm_it_replaces(M):
    """PropMLP: dZ of the last trunk layer = mask * (d_raw (x) w_density): bit for bit what mip360_linear_fm act 2 gives for an
    operand whose only non-zero column is d_raw."""
    rs = np.random.RandomState(4)
    m, n, k = 1024, 256, 160
    act = round_bf16(rs.randn(m, n).astype(np.float32))
    mask = M.fm_mask_buffer(m, n, dev())
    ident = np.zeros((n, n), np.float32)
    np.fill_diagonal(ident, 1.0)
    sink = M.fm_buffer(m, n, dev())
    M.linear_fm(M.to_fm(bf(act)), M.to_fm(bf(ident)), torch.zeros(n, device=dev()), 1, m, n, n, sink, mask)     # mask = (act > 0)
    z = round_bf16(rs.randn(m).astype(np.float32))
    w = round_bf16(rs.randn(n).astype(np.float32))
    heads = np.zeros((m, k), np.float32)
    heads[:, 0] = z
    wmat = np.zeros((n, k), np.float32)
    wmat[:, 0] = w
    want = M.fm_buffer(m, n, dev())
    M.linear_fm(M.to_fm(bf(heads)), M.to_fm(bf(wmat)), None, 2, m, n, k, want, mask)
    got = M.fm_buffer(m, n, dev())
    z16, w16 = bf(z), bf(w)
    M._check(M.lib().mip360_outer_masked_fm(M._stream(), m, n, M._p(z16), M._p(w16), M._p(mask), M._p(got), n), 'outer_masked_fm')
    np.testing.assert_array_equal(N(got), N(want))
    np.testing.assert_array_equal(N(M.from_fm(got, m, n)), round_bf16((z[:, None] * w[None, :]).astype(np.float32)) * (act > 0))


def test_pack_weight_fm_and_cast_encode_fm_equal_the_converted_row_major_results(M):
    rs = np.random.RandomState(6)
    n_in, n_out, ld_f, ld_b, col0, brows = 504, 256, 512, 320, 64, 256
    kern = torch.from_numpy(rs.randn(n_in, n_out).astype(np.float32)).to(dev())
    fwd, bwd = torch.zeros(n_out, ld_f, dtype=torch.bfloat16, device=dev()), torch.zeros(n_in, n_out, dtype=torch.bfloat16, device=dev())
    fwd_fm = torch.zeros(n_out * ld_f, dtype=torch.bfloat16, device=dev())
    bwd_fm = torch.zeros(brows * ld_b, dtype=torch.bfloat16, device=dev())
    M._check(M.lib().mip360_pack_weight_fm(M._stream(), n_in, n_out, M._p(kern), M._p(fwd), ld_f, M._p(bwd), n_out, M._p(fwd_fm), ld_f,
                                           M._p(bwd_fm), ld_b, brows, col0), 'pack_weight_fm')
    np.testing.assert_array_equal(N(fwd)[:, :n_in], round_bf16(N(kern).T))
    np.testing.assert_array_equal(N(fwd_fm), N(M.to_fm(fwd)))
    want_b = torch.zeros(brows * ld_b, dtype=torch.bfloat16, device=dev())
    M.to_fm(bwd[:brows].contiguous(), out=want_b, ld=ld_b, col0=col0)
    np.testing.assert_array_equal(N(bwd_fm), N(want_b))
    # the encoding written straight into the fm tensor == the row-major rows pushed through to_fm
    n, S, W = 24, 32, 256
    rays = _rays(rs, n)
    s = np.sort(rs.rand(n, S + 1), -1).astype(np.float32)
    _, s_to_t = O.construct_ray_warps('reciprocal', rays['near'], np.full((n, 1), 30., np.float32))
    tdist = s_to_t(s).astype(np.float32)
    basis = T(O.pos_basis_t())
    rm = torch.zeros(n * S, W + 512, dtype=torch.bfloat16, device=dev())
    M.cast_encode(T(tdist), T(rays['origins']), T(rays['directions']), T(rays['radii']), basis, out=rm[:, W:], ld=W + 512)
    fm = torch.zeros(n * S * (W + 512), dtype=torch.bfloat16, device=dev())
    M.cast_encode_fm(T(tdist), T(rays['origins']), T(rays['directions']), T(rays['radii']), basis, fm, W, W + 512)
    np.testing.assert_array_equal(N(fm), N(M.to_fm(rm)))


# ------------------------------------------------------------------------------------------------------------ one MLP
@pytest.mark.parametrize('which', ['prop', 'nerf'])
def test_mlp_forward_backward_fm_matches_row_major_and_reference(M, which):
    rs = np.random.RandomState(12)
    n, S = 16, 32
    cfg = O.PROP_CFG if which == 'prop' else O.NERF_CFG
    params = O.init_mlp_params(cfg, rs)
    params = [(w, (rs.randn(*b.shape) * 0.05).astype(np.float32)) for w, b in params]
    rays = _rays(rs, n)
    s = np.sort(rs.rand(n, S + 1), -1).astype(np.float32)
    _, s_to_t = O.construct_ray_warps('reciprocal', rays['near'], np.full((n, 1), 30., np.float32))
    tdist = s_to_t(s).astype(np.float32)
    basis = T(O.pos_basis_t())
    mcfg = M.PROP_CFG if which == 'prop' else M.NERF_CFG
    W, rows = cfg['net_width'], n * S
    g_d = rs.randn(n, S).astype(np.float32)
    g_c = rs.randn(n, S, 3).astype(np.float32) if which == 'nerf' else None
    args = (T(tdist), T(rays['origins']), T(rays['directions']), T(rays['radii']), basis)
    res = {}
    for kind in ('fm', 'rm'):
        tm = M.TrainableMLP(params, mcfg, dev())
        assert tm.w_fm, 'the fm operand copies exist for the 256- / 1024-wide MLPs'
        if kind == 'fm':
            buf = M.fm_buffer(rows, W + 512, dev())
            M.cast_encode_fm(*args, buf, W, W + 512)
            density, rgb, saved = M.mlp_forward_train_fm(tm, buf, rows, T(rays['viewdirs']), n, S)
            assert saved['fm']
        else:
            buf = torch.empty(rows, W + 512, dtype=torch.bfloat16, device=dev())
            M.cast_encode(*args, out=buf[:, W:], ld=W + 512)
            enc = N(buf[:, W:W + 504])
            density, rgb, saved = M.mlp_forward_train(tm, buf, rows, T(rays['viewdirs']), n, S)
        M.mlp_backward(tm, saved, rows, T(g_d).reshape(-1), None if g_c is None else T(g_c).reshape(-1, 3), [None, None])
        res[kind] = (N(density), None if rgb is None else N(rgb), [(N(tm.kernel(t, tm.grads)).astype(np.float64), N(tm.bias(t, tm.grads)).astype(np.float64))
                                                                     for t in range(len(params))])
    rel = lambda a, b: np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
    # same bf16 roundings at the same places, but float32 sums in another order and the bias as bf16 hi + lo (2^-17): an
    # activation on a rounding boundary lands on the other side and the difference travels down the dX chain -- the two
    # paths differ from each other by what either differs from the bf16-rounding reference (2-5 % on the 8-layer NerfMLP)
    np.testing.assert_allclose(res['fm'][0], res['rm'][0], rtol=2e-2, atol=1e-4)
    assert (np.abs(res['fm'][0] - res['rm'][0]) <= 1e-4 * np.abs(res['rm'][0]) + 1e-6).mean() > 0.75
    if which == 'nerf':
        np.testing.assert_allclose(res['fm'][1], res['rm'][1], rtol=0, atol=1.5e-2)
    for t in range(len(params)):
        assert rel(res['fm'][2][t][0], res['rm'][2][t][0]) < (6e-2 if which == 'nerf' else 1e-2), ('kernel fm vs rm', t, rel(res['fm'][2][t][0], res['rm'][2][t][0]))
    ref16 = _mlp_bf16_fwd_bwd(params, cfg, enc, np.repeat(rays['viewdirs'], S, 0), g_d, g_c)
    for t in range(len(params)):
        mine_k, mine_b = res['fm'][2][t]
        # The reference rounds where the row-major kernels round and sums in float64; the row-major path sits 2.5-2.8 % from it
        # on the NerfMLP (tests/test_gpu_mip360.py bounds that by 3e-2), the fm path 3.8-4.5 %: measured against the float64
        # oracle both are equally far from the true gradient (15.8 % on these random upstream gradients), and one bf16 ulp
        # on 0.01 % of the encoding moves the row-major gradients by 2.6 % -- the bound is that sensitivity, not a kernel error.
        e_fm, e_rm = rel(mine_k, ref16[t][0]), rel(res['rm'][2][t][0], ref16[t][0])
        assert e_fm < max(3e-2, 2 * e_rm), ('kernel vs bf16 reference', t, e_fm, e_rm)
        b_fm, b_rm = rel(mine_b, ref16[t][1]), rel(res['rm'][2][t][1], ref16[t][1])
        assert b_fm < max(3e-2, 2 * b_rm) or np.abs(mine_b - ref16[t][1]).max() < 1e-2 * np.abs(ref16[t][1]).max() + 1e-3, ('bias', t, b_fm, b_rm)


def test_trainer_takes_the_same_steps_on_both_paths(M, monkeypatch):
    """Mip360Trainer with USE_FM off (the row-major kernels, as MIP360_NO_FM=1 selects) against the default: losses of three
    steps agree to bf16 grade and the data term decreases."""
    from outdoor_nerf_depth_amd import mip360 as mod
    rs = np.random.RandomState(3)
    n = 128
    rays = {k: T(v) for k, v in _rays(rs, n).items()}
    gt = T(rs.rand(n, 3).astype(np.float32))
    sup = T((0.5 + rs.rand(n)).astype(np.float32))
    losses = {}
    for kind in ('fm', 'rm'):
        monkeypatch.setattr(mod, 'USE_FM', kind == 'fm')
        prs = np.random.RandomState(7)
        tr = mod.Mip360Trainer(O.init_mlp_params(O.PROP_CFG, prs), O.init_mlp_params(O.NERF_CFG, prs), dev(), max_steps=1000)
        assert bool(tr.nerf.w_fm) == (kind == 'fm')
        out = []
        for step in range(3):
            jit = [T(np.random.RandomState(100 + step).rand(n).astype(np.float32)) for _ in range(3)]
            out.append(N(tr.train_step(rays, gt, sup, jitter01=jit)))
        losses[kind] = np.array(out)
    assert np.isfinite(losses['fm']).all() and np.isfinite(losses['rm']).all()
    np.testing.assert_allclose(losses['fm'], losses['rm'], rtol=3e-2, atol=1e-4)
    assert losses['fm'][-1, 1] < losses['fm'][0, 1]                                 # the data term goes down


def test_model_forward_on_both_paths(M, monkeypatch):
    """Mip360Model.forward (inference: no masks kept) with the trunk fm against the row-major kernels: renderings to bf16 grade."""
    from outdoor_nerf_depth_amd import mip360 as mod
    rs = np.random.RandomState(9)
    n = 64
    rays = {k: T(v) for k, v in _rays(rs, n).items()}
    prs = np.random.RandomState(1)
    pp, pn = O.init_mlp_params(O.PROP_CFG, prs), O.init_mlp_params(O.NERF_CFG, prs)
    out = {}
    for kind in ('fm', 'rm'):
        monkeypatch.setattr(mod, 'USE_FM', kind == 'fm')
        model = mod.Mip360Model(pp, pn, dev())
        assert bool(model.nerf.w_fm) == (kind == 'fm')
        rend, hist = model.forward(rays, train_frac=0.5)
        out[kind] = (N(rend[-1]['rgb']), N(rend[-1]['distance_mean']), N(hist[-1]['weights']))
    np.testing.assert_allclose(out['fm'][0], out['rm'][0], rtol=0, atol=5e-3)
    np.testing.assert_allclose(out['fm'][1], out['rm'][1], rtol=2e-2, atol=1e-3)
    np.testing.assert_allclose(out['fm'][2], out['rm'][2], rtol=0, atol=5e-3)


def test_deferred_updates_and_concurrent_backward_change_no_bit(M):
    """Mip360Trainer options that only move work between streams: proposal backward on its own stream (default) vs after the NeRF
    level's; updates joined at the end of every step (default) vs pipelined under the next step (defer_update + flush()).
    Parameters and Adam moments after three steps are bit-identical."""
    rs = np.random.RandomState(5)
    n = 64
    rays = {k: T(v) for k, v in _rays(rs, n).items()}
    gt = T(rs.rand(n, 3).astype(np.float32))
    sup = T((0.5 + rs.rand(n)).astype(np.float32))
    jit = [[T(np.random.RandomState(10 * s + l).rand(n).astype(np.float32)) for l in range(3)] for s in range(3)]
    finals = []
    for concurrent, defer in ((True, False), (False, False), (True, True)):
        prs = np.random.RandomState(7)
        tr = M.Mip360Trainer(O.init_mlp_params(O.PROP_CFG, prs), O.init_mlp_params(O.NERF_CFG, prs), dev(), max_steps=1000)
        tr.concurrent_prop_backward, tr.defer_update = concurrent, defer
        for s in range(3):
            tr.train_step(rays, gt, sup, jitter01=jit[s])
        tr.flush()
        finals.append([N(t) for t in (tr.nerf.flat, tr.prop.flat, tr.nerf.mu, tr.prop.nu)])
    for other in finals[1:]:
        for a, b in zip(finals[0], other):
            np.testing.assert_array_equal(a, b)


def test_row_counts_the_fm_kernels_do_not_take_fall_back_to_row_major(M, monkeypatch):
    """33 rays x 64 / 32 samples = 2112 / 1056 rows: not multiples of 256, so every level runs on the row-major kernels although
    the fm operand copies exist -- bit-identical to a trainer built with USE_FM off, and the lazily skipped row-major weight
    copies are refreshed before those kernels read them (the second and third step would diverge otherwise)."""
    from outdoor_nerf_depth_amd import mip360 as mod
    rs = np.random.RandomState(8)
    n = 33
    rays = {k: T(v) for k, v in _rays(rs, n).items()}
    gt = T(rs.rand(n, 3).astype(np.float32))
    sup = T((0.5 + rs.rand(n)).astype(np.float32))
    jit = [T(rs.rand(n).astype(np.float32)) for _ in range(3)]
    finals = []
    for use_fm in (True, False):
        monkeypatch.setattr(mod, 'USE_FM', use_fm)
        prs = np.random.RandomState(7)
        tr = mod.Mip360Trainer(O.init_mlp_params(O.PROP_CFG, prs), O.init_mlp_params(O.NERF_CFG, prs), dev(), max_steps=1000)
        assert bool(tr.nerf.w_fm) == use_fm
        hist = [N(tr.train_step(rays, gt, sup, jitter01=jit)) for _ in range(3)]
        assert np.isfinite(hist).all()
        finals.append((N(tr.nerf.flat), N(tr.prop.flat)))
    np.testing.assert_array_equal(finals[0][0], finals[1][0])
    np.testing.assert_array_equal(finals[0][1], finals[1][1])
