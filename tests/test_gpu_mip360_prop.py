"""mip360_prop_mlp_fm (csrc/mip360_prop.hip): the PropMLP forward as one launch, through the C ABI.

* every layer output against a float64 evaluation of the same bf16 operands chained through the kernel's own (bf16) outputs, and
  against the launches it replaces (mip360_linear_fm act 1 x 4): equal up to the summation order, i.e. a bf16 ulp on a small
  fraction of the elements;
* the ReLU masks in mip360_linear_fm's format: the masked dX layer (act 2) driven by the fused kernel's mask zeroes exactly the
  elements whose saved activation is zero;
* density == mip360_rowdot_fm of the saved last layer, bit for bit; inference mode (nothing saved) == training mode;
* mip360_prop_mlp_bwd_fm (the dX chain as one launch) == mip360_outer_masked_fm + 3 x mip360_linear_fm act 2, bit for bit;
* mip360_view_branch_fm (the NerfMLP's view branch, forward) against from_fm + dir_encode + two row-major GEMMs;
* the training step of Mip360Trainer with and without the fused launch: same losses / parameters to summation-order grade.
"""
import numpy as np
import pytest
import torch

from tests.test_gpu_mip360_fm import M, N, T, bf, dev, round_bf16   # noqa: F401  (fixtures / helpers)

pytestmark = pytest.mark.gpu


def _problem(M, rows, seed, ld=768, col0=256):
    rs = np.random.RandomState(seed)
    x = round_bf16((rs.randn(rows, 512) * 0.7).astype(np.float32))
    x[:, 504:] = 0
    ws = [round_bf16((rs.randn(256, 512 if l == 0 else 256) * np.sqrt(2.0 / (512 if l == 0 else 256))).astype(np.float32)) for l in range(4)]
    bs = [(rs.randn(256) * 0.1).astype(np.float32) for _ in range(4)]
    wd = round_bf16((rs.randn(1, 256) / 16).astype(np.float32))
    bd = np.array([0.3], np.float32)
    enc = torch.full((rows * ld,), 5.0, dtype=torch.bfloat16, device=dev())
    M.to_fm(bf(x), out=enc, ld=ld, col0=col0)
    return dict(x=x, ws=ws, bs=bs, wd=wd, bd=bd, enc=enc, ld=ld, col0=col0,
                w_fm=[M.to_fm(bf(w)) for w in ws], b_t=[T(b) for b in bs], wd_t=bf(wd), bd_t=T(bd))


def _fused(M, P, rows, train=True):
    hs = [M.fm_buffer(rows, 256, dev()) for _ in range(4)] if train else None
    masks = [M.fm_mask_buffer(rows, 256, dev()) for _ in range(4)] if train else None
    density = torch.empty(rows, 1, device=dev())
    M.prop_mlp_fm(P['enc'], P['col0'], P['ld'], rows, P['w_fm'], [512, 256, 256, 256], P['b_t'], P['wd_t'], P['bd_t'], density,
                  h=hs, masks=masks)
    return hs, masks, density


@pytest.mark.parametrize('rows', [256, 66 * 256])
def test_prop_mlp_fm_layers_masks_and_density(M, rows):
    P = _problem(M, rows, rows)
    hs, masks, density = _fused(M, P, rows)
    assert (N(M.from_fm(P['enc'], rows, 256, ld=P['ld'], col0=0)) == 5.0).all()      # nothing outside the operand window is touched
    a = P['x'].astype(np.float64)
    a_t, a_col0, a_ld, a_k = P['enc'], P['col0'], P['ld'], 512
    for l in range(4):
        got = N(M.from_fm(hs[l], rows, 256))
        ref = np.maximum(a @ P['ws'][l].astype(np.float64).T + P['bs'][l], 0)
        np.testing.assert_allclose(got, ref, rtol=2 ** -7, atol=2e-3)
        assert (got >= 0).all() and not np.signbit(got).any()
        # the launch it replaces, on the same operand
        want_fm, want_mask = M.fm_buffer(rows, 256, dev()), M.fm_mask_buffer(rows, 256, dev())
        M.linear_fm(a_t, P['w_fm'][l], P['b_t'][l], 1, rows, 256, a_k, want_fm, want_mask, lda=a_ld, a_col0=a_col0)
        want = N(M.from_fm(want_fm, rows, 256))
        diff = got != want
        assert diff.mean() < 2e-3, diff.mean()                              # summation order: a bf16 ulp here and there
        np.testing.assert_allclose(got, want, rtol=2 ** -7, atol=1e-6)
        # the mask drives linear_fm's masked dX exactly like the saved activation
        eye = round_bf16(np.eye(256, dtype=np.float32))
        ones_fm = M.to_fm(bf(np.ones((rows, 256), np.float32)))
        out = M.fm_buffer(rows, 256, dev())
        M.linear_fm(ones_fm, M.to_fm(bf(eye)), None, 2, rows, 256, 256, out, masks[l])
        np.testing.assert_array_equal(N(M.from_fm(out, rows, 256)) != 0, got != 0)
        a = got.astype(np.float64)
        a_t, a_col0, a_ld, a_k = hs[l], 0, 256, 256
    # density: the row-dot launch on the saved last layer, bit for bit; and the float64 value
    want_d = torch.empty(rows, 1, device=dev())
    M._check(M.lib().mip360_rowdot_fm(M._stream(), rows, 256, M._fm_ptr(hs[3], 0), 256, M._p(P['wd_t']), M._p(P['bd_t']), 2, M.DENSITY_BIAS,
                                      M._p(want_d), 1), 'rowdot_fm')
    np.testing.assert_array_equal(N(density), N(want_d))
    raw = a @ P['wd'].astype(np.float64).T + P['bd'] + M.DENSITY_BIAS
    np.testing.assert_allclose(N(density), np.logaddexp(raw, 0), rtol=1e-5, atol=1e-6)
    # inference mode: nothing saved, the same density
    _, _, d_inf = _fused(M, P, rows, train=False)
    np.testing.assert_array_equal(N(d_inf), N(density))


@pytest.mark.parametrize('rows', [256, 66 * 256])
def test_prop_mlp_bwd_fm_equals_the_launches_it_replaces(M, rows):
    """dZ_3 .. dZ_0 of mip360_prop_mlp_bwd_fm against mip360_outer_masked_fm + three mip360_linear_fm (act 2) on the masks the fused
    forward wrote: bit for bit (same products, same MFMA sequence per element); and dZ_2 against float64."""
    P = _problem(M, rows, rows + 1)
    hs, masks, _ = _fused(M, P, rows)
    rs = np.random.RandomState(rows)
    z = round_bf16((rs.randn(rows) * 0.05).astype(np.float32))
    wbs = [None] + [round_bf16(np.ascontiguousarray(P['ws'][l].T)) for l in range(1, 4)]        # [in, out] = W_l^T: dX = dZ W_l
    wb_fm = [None] + [M.to_fm(bf(w)) for w in wbs[1:]]
    dz = [M.fm_buffer(rows, 256, dev()) for _ in range(4)]
    M.prop_mlp_bwd_fm(rows, bf(z), P['wd_t'], masks, wb_fm, [0, 256, 256, 256], dz)
    want = [None] * 4
    want[3] = M.fm_buffer(rows, 256, dev())
    M._check(M.lib().mip360_outer_masked_fm(M._stream(), rows, 256, M._p(bf(z)), M._p(P['wd_t']), M._p(masks[3]), M._p(want[3]), 256),
             'outer_masked_fm')
    for l in (3, 2, 1):
        want[l - 1] = M.fm_buffer(rows, 256, dev())
        M.linear_fm(want[l], wb_fm[l], None, 2, rows, 256, 256, want[l - 1], masks[l - 1])
    for l in range(4):
        np.testing.assert_array_equal(N(dz[l]).view(np.uint16) if N(dz[l]).dtype != np.float32 else N(dz[l]), N(want[l]).view(np.uint16) if N(want[l]).dtype != np.float32 else N(want[l]))
    d3 = N(M.from_fm(dz[3], rows, 256)).astype(np.float64)
    h2 = N(M.from_fm(hs[2], rows, 256))
    ref = (d3 @ P['ws'][3].astype(np.float64)) * (h2 != 0)
    np.testing.assert_allclose(N(M.from_fm(dz[2], rows, 256)), ref, rtol=2 ** -7, atol=1e-5)
    h3 = N(M.from_fm(hs[3], rows, 256))
    np.testing.assert_allclose(d3, round_bf16((z[:, None] * P['wd']).astype(np.float32)) * (h3 != 0), rtol=0, atol=0)


def test_prop_mlp_fm_rejects_shapes_it_does_not_take(M):
    P = _problem(M, 256, 3)
    density = torch.empty(256, 1, device=dev())
    with pytest.raises(M.Mip360Error):                                   # rows not a multiple of 256
        M.prop_mlp_fm(P['enc'], P['col0'], P['ld'], 224, P['w_fm'], [512, 256, 256, 256], P['b_t'], P['wd_t'], P['bd_t'], density)
    with pytest.raises(M.Mip360Error):                                   # operand window beyond the tensor
        M.prop_mlp_fm(P['enc'], 512, P['ld'], 256, P['w_fm'], [512, 256, 256, 256], P['b_t'], P['wd_t'], P['bd_t'], density)
    with pytest.raises(M.Mip360Error):                                   # first layer narrower than the 512 operand columns
        M.prop_mlp_fm(P['enc'], P['col0'], P['ld'], 256, P['w_fm'], [256, 256, 256, 256], P['b_t'], P['wd_t'], P['bd_t'], density)


def test_trainer_step_with_and_without_the_fused_prop_mlp(M, monkeypatch):
    """Three optimisation steps of Mip360Trainer on 64 rays (4096 / 2048 rows per level): the fused PropMLP forward against the
    five launches it replaces.  Same sample positions would need bit-equal densities; the layers differ in summation order, so
    the comparison is loss-level (1e-3) and update-level (Adam steps of size lr: a fraction of the components may flip sign)."""
    from oracle import mip360_oracle as O
    from tests.test_gpu_mip360 import _rays
    rs = np.random.RandomState(5)
    n = 64
    rays = {k: T(v) for k, v in _rays(rs, n).items()}
    gt = T(rs.rand(n, 3).astype(np.float32))
    sup = T((0.5 + rs.rand(n)).astype(np.float32))
    jit = [[T(np.random.RandomState(10 * s_ + l).rand(n).astype(np.float32)) for l in range(3)] for s_ in range(3)]
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(M, 'USE_FUSED_PROP', fused)
        monkeypatch.setattr(M, 'USE_FUSED_VIEW', fused)
        prs = np.random.RandomState(7)
        tr = M.Mip360Trainer(O.init_mlp_params(O.PROP_CFG, prs), O.init_mlp_params(O.NERF_CFG, prs), dev(), max_steps=1000)
        hist = [N(tr.train_step(rays, gt, sup, jitter01=jit[s_])) for s_ in range(3)]
        tr.flush()
        out[fused] = (np.stack(hist), N(tr.prop.flat), N(tr.nerf.flat))
    assert np.isfinite(out[True][0]).all()
    np.testing.assert_allclose(out[True][0][0], out[False][0][0], rtol=2e-3, atol=1e-6)       # first step: same weights
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=2e-2, atol=1e-5)
    lr = float(M.learning_rate(3))
    for a_, b_ in ((out[True][1], out[False][1]), (out[True][2], out[False][2])):
        assert np.abs(a_ - b_).max() <= 3 * 2.5 * lr                     # three Adam steps, each bounded by ~lr
        assert np.mean(np.abs(a_ - b_) > 0.5 * lr) < 0.1


def test_view_branch_fm_equals_the_launches_it_replaces(M):
    """mip360_view_branch_fm against mip360_from_fm + mip360_dir_encode + two mip360_linear_bf16 (act 1 / act 3) on the same operands:
    view_in bit for bit (a copy of the bottleneck + dir_encode's arithmetic), h up to the summation order, rgb to 1e-5; and against
    float64.  Inference form (nothing but rgb written) gives the same rgb."""
    rs = np.random.RandomState(21)
    n_rays, S = 64, 32
    rows = n_rays * S
    bott = round_bf16(rs.randn(rows, 256).astype(np.float32))
    vd = rs.randn(n_rays, 3).astype(np.float32)
    vd /= np.linalg.norm(vd, axis=-1, keepdims=True)
    w1 = round_bf16((rs.randn(128, 288) * np.sqrt(2.0 / 283)).astype(np.float32))
    w1[:, 283:] = 0
    w2 = round_bf16((rs.randn(3, 128) / np.sqrt(128)).astype(np.float32))
    b1, b2 = (rs.randn(128) * 0.1).astype(np.float32), (rs.randn(3) * 0.1).astype(np.float32)
    bott_fm = M.to_fm(bf(bott))
    w1_fm = M.to_fm(bf(w1))
    w2p = np.zeros((32, 128), np.float32)
    w2p[:3] = w2
    w2_fm = M.to_fm(bf(w2p))

    class PK(object):
        w_fm = {2: w1_fm, 3: w2_fm}
        b = {2: T(b1), 3: T(b2)}
    view_in = torch.full((rows, 288), 9.0, dtype=torch.bfloat16, device=dev())
    h = torch.full((rows, 128), 9.0, dtype=torch.bfloat16, device=dev())
    rgb = torch.empty(rows, 3, device=dev())
    M.view_branch_fm(PK, 0, bott_fm, rows, S, T(vd), view_in, h, rgb)
    # the launches it replaces
    want_in = torch.empty(rows, 288, dtype=torch.bfloat16, device=dev())
    M.from_fm(bott_fm, rows, 256, out=want_in)
    M._check(M.lib().mip360_dir_encode(M._stream(), n_rays, S, M._p(T(vd)), M._p(want_in), 288, 256, 32), 'dir_encode')
    want_h = torch.empty(rows, 128, dtype=torch.bfloat16, device=dev())
    M.linear(want_in, bf(w1), T(b1), act=1, out_bf16=want_h, m=rows, n=128, k=288)
    want_rgb = torch.empty(rows, 3, device=dev())
    M.linear(want_h, bf(w2), T(b2), act=3, act_param=M.RGB_PADDING, out_f32=want_rgb, m=rows, n=3, k=128)
    np.testing.assert_array_equal(N(view_in), N(want_in))
    gh, wh = N(h), N(want_h)
    assert (gh != wh).mean() < 2e-3
    np.testing.assert_allclose(gh, wh, rtol=2 ** -7, atol=1e-6)
    np.testing.assert_allclose(N(rgb), N(want_rgb), rtol=0, atol=2e-4)
    # float64 on the kernel's own bf16 intermediate
    x = N(view_in).astype(np.float64)
    ref_h = np.maximum(x @ w1.astype(np.float64).T + b1, 0)
    np.testing.assert_allclose(gh, ref_h, rtol=2 ** -7, atol=2e-3)
    raw = gh.astype(np.float64) @ w2.astype(np.float64).T + b2
    np.testing.assert_allclose(N(rgb), 1 / (1 + np.exp(-raw)) * (1 + 2 * M.RGB_PADDING) - M.RGB_PADDING, rtol=0, atol=2e-6)
    rgb2 = torch.empty(rows, 3, device=dev())
    M.view_branch_fm(PK, 0, bott_fm, rows, S, T(vd), None, None, rgb2)
    np.testing.assert_array_equal(N(rgb2), N(rgb))
    with pytest.raises(M.Mip360Error):
        M.view_branch_fm(PK, 0, bott_fm, rows - 32, S, T(vd), None, None, rgb2)


def test_grad_weight_fm_multi_equals_the_single_problem_launches(M):
    """mip360_grad_weight_fm_multi (the PropMLP's four weight-gradient problems in one launch, 48 row slices each) against
    mip360_grad_weight_fm per problem (its own slice count) and float64: kernel and bias gradients."""
    import ctypes as C
    rs = np.random.RandomState(31)
    m, W = 6144, 256                                       # 192 chunks of 32 rows: 48 slices of 4
    n_ins = [512, 256, 256, 256]
    hs = [round_bf16(rs.randn(m, k).astype(np.float32)) for k in n_ins]
    dzs = [round_bf16((rs.randn(m, W) * 0.1).astype(np.float32)) for _ in n_ins]
    h_fm = [M.to_fm(bf(h)) for h in hs]
    dz_fm = [M.to_fm(bf(d)) for d in dzs]
    ks = 48
    sizes = [ks * (k * W + W) for k in n_ins]
    slabs = [torch.empty(sz, device=dev()) for sz in sizes]
    ci = lambda v: (C.c_int * 4)(*[int(x) for x in v])
    cp = lambda ps: (C.c_void_p * 4)(*ps)
    M._check(M.lib().mip360_grad_weight_fm_multi(M._stream(), 4, m, ks, ci(n_ins), ci([W] * 4), cp([M._p(t) for t in h_fm]), ci(n_ins),
                                                 cp([M._p(t) for t in dz_fm]), ci([W] * 4), cp([M._p(t) for t in slabs])), 'multi')
    scratch = [None, None]
    for i, k in enumerate(n_ins):
        out, bias = torch.empty(k, W, device=dev()), torch.empty(W, device=dev())
        M._check(M.lib().mip360_grad_weight_reduce(M._stream(), k, k, W, ks, M._p(slabs[i]), M._p(out), W, 1.0, M._p(bias)), 'reduce')
        want, want_b = torch.empty(k, W, device=dev()), torch.empty(W, device=dev())
        M._grad_weight_fm(h_fm[i], 0, k, dz_fm[i], W, m, k, W, want, scratch, want_b)
        ref = hs[i].astype(np.float64).T @ dzs[i].astype(np.float64)
        np.testing.assert_allclose(N(out), ref, rtol=1e-4, atol=2e-3)
        np.testing.assert_allclose(N(out), N(want), rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(N(bias), dzs[i].astype(np.float64).sum(0), rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(N(bias), N(want_b), rtol=1e-4, atol=1e-3)
    with pytest.raises(M.Mip360Error):
        M._check(M.lib().mip360_grad_weight_fm_multi(M._stream(), 4, m, 300, ci(n_ins), ci([W] * 4), cp([M._p(t) for t in h_fm]), ci(n_ins),
                                                     cp([M._p(t) for t in dz_fm]), ci([W] * 4), cp([M._p(t) for t in slabs])), 'multi')


def test_view_branch_bwd_fm_equals_the_launches_it_replaces(M):
    """mip360_view_branch_bwd_fm against mip360_head_backward + mip360_linear_bf16 act 4 + mip360_linear_bf16 + mip360_to_fm: d_pre and
    the d-raw-density column bit for bit (same per-row arithmetic), d_hz / d_bott up to the summation order of 3 / 128 products."""
    rs = np.random.RandomState(23)
    rows = 2048
    density = (rs.rand(rows) * 3).astype(np.float32)
    g_density = (rs.randn(rows) * 0.01).astype(np.float32)
    rgb = rs.rand(rows, 3).astype(np.float32)
    g_rgb = (rs.randn(rows, 3) * 0.01).astype(np.float32)
    h = np.maximum(round_bf16(rs.randn(rows, 128).astype(np.float32)), 0)
    wb3 = round_bf16(np.concatenate([(rs.randn(128, 3) / 11).astype(np.float32), np.zeros((128, 29), np.float32)], 1))     # [in 128, out 3 -> 32]
    wb2 = round_bf16((rs.randn(288, 128) / 17).astype(np.float32))                                                           # [in 288, out 128]
    head_k = 320
    t_den, t_gden, t_rgb, t_grgb, t_h = T(density), T(g_density), T(rgb), T(g_rgb), bf(h)      # (kept alive across the launches)
    wb3_fm, wb2_fm, t_wb3, t_wb2 = M.to_fm(bf(wb3)), M.to_fm(bf(wb2)), bf(wb3), bf(wb2)
    d_pre = torch.empty(rows, 32, dtype=torch.bfloat16, device=dev())
    d_hz = torch.empty(rows, 128, dtype=torch.bfloat16, device=dev())
    heads_fm = M.fm_buffer(rows, head_k, dev())
    M._check(M.lib().mip360_view_branch_bwd_fm(M._stream(), rows, M._p(t_den), M._p(t_gden), M._p(t_rgb), M._p(t_grgb), M.RGB_PADDING,
                                               M._p(t_h), 128, M._p(wb3_fm), 32, M._p(wb2_fm), 128, M._p(d_pre), M._p(d_hz),
                                               128, M._p(heads_fm)), 'view_branch_bwd_fm')
    # the launches it replaces
    heads = torch.empty(rows, head_k, dtype=torch.bfloat16, device=dev())
    want_pre = torch.empty(rows, 32, dtype=torch.bfloat16, device=dev())
    M._check(M.lib().mip360_head_backward(M._stream(), rows, M._p(t_den), M._p(t_gden), M._p(t_rgb), M._p(t_grgb), M.RGB_PADDING,
                                          M._p(heads), head_k, 256, head_k, M._p(want_pre)), 'head_backward')
    want_hz = torch.empty(rows, 128, dtype=torch.bfloat16, device=dev())
    M.linear(want_pre, t_wb3, None, act=4, out_bf16=want_hz, m=rows, n=128, k=32, aux=t_h)
    M.linear(want_hz, t_wb2, None, act=0, out_bf16=heads, m=rows, n=256, k=128)
    np.testing.assert_array_equal(N(d_pre), N(want_pre))
    gz, wz = N(d_hz), N(want_hz)
    assert ((gz == 0) == (h == 0) | (gz == 0)).all() and (gz[h == 0] == 0).all()
    np.testing.assert_allclose(gz, wz, rtol=2 ** -7, atol=1e-7)
    got = N(M.from_fm(heads_fm, rows, head_k))
    want = N(heads)
    np.testing.assert_array_equal(got[:, 256:], want[:, 256:])                    # d raw density, then zeros
    np.testing.assert_allclose(got[:, :256], want[:, :256], rtol=2 ** -6, atol=2e-6)
    ref = (gz.astype(np.float64) @ wb2[:256].astype(np.float64).T)
    np.testing.assert_allclose(got[:, :256], ref, rtol=2 ** -7, atol=1e-6)
