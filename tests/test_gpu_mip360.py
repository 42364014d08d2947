"""GPU parity tests of the MipNeRF-360 kernels (BASELINE config 5) through the C ABI of
include/mip360_hip.h, against oracle/mip360_oracle.py (pinned by the upstream unit-test properties,
tests/test_mip360_oracle.py) on the same seeded inputs.

Tolerances: float32 ray-side kernels (resampling, compositing, losses) 1e-5 relative / 2e-6 absolute on [0,1]
quantities; the IPE features scale their absolute tolerance with the frequency (a float32 argument of 2^k x carries
2^k ulp(x) of phase error in the reference too); the bf16 dense layers are compared with a bf16-operand numpy
restatement at 2e-3 and with the float32 oracle at bf16-grade bounds."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import mip360_oracle as O                                    # noqa: E402
from oracle.nerfpp_oracle import round_bf16                               # noqa: E402


def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def N(t):
    return t.detach().float().cpu().numpy()


@pytest.fixture(scope='module')
def M():
    dev()
    from outdoor_nerf_depth_amd import mip360
    return mip360


def _rays(rs, n):
    d = rs.randn(n, 3).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return dict(origins=(rs.randn(n, 3) * 0.3).astype(np.float32), directions=d, viewdirs=d.copy(),
                radii=np.full((n, 1), 2e-3, np.float32), near=np.full((n, 1), 0.2, np.float32),
                far=np.full((n, 1), 1e6, np.float32))


# ---------------------------------------------------------------------------------------------------- resampling
@pytest.mark.parametrize('jitter', [False, True])
def test_resample_levels_match_oracle(M, jitter):
    """models.py:158-208: level 0 (one unit interval), then dilation 0.0025 + 0.5/1 and 0.0025 + 0.5/64 on 64-bin
    histograms with zero-width bins and zero weights in them."""
    rs = np.random.RandomState(0)
    n = 37
    near, far = np.full((n, 1), 0.2, np.float32), np.full((n, 1), 1e6, np.float32)
    _, s_to_t = O.construct_ray_warps('reciprocal', near, far)
    sd = np.tile(np.array([[0., 1.]], np.float32), (n, 1))
    w = np.ones((n, 1), np.float32)
    prod = 1
    for lvl, ns in enumerate((64, 64, 32)):
        dilation = np.float32(0.0025 + 0.5 / prod)
        prod *= ns
        anneal = np.float32(10 * 0.3 / (9 * 0.3 + 1))
        jit = rs.rand(n, 1).astype(np.float32) if jitter else None
        got_s, got_t = M.resample(T(sd), T(w), float(dilation) if lvl > 0 else 0.0, float(anneal), ns, T(near), T(far),
                                  None if jit is None else T(jit))
        sd_o, w_o = sd, w
        if lvl > 0:
            sd_o, w_o = O.max_dilate_weights(sd, w, dilation, domain=(0., 1.), renormalize=True)
            sd_o, w_o = sd_o[..., 1:-1], w_o[..., 1:-1]
        with np.errstate(divide='ignore'):
            logits = np.where(sd_o[..., 1:] > sd_o[..., :-1], anneal * np.log(w_o), -np.inf).astype(np.float32)
        want = O.sample_intervals(sd_o.astype(np.float32), logits, ns, jit, True, domain=(0., 1.)).astype(np.float32)
        np.testing.assert_allclose(N(got_s), want, rtol=1e-5, atol=3e-6, err_msg='level %d sdist' % lvl)
        # the warp is checked on the kernel's own intervals (1/t = s/far + (1-s)/near amplifies the 3e-6 of sdist by 1/near)
        np.testing.assert_allclose(1.0 / N(got_t), 1.0 / s_to_t(N(got_s)), rtol=2e-6, atol=1e-9, err_msg='level %d 1/tdist' % lvl)
        np.testing.assert_allclose(1.0 / N(got_t), 1.0 / s_to_t(want), rtol=1e-4, atol=3e-5, err_msg='level %d 1/tdist' % lvl)
        assert (np.diff(N(got_s)) >= 0).all()
        # next level's input: the oracle's intervals with a spiky weight vector (exact zeros included)
        sd = want
        w = O.softmax(rs.randn(n, ns) * 3).astype(np.float32)
        w[:, ::9] = 0
        w /= w.sum(-1, keepdims=True)


# ---------------------------------------------------------------------------------------------------- featurisation
def test_cast_encode_matches_oracle(M):
    rs = np.random.RandomState(1)
    n, S = 29, 32
    rays = _rays(rs, n)
    s = np.sort(rs.rand(n, S + 1), -1).astype(np.float32)
    _, s_to_t = O.construct_ray_warps('reciprocal', rays['near'], np.full((n, 1), 50., np.float32))
    tdist = s_to_t(s).astype(np.float32)
    basis = O.pos_basis_t()
    enc = N(M.cast_encode(T(tdist), T(rays['origins']), T(rays['directions']), T(rays['radii']), T(basis), bf16=False))
    assert enc.shape == (n * S, 512) and not enc[:, 504:].any()
    f64 = lambda a: np.asarray(a, np.float64)
    means, covs = O.cast_rays(f64(tdist), f64(rays['origins']), f64(rays['directions']), f64(rays['radii']), 'cone', diag=False)
    m, cv = O.track_linearize_contract(means, covs)
    lm, lv = O.lift_and_diagonalize(m, cv, f64(basis))
    want = O.integrated_pos_enc(lm, lv, 0, 12).reshape(n * S, 504)
    # absolute tolerance grows with the frequency: column k*21 + j (and 252 + ...) has scale 2^k
    scale = np.tile(np.repeat(2.0 ** np.arange(12), 21), 2)
    err = np.abs(enc[:, :504] - want)
    assert (err <= 2e-6 + 3e-6 * scale[None, :] * np.maximum(1.0, np.abs(np.tile(np.repeat(lm.reshape(n * S, 1, 21), 12, 1).reshape(n * S, 252), 2)))).all(), err.max()
    assert err[:, :21 * 4].max() < 2e-5                                # the low degrees are tight
    # bf16 output = rounding of the same values up to the bf16 path's own evaluation error (hardware sine / cosine / exp2 at the
    # degrees 0, 2, 5, 8, angle doubling in between: a few 1e-6 at degree 0, x <= 2.8 per doubling -- the numpy twin in
    # tests/test_layout_emulation.py -- i.e. half the float32 path's frequency-scaled tolerance above, far below the bf16
    # grid), written through a strided view (the MLP's skip buffer)
    buf = torch.full((n * S + 1, 256 + 512), 7.0, dtype=torch.bfloat16, device=dev())
    M.cast_encode(T(tdist), T(rays['origins']), T(rays['directions']), T(rays['radii']), T(basis), out=buf[:, 256:], ld=768)
    got = N(buf[:n * S, 256:256 + 504]).astype(np.float64)
    assert (np.abs(got - enc[:, :504]) <= 2 ** -8 * np.abs(enc[:, :504]) + 0.5 * (2e-6 + 3e-6 * scale[None, :])).all()
    np.testing.assert_allclose(got[:, :21], enc[:, :21], rtol=2 ** -8, atol=8e-6)      # degree 0: the rounding and nothing else
    assert not N(buf[:n * S, 256 + 504:]).any()                        # K padding of the window
    assert (N(buf[:, :256]) == 7).all() and (N(buf[n * S]) == 7).all()   # nothing outside the window is touched


# ---------------------------------------------------------------------------------------------------- dense layers
# (512, 512, 192) and (768, 256, 1024): whole 256 x 256 tiles and K % 64 == 0 -> the persistent ping-pong kernel (more tiles than
# one round would need on a small grid is covered by the MLP tests); the others -> the ring / register-staged kernels
@pytest.mark.parametrize('m,n,k', [(300, 70, 96), (128, 128, 32), (1000, 1, 1024), (257, 1024, 1536), (512, 512, 192), (768, 256, 1024)])
def test_linear_bf16_against_numpy(M, m, n, k):
    rs = np.random.RandomState(m + n)
    a = round_bf16(rs.randn(m, k).astype(np.float32))
    w = round_bf16((rs.randn(n, k) / np.sqrt(k)).astype(np.float32))
    b = rs.randn(n).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64).T + b
    ta, tw = T(a).to(torch.bfloat16), T(w).to(torch.bfloat16)
    for act, fn in ((0, lambda v: v), (1, lambda v: np.maximum(v, 0)), (2, lambda v: np.logaddexp(v - 1.0, 0)),
                    (3, lambda v: 1 / (1 + np.exp(-v)) * 1.002 - 0.001)):
        o32 = torch.empty(m, n, device=dev())
        o16 = torch.empty(m, n + 8, dtype=torch.bfloat16, device=dev())
        M.linear(ta, tw, T(b), act=act, act_param={2: -1.0, 3: 0.001}.get(act, 0.0), out_bf16=o16[:, :n] if n > 1 else None,
                 out_f32=o32)
        np.testing.assert_allclose(N(o32), fn(ref), rtol=2e-5, atol=2e-5)
        if n > 1:
            np.testing.assert_allclose(N(o16[:, :n]), fn(ref), rtol=2 ** -7, atol=1e-3)
            o16b = torch.empty(m, n + 8, dtype=torch.bfloat16, device=dev())      # bf16 output alone: the LDS-staged epilogue
            M.linear(ta, tw, T(b), act=act, act_param={2: -1.0, 3: 0.001}.get(act, 0.0), out_bf16=o16b[:, :n])
            np.testing.assert_allclose(N(o16b[:, :n]), fn(ref), rtol=2 ** -7, atol=1e-3)
    # strided A (a column window of a wider buffer), as the skip / view layers use it
    wide = torch.zeros(m, k + 64, dtype=torch.bfloat16, device=dev())
    wide[:, 64:] = ta
    o32 = torch.empty(m, n, device=dev())
    M.linear(wide[:, 64:], tw, None, act=0, out_f32=o32)
    np.testing.assert_allclose(N(o32), ref - b, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('m,n,k', [(1000, 320, 64), (512, 1024, 1024), (77, 256, 32)])
def test_linear_relu_bit_mask_equals_the_saved_activation_path(M, m, n, k):
    """mip360_linear_relu_mask_bf16 / mip360_linear_masked_bf16 (one bit per element) against act 1 / act 4 (mask re-derived
    from the saved bf16 output): identical bf16 results bit for bit, mask bits = (output > 0); ragged M and N tiles."""
    rs = np.random.RandomState(m + n + k)
    ta = T(round_bf16(rs.randn(m, k).astype(np.float32))).to(torch.bfloat16)
    tw = T(round_bf16((rs.randn(n, k) / np.sqrt(k)).astype(np.float32))).to(torch.bfloat16)
    b = T(rs.randn(n).astype(np.float32) * 0.3)
    h_ref = torch.empty(m, n, dtype=torch.bfloat16, device=dev())
    M.linear(ta, tw, b, act=1, out_bf16=h_ref)
    h = torch.zeros(m, n, dtype=torch.bfloat16, device=dev())
    mask, ld = M.relu_mask_buffer(m, n, dev())
    mask.zero_()
    M.linear_relu_mask(ta, tw, b, h, mask, ld)
    assert torch.equal(h.view(torch.int16), h_ref.view(torch.int16))
    bits = N(mask).reshape(-1, ld)[:(n + 7) // 8, :m]                       # [column byte, row]
    want = np.zeros(((n + 7) // 8 * 8, m), np.uint8)
    want[:n] = (N(h_ref.float()) > 0).T
    want = np.packbits(want.reshape(-1, 8, m), axis=1, bitorder='little')[:, 0]
    np.testing.assert_array_equal(bits, want)
    # dX step: dz [m, kk] times a [n, kk] transposed kernel, masked by this layer's pattern
    kk = 128 if m % 256 == 0 else 96                                       # 128: act 4 and the bit-mask dX both on the ping-pong kernel
    dz = T(round_bf16(rs.randn(m, kk).astype(np.float32))).to(torch.bfloat16)
    wb = T(round_bf16(rs.randn(n, kk).astype(np.float32))).to(torch.bfloat16)
    want16 = torch.empty(m, n, dtype=torch.bfloat16, device=dev())
    M.linear(dz, wb, None, act=4, out_bf16=want16, aux=h_ref)
    got16 = torch.empty(m, n, dtype=torch.bfloat16, device=dev())
    M.linear_masked(dz, wb, got16, mask, ld)
    assert torch.equal((got16.float() + 0.0).view(torch.int32), (want16.float() + 0.0).view(torch.int32))    # -0 == +0
    assert float((got16 != 0).float().mean()) > 0.2


def _mlp_bf16_reference(params, cfg, enc504, viewdirs_rows):
    """MLP.__call__ with every dense layer's operands rounded to bfloat16 (what the matrix cores see), float64 sums."""
    c = dict(O.MLP_DEFAULTS, **cfg)
    r = lambda a: round_bf16(np.asarray(a, np.float32)).astype(np.float64)
    x = r(enc504)
    inputs = x
    k = 0
    for i in range(c['net_depth']):
        W, b = params[k]; k += 1
        x = r(np.maximum(x @ r(W) + b, 0))
        if i % c['skip_layer'] == 0 and i > 0:
            x = np.concatenate([x, inputs], -1)
    W, b = params[k]; k += 1
    density = np.logaddexp((x @ r(W) + b)[..., 0] - 1.0, 0)
    if c['disable_rgb']:
        return density, None
    W, b = params[k]; k += 1
    bott = r(x @ r(W) + b)
    x = np.concatenate([bott, r(O.pos_enc(viewdirs_rows, 0, 4, True))], -1)
    W, b = params[k]; k += 1
    x = r(np.maximum(x @ r(W) + b, 0))
    W, b = params[k]; k += 1
    rgb = 1 / (1 + np.exp(-(x @ r(W) + b))) * 1.002 - 0.001
    return density, rgb


@pytest.mark.parametrize('which', ['prop', 'nerf'])
def test_mlp_forward_matches_bf16_reference_and_oracle(M, which):
    rs = np.random.RandomState(3)
    n, S = 11, 32
    cfg = O.PROP_CFG if which == 'prop' else O.NERF_CFG
    params = O.init_mlp_params(cfg, rs)
    params = [(w, (rs.randn(*b.shape) * 0.1).astype(np.float32)) for w, b in params]     # non-zero biases
    rays = _rays(rs, n)
    s = np.sort(rs.rand(n, S + 1), -1).astype(np.float32)
    _, s_to_t = O.construct_ray_warps('reciprocal', rays['near'], np.full((n, 1), 30., np.float32))
    tdist = s_to_t(s).astype(np.float32)
    basis = O.pos_basis_t()
    pk = M.PackedMLP(params, M.PROP_CFG if which == 'prop' else M.NERF_CFG, dev())
    W = cfg['net_width']
    buf = torch.empty(n * S, W + 512, dtype=torch.bfloat16, device=dev())
    M.cast_encode(T(tdist), T(rays['origins']), T(rays['directions']), T(rays['radii']), T(basis), out=buf[:, W:], ld=W + 512)
    enc = N(buf[:, W:W + 504])
    density, rgb = M.mlp_forward(pk, buf, n * S, T(rays['viewdirs']), n, S)
    vd_rows = np.repeat(rays['viewdirs'], S, 0)
    d_ref, rgb_ref = _mlp_bf16_reference(params, cfg, enc, vd_rows)
    # same operands, float32 vs float64 sums: identical except where a hidden unit sits on a bf16 rounding boundary
    np.testing.assert_allclose(N(density), d_ref, rtol=2e-2, atol=1e-4)
    assert (np.abs(N(density) - d_ref) <= 1e-4 * np.abs(d_ref) + 1e-6).mean() > 0.75
    means, covs = O.cast_rays(tdist, rays['origins'], rays['directions'], rays['radii'], 'cone', diag=False)
    full = O.mlp_forward(params, cfg, means, covs, rays['viewdirs'], basis)
    np.testing.assert_allclose(N(density).reshape(n, S), full['density'], rtol=5e-2, atol=5e-3)
    if which == 'nerf':
        np.testing.assert_allclose(N(rgb), rgb_ref, rtol=0, atol=1.5e-2)              # isolated bf16 rounding flips
        assert (np.abs(N(rgb) - rgb_ref) <= 1e-3).mean() > 0.9
        np.testing.assert_allclose(N(rgb).reshape(n, S, 3), full['rgb'], rtol=0, atol=2e-2)
    else:
        assert rgb is None


# ---------------------------------------------------------------------------------------------------- compositing
@pytest.mark.parametrize('S,opaque', [(32, True), (64, True), (17, False)])
def test_render_level_forward_and_backward(M, S, opaque):
    rs = np.random.RandomState(S)
    n = 23
    density = np.exp(rs.randn(n, S)).astype(np.float32)
    rgbs = rs.rand(n, S, 3).astype(np.float32)
    tdist = (0.2 + np.cumsum(np.exp(rs.randn(n, S + 1) * 0.5) * 0.1, -1)).astype(np.float32)
    dirs = rs.randn(n, 3).astype(np.float32)
    r = M.render_level(T(density), T(rgbs), T(tdist), T(dirs), opaque, 1.0)
    w_o = O.compute_alpha_weights(density, tdist, dirs, opaque)[0]
    rend = O.volumetric_rendering(rgbs, w_o, tdist, 1.0, np.full((n, 1), 1e6, np.float32))
    np.testing.assert_allclose(N(r['weights']), w_o, rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(N(r['rgb']), rend['rgb'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(N(r['acc']), rend['acc'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(N(r['distance_mean']), rend['distance_mean'], rtol=3e-5)
    np.testing.assert_allclose(N(r['depth']), rend['depth'], rtol=3e-5)
    # backward against float64 finite differences of the oracle's forward
    g_w, g_rgb, g_dm = rs.randn(n, S).astype(np.float32), rs.randn(n, 3).astype(np.float32), rs.randn(n).astype(np.float32)
    gd, grgbs = M.render_level_backward(T(density), T(rgbs), T(tdist), T(dirs), T(g_w), T(g_rgb), T(g_dm), opaque, 1.0)

    def scalar(dens, cols):
        w = O.compute_alpha_weights(dens, tdist.astype(np.float64), dirs.astype(np.float64), opaque)[0]
        rr = O.volumetric_rendering(cols, w, tdist.astype(np.float64), 1.0, np.full((n, 1), 1e6))
        return (w * g_w).sum() + (rr['rgb'] * g_rgb).sum() + (rr['distance_mean'] * g_dm).sum()

    d64, c64 = density.astype(np.float64), rgbs.astype(np.float64)
    num = np.zeros_like(d64)
    for i in range(0, n, 5):
        for s_ in range(S):
            dp, dm_ = d64.copy(), d64.copy()
            h = 1e-6 * max(1.0, d64[i, s_])
            dp[i, s_] += h
            dm_[i, s_] -= h
            num[i, s_] = (scalar(dp, c64) - scalar(dm_, c64)) / (2 * h)
    sel = slice(0, n, 5)
    np.testing.assert_allclose(N(gd)[sel], num[sel], rtol=2e-3, atol=2e-5 * np.abs(num[sel]).max())
    np.testing.assert_allclose(N(grgbs), w_o[..., None] * g_rgb[:, None, :], rtol=2e-5, atol=1e-7)


# ---------------------------------------------------------------------------------------------------- losses
@pytest.mark.parametrize('depth_kind', ['mse', 'l1'])
def test_losses_and_gradients_match_oracle(M, depth_kind):
    rs = np.random.RandomState(7)
    n, Sn, Sp = 41, 32, 64
    mk_s = lambda S: np.sort(rs.rand(n, S + 1), -1).astype(np.float32)
    mk_w = lambda S: (O.softmax(rs.randn(n, S) * 2) * rs.uniform(0.5, 1.0, (n, 1))).astype(np.float32)
    sd_n, w_n = mk_s(Sn), mk_w(Sn)
    sd_p, w_p = [mk_s(Sp), mk_s(Sp)], [mk_w(Sp), mk_w(Sp)]
    rgb, gt = rs.rand(n, 3).astype(np.float32), rs.rand(n, 3).astype(np.float32)
    dm = rs.uniform(1, 6, n).astype(np.float32)
    sup = np.where(rs.rand(n) < .6, rs.uniform(1, 6, n), 0).astype(np.float32)
    dm_p = [rs.uniform(1, 6, n).astype(np.float32) for _ in range(2)]
    sc, g_rgb, g_dm, g_wn, g_wp, g_dmp = M.losses(T(rgb), T(gt), T(dm), T(sup), T(sd_n), T(w_n), [T(x) for x in sd_p],
                                                  [T(x) for x in w_p], depth_loss_type=depth_kind, lambda_depth=0.1,
                                                  depth_weight=2.0, dm_prop=[T(x) for x in dm_p], prop_depth_weight=1.0)
    rend = [dict(rgb=rgb, distance_mean=dm_p[0]), dict(rgb=rgb, distance_mean=dm_p[1]), dict(rgb=rgb, distance_mean=dm)]
    hist = [dict(sdist=sd_p[0], weights=w_p[0]), dict(sdist=sd_p[1], weights=w_p[1]), dict(sdist=sd_n, weights=w_n, tdist=sd_n)]
    data_loss, st = O.compute_data_loss(gt, sup, rend, hist, np.ones((n, 3), np.float32), depth_loss_type=depth_kind,
                                        lambda_depth=0.1)
    inter, dist = O.interlevel_loss(hist), O.distortion_loss(hist)
    total = data_loss + 0.1 * st['depth_losses'].sum() + inter + dist      # + stats['loss_disp_mse'] (train_utils.py:143, :268-269)
    s = N(sc)
    np.testing.assert_allclose(s[1], np.sqrt((rgb - gt) ** 2 + 1e-6).mean(), rtol=1e-5)
    np.testing.assert_allclose(s[2], st['depth_losses'][-1], rtol=1e-5)
    np.testing.assert_allclose(s[3], inter, rtol=2e-5)
    np.testing.assert_allclose(s[4], dist, rtol=2e-5)
    np.testing.assert_allclose(s[0], total, rtol=2e-5)
    # gradients: closed forms of the oracle (themselves checked by finite differences in tests/test_mip360_oracle.py)
    resid = rgb - gt
    np.testing.assert_allclose(N(g_rgb), resid / np.sqrt(resid ** 2 + 1e-6) / (3 * n), rtol=1e-5, atol=1e-9)
    m = (sup > 0).astype(np.float32)
    diff = m * dm - m * sup
    want = 0.2 * (2 * diff if depth_kind == 'mse' else np.sign(diff)) * m / n
    np.testing.assert_allclose(N(g_dm), want, rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(s[5], st['depth_losses'][:-1].sum(), rtol=1e-5)
    for k in range(2):
        dk = m * dm_p[k] - m * sup
        np.testing.assert_allclose(N(g_dmp[k]), 0.1 * (2 * dk if depth_kind == 'mse' else np.sign(dk)) * m / n, rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(N(g_wn), 0.01 * O.lossfun_distortion_grad_w(sd_n, w_n) / n, rtol=2e-5, atol=1e-9)
    for k in range(2):
        np.testing.assert_allclose(N(g_wp[k]), O.lossfun_outer_grad_w_env(sd_n, w_n, sd_p[k], w_p[k]) / (n * Sn), rtol=2e-5,
                                   atol=1e-9)


# ---------------------------------------------------------------------------------------------------- whole model
def test_model_forward_matches_oracle(M):
    """Model.__call__ for configs/360.gin (64 / 64 / 32 samples) against the oracle with the same per-level jitter.
    The dense layers run in bf16, and later levels re-sample from earlier weights, so the comparison is bf16-grade;
    level 0's intervals do not depend on any MLP and are float32-tight."""
    rs = np.random.RandomState(5)
    n = 9
    rays = _rays(rs, n)
    prop, nerf = O.init_mlp_params(O.PROP_CFG, rs), O.init_mlp_params(O.NERF_CFG, rs)
    jit = [rs.rand(n, 1).astype(np.float32) for _ in range(3)]
    model = M.Mip360Model(prop, nerf, dev())
    rend, hist = model.forward({k: T(v) for k, v in rays.items()}, train_frac=0.3, jitter01=[T(j) for j in jit])
    rend_o, hist_o = O.model_forward(prop, nerf, rays, train_frac=0.3, jitter01=jit)
    assert [h['weights'].shape[1] for h in hist] == [64, 64, 32]
    np.testing.assert_allclose(N(hist[0]['sdist']), hist_o[0]['sdist'], rtol=1e-5, atol=3e-6)
    np.testing.assert_allclose(N(hist[0]['weights']), hist_o[0]['weights'], rtol=0, atol=2e-2)
    for lvl in (1, 2):
        assert np.abs(N(hist[lvl]['sdist']) - hist_o[lvl]['sdist']).max() < 2e-2
        np.testing.assert_allclose(N(hist[lvl]['weights']).sum(-1), 1, atol=1e-4)
    np.testing.assert_allclose(N(rend[-1]['rgb']), rend_o[-1]['rgb'], rtol=0, atol=3e-2)
    for h in hist:
        s = N(h['sdist'])
        assert (np.diff(s) >= 0).all() and s.min() >= 0 and s.max() <= 1


# ---------------------------------------------------------------------------------------------------- training side
@pytest.mark.parametrize('m,n_in,n_out,ldz', [(1000, 512, 1024, 1024), (777, 128, 3, 32), (4096, 1536, 256, 288), (300, 256, 1, 32),
                                               (8192, 1024, 1024, 1024), (2080, 256, 512, 512)])
def test_grad_weight_and_bias_against_numpy(M, m, n_in, n_out, ldz):
    """dK = H^T dZ (split-K MFMA with transposed LDS reads) and db = column sums, bf16 operands, float32 result."""
    rs = np.random.RandomState(m)
    h = round_bf16(rs.randn(m, n_in).astype(np.float32))
    dz_full = np.zeros((m, ldz), np.float32)
    dz_full[:, :n_out] = round_bf16(rs.randn(m, n_out).astype(np.float32))
    th, tz = T(h).to(torch.bfloat16), T(dz_full).to(torch.bfloat16)
    out = torch.empty(n_in, n_out, device=dev())
    scratch = [None, None]
    fused_b = torch.empty(n_out, device=dev())
    M._grad_weight(th, tz, n_in, n_out, out, scratch, fused_b)
    ref = h.astype(np.float64).T @ dz_full[:, :n_out].astype(np.float64)
    np.testing.assert_allclose(N(out), ref, rtol=2e-4, atol=2e-4 * np.sqrt(m))
    np.testing.assert_allclose(N(fused_b), dz_full[:, :n_out].astype(np.float64).sum(0), rtol=2e-4, atol=1e-3)
    b = torch.empty(n_out, device=dev())
    M._grad_bias(tz, n_out, b, scratch)
    np.testing.assert_allclose(N(b), dz_full[:, :n_out].astype(np.float64).sum(0), rtol=2e-4, atol=1e-3)


def _mlp_bf16_fwd_bwd(params, cfg, enc504, vd_rows, g_d, g_c):
    """Forward + closed-form backward of MLP.__call__ in float64 with bfloat16 rounding at exactly the points where the
    HIP path rounds (every GEMM operand: activations after ReLU, the bottleneck, dZ after its mask, the head
    derivatives).  Returns the per-tensor (d kernel, d bias) list."""
    c = dict(O.MLP_DEFAULTS, **cfg)
    r = lambda a: round_bf16(np.asarray(a, np.float32)).astype(np.float64)
    D, W = c['net_depth'], c['net_width']
    Ws = [r(w) for w, _ in params]
    bs = [np.asarray(b, np.float64) for _, b in params]
    x = r(enc504)
    inputs = x
    ins, Hs = [], []
    for i in range(D):
        ins.append(x)
        h = r(np.maximum(x @ Ws[i] + bs[i], 0))
        Hs.append(h)
        x = np.concatenate([h, inputs], -1) if (i % c['skip_layer'] == 0 and i > 0) else h
    density = np.logaddexp((x @ Ws[D] + bs[D])[:, 0] - 1.0, 0).astype(np.float32).astype(np.float64)
    grads = [None] * len(params)
    d_raw = r(g_d.reshape(-1) * (1 - np.exp(-density)))[:, None]
    grads[D] = (x.T @ d_raw, d_raw.sum(0))
    d_trunk = d_raw @ Ws[D].T
    if not c['disable_rgb']:
        bott = r(x @ Ws[D + 1] + bs[D + 1])
        vin = np.concatenate([bott, r(O.pos_enc(vd_rows, 0, 4, True))], -1)
        h = r(np.maximum(vin @ Ws[D + 2] + bs[D + 2], 0))
        s_ = 1 / (1 + np.exp(-(h @ Ws[D + 3] + bs[D + 3])))
        rgb = (s_ * 1.002 - 0.001).astype(np.float32).astype(np.float64)
        s2 = (rgb + 0.001) / 1.002
        d_pre = r(g_c.reshape(-1, 3) * 1.002 * s2 * (1 - s2))
        grads[D + 3] = (h.T @ d_pre, d_pre.sum(0))
        d_hz = r((d_pre @ Ws[D + 3].T) * (h > 0))
        grads[D + 2] = (vin.T @ d_hz, d_hz.sum(0))
        d_bott = r((d_hz @ Ws[D + 2].T)[:, :256])
        grads[D + 1] = (x.T @ d_bott, d_bott.sum(0))
        d_trunk = d_trunk + d_bott @ Ws[D + 1].T
    dz = r(d_trunk[:, :W] * (Hs[D - 1] > 0))
    for i in reversed(range(D)):
        grads[i] = (ins[i].T @ dz, dz.sum(0))
        if i > 0:
            dz = r((dz @ Ws[i].T)[:, :W] * (Hs[i - 1] > 0))
    return grads


@pytest.mark.parametrize('which', ['prop', 'nerf'])
def test_mlp_backward_matches_oracle(M, which):
    """Parameter gradients of one MLP (dX chain with ReLU masks, stacked head GEMM, weight / bias gradient GEMMs):
    tight against a float64 restatement that rounds to bfloat16 where the kernels do (structure check), bf16-grade
    against the oracle's closed-form float64 backward (oracle/mip360_oracle.py: mlp_backward) on the same frustums."""
    rs = np.random.RandomState(11)
    n, S = 16, 32
    cfg = O.PROP_CFG if which == 'prop' else O.NERF_CFG
    params = O.init_mlp_params(cfg, rs)
    params = [(w, (rs.randn(*b.shape) * 0.05).astype(np.float32)) for w, b in params]
    rays = _rays(rs, n)
    s = np.sort(rs.rand(n, S + 1), -1).astype(np.float32)
    _, s_to_t = O.construct_ray_warps('reciprocal', rays['near'], np.full((n, 1), 30., np.float32))
    tdist = s_to_t(s).astype(np.float32)
    basis = O.pos_basis_t()
    tm = M.TrainableMLP(params, M.PROP_CFG if which == 'prop' else M.NERF_CFG, dev())
    W = cfg['net_width']
    rows = n * S
    buf = torch.empty(rows, W + 512, dtype=torch.bfloat16, device=dev())
    M.cast_encode(T(tdist), T(rays['origins']), T(rays['directions']), T(rays['radii']), T(basis), out=buf[:, W:], ld=W + 512)
    enc = N(buf[:, W:W + 504])
    density, rgb, saved = M.mlp_forward_train(tm, buf, rows, T(rays['viewdirs']), n, S)
    g_d = rs.randn(n, S).astype(np.float32)
    g_c = rs.randn(n, S, 3).astype(np.float32) if which == 'nerf' else None
    M.mlp_backward(tm, saved, rows, T(g_d).reshape(-1), None if g_c is None else T(g_c).reshape(-1, 3), [None, None])
    ref16 = _mlp_bf16_fwd_bwd(params, cfg, enc, np.repeat(rays['viewdirs'], S, 0), g_d, g_c)
    means, covs = O.cast_rays(tdist.astype(np.float64), rays['origins'].astype(np.float64), rays['directions'].astype(np.float64),
                              rays['radii'].astype(np.float64), 'cone', diag=False)
    cache = {}
    p64 = [(w.astype(np.float64), b.astype(np.float64)) for w, b in params]
    out = O.mlp_forward(p64, cfg, means, covs, rays['viewdirs'].astype(np.float64), basis.astype(np.float64), cache=cache)
    np.testing.assert_allclose(N(density).reshape(n, S), out['density'], rtol=5e-2, atol=5e-3)
    ref64 = O.mlp_backward(p64, cache, g_d.astype(np.float64), None if g_c is None else g_c.astype(np.float64))
    rel = lambda a, b: np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
    for t in range(len(params)):
        mine_k, mine_b = N(tm.kernel(t, tm.grads)).astype(np.float64), N(tm.bias(t, tm.grads)).astype(np.float64)
        assert rel(mine_k, ref16[t][0]) < 3e-2, ('kernel vs bf16 reference', t, rel(mine_k, ref16[t][0]))
        assert rel(mine_b, ref16[t][1]) < 3e-2 or np.abs(mine_b - ref16[t][1]).max() < 1e-2 * np.abs(ref16[t][1]).max() + 1e-3, ('bias', t)
        assert rel(mine_k, ref64[t][0]) < 0.25, ('kernel vs float64 oracle', t, rel(mine_k, ref64[t][0]))


def test_trainer_steps_reduce_the_loss_and_are_deterministic(M):
    """train_utils.create_train_step end to end on a toy target: 25 steps with fixed rays and jitter bring the total
    loss down, every scalar stays finite, the clipping multiplier equals min(1, max_norm / ||g||), and two trainers fed
    the same inputs end with bit-identical parameters (fixed-order reductions everywhere)."""
    rs = np.random.RandomState(2)
    n = 64
    rays = {k: T(v) for k, v in _rays(rs, n).items()}
    gt = T(np.tile(np.array([[0.2, 0.5, 0.7]], np.float32), (n, 1)))
    sup = T(np.where(rs.rand(n) < .5, rs.uniform(1, 4, n), 0).astype(np.float32))
    jit = [T(rs.rand(n).astype(np.float32)) for _ in range(3)]
    init = (O.init_mlp_params(O.PROP_CFG, np.random.RandomState(0)), O.init_mlp_params(O.NERF_CFG, np.random.RandomState(1)))
    finals = []
    for rep in range(2):
        tr = M.Mip360Trainer(init[0], init[1], dev(), max_steps=2000, grad_max_norm=0.0)
        hist = []
        for _ in range(25):
            sc = tr.train_step(rays, gt, sup, jitter01=jit)
            hist.append(N(sc))
        hist = np.array(hist)
        assert np.isfinite(hist).all()
        assert hist[-5:, 1].mean() < 0.6 * hist[:3, 1].mean(), hist[:, 1]          # the data term goes down
        finals.append((N(tr.nerf.flat), N(tr.prop.flat)))
    np.testing.assert_array_equal(finals[0][0], finals[1][0])
    np.testing.assert_array_equal(finals[0][1], finals[1][1])
    tr = M.Mip360Trainer(init[0], init[1], dev(), max_steps=2000, grad_max_norm=0.001)
    before = N(tr.nerf.flat).copy()
    tr.train_step(rays, gt, sup, jitter01=jit)
    clip = N(tr.clip)
    norm = np.linalg.norm(N(tr.nerf.grads).astype(np.float64))
    np.testing.assert_allclose(clip[0, 1], norm, rtol=1e-4)
    np.testing.assert_allclose(clip[0, 0], min(1.0, 0.001 / (np.finfo(np.float32).eps + norm)), rtol=1e-4)
    # first Adam step with clipped gradients: |delta| = lr * |g| / (|g| + eps_hat) <= lr
    delta = np.abs(N(tr.nerf.flat) - before)
    assert delta.max() <= M.learning_rate(1, max_steps=2000) * 1.001 and delta.max() > 0


# ------------------------------------------------------------------------------------------- data parallel (pmean)
def _dp_inputs(rank, n=48):
    rs = np.random.RandomState(100 + rank)
    rays = _rays(rs, n)
    gt = rs.rand(n, 3).astype(np.float32)
    sup = np.where(rs.rand(n) < .5, rs.uniform(1, 4, n), 0).astype(np.float32)
    jit = [rs.rand(n).astype(np.float32) for _ in range(3)]
    return rays, gt, sup, jit


def _dp_init():
    return O.init_mlp_params(O.PROP_CFG, np.random.RandomState(0)), O.init_mlp_params(O.NERF_CFG, np.random.RandomState(1))


def _dp_worker(rank, world, port, out_dir, backend):
    import os
    import torch.distributed as dist
    from outdoor_nerf_depth_amd import mip360 as M3
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group(backend, rank=rank, world_size=world)
    d = torch.device('cuda:%d' % (rank if backend == 'nccl' else 0))
    torch.cuda.set_device(d)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)
    prop, nerf = _dp_init()
    tr = M3.Mip360Trainer(prop, nerf, d, max_steps=2000, world_size=world)
    rays, gt, sup, jit = _dp_inputs(rank)
    rays = {k: to(v) for k, v in rays.items()}
    for _ in range(2):
        tr.train_step(rays, to(gt), to(sup), jitter01=[to(j) for j in jit])
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, 'mip360_rank%d.npy' % rank),
            np.concatenate([tr.nerf.flat.cpu().numpy(), tr.prop.flat.cpu().numpy()]))
    dist.barrier()
    dist.destroy_process_group()


def _dp_run_and_check(M, tmp_path, backend):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path), backend), nprocs=2, join=True)
    got = [np.load(str(tmp_path / ('mip360_rank%d.npy' % r))) for r in range(2)]
    np.testing.assert_array_equal(got[0], got[1])                       # every rank applies the identical update
    # single process: each rank's gradients from its own trainer, averaged by hand (train_utils.py:340-342), then the
    # same clip + Adam
    prop, nerf = _dp_init()
    trs = [M.Mip360Trainer(prop, nerf, dev(), max_steps=2000) for _ in range(3)]
    main = trs[2]
    for step in range(2):
        for r in range(2):
            rays, gt, sup, jit = _dp_inputs(r)
            t = trs[r]
            t.nerf.flat.copy_(main.nerf.flat); t.prop.flat.copy_(main.prop.flat)
            t.nerf.repack(); t.prop.repack()
            t.step = main.step
            apply = t.apply_gradients
            t.overlap_update = False                                    # (the overlapped path calls _apply_one itself)
            t.apply_gradients = lambda: None                            # gradients only
            t.train_step({k: T(v) for k, v in rays.items()}, T(gt), T(sup), jitter01=[T(j) for j in jit])
            t.apply_gradients = apply
        main.step += 1
        for name in ('nerf', 'prop'):
            g = getattr(trs[0], name).grads + getattr(trs[1], name).grads
            getattr(main, name).grads.copy_(g.div_(2))
        main.apply_gradients()
    want = np.concatenate([N(main.nerf.flat), N(main.prop.flat)])
    np.testing.assert_array_equal(got[0], want)


def test_two_ranks_pmean_matches_hand_averaged_gradients_gloo(M, tmp_path):
    """train_utils.py:340-342 (jax.lax.pmean over the device axis): two ranks sharing cuda:0, gloo all-reduce."""
    _dp_run_and_check(M, tmp_path, 'gloo')


def test_two_ranks_pmean_rccl(M, tmp_path):
    """The same over RCCL (backend nccl), one GPU per rank."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs')
    _dp_run_and_check(M, tmp_path, 'nccl')
