"""the MipNeRF-360 oracle (oracle/mip360_oracle.py) pinned by the reference project's OWN unit
tests.  nerf-methods/mipnerf360 is JAX and cannot be imported here, so every test below re-runs, with numpy RNG,
the closed-form / brute-force / round-trip property of the reference test it cites (mipnerf360/tests/*_test.py) --
same constructions, same tolerances -- against the numpy restatement.  `generate_basis` is additionally compared
with the output of the imported reference (pure numpy upstream), tests/golden/mip360_basis.npz.

The closed-form gradients the HIP kernels need (upstream: jax autograd) are checked against float64 finite
differences.  The depth-loss additions (train_utils.py:108-129, internal/depth_loss.py) have no upstream tests:
PARITY UNPINNED, self-consistency only."""
import numpy as np
import pytest

from oracle import mip360_oracle as M


def stable_pos_enc(x, n):
    """coord_test.py:34-44: posenc by repeated doubling of a rotation matrix (float64), exact at any degree."""
    s, c = np.sin(x), np.cos(x)
    rot = np.array([[c, -s], [s, c]], np.float64)
    out = []
    for _ in range(n):
        out.append(rot[::-1, 0, :])
        rot = np.einsum('ijn,jkn->ikn', rot, rot)
    return np.reshape(np.transpose(np.stack(out, 0), [2, 1, 0]), [-1, 2 * n])


# ------------------------------------------------------------------------------------------------ geopoly
def test_generate_basis_matches_imported_reference(golden):
    g = golden('mip360_basis')
    for tess in (1, 2, 3):
        np.testing.assert_array_equal(M.generate_basis('icosahedron', tess), g['icosahedron_%d' % tess])
    b = M.pos_basis_t()
    assert b.shape == (3, 21) and b.dtype == np.float32
    np.testing.assert_allclose(np.linalg.norm(b, axis=0), 1, atol=1e-6)


# ------------------------------------------------------------------------------------------------ coord
def test_contract_matches_special_case():
    """coord_test.py:61-69 (Figure 2 of arXiv:2111.12077): uniform s -> contracted t is uniformly spaced."""
    n = 10
    _, s_to_t = M.construct_ray_warps('reciprocal', np.float32(1), np.float32(np.inf))
    s = np.linspace(0, 1 - np.finfo(np.float32).eps, n + 1).astype(np.float32)
    tc = M.contract(s_to_t(s)[:, None])[:, 0]
    np.testing.assert_allclose(np.diff(tc), np.full(n, 1 / n), atol=1e-5, rtol=1e-5)


def test_contract_is_bounded_and_noop_inside_unit_ball():
    """coord_test.py:71-92."""
    rs = np.random.RandomState(0)
    x = np.where(rs.rand(10000, 3) < .5, 1, -1) * np.exp(rs.uniform(-10, 10, (10000, 3)))
    assert np.max(M.contract(x.astype(np.float32))) <= 2
    x = rs.randn(10000, 3).astype(np.float32)
    xc = x / np.maximum(1, np.linalg.norm(x, axis=-1, keepdims=True))
    np.testing.assert_allclose(M.contract(xc), xc, atol=1e-5, rtol=1e-5)


def test_inv_contract_inverts_contract_and_jacobians_are_finite():
    """coord_test.py:94-112 (finite gradients at x = 0; round trip)."""
    x = np.stack(np.meshgrid(*[np.linspace(-4, 4, 11)] * 2), -1).astype(np.float32)
    np.testing.assert_allclose(M.inv_contract(M.contract(x)), x, atol=1e-5, rtol=1e-5)
    assert np.isfinite(M.contract_jacobian(x)).all()


def test_contract_jacobian_matches_finite_differences():
    rs = np.random.RandomState(1)
    x = rs.randn(200, 3) * np.exp(rs.uniform(-2, 3, (200, 1)))
    J = M.contract_jacobian(x)
    h = 1e-6
    for k in range(3):
        e = np.zeros(3); e[k] = h
        num = (M.contract(x + e) - M.contract(x - e)) / (2 * h)
        np.testing.assert_allclose(J[..., :, k], num, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('n,tol', [(5, 1e-5), (10, 1e-4), (15, 0.005), (20, 0.2), (25, 2), (30, 2)])
def test_pos_enc_against_stable_implementation(n, tol):
    """coord_test.py:114-130."""
    x = np.linspace(-np.pi, np.pi, 10001)
    z = M.pos_enc(x[:, None].astype(np.float32), 0, n, append_identity=False)
    assert np.max(np.abs(z - stable_pos_enc(x, n))) < tol


def test_pos_enc_matches_integrated_with_zero_variance():
    """coord_test.py:132-143."""
    x = np.linspace(-np.pi, np.pi, 10000).astype(np.float32)
    z_ipe = M.integrated_pos_enc(x, np.zeros_like(x), 0, 10)
    z_pe = M.pos_enc(x, 0, 10, append_identity=False)
    np.testing.assert_allclose(z_pe, z_ipe, atol=1e-4)


def test_track_linearize_on_affine_maps():
    """coord_test.py:145-178: pushing Gaussians through an affine map must give (A mu + b, A cov A^T); the oracle's
    Jacobian formulation J cov J^T is exercised with J = A."""
    rs = np.random.RandomState(0)
    for _ in range(30):
        din, dout = rs.randint(1, 10), rs.randint(1, 10)
        mean = rs.randn(20, din)
        half = rs.randn(20, din, din)
        cov = half @ np.swapaxes(half, -1, -2)
        A, b = rs.randn(dout, din), rs.randn(dout)
        m, c = M.track_linearize_affine(A, b, mean, cov)
        np.testing.assert_allclose(m, (A @ mean[..., None])[..., 0] + b, atol=1e-9)
        np.testing.assert_allclose(c, np.einsum('ij,njk,lk->nil', A, cov, A), atol=1e-9)
    # contract is linear (identity) inside the unit ball: track_linearize_contract is then a no-op
    mean = rs.uniform(-.5, .5, (20, 3))
    half = rs.randn(20, 3, 3) * 0.1
    cov = half @ np.swapaxes(half, -1, -2)
    m, c = M.track_linearize_contract(mean, cov)
    np.testing.assert_allclose(m, mean)
    np.testing.assert_allclose(c, cov, atol=1e-12)


@pytest.mark.parametrize('fn', ['reciprocal', 'log', 'sqrt'])
def test_construct_ray_warps_extents(fn):
    """coord_test.py:180-200."""
    rs = np.random.RandomState(0)
    t_near = np.exp(rs.randn(100))
    t_far = t_near + np.exp(rs.randn(100))
    t_to_s, s_to_t = M.construct_ray_warps(fn, t_near, t_far)
    np.testing.assert_allclose(t_to_s(t_near), 0, atol=1e-5)
    np.testing.assert_allclose(t_to_s(t_far), 1, atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(s_to_t(np.zeros(100)), t_near, atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(s_to_t(np.ones(100)), t_far, atol=1e-5, rtol=1e-5)


def test_construct_ray_warps_special_reciprocal():
    """coord_test.py:202-226: closed form for fn = 1/x."""
    rs = np.random.RandomState(0)
    t_near = np.exp(rs.randn(100))
    t_far = t_near + np.exp(rs.randn(100))
    u, s = rs.rand(100), rs.rand(100)
    t = t_near * (1 - u) + t_far * u
    t_to_s, s_to_t = M.construct_ray_warps('reciprocal', t_near, t_far)
    np.testing.assert_allclose(t_to_s(t), (t_far * (t - t_near)) / (t * (t_far - t_near)), atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(s_to_t(s), 1 / (s / t_far + (1 - s) / t_near), atol=1e-5, rtol=1e-5)


def test_expected_sin_and_integrated_pos_enc_against_sampling():
    """coord_test.py:228-265."""
    rs = np.random.RandomState(0)
    z = rs.randn(100000)
    for mu, var in [(0, 1), (1, 3), (-2, .2), (10, 10)]:
        np.testing.assert_allclose(M.expected_sin(mu, var), np.mean(np.sin(np.sqrt(var) * z + mu)), atol=1e-2)
    for _ in range(5):
        mean = rs.randn(2)
        half = rs.randn(2, 2)
        cov = half @ half.T
        enc = M.integrated_pos_enc(mean, np.diag(cov), 0, 4)
        samples = rs.multivariate_normal(mean, cov, 100000)
        enc_samples = np.concatenate([stable_pos_enc(x, 4) for x in samples.T], -1)
        gt = np.mean(enc_samples, 0).reshape([2, 8]).T.reshape(-1)
        np.testing.assert_allclose(enc, gt, rtol=1e-2, atol=1e-2)


# ------------------------------------------------------------------------------------------------ stepfun
def _inner_ref(t0, t1, w1):
    """stepfun_test.py:26-38 reference loops."""
    return np.array([sum(w1[j] for j in range(len(t1) - 1) if t1[j] >= t0[i] and t1[j + 1] < t0[i + 1])
                     for i in range(len(t0) - 1)])


def _outer_ref(t0, t1, w1):
    return np.array([sum(w1[j] for j in range(len(t1) - 1) if t1[j + 1] >= t0[i] and t1[j] <= t0[i + 1])
                     for i in range(len(t0) - 1)])


def test_searchsorted_in_and_out_of_bounds_and_reference():
    """stepfun_test.py:55-146."""
    rs = np.random.RandomState(0)
    for _ in range(10):
        n, m = rs.randint(10, 100), rs.randint(10, 100)
        v = rs.uniform(1e-7, 1 - 1e-7, n)
        a = np.concatenate([[0.], np.sort(rs.rand(m)), [1.]])
        lo, hi = M.searchsorted(a, v)
        assert (a[lo] <= v).all() and (v < a[hi]).all()
        np.testing.assert_array_equal(np.searchsorted(a, v, side='right'), hi)
        a = np.sort(rs.uniform(1, 2, m))
        for q, want in ((rs.uniform(0., .9, n), 0), (rs.uniform(2.1, 3, n), m - 1)):
            lo, hi = M.searchsorted(a, q)
            assert (lo == want).all() and (hi == want).all()
    a = np.sort(rs.uniform(-4, 4, 10))
    v = rs.uniform(-6, 6, 100)
    lo, hi = M.searchsorted(a, v)
    for x, i0, i1 in zip(v, lo, hi):
        if x < a.min():
            want = (0, 0)
        elif x > a.max():
            want = (9, 9)
        else:
            want = (np.argmax(np.where(x >= a, a, -np.inf)), np.argmin(np.where(x < a, a, np.inf)))
        assert (i0, i1) == want


@pytest.mark.parametrize('mode,delta', [('front', 0.), ('front', .05), ('front', .099), ('back', 1e-6), ('back', .05),
                                        ('back', .099), ('before', 1e-6), ('after', 0.)])
def test_query(mode, delta):
    """stepfun_test.py:148-199."""
    rs = np.random.RandomState(0)
    n, d = 10, 8
    t = -d / 2 + np.cumsum(rs.uniform(0.1, 1., (n, d + 1)), -1)
    y = rs.randn(n, d)
    q = lambda tq: M.query(tq, t, y, outside_value=-10.)
    if mode == 'front':
        np.testing.assert_array_equal(q(t[..., :-1] + delta), y)
    elif mode == 'back':
        np.testing.assert_array_equal(q(t[..., 1:] - delta), y)
    elif mode == 'before':
        np.testing.assert_array_equal(q(t.min(-1)[:, None] + np.linspace(-10, -delta, 100)[None]), -10.)
    else:
        np.testing.assert_array_equal(q(t.max(-1)[:, None] + np.linspace(delta, 10, 100)[None]), -10.)


def test_distortion_loss_against_sampling_and_interval_distortion():
    """stepfun_test.py:201-273."""
    rs = np.random.RandomState(0)
    n, d = 10, 8
    t = np.sort(rs.uniform(-3, 3, (n, d + 1)), -1)
    logits = 2 * rs.randn(n, d)
    w = M.softmax(logits)
    losses = M.lossfun_distortion(t, w)
    samples = M.sample(t, logits, 10000, jitter01=rs.rand(n, 10000), single_jitter=False)
    stoch = np.array([np.mean(np.abs(s[:, None] - s[None, :])) for s in samples])
    np.testing.assert_allclose(losses, stoch, atol=2e-3, rtol=2e-3)       # upstream 1e-4 with 10k jittered samples; ours ~1e-3
    dd = M.interval_distortion(t[..., :-1, None], t[..., 1:, None], t[..., None, :-1], t[..., None, 1:])
    np.testing.assert_allclose(losses, np.sum(w[:, None, :] * w[:, :, None] * dd, (-1, -2)), atol=1e-6, rtol=1e-4)
    # interval_distortion against brute force (stepfun_test.py:227-250)
    t0, t1 = np.sort(rs.uniform(-3, 3, (3, 8)), -1), np.sort(rs.uniform(-3, 3, (3, 8)), -1)
    dist = M.interval_distortion(t0[..., :-1], t0[..., 1:], t1[..., :-1], t1[..., 1:])
    for i in range(3):
        for j in range(7):
            brute = np.mean(np.abs(np.linspace(t0[i, j], t0[i, j + 1], 2001)[:, None] -
                                   np.linspace(t1[i, j], t1[i, j + 1], 2001)[None, :]))
            np.testing.assert_allclose(dist[i, j], brute, atol=1e-5, rtol=2e-3)


def test_max_dilate_is_the_max_of_shifted_queries():
    """stepfun_test.py:275-300."""
    rs = np.random.RandomState(0)
    n, d, dilation = 20, 8, 0.53
    t = np.cumsum(rs.randint(1, 10, (n, d + 1)), -1) / 10
    w = M.softmax(rs.randn(n, d))
    td, wd = M.max_dilate(t, w, dilation)
    tq = (np.arange((d + 4) * 10) - 2.5) / 10
    wq = M.query(np.broadcast_to(tq, (n, tq.size)), t, w)
    wdq = M.query(np.broadcast_to(tq, (n, tq.size)), td, wd)
    mask = np.abs(tq[None, :] - tq[:, None]) <= dilation
    for i in range(n):
        np.testing.assert_array_equal(wdq[i], np.max(mask * wq[i], -1))


@pytest.mark.parametrize('randomized,single_jitter', [(False, None), (True, False), (True, True)])
def test_sample_reproduces_its_distribution(randomized, single_jitter):
    """stepfun_test.py:302-383 (histogram of many samples == the PDF) and :474-495 (single bin)."""
    rs = np.random.RandomState(0)
    num_bins, num_samples = 16, 200000
    for _ in range(3):
        delta = np.round(1e5 * np.exp(rs.uniform(-3, 3, num_bins + 1))) * (rs.rand(num_bins + 1) < 0.9)
        bins = np.cumsum(delta) / 1e5 + rs.randn() * num_bins / 2
        logits = rs.randn(num_bins) * 2
        logits = np.where(np.diff(bins) > 0, logits, -np.inf)
        jit = None if not randomized else rs.rand(1, 1 if single_jitter else num_samples)
        s = M.sample(bins[None], logits[None], num_samples, jitter01=jit, single_jitter=bool(single_jitter))[0]
        assert (np.diff(s) >= 0).all() and s.min() >= bins[0] and s.max() <= bins[-1]
        hist = np.histogram(s, bins=np.unique(bins))[0] / num_samples
        w = M.softmax(logits)
        keep = np.diff(bins) > 0
        np.testing.assert_allclose(hist, w[keep], atol=2e-3)
    s = M.sample(np.array([[0., 1., 3.]]), np.array([[0., -np.inf]]), 1000,
                 jitter01=None if not randomized else rs.rand(1, 1), single_jitter=True)
    assert (s >= 0).all() and (s <= 1).all()


def test_sample_intervals_cover_the_bins_they_came_from():
    """stepfun_test.py:497-586: intervals sampled from a one-bin step function tile that bin; first / last edges are
    clamped to the domain."""
    t = np.array([[1., 2.]])
    out = M.sample_intervals(t, np.array([[0.]]), 10, domain=(1., 2.))
    assert out.shape == (1, 11) and (np.diff(out) > 0).all()
    np.testing.assert_allclose(out[0, [0, -1]], [1., 2.], atol=1e-6)
    np.testing.assert_allclose(np.diff(out)[0, 1:-1], 0.1, atol=1e-5)
    rs = np.random.RandomState(0)
    t = np.sort(rs.rand(4, 9), -1)
    out = M.sample_intervals(t, rs.randn(4, 8), 32, jitter01=rs.rand(4, 1), single_jitter=True, domain=(0., 1.))
    assert (np.diff(out) >= 0).all() and out.min() >= 0 and out.max() <= 1


@pytest.mark.parametrize('num_ablate,is_all_zero', [(0, True), (2, False)])
def test_lossfun_outer(num_ablate, is_all_zero):
    """stepfun_test.py:588-622: a histogram that is an upper envelope of the other gives zero loss."""
    rs = np.random.RandomState(0)
    n, d = 5, 10
    t = np.sort(rs.rand(n, d + 1), -1)
    w = M.softmax(rs.randn(n, d))
    # the envelope: the same step function on a coarser grid (every other edge) -- outer measure >= w
    t_env = t[:, ::2]
    w_env = w.reshape(n, d // 2, 2).sum(-1)
    if num_ablate:
        w_env[:, :num_ablate] = 0
    loss = M.lossfun_outer(t, w, t_env, w_env)
    assert (loss >= 0).all()
    assert (np.abs(loss) < 1e-7).all() == is_all_zero


def test_inner_outer_against_reference_loops_and_self():
    """stepfun_test.py:624-737."""
    rs = np.random.RandomState(0)
    for _ in range(10):
        d0, d1 = rs.randint(2, 12), rs.randint(2, 12)
        t0, t1 = np.sort(rs.rand(d0 + 1)), np.sort(rs.rand(d1 + 1))
        w1 = np.exp(rs.randn(d1))
        inner, outer = M.inner_outer(t0[None], t1[None], w1[None])
        np.testing.assert_allclose(outer[0], _outer_ref(t0, t1, w1), atol=1e-6, rtol=1e-6)
        np.testing.assert_allclose(inner[0], _inner_ref(t0, t1, w1), atol=1e-6, rtol=1e-6)
        assert (inner <= outer + 1e-9).all()
    # :624-655 two histograms of the same points bound each other
    for _ in range(10):
        d0, d1, npts = rs.randint(10, 20, 3)
        t0, t1 = np.sort(rs.rand(d0 + 1)), np.sort(rs.rand(d1 + 1))
        pts = rs.uniform(max(t0.min(), t1.min()) + 0.1, min(t0.max(), t1.max()) - 0.1, npts)
        w0 = np.array([np.sum((pts >= t0[i]) & (pts < t0[i + 1])) for i in range(d0)], np.float64)
        w1 = np.array([np.sum((pts >= t1[i]) & (pts < t1[i + 1])) for i in range(d1)], np.float64)
        i0, o0 = M.inner_outer(t0[None], t1[None], w1[None])
        i1, o1 = M.inner_outer(t1[None], t0[None], w0[None])
        assert (i0[0] <= w0).all() and (w0 <= o0[0]).all() and (i1[0] <= w1).all() and (w1 <= o1[0]).all()
    # :657-697 invariance to monotonic maps of t; zero against itself
    t = np.sort(rs.rand(3, 9), -1)
    w = M.softmax(rs.randn(3, 8))
    t_b, w_b = np.sort(rs.rand(3, 12), -1), np.exp(rs.randn(3, 11))
    np.testing.assert_array_equal(M.lossfun_outer(t, w, t_b, w_b), M.lossfun_outer(1 + t ** 3, w, 1 + t_b ** 3, w_b))
    assert (M.lossfun_outer(t, w, t, w) < 1e-10).all()


def test_weighted_percentile():
    """stepfun_test.py:739-790."""
    rs = np.random.RandomState(0)
    for _ in range(5):
        d = rs.randint(5, 20)
        t = np.sort(rs.randn(d + 1))
        w = M.softmax(rs.randn(d))
        ps = np.linspace(1, 99, 11)
        got = M.weighted_percentile(t[None], w[None], ps)[0]
        s = M.sample(t[None], np.log(w)[None], 200000)[0]
        np.testing.assert_allclose(got, np.percentile(s, ps), atol=5e-3, rtol=5e-3)


@pytest.mark.parametrize('use_avg', [False, True])
def test_resample(use_avg):
    """stepfun_test.py:792-930: resampling onto itself is a no-op; 2x down-sampling sums (or averages) pairs; a
    single interval spanning everything gives the total."""
    rs = np.random.RandomState(0)
    d = 32
    tp = np.sort(rs.rand(4, d + 1), -1)
    vp = rs.rand(4, d)
    np.testing.assert_allclose(M.resample(tp, tp, vp, use_avg=use_avg), vp, atol=1e-6)
    t2 = tp[:, ::2]
    got = M.resample(t2, tp, vp, use_avg=use_avg)
    if use_avg:
        wp = np.diff(tp, axis=-1)
        want = (vp * wp).reshape(4, d // 2, 2).sum(-1) / wp.reshape(4, d // 2, 2).sum(-1)
    else:
        want = vp.reshape(4, d // 2, 2).sum(-1)
    np.testing.assert_allclose(got, want, atol=1e-6)
    whole = M.resample(tp[:, [0, -1]], tp, vp, use_avg=use_avg)[:, 0]
    want = (vp * np.diff(tp, axis=-1)).sum(-1) / np.diff(tp, axis=-1).sum(-1) if use_avg else vp.sum(-1)
    np.testing.assert_allclose(whole, want, atol=1e-6)


# ------------------------------------------------------------------------------------------------ render
def _sample_conical_frustum(rs, n, d, t0, t1, r):
    """render_test.py:66-93: uniform samples inside a conical frustum along direction d."""
    t = (t0 ** 3 + rs.rand(n) * (t1 ** 3 - t0 ** 3)) ** (1 / 3)
    theta = rs.rand(n) * 2 * np.pi
    rad = r * t * np.sqrt(rs.rand(n))
    dn = d / np.linalg.norm(d)
    a = np.cross(dn, [1., 0, 0]) if abs(dn[0]) < .9 else np.cross(dn, [0, 1., 0])
    a /= np.linalg.norm(a)
    b = np.cross(dn, a)
    return t[:, None] * d[None] + np.linalg.norm(d) * rad[:, None] * (np.cos(theta)[:, None] * a + np.sin(theta)[:, None] * b)


def test_conical_frustum_gaussian_matches_sample_moments():
    """render_test.py:279-331: mean / covariance of conical_frustum_to_gaussian == sample moments; the stable and
    unstable parametrisations agree."""
    rs = np.random.RandomState(0)
    for _ in range(5):
        d = rs.randn(3)
        t0 = np.exp(rs.uniform(-1, 1))
        t1 = t0 + np.exp(rs.uniform(-1, 1))
        r = np.exp(rs.uniform(-3, -1))
        mean, cov = M.conical_frustum_to_gaussian(d, np.array([t0]), np.array([t1]), r, diag=False)
        pts = _sample_conical_frustum(rs, 400000, d, t0, t1, r)
        np.testing.assert_allclose(mean[0], pts.mean(0), atol=2e-2 * t1, rtol=2e-2)
        np.testing.assert_allclose(cov[0], np.cov(pts.T), atol=2e-2 * t1 ** 2 * max(1, (d ** 2).sum()), rtol=5e-2)
        m2, c2 = M.conical_frustum_to_gaussian(d, np.array([t0]), np.array([t1]), r, diag=False, stable=False)
        np.testing.assert_allclose(mean, m2, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(cov, c2, rtol=1e-5, atol=1e-9)
        _, cd = M.conical_frustum_to_gaussian(d, np.array([t0]), np.array([t1]), r, diag=True)
        np.testing.assert_allclose(cd[0], np.diag(cov[0]), rtol=1e-9)           # :354-370 lift_gaussian diag == diag(full)


def test_cylinder_gaussian_matches_sample_moments():
    """render_test.py:333-352."""
    rs = np.random.RandomState(0)
    d = rs.randn(3)
    t0, t1, r = 0.7, 1.9, 0.3
    mean, cov = M.cylinder_to_gaussian(d, np.array([t0]), np.array([t1]), r, diag=False)
    n = 400000
    t = t0 + rs.rand(n) * (t1 - t0)
    theta, rad = rs.rand(n) * 2 * np.pi, r * np.sqrt(rs.rand(n))
    dn = d / np.linalg.norm(d)
    a = np.cross(dn, [1., 0, 0]); a /= np.linalg.norm(a)
    b = np.cross(dn, a)
    pts = t[:, None] * d[None] + rad[:, None] * (np.cos(theta)[:, None] * a + np.sin(theta)[:, None] * b)
    np.testing.assert_allclose(mean[0], pts.mean(0), atol=1e-2)
    np.testing.assert_allclose(cov[0], np.cov(pts.T), atol=1e-2)


@pytest.mark.parametrize('log_density_log_mult', [-100, -10, 0, 10])
@pytest.mark.parametrize('tvals_log_mult', [-100, -10, 0, 10])
def test_alpha_weights_finite(log_density_log_mult, tvals_log_mult):
    """render_test.py:407-441 (finite outputs and gradients over 87 decades of density / distance)."""
    rs = np.random.RandomState(0)
    n, d = 20, 128
    density = np.exp(log_density_log_mult + rs.randn(n, d))
    tvals = np.exp(tvals_log_mult) * np.sort(2 * rs.rand(n, d + 1) - 1, -1)
    dirs = rs.randn(n, 3)
    w, a, tr = M.compute_alpha_weights(density, tvals, dirs)
    g = M.alpha_weights_backward(density, tvals, dirs, np.ones_like(w))
    for x in (w, a, tr, g):
        assert np.isfinite(x).all()


def test_alpha_weights_delta_correct():
    """render_test.py:443-460: one interval with a huge density takes all the weight."""
    rs = np.random.RandomState(0)
    n, d = 100, 128
    r = rs.randn(n, d)
    mask = r == r.max(-1, keepdims=True)
    tvals = np.sort(2 * rs.rand(n, d + 1) - 1, -1)
    w, a, _ = M.compute_alpha_weights(1e10 * mask, tvals, rs.randn(n, 3))
    np.testing.assert_allclose(w, mask.astype(np.float64), atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(a, mask.astype(np.float64), atol=1e-5, rtol=1e-5)


# ------------------------------------------------------------------------------------------------ closed-form gradients
def _fd(f, x, h=1e-6):
    g = np.zeros_like(x)
    it = np.nditer(x, flags=['multi_index'])
    for _ in it:
        i = it.multi_index
        xp, xm = x.copy(), x.copy()
        xp[i] += h
        xm[i] -= h
        g[i] = (f(xp) - f(xm)) / (2 * h)
    return g


def test_closed_form_gradients_match_finite_differences():
    rs = np.random.RandomState(0)
    t = np.sort(rs.rand(3, 9), -1)
    w = M.softmax(rs.randn(3, 8))
    np.testing.assert_allclose(M.lossfun_distortion_grad_w(t, w), _fd(lambda x: M.lossfun_distortion(t, x).sum(), w),
                               rtol=1e-6, atol=1e-8)
    t_env = np.sort(rs.rand(3, 7), -1)
    w_env = M.softmax(rs.randn(3, 6)) * 0.7
    np.testing.assert_allclose(M.lossfun_outer_grad_w_env(t, w, t_env, w_env),
                               _fd(lambda x: M.lossfun_outer(t, w, t_env, x).sum(), w_env), rtol=1e-5, atol=1e-8)
    density = np.exp(rs.randn(3, 8))
    dirs = rs.randn(3, 3)
    g_w = rs.randn(3, 8)
    for opaque in (False, True):
        f = lambda x: (M.compute_alpha_weights(x, t, dirs, opaque)[0] * g_w).sum()
        np.testing.assert_allclose(M.alpha_weights_backward(density, t, dirs, g_w, opaque), _fd(f, density),
                                   rtol=1e-5, atol=1e-8)


# ------------------------------------------------------------------------------------------------ model + losses (self-consistency)
def _rays(rs, n):
    d = rs.randn(n, 3).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return dict(origins=(rs.randn(n, 3) * 0.3).astype(np.float32), directions=d, viewdirs=d,
                radii=np.full((n, 1), 2e-3, np.float32), near=np.full((n, 1), 0.2, np.float32),
                far=np.full((n, 1), 1e6, np.float32))


def test_model_forward_shapes_and_invariants():
    """Model.__call__ (models.py:76-303) on configs/360.gin: 64 / 64 / 32 samples, weights sum <= 1 (== 1 with the
    opaque background), sdist inside [0, 1] and sorted, tdist inside [near, far], rgb inside the padded range."""
    rs = np.random.RandomState(0)
    n = 6
    rays = _rays(rs, n)
    prop, nerf = M.init_mlp_params(M.PROP_CFG, rs), M.init_mlp_params(M.NERF_CFG, rs)
    assert [p[0].shape for p in prop] == [(504, 256), (256, 256), (256, 256), (256, 256), (256, 1)]
    assert nerf[5][0].shape == (1024 + 504, 1024) and nerf[8][0].shape == (1024, 1) and nerf[10][0].shape == (256 + 27, 128)
    rend, hist = M.model_forward(prop, nerf, rays, train_frac=0.3, jitter01=[rs.rand(n, 1) for _ in range(3)])
    assert [h['weights'].shape[-1] for h in hist] == [64, 64, 32]
    for h in hist:
        assert (np.diff(h['sdist']) >= 0).all() and h['sdist'].min() >= 0 and h['sdist'].max() <= 1
        assert (h['tdist'] >= 0.2 - 1e-6).all() and np.isfinite(h['tdist'][..., :-1]).all()
        np.testing.assert_allclose(h['weights'].sum(-1), 1, atol=1e-5)
    rgb = rend[-1]['rgb']
    assert rgb.shape == (n, 3) and rgb.min() >= -0.001 - 1e-6 and rgb.max() <= 1.001 + 1e-6
    # determinism without jitter
    r1, _ = M.model_forward(prop, nerf, rays)
    r2, _ = M.model_forward(prop, nerf, rays)
    np.testing.assert_array_equal(r1[-1]['rgb'], r2[-1]['rgb'])
    # losses are finite, interlevel >= 0, distortion >= 0
    gt = rs.rand(n, 3).astype(np.float32)
    sup = np.where(rs.rand(n) < .5, rs.uniform(1, 5, n), 0).astype(np.float32)
    for kind in ('mse', 'l1'):
        loss, stats = M.compute_data_loss(gt, sup, rend, hist, rays['directions'], depth_loss_type=kind)
        assert np.isfinite(loss) and len(stats['depth_losses']) == 3
    assert M.interlevel_loss(hist) >= 0 and M.distortion_loss(hist) >= 0


def test_depth_terms_as_written_upstream():
    """PARITY UNPINNED (no upstream tests): train_utils.py:108-129 -- the masked difference is averaged over ALL rays
    -- and depth_loss.py's `.sum(-2)` quirk."""
    dm = np.array([1., 2., 3., 4.])
    sup = np.array([1.5, 0., 2., 0.])
    rend = [dict(rgb=np.zeros((4, 3)), distance_mean=dm)]
    hist = [dict(weights=np.full((4, 4), .25), tdist=np.tile(np.linspace(1, 2, 5), (4, 1)))]
    _, st = M.compute_data_loss(np.zeros((4, 3)), sup, rend, hist, np.ones((4, 3)), depth_loss_type='mse')
    np.testing.assert_allclose(st['depth_losses'][0], ((1 - 1.5) ** 2 + (3 - 2) ** 2) / 4)
    _, st = M.compute_data_loss(np.zeros((4, 3)), sup, rend, hist, np.ones((4, 3)), depth_loss_type='l1')
    np.testing.assert_allclose(st['depth_losses'][0], (0.5 + 1.0) / 4)
    kl = M.depth_loss(hist[0]['weights'], hist[0]['tdist'], sup, dm, 0.01, np.ones((4, 3)), 'kl')
    assert np.isfinite(kl) and kl > 0


@pytest.mark.parametrize('which', ['prop', 'nerf'])
def test_mlp_backward_matches_finite_differences(which):
    """The closed-form MLP backward (upstream: jax.grad) against float64 central differences, on a narrow copy of the
    360.gin networks (same depth, skip, heads; width 24) so every parameter can be probed."""
    rs = np.random.RandomState(0)
    cfg = dict(M.PROP_CFG if which == 'prop' else M.NERF_CFG, net_width=24, bottleneck_width=8, net_width_viewdirs=6)
    params = [(rs.randn(*w.shape) * 0.3, rs.randn(*b.shape) * 0.1) for w, b in
              [(np.zeros(sh), np.zeros(sh[1])) for sh in M.mlp_param_shapes(cfg)]]
    n, S = 3, 5
    means = rs.randn(n, S, 3) * 1.5
    half = rs.randn(n, S, 3, 3) * 0.05
    covs = half @ np.swapaxes(half, -1, -2)
    vd = rs.randn(n, 3)
    vd /= np.linalg.norm(vd, axis=-1, keepdims=True)
    basis = M.pos_basis_t().astype(np.float64)
    g_d, g_c = rs.randn(n, S), rs.randn(n, S, 3)

    def scalar(ps):
        out = M.mlp_forward(ps, cfg, means, covs, vd, basis)
        return (out['density'] * g_d).sum() + (out['rgb'] * g_c).sum()

    cache = {}
    M.mlp_forward(params, cfg, means, covs, vd, basis, cache=cache)
    grads = M.mlp_backward(params, cache, g_d, None if cfg['disable_rgb'] else g_c)
    for li, (W, b) in enumerate(params):
        for which_p, arr, g in ((0, W, grads[li][0]), (1, b, grads[li][1])):
            idxs = [tuple(rs.randint(0, d) for d in arr.shape) for _ in range(6)]
            for idx in idxs:
                h = 1e-6
                pp = [(w.copy(), bb.copy()) for w, bb in params]
                pm = [(w.copy(), bb.copy()) for w, bb in params]
                pp[li][which_p][idx] += h
                pm[li][which_p][idx] -= h
                num = (scalar(pp) - scalar(pm)) / (2 * h)
                np.testing.assert_allclose(g[idx], num, rtol=2e-4, atol=1e-7, err_msg='layer %d %s %r' % (li, 'Wb'[which_p], idx))


# ================================================================================================ round 3
# Literal-valued / closed-form cases of mipnerf360/tests/{math,coord,stepfun,render}_test.py that were not restated yet,
# and the pieces added to the oracle this round (train step, kl / urf gradients).
def test_stable_pos_enc_on_multiples_of_half_pi():
    """coord_test.py:33-59: the doubling-rotation pos_enc used as the high-degree reference, on x = k pi / 2 (literal
    expected values), and pos_enc against it at degrees where float64 is exact."""
    def stable_pos_enc(x, n):
        sin_x, cos_x = np.sin(x), np.cos(x)
        out = []
        rot = np.array([[cos_x, -sin_x], [sin_x, cos_x]], dtype='double')
        for _ in range(n):
            out.append(rot[::-1, 0, :])
            rot = np.einsum('ijn,jkn->ikn', rot, rot)
        return np.reshape(np.transpose(np.stack(out, 0), [2, 1, 0]), [-1, 2 * n])
    n = 10
    x = np.linspace(-np.pi, np.pi, 5)
    z = stable_pos_enc(x, n).reshape([-1, 2, n])
    z0, z1 = np.zeros_like(z[:, 0, :]), np.ones_like(z[:, 1, :])
    z0[:, 0] = [0, -1, 0, 1, 0]
    z1[:, 0] = [-1, 0, 1, 0, -1]
    z1[:, 1] = [1, -1, 1, -1, 1]
    np.testing.assert_allclose(z, np.stack([z0, z1], axis=1), atol=1e-10)
    xs = np.linspace(-1, 1, 7)[:, None]
    np.testing.assert_allclose(M.pos_enc(xs, 0, 6, append_identity=False), stable_pos_enc(xs[:, 0], 6), atol=1e-9)


def test_safe_sin_is_accurate_and_never_nan():
    """math_test.py:24-47 (safe_sin; safe_cos is not restated -- the path uses safe_sin only, coord.py:99)."""
    for max_exp, check in ((10, True), (60, False)):
        x = 10 ** np.linspace(-30, max_exp, 10000)
        x = np.concatenate([-x[::-1], np.array([0]), x])
        y = M.safe_sin(x)
        assert not np.isnan(y).any()
        if check:
            assert np.abs(y - np.sin(x)).max() < 1e-4


def test_learning_rate_decay_anchor_points():
    """math_test.py:72-153: lr(0) = lr_init (x lr_delay_mult when delayed), lr(max) = lr_final, lr(max / 2) = geometric
    mean, flat past the end, and the delayed schedule joins the plain one at lr_delay_steps."""
    rs = np.random.RandomState(0)
    for _ in range(10):
        lr_init = np.exp(rs.randn() - 3)
        lr_final = lr_init * np.exp(rs.randn() - 5)
        max_steps = int(np.ceil(100 + 100 * np.exp(rs.randn())))
        f = lambda s, **kw: M.learning_rate_decay(s, lr_init, lr_final, max_steps, **kw)
        np.testing.assert_allclose(f(0), lr_init, atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(f(max_steps), lr_final, atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(f(max_steps / 2), np.sqrt(lr_init * lr_final), atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(f(max_steps + 100), lr_final, atol=1e-5, rtol=1e-5)
        delay_steps = int(rs.uniform(0.1, 0.4) * max_steps)
        delay_mult = np.exp(rs.randn() - 3)
        kw = dict(lr_delay_steps=delay_steps, lr_delay_mult=delay_mult)
        np.testing.assert_allclose(f(0, **kw), delay_mult * lr_init, atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(f(max_steps, **kw), lr_final, atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(f(delay_steps, **kw), f(delay_steps), atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(f(max_steps / 2, **kw), np.sqrt(lr_init * lr_final), atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(f(max_steps + 100, **kw), lr_final, atol=1e-5, rtol=1e-5)
    # the Config defaults of this fork (configs.py:91,118-121) at the literal anchor points
    lr = lambda s: M.learning_rate_decay(s, 2e-3, 2e-5, 250000, 512, 0.01)
    np.testing.assert_allclose([lr(0), lr(250000), lr(125000)], [2e-5, 2e-5, 2e-4], rtol=1e-6)


def test_sorted_interp_matches_numpy_interp():
    """math_test.py:155-178 (`sort` variant: the branch invert_cdf uses)."""
    rs = np.random.RandomState(0)
    n, d0, d1 = 100, 10, 20
    x = rs.randn(n, d0)
    xp, fp = np.sort(rs.randn(n, d1), -1), np.sort(rs.randn(n, d1), -1)
    z = M.sorted_interp(x, xp, fp)
    np.testing.assert_allclose(z, np.stack([np.interp(x[i], xp[i], fp[i]) for i in range(n)]), atol=1e-5, rtol=1e-5)


def test_sample_single_bin_literal():
    """stepfun_test.py:477-495: bins [0, 1, 3, 6, 10], one-hot weights: all 625 samples inside the hot bin."""
    bins = np.array([0, 1, 3, 6, 10], np.float32)
    rs = np.random.RandomState(0)
    for jit, single in ((None, False), (rs.rand(1, 625), False), (rs.rand(1, 1), True)):
        for i in range(len(bins) - 1):
            w = np.zeros(len(bins) - 1, np.float32)
            w[i] = 1.
            with np.errstate(divide='ignore'):
                s = M.sample(bins[None], np.log(w[None]), 625, jitter01=jit, single_jitter=single)[0]
            assert (s >= bins[i]).all() and (s <= bins[i + 1]).all()


@pytest.mark.parametrize('randomized,bound_domain', [(False, False), (True, False), (False, True), (True, True)])
def test_sample_intervals_unbiased_literal(randomized, bound_domain):
    """stepfun_test.py:542-577: t = [-2.5 .. 2.5], logits [0, 0, 100, 0, 0] -- one interval [-0.5, 0.5]."""
    n, d = 1000, 64
    domain = (-0.5, 0.5) if bound_domain else (-np.inf, np.inf)
    t = np.tile(np.array([-2.5, -1.5, -0.5, 0.5, 1.5, 2.5])[None], (n, 1))
    logits = np.tile(np.array([0, 0, 100., 0, 0])[None], (n, 1))
    jit = np.random.RandomState(0).rand(n, 1) if randomized else None
    ts = M.sample_intervals(t, logits, d, jitter01=jit, single_jitter=True, domain=domain)
    if randomized:
        assert np.abs(ts.mean(-1)).max() < 0.5 / d
        np.testing.assert_allclose(np.mean(ts[:, 0] > -0.5), 0.5, atol=3.0 / d)       # binomial spread of 1000 draws
        np.testing.assert_allclose(np.mean(ts[:, -1] < 0.5), 0.5, atol=3.0 / d)
    else:
        np.testing.assert_allclose(ts.mean(-1), np.zeros(n), atol=1e-5, rtol=1e-5)
    if bound_domain and randomized:
        # upstream asserts the MEDIAN of the outer edges is +-0.5 (about half of the draws are clamped to the domain, so its
        # median sits on the clamp for its PRNG key); seed-independent form: the 40 % / 60 % quantiles are the clamp value
        np.testing.assert_allclose(np.quantile(ts[:, 0], 0.4), -0.5, atol=1e-4)
        np.testing.assert_allclose(np.quantile(ts[:, -1], 0.6), 0.5, atol=1e-4)
        assert ts.min() >= -0.5 and ts.max() <= 0.5


def test_sample_single_interval_is_a_linspace():
    """stepfun_test.py:579-586: t = 1..6, logits [0, 0, 100, 0, 0] -> linspace(3, 4, 11)."""
    t = np.array([1, 2, 3, 4, 5, 6], np.float64)
    got = M.sample_intervals(t, np.array([0, 0, 100, 0, 0], np.float64), 10, single_jitter=True)
    np.testing.assert_allclose(got, np.linspace(3, 4, 11), atol=1e-5, rtol=1e-5)


def test_lossfun_outer_monotonic_and_self_zero():
    """stepfun_test.py:657-697: invariant under a monotonic map of t (bit for bit); zero against itself."""
    rs = np.random.RandomState(0)
    curve = lambda x: 1 + x ** 3
    for _ in range(10):
        d0, d1 = rs.randint(10, 20, 2)
        t0, t1 = np.sort(rs.rand(d0 + 1)), np.sort(rs.rand(d1 + 1))
        w0, w1 = np.exp(rs.randn(d0)), np.exp(rs.randn(d1))
        np.testing.assert_array_equal(M.lossfun_outer(t0, w0, t1, w1), M.lossfun_outer(curve(t0), w0, curve(t1), w1))
        assert (M.lossfun_outer(t0, w0, t0, w0) < 1e-10).all()


@pytest.mark.parametrize('use_avg', [False, True])
def test_resample_entire_domain_and_single_span(use_avg):
    """stepfun_test.py:856-896: an interval covering everything sums all values; a sub-span of one bin returns that bin's
    value (average) or its covered fraction (sum)."""
    rs = np.random.RandomState(0)
    d = 32
    tp, vp = np.sort(rs.randn(d + 1)), rs.randn(d)
    if not use_avg:
        np.testing.assert_allclose(M.resample(np.array([-1e6, 1e6]), tp, vp)[0], vp.sum(), atol=1e-4)
    pad = (tp[d // 2 + 1] - tp[d // 2]) / 4
    t = np.array([tp[d // 2] + pad, tp[d // 2 + 1] - pad])
    np.testing.assert_allclose(M.resample(t, tp, vp, use_avg=use_avg)[0], vp[d // 2] * (1.0 if use_avg else 0.5), atol=1e-4)


@pytest.mark.parametrize('fn', ['cylinder_to_gaussian', 'conical_frustum_to_gaussian'])
def test_gaussian_scaling_literal(fn):
    """render_test.py:200-258: d = (0, 0, 1), t0 = 0.3, t1 = 0.7, radius = 0.4; scaling d by 2.7 scales the mean by 2.7,
    the covariance along the ray by 2.7^2 and leaves the perpendicular covariance alone."""
    d = np.array([0., 0., 1.])
    t0, t1, radius = np.array([0.3]), np.array([0.7]), np.array([0.4])
    f = getattr(M, fn)
    mean, cov = f(d, t0, t1, radius, False)
    scale = 2.7
    mean_s, cov_s = f(scale * d, t0, t1, radius, False)
    np.testing.assert_allclose(scale * mean, mean_s, atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(scale ** 2 * cov[..., 2, 2], cov_s[..., 2, 2], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(cov[..., :2, :2], cov_s[..., :2, :2], atol=1e-5, rtol=1e-5)
    if fn == 'cylinder_to_gaussian':          # closed form: mean (t0 + t1) / 2, var_t = (t1 - t0)^2 / 12, var_r = r^2 / 4
        np.testing.assert_allclose(mean[0], [0, 0, 0.5], atol=1e-12)
        np.testing.assert_allclose(np.diag(cov[0]), [0.04, 0.04, 0.16 / 12], atol=1e-12)


def test_conical_frustum_stable_matches_the_textbook_form():
    """render_test.py:320-331: the `stable` re-parameterisation equals the direct moments for well-conditioned frusta."""
    rs = np.random.RandomState(0)
    n = 50
    d = rs.randn(n, 3)
    t0 = np.exp(rs.uniform(-1, 1, n))
    t1 = t0 + np.exp(rs.uniform(-1, 1, n))
    r = np.exp(rs.uniform(-3, -1, n))
    a = M.conical_frustum_to_gaussian(d, t0, t1, r, True, stable=True)
    b = M.conical_frustum_to_gaussian(d, t0, t1, r, True, stable=False)
    for x, y in zip(a, b):
        np.testing.assert_allclose(x, y, atol=1e-7, rtol=1e-5)


def test_volumetric_rendering_backward_matches_finite_differences():
    rs = np.random.RandomState(0)
    n, S = 6, 8
    d = rs.randn(n, 3)
    td = np.sort(rs.uniform(0.5, 6, (n, S + 1)), -1)
    dens, rgbs = rs.rand(n, S) * 2, rs.rand(n, S, 3)
    far = np.full((n, 1), 1e6)
    g_rgb, g_dm, g_wx = rs.randn(n, 3), rs.randn(n), rs.randn(n, S)
    for opaque in (True, False):
        def f(dd):
            w = M.compute_alpha_weights(dd, td, d, opaque)[0]
            r = M.volumetric_rendering(rgbs, w, td, 1.0, far)
            return (r['rgb'] * g_rgb).sum() + (r['distance_mean'] * g_dm).sum() + (w * g_wx).sum()
        w = M.compute_alpha_weights(dens, td, d, opaque)[0]
        gw, g_rgbs = M.volumetric_rendering_backward(rgbs, w, td, 1.0, g_rgb, g_dm)
        gd = M.alpha_weights_backward(dens, td, d, gw + g_wx, opaque)
        num = np.zeros_like(dens)
        for i in range(n):
            for j in range(S):
                a, b = dens.copy(), dens.copy()
                a[i, j] += 1e-6
                b[i, j] -= 1e-6
                num[i, j] = (f(a) - f(b)) / 2e-6
        np.testing.assert_allclose(gd, num, rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(g_rgbs, w[..., None] * g_rgb[:, None, :])


@pytest.mark.parametrize('kind', ['kl', 'urf'])
@pytest.mark.parametrize('n,S', [(6, 6), (1, 9)])
def test_depth_loss_grads_match_finite_differences(kind, n, S):
    """closed-form gradient of internal/depth_loss.py (with its `.sum(-2)` / mask broadcast) vs central differences."""
    rs = np.random.RandomState(n + S)
    td = np.sort(rs.uniform(0.5, 6, (n, S + 1)), -1)
    w = M.softmax(rs.randn(n, S))
    sup = np.where(rs.rand(n) < .7, rs.uniform(1, 5, n), 0.)
    sup[0] = 3.0
    pred, dirs, sigma = rs.uniform(1, 5, n), rs.randn(n, 3), 0.7
    gw, gd = M.depth_loss_grads(w, td, sup, pred, sigma, dirs, kind)
    f = lambda w_, p_: M.depth_loss(w_, td, sup, p_, sigma, dirs, kind)
    for idx in [(0, 0), (n - 1, S - 1), (n // 2, S // 2)]:
        a, b = w.copy(), w.copy()
        a[idx] += 1e-7
        b[idx] -= 1e-7
        np.testing.assert_allclose(gw[idx], (f(a, pred) - f(b, pred)) / 2e-7, rtol=1e-4, atol=1e-9)
    for i in range(n):
        a, b = pred.copy(), pred.copy()
        a[i] += 1e-6
        b[i] -= 1e-6
        np.testing.assert_allclose(gd[i], (f(w, a) - f(w, b)) / 2e-6, rtol=1e-5, atol=1e-9)
    with pytest.raises(ValueError):
        M.depth_loss_grads(np.ones((5, 3)), np.ones((5, 4)), np.ones(5), np.ones(5), 0.1, np.ones((5, 3)), kind)


def test_apply_gradients_is_optax_adam_with_clipping_and_nan_to_num():
    """train_utils.py:215-236, :344-347, :371-395 on hand-computable numbers: one 2-element 'MLP' per name."""
    mk = lambda v: [(np.array([[v]], np.float64), np.array([0.0]))]
    st = M.new_train_state(mk(1.0), mk(-2.0))
    grads = dict(prop=[(np.array([[3e-4]]), np.array([4e-4]))], nerf=[(np.array([[30.0]]), np.array([40.0]))])
    mult = M.apply_gradients(st, grads, max_steps=1000, grad_max_norm=1e-3, adam_eps=1e-6)
    np.testing.assert_allclose(mult['prop'], 1.0)                                    # |g| = 5e-4 < 1e-3: untouched
    np.testing.assert_allclose(mult['nerf'], 1e-3 / (M.EPS32 + 50.0), rtol=1e-12)    # |g| = 50: scaled to 1e-3
    lr0 = M.learning_rate_decay(0, 2e-3, 2e-5, 1000, 512, 0.01)                      # schedule at the PRE-increment count
    g = 3e-4
    np.testing.assert_allclose(st['prop'][0][0][0, 0], 1.0 - lr0 * g / (abs(g) + 1e-6), rtol=1e-12)   # first step: m_hat / sqrt(v_hat) = sign
    g = 30.0 * mult['nerf']
    np.testing.assert_allclose(st['nerf'][0][0][0, 0], -2.0 - lr0 * g / (abs(g) + 1e-6), rtol=1e-12)
    assert st['count'] == 1
    # a NaN anywhere in an MLP's gradient: multiplier NaN, every entry nan_to_num'ed to 0 -- moments decay, nothing breaks
    mu0 = st['nerf'] and st['mu']['nerf'][0][0].copy()
    p0 = st['nerf'][0][0].copy()
    mult = M.apply_gradients(st, dict(prop=grads['prop'], nerf=[(np.array([[np.nan]]), np.array([1.0]))]), max_steps=1000)
    assert np.isnan(mult['nerf'])
    np.testing.assert_allclose(st['mu']['nerf'][0][0], 0.9 * mu0)
    assert np.isfinite(st['nerf'][0][0]).all() and st['nerf'][0][0] != p0            # the decayed momentum still moves it


def test_loss_and_grads_total_and_gradient_against_finite_differences():
    """loss_fn + value_and_grad (train_utils.py:258-333) at toy sizes (16 / 8 samples): the total is the sum of the four
    terms with the depth term counted as upstream counts it, and the NeRF-MLP gradient matches central differences once
    the stop_gradient'ed interlevel term is switched off (the proposal MLP's would also see the re-sampling)."""
    rs = np.random.RandomState(0)
    n = 8
    d = rs.randn(n, 3)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    rays = dict(origins=rs.randn(n, 3) * 0.3, directions=d, viewdirs=d.copy(), radii=np.full((n, 1), 2e-3),
                near=np.full((n, 1), 0.2), far=np.full((n, 1), 1e6))
    small = dict(net_width=16, bottleneck_width=8, net_width_viewdirs=8)
    saved = dict(M.PROP_CFG), dict(M.NERF_CFG)
    M.PROP_CFG.update(small)
    M.NERF_CFG.update(small)
    try:
        pp = [(w.astype(np.float64), b.astype(np.float64)) for w, b in M.init_mlp_params(M.PROP_CFG, np.random.RandomState(0))]
        pn = [(w.astype(np.float64), b.astype(np.float64)) for w, b in M.init_mlp_params(M.NERF_CFG, np.random.RandomState(1))]
        gt = rs.rand(n, 3)
        sup = np.where(rs.rand(n) < .6, rs.uniform(1, 4, n), 0.)
        jit = [rs.rand(n, 1) for _ in range(3)]
        kw = dict(num_prop_samples=16, num_nerf_samples=8)
        st, _ = M.loss_and_grads(pp, pn, rays, gt, sup, 0.3, jit, **kw)
        np.testing.assert_allclose(st['loss'], st['data'] + 0.1 * st['depth_losses'].sum() + st['interlevel'] + st['distortion'])
        for dl in ('mse', 'l1'):
            kw2 = dict(kw, interlevel_loss_mult=0.0, depth_loss_type=dl)
            _, g = M.loss_and_grads(pp, pn, rays, gt, sup, 0.3, jit, **kw2)
            total = lambda pn_: M.loss_and_grads(pp, pn_, rays, gt, sup, 0.3, jit, **kw2)[0]['loss']
            for li in (0, 4, len(pn) - 1):
                idx = tuple(rs.randint(0, s) for s in pn[li][0].shape)
                a = [(w.copy(), b.copy()) for w, b in pn]
                b_ = [(w.copy(), b.copy()) for w, b in pn]
                a[li][0][idx] += 1e-6
                b_[li][0][idx] -= 1e-6
                np.testing.assert_allclose(g['nerf'][li][0][idx], (total(a) - total(b_)) / 2e-6, rtol=2e-4, atol=1e-9)
    finally:
        M.PROP_CFG.clear(); M.PROP_CFG.update(saved[0])
        M.NERF_CFG.clear(); M.NERF_CFG.update(saved[1])
