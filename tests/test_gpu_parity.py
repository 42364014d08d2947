"""GPU parity tests: the HIP path (through the C ABI) against the committed golden vectors of the
reference and against the numpy oracle on the same seeded inputs.

Tolerances (north_star): 1e-4 relative float32 for rendered RGB / expected depth / loss in the
split-bf16 "parity" precision, integer sample bins bit-exact.  The single-pass bf16 "speed"
precision is checked against looser, stated bounds.
"""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import nerfpp_oracle as O                                   # noqa: E402


def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def N(t):
    return t.detach().cpu().numpy()


def close(a, b, rtol=1e-4, atol=1e-6):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=False)


@pytest.fixture(scope='module')
def ops():
    dev()
    from outdoor_nerf_depth_amd import ops as _ops
    return _ops


@pytest.fixture(scope='module')
def levels():
    return O.init_params_like_reference(2)


def flat(level):
    return np.concatenate([level[k].reshape(-1) for k in O.param_order()]).astype(np.float32)


def unflat(vec):
    out, off = {}, 0
    shapes = {}
    for net, in_ch in (('fg_net', O.FG_IN), ('bg_net', O.BG_IN)):
        for k, s in O.mlp_param_shapes(in_ch, O.DIR_IN).items():
            shapes['%s.%s' % (net, k)] = s
    for k in O.param_order():
        n = int(np.prod(shapes[k]))
        out[k] = vec[off:off + n].reshape(shapes[k])
        off += n
    return out


# ----------------------------------------------------------------------------------------- sampling
def test_intersect_coarse_perturb_bit_exact(ops, golden):
    g = golden('sampling')
    far = N(ops.intersect_sphere(T(g['ray_o']), T(g['ray_d'])))
    close(far, g['fg_far'], 2e-6, 0)
    np.testing.assert_array_equal(far, O.intersect_sphere(g['ray_o'], g['ray_d']))
    far2, fg, bg = ops.sample_coarse(T(g['ray_o']), T(g['ray_d']), T(g['min_depth']), 64, perturb=False)
    far_o = O.intersect_sphere(g['ray_o'], g['ray_d'])
    fg_o, bg_o = O.coarse_depths(g['min_depth'], far_o, 64)
    np.testing.assert_array_equal(N(fg), fg_o)
    np.testing.assert_array_equal(N(bg), g['bg_coarse'])
    far3, fgp, bgp = ops.sample_coarse(T(g['ray_o']), T(g['ray_d']), T(g['min_depth']), 64,
                                       t_rand_fg=T(g['t_fg']), t_rand_bg=T(g['t_bg']))
    np.testing.assert_array_equal(N(fgp), O.perturb_samples(fg_o, g['t_fg']))
    np.testing.assert_array_equal(N(bgp), g['bg_perturbed'])
    close(N(fgp), g['fg_perturbed'], 1e-6, 1e-9)
    np.testing.assert_array_equal(N(ops.perturb_samples(T(g['fg_coarse']), T(g['t_fg']))), g['fg_perturbed'])


def test_intersect_sphere_raises_like_the_reference(ops):
    with pytest.raises(Exception, match='unit sphere'):
        ops.intersect_sphere(T(np.array([[2., 0, 0]], np.float32)), T(np.array([[0., 0, 1]], np.float32)))


@pytest.mark.parametrize('tag', ['rand', 'det'])
def test_sample_pdf_integer_bins_bit_exact(ops, golden, tag):
    g = golden('sampling')
    u = None if tag == 'det' else T(g['u_rand'])
    samples, above = ops.sample_pdf(T(g['bins']), T(g['weights']), 128, det=(tag == 'det'), u=u, return_inds=True)
    s_o, a_o = O.sample_pdf(g['bins'], g['weights'], g['u_' + tag])
    np.testing.assert_array_equal(N(above), a_o)                 # bit-exact vs the oracle, all inputs
    np.testing.assert_array_equal(N(samples), s_o)
    safe = g['margin_' + tag] >= 1e-5                            # vs the reference: away from cdf edges
    np.testing.assert_array_equal(N(above)[safe], g['above_' + tag][safe])
    same = N(above) == g['above_' + tag]
    assert same.mean() > 0.999
    close(N(samples)[same], g['samples_' + tag][same], 1e-4, 1e-6)


def test_sample_fine_merge_sorted_and_matches_oracle(ops, golden):
    g = golden('forward')
    for zk, wk, uk, refk in (('fg_z0', 'L0.fg_weights', 'u_fg', 'fg_z1'), ('bg_z0', 'L0.bg_weights', 'u_bg', 'bg_z1')):
        merged, samples, above = ops.sample_fine(T(g[zk]), T(g[wk]), 128, u=T(g[uk]), return_all=True)
        m_o, s_o, a_o = O.fine_depths(g[zk], g[wk], g[uk])
        np.testing.assert_array_equal(N(above), a_o)
        np.testing.assert_array_equal(N(merged), m_o)
        assert (np.diff(N(merged), axis=-1) >= 0).all()
        bad = np.abs(N(merged) - g[refk]) > 1e-6 + 1e-5 * np.abs(g[refk])
        assert bad.mean() < 2e-3
    det = ops.sample_fine(T(g['fg_z0']), T(g['L0.fg_weights']), 128, det=True)
    m_o, _, _ = O.fine_depths(g['fg_z0'], g['L0.fg_weights'], np.broadcast_to(O.torch_linspace(0, 1, 128), (12, 128)))
    np.testing.assert_array_equal(N(det), m_o)
    # both volumes in one launch == the two single calls, bit for bit (random and det)
    fg_m, bg_m = ops.sample_fine_pair(T(g['fg_z0']), T(g['L0.fg_weights']), T(g['bg_z0']), T(g['L0.bg_weights']), 128,
                                      u_fg=T(g['u_fg']), u_bg=T(g['u_bg']))
    np.testing.assert_array_equal(N(fg_m), O.fine_depths(g['fg_z0'], g['L0.fg_weights'], g['u_fg'])[0])
    np.testing.assert_array_equal(N(bg_m), O.fine_depths(g['bg_z0'], g['L0.bg_weights'], g['u_bg'])[0])
    fg_d, bg_d = ops.sample_fine_pair(T(g['fg_z0']), T(g['L0.fg_weights']), T(g['bg_z0']), T(g['L0.bg_weights']), 128,
                                      det=True)
    np.testing.assert_array_equal(N(fg_d), N(det))
    np.testing.assert_array_equal(N(bg_d), N(ops.sample_fine(T(g['bg_z0']), T(g['L0.bg_weights']), 128, det=True)))


@pytest.mark.parametrize('n_rays,S_old,n_new', [(1, 64, 128), (7, 64, 128), (5, 33, 17), (9, 100, 156)])
def test_sample_fine_ragged_shapes_bit_exact(ops, n_rays, S_old, n_new):
    """ray counts that do not fill a 4-ray block, sample counts that are not multiples of 64: the fused
    mids -> sample_pdf -> merge kernel against the oracle, bit for bit, single and paired launches."""
    rs = np.random.RandomState(n_rays * 1000 + S_old)
    z = np.sort(rs.rand(n_rays, S_old).astype(np.float32) * 3 + 0.1, axis=-1)
    w = (rs.rand(n_rays, S_old).astype(np.float32) ** 4)
    w[:, ::7] = 0
    u = rs.rand(n_rays, n_new).astype(np.float32)
    m_o, s_o, a_o = O.fine_depths(z, w, u)
    merged, samples, above = ops.sample_fine(T(z), T(w), n_new, u=T(u), return_all=True)
    np.testing.assert_array_equal(N(above), a_o)
    np.testing.assert_array_equal(N(samples), s_o)
    np.testing.assert_array_equal(N(merged), m_o)
    z2 = np.sort(rs.rand(n_rays, S_old).astype(np.float32), axis=-1)
    w2 = rs.rand(n_rays, S_old).astype(np.float32)
    u2 = rs.rand(n_rays, n_new).astype(np.float32)
    a, b = ops.sample_fine_pair(T(z), T(w), T(z2), T(w2), n_new, u_fg=T(u), u_bg=T(u2))
    np.testing.assert_array_equal(N(a), m_o)
    np.testing.assert_array_equal(N(b), O.fine_depths(z2, w2, u2)[0])


# ----------------------------------------------------------------------------------------- forward
# single-pass bf16: elementwise sanity bound; the measured errors (profiles/r02_bf16_error_report.json: rgb 2e-4, depth 7e-5,
# fg_weights 4.7e-3 of the tensor's max) are held to 2x by tests/test_gpu_round2.py::test_bf16_errors_within_twice_...
RET_TOL = {2: dict(rtol=1e-4, atol=2e-6), 1: dict(rtol=2e-2, atol=1.5e-3)}


@pytest.mark.parametrize('prec', [2, 1])
def test_level_forward_matches_reference(ops, golden, levels, prec):
    g = golden('forward')
    for m, (fz, bz) in enumerate((('fg_z0', 'bg_z0'), ('fg_z1', 'bg_z1'))):
        eng = ops.LevelEngine(T(flat(levels[m])), precision=prec)
        ret = eng.forward(T(g['ray_o']), T(g['ray_d']), T(g['fg_far']), T(g[fz]), T(g[bz]))
        assert list(ret.keys()) == list(ops.RET_KEYS)
        for k, v in ret.items():
            ref = g['L%d.%s' % (m, k)]
            tol = dict(RET_TOL[prec])
            if k in ('bg_depth', 'depth'):      # sums of terms up to 1e6 (1/(z+eps)): scale atol
                tol['atol'] = tol['atol'] * max(1.0, np.abs(ref).max())
            np.testing.assert_allclose(N(v), ref, err_msg='L%d.%s' % (m, k), **tol)


@pytest.mark.parametrize('n_rays,S', [(1, 64), (5, 64), (7, 192), (33, 33), (3, 256)])
def test_level_forward_ragged_sizes_match_oracle(ops, levels, n_rays, S):
    """rows not a multiple of the 128/256-row tile, S not a multiple of 64, single ray."""
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    b = SyntheticKitti().random_batch(n_rays, np.random.RandomState(n_rays + S))
    rs = np.random.RandomState(S)
    far = O.intersect_sphere(b['ray_o'], b['ray_d'])
    fg, bg = O.coarse_depths(b['min_depth'], far, S)
    fg = O.perturb_samples(fg, rs.rand(n_rays, S).astype(np.float32))
    bg = O.perturb_samples(bg, rs.rand(n_rays, S).astype(np.float32))
    ref = O.nerf_forward(levels[0], b['ray_o'], b['ray_d'], far, fg, bg)
    eng = ops.LevelEngine(T(flat(levels[0])), precision=2)
    ret = eng.forward(T(b['ray_o']), T(b['ray_d']), T(far), T(fg), T(bg))
    for k in ('rgb', 'fg_weights', 'bg_weights', 'fg_dists', 'fg_depth', 'bg_lambda'):
        np.testing.assert_allclose(N(ret[k]), ref[k], rtol=2e-4, atol=3e-6, err_msg=k)


# ----------------------------------------------------------------------------------------- losses
def test_losses_match_reference(ops, golden):
    g = golden('losses')
    n, S = g['w'].shape
    ret = dict(rgb=T(g['x']), depth=T(g['pred']), fg_weights=T(g['w']), fg_dists=T(g['lengths']))
    sc, g_rgb, g_depth, g_w = ops.loss_and_grads(ret, T(g['y']), T(g['gt']), 'mse', 1.0)
    close(N(sc)[1], g['img2mse'], 1e-5, 0)
    close(N(sc)[2], g['mse'], 1e-5, 0)
    close(N(sc)[0], g['img2mse'] + g['mse'], 1e-5, 0)
    assert N(sc)[3] == (g['gt'] > 0).sum()
    sc, _, _, _ = ops.loss_and_grads(ret, T(g['y']), T(g['gt']), 'l1', 1.0)
    close(N(sc)[2], g['l1'], 1e-5, 0)
    sc, _, _, g_w = ops.loss_and_grads(ret, T(g['y']), T(g['gt']), 'kl', 1.0, kl_sigma=float(g['sigma']),
                                       fg_z_vals=T(g['steps']), fg_far_depth=T(g['far']))
    close(N(sc)[2], g['kl'], 2e-5, 0)
    # empty masks: NaN / NaN / 0 like the reference
    zero = T(np.zeros_like(g['gt']))
    assert np.isnan(N(ops.loss_and_grads(ret, T(g['y']), zero, 'mse', 1.0)[0])[2])
    assert np.isnan(N(ops.loss_and_grads(ret, T(g['y']), zero, 'l1', 1.0)[0])[2])
    sc, _, g_depth, g_w = ops.loss_and_grads(ret, T(g['y']), zero, 'kl', 1.0, kl_sigma=float(g['sigma']),
                                             fg_z_vals=T(g['steps']), fg_far_depth=T(g['far']))
    assert N(sc)[2] == 0 and not N(g_w).any() and not N(g_depth).any()
    # gradients vs the oracle's closed forms
    retn = dict(rgb=g['x'], depth=g['pred'], fg_weights=g['w'], fg_dists=g['lengths'])
    for mode in ('mse', 'l1', 'kl'):
        _, g_rgb, g_depth, g_w = ops.loss_and_grads(ret, T(g['y']), T(g['gt']), mode, 0.3, kl_sigma=float(g['sigma']),
                                                    fg_z_vals=T(g['steps']), fg_far_depth=T(g['far']))
        _, _, _, o_rgb, o_depth, o_w = O.loss_and_grads(retn, g['steps'], g['far'], g['y'], g['gt'], True, mode,
                                                        0.3, float(g['sigma']))
        close(N(g_rgb), o_rgb, 1e-5, 1e-9)
        close(N(g_depth), o_depth, 1e-5, 1e-9)
        if mode == 'kl':
            close(N(g_w), o_w, 2e-5, 1e-9)


# ----------------------------------------------------------------------------------------- backward
# max |diff| / RMS of the tensor, and relative L2 error over the sampled entries.  The float32
# reference itself is ~1e-1*RMS noisy on these cancelling sums (tests/test_oracle_golden.py::
# test_level_gradients), so the comparison is against the float64 run of the reference.  Single-pass
# bf16 (8 mantissa bits) is expected to be ~10x noisier than split-bf16.
# bf16: 2x the worst values measured on MI355X over the four loss modes (profiles/r02_bf16_error_report.json:
# max|diff|/RMS 0.62, relative L2 0.10)
GRAD_TOL = {2: (5e-2, 2e-2), 1: (1.25, 0.2)}


@pytest.mark.parametrize('prec', [2, 1])
@pytest.mark.parametrize('mode', ['rgbonly', 'mse', 'l1', 'kl'])
def test_level_gradients_match_reference(ops, golden, levels, mode, prec):
    g = golden('grads_' + mode)
    for m in range(2):
        fz, bz = g['L%d.fg_z' % m], g['L%d.bg_z' % m]
        eng = ops.LevelEngine(T(flat(levels[m])), precision=prec)
        ret = eng.forward(T(g['ray_o']), T(g['ray_d']), T(g['fg_far']), T(fz), T(bz), training=True)
        sc, g_rgb, g_depth, g_w = ops.loss_and_grads(
            ret, T(g['rgb_gt']), T(g['depth_sup']), mode, float(g['lambda_depth']),
            kl_sigma=float(g['depth_sigma_scaled']), fg_z_vals=T(fz), fg_far_depth=T(g['fg_far']))
        if prec == 2:
            close(N(sc)[0], g['L%d.loss' % m], 2e-4, 0)
            close(N(ret['rgb']), g['L%d.rgb' % m], 1e-4, 2e-6)
        grads = unflat(N(eng.backward(g_rgb, g_depth, g_w)))
        for k in O.param_order():
            mine = grads[k].reshape(-1)[g['L%d.%s.idx' % (m, k)]]
            ref64 = g['L%d.%s.g64' % (m, k)]
            rms = g['L%d.%s.norm64' % (m, k)] / np.sqrt(grads[k].size) + 1e-12
            err = np.abs(mine - ref64).max() / rms
            rel_l2 = np.linalg.norm(mine - ref64) / (np.linalg.norm(ref64) + 1e-30)
            assert err <= GRAD_TOL[prec][0], (k, m, err)
            # 1- and 3-element tensors (sigma / rgb biases) are single cancelling sums: the oracle
            # itself is 3e-2 off there (tests/test_oracle_golden.py), allow 3x the bound
            l2_tol = GRAD_TOL[prec][1] * (3 if mine.size <= 3 else 1)
            assert rel_l2 <= l2_tol or err <= 1e-3, (k, m, rel_l2)
            n_mine = np.linalg.norm(grads[k].astype(np.float64))
            # (the norm of a 1- or 3-element tensor IS its single cancelling sums: the same 3x as above.  bg sigma bias,
            # bf16: 2.7e-6 before the remap layer was folded into the colour head, 1.3e-6 after, 2.9e-6 in float64)
            n_tol = (0.06 if prec == 2 else 0.3) * (3 if mine.size <= 3 else 1)
            assert abs(n_mine - g['L%d.%s.norm64' % (m, k)]) <= n_tol * g['L%d.%s.norm64' % (m, k)] + 1e-12, k


def test_backward_matches_oracle_elementwise(ops, golden, levels):
    """Same inputs through the oracle's closed-form backward: tighter than the reference check."""
    g = golden('grads_l1')
    fz, bz = g['L0.fg_z'], g['L0.bg_z']
    cache = {}
    ret_o = O.nerf_forward(levels[0], g['ray_o'], g['ray_d'], g['fg_far'], fz, bz, cache=cache)
    _, _, _, o_rgb, o_depth, o_w = O.loss_and_grads(ret_o, fz, g['fg_far'], g['rgb_gt'], g['depth_sup'], True, 'l1',
                                                    0.1, 0.01)
    g_o = O.nerf_backward(cache, o_rgb, o_depth, o_w)
    eng = ops.LevelEngine(T(flat(levels[0])), precision=2)
    eng.forward(T(g['ray_o']), T(g['ray_d']), T(g['fg_far']), T(fz), T(bz), training=True)
    grads = unflat(N(eng.backward(T(o_rgb), T(o_depth), None)))
    for k in O.param_order():
        rms = np.sqrt((g_o[k].astype(np.float64) ** 2).mean()) + 1e-12
        assert np.abs(grads[k] - g_o[k]).max() <= 8e-2 * rms, (k, np.abs(grads[k] - g_o[k]).max() / rms)
        rel = np.linalg.norm(grads[k] - g_o[k]) / (np.linalg.norm(g_o[k]) + 1e-30)
        assert rel <= 1e-2, (k, rel)
    # linearity of the backward in the upstream gradients
    g2 = unflat(N(eng.backward(T(2 * o_rgb), T(2 * o_depth), None)))
    for k in ('fg_net.base_layers.3.0.weight', 'bg_net.rgb_layers.0.weight'):
        np.testing.assert_allclose(g2[k], 2 * grads[k], rtol=2e-2, atol=2e-3 * np.abs(grads[k]).max())


@pytest.mark.parametrize('n_rays,S,mode', [(7, 192, 'mse'), (33, 33, 'kl'), (5, 64, 'l1'), (270, 64, 'mse')])
def test_training_ragged_sizes_match_oracle(ops, levels, n_rays, S, mode):
    """rows = n_rays*S not a multiple of the 256-row (bf16) / 128-row (split-bf16) tile: the last tile's
    tail rows, the loader wave's hand-off rows and the padding rows the weight-gradient GEMMs read must
    all be right.  Both precisions against the oracle's closed-form forward + backward."""
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    b = SyntheticKitti().random_batch(n_rays, np.random.RandomState(7 * n_rays + S))
    b['depth_sup'][: max(1, n_rays // 3)] = np.float32(0.05)
    rs = np.random.RandomState(S)
    far = O.intersect_sphere(b['ray_o'], b['ray_d'])
    fg, bg = O.coarse_depths(b['min_depth'], far, S)
    fg = O.perturb_samples(fg, rs.rand(n_rays, S).astype(np.float32))
    bg = O.perturb_samples(bg, rs.rand(n_rays, S).astype(np.float32))
    cache = {}
    ret_o = O.nerf_forward(levels[0], b['ray_o'], b['ray_d'], far, fg, bg, cache=cache)
    _, _, _, o_rgb, o_depth, o_w = O.loss_and_grads(ret_o, fg, far, b['rgb'], b['depth_sup'], True, mode, 0.1, 0.01)
    g_o = O.nerf_backward(cache, o_rgb, o_depth, o_w)
    for prec in (2, 1):
        eng = ops.LevelEngine(T(flat(levels[0])), precision=prec)
        ret = eng.forward(T(b['ray_o']), T(b['ray_d']), T(far), T(fg), T(bg), training=True)
        sc, g_rgb, g_depth, g_w = ops.loss_and_grads(ret, T(b['rgb']), T(b['depth_sup']), mode, 0.1, kl_sigma=0.01,
                                                     fg_z_vals=T(fg), fg_far_depth=T(far))
        tol = RET_TOL[prec]
        np.testing.assert_allclose(N(ret['rgb']), ret_o['rgb'], **tol)
        np.testing.assert_allclose(N(ret['fg_weights']), ret_o['fg_weights'], rtol=tol['rtol'] * 2, atol=tol['atol'] * 2)
        # the same upstream gradients for both paths: what differs is only the MLP backward + dW
        grads = unflat(N(eng.backward(T(o_rgb), T(o_depth), None if o_w is None else T(o_w))))
        for k in O.param_order():
            rms = np.sqrt((g_o[k].astype(np.float64) ** 2).mean()) + 1e-12
            err = np.abs(grads[k] - g_o[k]).max() / rms
            rel = np.linalg.norm(grads[k] - g_o[k]) / (np.linalg.norm(g_o[k]) + 1e-30)
            # bf16: the max-norm of a sparse gradient tensor is dominated by single roundings; the L2 bound is the check
            assert err <= (1.2e-1 if prec == 2 else 4.0), (prec, k, err)       # float32 oracle: ~1e-1 RMS noise on tiny batches
            assert rel <= (1e-2 if prec == 2 else 0.3), (prec, k, rel)


# ----------------------------------------------------------------------------------------- optimiser
def test_adam_matches_torch_optim_golden(ops, golden):
    g = golden('adam_unit')
    p = T(g['p0'].copy())
    ea, eas = torch.zeros_like(p), torch.zeros_like(p)
    for i in range(4):
        ops.adam_step(p, T(g['grads'][i]), ea, eas, i + 1)
        np.testing.assert_allclose(N(p), g['p_after'][i], rtol=2e-7, atol=2e-9)
    np.testing.assert_allclose(N(ea), g['exp_avg'], rtol=1e-4, atol=1e-12)
    np.testing.assert_allclose(N(eas), g['exp_avg_sq'], rtol=1e-5, atol=1e-25)


def run_train_step(ops, engines, opt, step, batch, uni, mode='mse', lambda_depth=0.1, kl_sigma=0.01,
                   grad_hook=None):
    """ddp_train_nerf.py:432-498 on the HIP path with replayed uniforms."""
    logs = []
    ray_o, ray_d = T(batch['ray_o']), T(batch['ray_d'])
    far, fg_z, bg_z = ops.sample_coarse(ray_o, ray_d, T(batch['min_depth']), 64, t_rand_fg=T(uni['t_fg']),
                                        t_rand_bg=T(uni['t_bg']))
    ret = None
    for m, eng in enumerate(engines):
        if m == 1:
            fg_z = ops.sample_fine(fg_z, ret['fg_weights'], 128, u=T(uni['u_fg']))
            bg_z = ops.sample_fine(bg_z, ret['bg_weights'], 128, u=T(uni['u_bg']))
        ret = eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=True)
        sc, g_rgb, g_depth, g_w = ops.loss_and_grads(ret, T(batch['rgb']), T(batch['depth_sup']), mode, lambda_depth,
                                                     kl_sigma=kl_sigma, fg_z_vals=fg_z, fg_far_depth=far)
        grads = eng.backward(g_rgb, g_depth, g_w)
        if grad_hook is not None:
            grads = grad_hook(m, grads)
        ops.adam_step(eng.params, grads, opt[m][0], opt[m][1], step)
        eng.repack()
        logs.append(N(sc))
    return logs


# Adam's first step is -lr * g / (|g| + eps) = -lr * sign(g): an element of the HIP path can differ from the reference's by 2 lr
# only where the SIGN of its gradient differs, i.e. where |g| is inside the gradient's own error.  The split-bf16 gradients are
# within 5e-2 * RMS (max-norm, per tensor) of the float64 reference (test_gradients_match_reference_f64), so every element
# whose own |g| exceeds FLOOR x the tensor's RMS must take the reference's step exactly; the others may flip and are counted.
# Step 3 is a continuous function of three gradients: lr * m / sqrt(v) moves by <= lr * (relative gradient error), so the
# elements above the floor in all three steps are held to STEP3_TOL with no outlier allowance.  (Until round 4 this test
# allowed 3 % / 10 % of ALL sampled elements outside 2e-5 / 2.5e-4 = half an Adam step.)
SIGN_FLOOR = 0.15
# measured (MI355X, round 5): step 1 worst 7.5e-9 on the 58 % of sampled elements above the floor (3 sign flips among the rest),
# step 3 worst 3.1e-5 on the 40 % above it in all three steps.  Gates = float rounding / 2 x measured.
STEP1_TOL, STEP3_TOL = 1e-7, 6e-5


def test_three_training_steps_match_reference(ops, golden, levels):
    g = golden('train_steps')
    engines = [ops.LevelEngine(T(flat(lv)), precision=2) for lv in levels]
    init = [unflat(N(e.params).copy()) for e in engines]
    opt = [(torch.zeros_like(e.params), torch.zeros_like(e.params)) for e in engines]
    above = [None, None]            # per level: dict name -> bool mask over the tensor, "above the floor in every step so far"
    stats = {}
    for step in range(1, 4):
        batch = {k: g['s%d.%s' % (step, k)] for k in ('ray_o', 'ray_d', 'rgb', 'depth_sup', 'min_depth')}
        uni = {k: g['s%d.%s' % (step, k)] for k in ('t_fg', 't_bg', 'u_fg', 'u_bg')}
        seen = {}

        def hook(m, grads):
            seen[m] = unflat(N(grads).copy())
            return grads
        logs = run_train_step(ops, engines, opt, step, batch, uni, grad_hook=hook)
        for m in range(2):
            close(logs[m][0], g['s%d.L%d.loss' % (step, m)], 5e-4, 0)
            close(logs[m][2], g['s%d.L%d.depth_loss' % (step, m)], 5e-4, 0)
            ok = {k: np.abs(v) > SIGN_FLOOR * np.sqrt(np.mean(v.astype(np.float64) ** 2)) for k, v in seen[m].items()}
            above[m] = ok if above[m] is None else {k: above[m][k] & ok[k] for k in ok}
        if step in (1, 3):
            tol = STEP1_TOL if step == 1 else STEP3_TOL
            worst, n_above, n_all, n_flip = 0.0, 0, 0, 0
            for m in range(2):
                now = unflat(N(engines[m].params))
                for k in O.param_order():
                    idx = g['after%d.L%d.%s.idx' % (step, m, k)]
                    mine, ref = now[k].reshape(-1)[idx], g['after%d.L%d.%s.val' % (step, m, k)]
                    sel = above[m][k].reshape(-1)[idx]
                    err = np.abs(mine - ref)
                    if sel.any():
                        worst = max(worst, float(err[sel].max()))
                        assert err[sel].max() <= tol, (step, m, k, float(err[sel].max()))
                    n_above += int(sel.sum()); n_all += sel.size
                    n_flip += int((err[~sel] > 0.5 * 5e-4).sum()) if step == 1 else 0
                    # nobody moves further than the Adam bound of `step` steps from where the reference is
                    assert err.max() <= 2 * 5e-4 * step * 1.001, (step, m, k, float(err.max()))
                    if step == 1:         # and everybody took a step of exactly lr in one direction or the other
                        d0 = np.abs(mine - init[m][k].reshape(-1)[idx])
                        assert np.all((np.abs(d0 - 5e-4) < 1e-6) | (d0 < 5e-4)), (m, k)
            stats[step] = dict(worst_above_floor=worst, frac_above_floor=n_above / n_all, sign_flips_below_floor=n_flip)
    print('three-step report', stats)
    assert stats[1]['frac_above_floor'] > 0.5 and stats[3]['frac_above_floor'] > 0.3, stats     # the gate covers the bulk


# ----------------------------------------------------------------------------------------- full size
@pytest.mark.parametrize('prec', [1, 2])
def test_full_size_properties(ops, levels, prec):
    """BASELINE sizes (1024 rays, 64 + 128 samples): size-independent properties."""
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    b = SyntheticKitti().random_batch(1024, np.random.RandomState(3))
    ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
    far, fg_z, bg_z = ops.sample_coarse(ray_o, ray_d, T(b['min_depth']), 64)
    e0 = ops.LevelEngine(T(flat(levels[0])), precision=prec)
    e1 = ops.LevelEngine(T(flat(levels[1])), precision=prec)
    r0 = e0.forward(ray_o, ray_d, far, fg_z, bg_z)
    fg1 = ops.sample_fine(fg_z, r0['fg_weights'], 128)
    bg1 = ops.sample_fine(bg_z, r0['bg_weights'], 128)
    assert fg1.shape == (1024, 192)
    assert bool((fg1[:, 1:] >= fg1[:, :-1]).all()) and bool((bg1[:, 1:] >= bg1[:, :-1]).all())
    # the 64 old depths survive the merge (multiset containment, checked through sums of sorted sets)
    assert bool((fg1.min(1)[0] <= fg_z.min(1)[0]).all()) and bool((fg1.max(1)[0] >= fg_z.max(1)[0]).all())
    r1 = e1.forward(ray_o, ray_d, far, fg1, bg1, training=True)
    for r in (r0, r1):
        rgb = N(r['rgb'])
        assert np.isfinite(rgb).all() and rgb.min() >= 0 and rgb.max() <= 1 + 1e-4
        w = N(r['fg_weights'])
        assert w.min() >= 0 and (w.sum(-1) + N(r['bg_lambda']) <= 1 + 2e-3).all()
        close(N(r['rgb']), N(r['fg_rgb']) + N(r['bg_rgb']), 1e-5, 1e-6)
    # determinism: same inputs -> bit-identical outputs and gradients
    r1b = e1.forward(ray_o, ray_d, far, fg1, bg1, training=True)
    assert torch.equal(r1['rgb'], r1b['rgb']) and torch.equal(r1['fg_weights'], r1b['fg_weights'])
    sc, g_rgb, g_depth, g_w = ops.loss_and_grads(r1, T(b['rgb']), T(b['depth_sup']), 'mse', 0.1)
    ga = e1.backward(g_rgb, g_depth, g_w).clone()
    gb = e1.backward(g_rgb, g_depth, g_w)
    assert torch.equal(ga, gb)
    assert bool(torch.isfinite(ga).all()) and float(ga.abs().max()) > 0


def test_autoexposure_training_steps_match_reference(ops, golden):
    """SURVEY 8a row a11 end to end: NerfppTrainer with optim_autoexpo (one cascade level, split-bf16)
    through the reference's 4 steps of tests/golden/autoexpo.npz -- losses, exposure parameters after
    every step, and a network tensor at the end (its gradient goes through the 1/scale^2 factor)."""
    from outdoor_nerf_depth_amd.trainer import NerfppTrainer
    from outdoor_nerf_depth_amd.model import init_level_params, state_dict_from_flat
    g = golden('autoexpo')
    names = [str(x) for x in g['names']]
    tr = NerfppTrainer(dev(), precision=2, cascade_samples=(64,), use_depth=True, depth_loss_type='mse',
                       lambda_depth=float(g['lambda_depth']), level_params=init_level_params(1),
                       optim_autoexpo=True, img_names=names, lambda_autoexpo=float(g['lambda_autoexpo']))
    for step in range(1, 5):
        img = int(g['s%d.img' % step])
        batch = {k: T(g['s%d.%s' % (step, k)]) for k in ('ray_o', 'ray_d', 'rgb', 'depth_sup', 'min_depth')}
        batch['img_name'] = names[img]
        sc = tr.train_step(batch, uniforms={'t_fg': T(g['s%d.t_fg' % step]), 't_bg': T(g['s%d.t_bg' % step])})
        close(N(sc[0])[0], g['s%d.loss' % step], 5e-4, 0)
        close(N(sc[0])[1], g['s%d.rgb_loss' % step], 5e-4, 1e-7)
        close(float(tr.last_autoexpo[0][0]), g['s%d.scale' % step], 1e-6, 0)
        tr.flush()                                    # the parameter updates run on the trainer's side stream
        np.testing.assert_allclose(N(tr.autoexpo[0].params), g['s%d.params_after' % step], rtol=2e-4, atol=2e-7)
    w = state_dict_from_flat(tr.engines[0].params)['module.nerf_net.fg_net.rgb_layers.2.weight']
    np.testing.assert_allclose(N(w), g['final.fg_rgb2_weight'], rtol=0, atol=2e-4)


def test_split_forward_bf16_backward_mode(ops, golden, levels):
    """PREC_SPLIT_FWD: the forward is the split-bf16 one bit for bit (so rendered outputs and loss keep the
    1e-4 contract); the backward is the single-pass bf16 chain over the hi planes it saved (bf16-mode
    gradient bounds against the float64 reference run)."""
    from outdoor_nerf_depth_amd import _lib as L
    g = golden('grads_mse')
    for m in range(2):
        fz, bz = g['L%d.fg_z' % m], g['L%d.bg_z' % m]
        e_split = ops.LevelEngine(T(flat(levels[m])), precision=2)
        e_hyb = ops.LevelEngine(T(flat(levels[m])), precision=L.PREC_SPLIT_FWD)
        e_bf16 = ops.LevelEngine(T(flat(levels[m])), precision=1)
        args = (T(g['ray_o']), T(g['ray_d']), T(g['fg_far']), T(fz), T(bz))
        r_s, r_h = e_split.forward(*args, training=True), e_hyb.forward(*args, training=True)
        for k in r_s:
            assert torch.equal(r_s[k], r_h[k]), k
        sc, g_rgb, g_depth, g_w = ops.loss_and_grads(r_h, T(g['rgb_gt']), T(g['depth_sup']), 'mse', 0.1)
        gh = unflat(N(e_hyb.backward(g_rgb, g_depth, g_w)))
        e_bf16.forward(*args, training=True)
        gb = unflat(N(e_bf16.backward(g_rgb, g_depth, g_w)))
        for k in O.param_order():
            ref = g['L%d.%s.g64' % (m, k)] if ('L%d.%s.g64' % (m, k)) in g else None
            # same upstream gradients, same masks up to bf16 sign flips: the two bf16 backward results agree closely
            rel = np.linalg.norm(gh[k] - gb[k]) / (np.linalg.norm(gb[k]) + 1e-30)
            assert rel <= 0.3, (m, k, rel)
            assert np.isfinite(gh[k]).all()
