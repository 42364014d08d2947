"""GPU parity tests added in round 2:

  * the in-kernel sampling RNG against the oracle's Philox, bit for bit, and the *_rng entry points
    against the explicit-uniform ones;
  * forward outputs AND losses (mse / l1 / kl) against the oracle at the sizes the bench runs
    (1024 rays x 64 and x 192 samples), split-bf16 at 1e-4, single-pass bf16 against the oracle run with
    bf16-rounded GEMM operands;
  * BASELINE config 1 (rgb-only, --cascade_samples 32,64, one 64x64 frame) step by step against the oracle with
    replayed uniforms, through NerfppTrainer and through the CLI;
  * measured single-pass-bf16 errors against the float64 reference run, bounded by 2x the values recorded in
    profiles/r02_bf16_error_report.json;
  * the per-step "cameras inside the unit sphere" exception of ddp_train_nerf.py:62-63 in the training loop.
"""
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import nerfpp_oracle as O                                   # noqa: E402
from tests.test_gpu_parity import T, N, flat, unflat, dev, close        # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def ops():
    dev()
    from outdoor_nerf_depth_amd import ops as _ops
    return _ops


@pytest.fixture(scope='module')
def levels():
    return O.init_params_like_reference(2)


# ------------------------------------------------------------------------------------------- RNG
def test_inkernel_rng_is_the_oracles_philox(ops):
    d = dev()
    for seed, step in ((777, 1), (1554, 12345), (2 ** 40 + 3, 2 ** 31 + 5)):
        for sid in range(4):
            got = N(ops.rng_uniform(seed, step, sid, (1000, 7), d)).reshape(-1)
            np.testing.assert_array_equal(got, O.philox_uniform(seed, step, sid, 7000))
    u = N(ops.rng_uniform(777, 3, 2, (1 << 20,), d))
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1 / 12.) < 1e-3


def test_rng_entry_points_equal_explicit_uniform_ones(ops, levels):
    """nerfpp_sample_coarse_rng / nerfpp_sample_fine_pair_rng draw exactly philox_uniform(seed, step, stream):
    feeding those uniforms to the explicit entry points gives bit-identical depths."""
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    n, seed, step = 300, 1554, 9
    b = SyntheticKitti().random_batch(n, np.random.RandomState(1))
    uni = O.step_uniforms(seed, step, n, 64, 128)
    ray_o, ray_d, md = T(b['ray_o']), T(b['ray_d']), T(b['min_depth'])
    far_a, fg_a, bg_a = ops.sample_coarse(ray_o, ray_d, md, 64, rng=(seed, step))
    far_b, fg_b, bg_b = ops.sample_coarse(ray_o, ray_d, md, 64, t_rand_fg=T(uni['t_fg']), t_rand_bg=T(uni['t_bg']))
    assert torch.equal(fg_a, fg_b) and torch.equal(bg_a, bg_b) and torch.equal(far_a, far_b)
    far_o = O.intersect_sphere(b['ray_o'], b['ray_d'])
    fg_o, bg_o = O.coarse_depths(b['min_depth'], far_o, 64)
    np.testing.assert_array_equal(N(fg_a), O.perturb_samples(fg_o, uni['t_fg']))
    eng = ops.LevelEngine(T(flat(levels[0])), precision=2)
    ret = eng.forward(ray_o, ray_d, far_a, fg_a, bg_a)
    f1, b1 = ops.sample_fine_pair(fg_a, ret['fg_weights'], bg_a, ret['bg_weights'], 128, rng=(seed, step))
    f2, b2 = ops.sample_fine_pair(fg_a, ret['fg_weights'], bg_a, ret['bg_weights'], 128, u_fg=T(uni['u_fg']),
                                  u_bg=T(uni['u_bg']))
    assert torch.equal(f1, f2) and torch.equal(b1, b2)
    m_o, _, _ = O.fine_depths(N(fg_a), N(ret['fg_weights']), uni['u_fg'])
    np.testing.assert_array_equal(N(f1), m_o)


# ------------------------------------------------------------------------------------------- full size
def _full_size_case(ops, levels, n=1024):
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    b = SyntheticKitti(depth_sup_type='mono_crop').random_batch(n, np.random.RandomState(11))
    b['depth_sup'][::7] = 0.0                                        # a sparse hole pattern in the dense prior
    uni = O.step_uniforms(777, 1, n, 64, 128)
    ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
    far, fg0, bg0 = ops.sample_coarse(ray_o, ray_d, T(b['min_depth']), 64, t_rand_fg=T(uni['t_fg']),
                                      t_rand_bg=T(uni['t_bg']))
    e0 = ops.LevelEngine(T(flat(levels[0])), precision=2)
    r0 = e0.forward(ray_o, ray_d, far, fg0, bg0)
    fg1, bg1 = ops.sample_fine_pair(fg0, r0['fg_weights'], bg0, r0['bg_weights'], 128, u_fg=T(uni['u_fg']),
                                    u_bg=T(uni['u_bg']))
    return b, far, (fg0, bg0), (fg1, bg1)


# measured on MI355X (profiles/r02_bf16_error_report.json): bf16 kernels vs the oracle with bf16-rounded operands
BF16_VS_BF16_ORACLE = dict(rtol=2e-2, atol=2e-3)


@pytest.mark.parametrize('level', [0, 1])
def test_full_size_forward_and_losses_match_oracle(ops, levels, level):
    """1024 rays x 64 (level 0) and x 192 (level 1) samples -- the sizes bench.py runs -- against the oracle's
    forward: every returned tensor at 1e-4 (split-bf16), and the loss head for mse, l1 and kl on those outputs."""
    b, far, z0, z1 = _full_size_case(ops, levels)
    fg_z, bg_z = (z0, z1)[level]
    ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
    ref = O.nerf_forward(levels[level], b['ray_o'], b['ray_d'], N(far), N(fg_z), N(bg_z))
    eng = ops.LevelEngine(T(flat(levels[level])), precision=2)
    ret = eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=True)
    for k in ops.RET_KEYS:
        tol = dict(rtol=1e-4, atol=2e-6)
        if k in ('bg_depth', 'depth'):
            tol['atol'] *= max(1.0, float(np.abs(ref[k]).max()))
        np.testing.assert_allclose(N(ret[k]), ref[k], err_msg=k, **tol)
    for mode in ('mse', 'l1', 'kl'):
        sc, g_rgb, g_depth, g_w = ops.loss_and_grads(ret, T(b['rgb']), T(b['depth_sup']), mode, 0.1, kl_sigma=0.01,
                                                     fg_z_vals=fg_z, fg_far_depth=far)
        loss, rgb_loss, depth_loss, o_rgb, o_depth, o_w = O.loss_and_grads(ref, N(fg_z), N(far), b['rgb'], b['depth_sup'],
                                                                            True, mode, 0.1, 0.01)
        close(N(sc)[1], rgb_loss, 1e-4, 0)
        close(N(sc)[2], depth_loss, 1e-4, 0)
        close(N(sc)[0], loss, 1e-4, 0)
        np.testing.assert_allclose(N(g_rgb), o_rgb, rtol=2e-4, atol=1e-9)
        np.testing.assert_allclose(N(g_depth), o_depth, rtol=2e-4, atol=1e-9)
        if mode == 'kl':
            # g_w = -lambda e dists / ((w + 1e-5) S) amplifies the 1e-4-level forward differences in w by w / (w + 1e-5)
            # where w ~ 0: the kernel's arithmetic is checked on ITS OWN forward outputs (tight), the cross-path value loosely
            own = {k: N(ret[k]) for k in ('rgb', 'depth', 'fg_weights', 'fg_dists')}
            k_w = O.loss_and_grads(own, N(fg_z), N(far), b['rgb'], b['depth_sup'], True, mode, 0.1, 0.01)[5]
            np.testing.assert_allclose(N(g_w), k_w, rtol=2e-5, atol=1e-9)
            np.testing.assert_allclose(N(g_w), o_w, rtol=2e-2, atol=1e-7)
    # single-pass bf16 at the same size, against the oracle run with bf16-rounded GEMM operands
    ref16 = O.nerf_forward(levels[level], b['ray_o'], b['ray_d'], N(far), N(fg_z), N(bg_z), bf16=True)
    e16 = ops.LevelEngine(T(flat(levels[level])), precision=1)
    r16 = e16.forward(ray_o, ray_d, far, fg_z, bg_z, training=True)
    for k in ('rgb', 'fg_rgb', 'fg_weights', 'bg_lambda', 'fg_depth'):
        np.testing.assert_allclose(N(r16[k]), ref16[k], err_msg='bf16 ' + k, **BF16_VS_BF16_ORACLE)
    for mode in ('mse', 'l1', 'kl'):
        sc, _, _, _ = ops.loss_and_grads(r16, T(b['rgb']), T(b['depth_sup']), mode, 0.1, kl_sigma=0.01, fg_z_vals=fg_z,
                                         fg_far_depth=far)
        loss = O.loss_and_grads(ref16, N(fg_z), N(far), b['rgb'], b['depth_sup'], True, mode, 0.1, 0.01)[0]
        close(N(sc)[0], loss, 2e-2, 0)


def test_bf16_gradients_match_bf16_oracle(ops, levels):
    """a tight gate for the precision behind the headline number.  The single-pass bf16 kernels
    against the oracle's closed-form backward evaluated with bf16-rounded weights / activations / dZ: a dropped
    K-slice, a wrong ReLU mask or a mis-packed weight block shows up as O(1) here, while agreement is bounded by
    bf16 rounding flips only."""
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    n, S = 270, 64
    b = SyntheticKitti().random_batch(n, np.random.RandomState(5))
    b['depth_sup'][:90] = np.float32(0.05)
    rs = np.random.RandomState(6)
    far = O.intersect_sphere(b['ray_o'], b['ray_d'])
    fg, bg = O.coarse_depths(b['min_depth'], far, S)
    fg = O.perturb_samples(fg, rs.rand(n, S).astype(np.float32))
    bg = O.perturb_samples(bg, rs.rand(n, S).astype(np.float32))
    cache = {}
    ret_o = O.nerf_forward(levels[0], b['ray_o'], b['ray_d'], far, fg, bg, cache=cache, bf16=True)
    _, _, _, o_rgb, o_depth, o_w = O.loss_and_grads(ret_o, fg, far, b['rgb'], b['depth_sup'], True, 'mse', 0.1, 0.01)
    g_o = O.nerf_backward(cache, o_rgb, o_depth, o_w, bf16=True)
    eng = ops.LevelEngine(T(flat(levels[0])), precision=1)
    ret = eng.forward(T(b['ray_o']), T(b['ray_d']), T(far), T(fg), T(bg), training=True)
    np.testing.assert_allclose(N(ret['rgb']), ret_o['rgb'], **BF16_VS_BF16_ORACLE)
    grads = unflat(N(eng.backward(T(o_rgb), T(o_depth), None)))
    worst = {}
    for k in O.param_order():
        rel = np.linalg.norm(grads[k] - g_o[k]) / (np.linalg.norm(g_o[k]) + 1e-30)
        worst[k] = rel
        assert rel <= (0.2 if grads[k].size <= 3 else 0.1), (k, rel)
    assert np.median(list(worst.values())) <= 0.05, worst


def test_bf16_errors_within_twice_the_recorded_measurement(ops, golden, levels):
    """the single-pass bf16 bounds are 2x the errors measured on MI355X and recorded in
    profiles/r02_bf16_error_report.json (tools/grad_error_report.py --json), not loose ceilings."""
    path = os.path.join(ROOT, 'profiles', 'r02_bf16_error_report.json')
    if not os.path.exists(path):
        pytest.skip('profiles/r02_bf16_error_report.json not recorded yet')
    rec = json.load(open(path))
    from tools.grad_error_report import measure
    now = measure(modes=('mse', 'kl'))
    for key, v in now['bf16'].items():
        lim = 2.0 * rec['bf16'][key] + 1e-6
        assert v <= lim, (key, v, rec['bf16'][key])


# ------------------------------------------------------------------------------------------- config 1
def _config1_scene():
    from outdoor_nerf_depth_amd.data_loader_split import synthetic_ray_samplers
    return synthetic_ray_samplers('train', 1, 'gt', 1, 64, 64)


def test_config1_rgbonly_32_coarse_samples_steps_match_oracle(ops):
    """BASELINE config 1: rgb-only, cascade_samples 32,64, one 64x64 frame.  NerfppTrainer with its in-kernel RNG
    for 3 steps against the oracle's train_step fed the same Philox uniforms: level-0 loss at 1e-4 per step (its
    inputs are bit-identical), level 1 within the band its re-sampled depths allow, parameters after the steps."""
    from outdoor_nerf_depth_amd.trainer import NerfppTrainer
    from outdoor_nerf_depth_amd.model import init_level_params
    sampler = _config1_scene()[0]
    assert (sampler.H, sampler.W) == (64, 64)
    n, seed = 256, 777
    tr = NerfppTrainer(dev(), precision=2, cascade_samples=(32, 64), use_depth=False, level_params=init_level_params(2),
                       seed=seed)
    lv = O.init_params_like_reference(2)
    opt = O.new_opt_state(lv)
    np.random.seed(3)
    for step in range(1, 4):
        b = sampler.random_sample(n)
        for k, v in b.items():                        # the one-frame scene used to normalise to NaN poses
            if isinstance(v, np.ndarray) and v.dtype.kind == 'f':
                assert np.isfinite(v).all(), 'non-finite %s in the config-1 batch' % k
        bd = {k: T(np.asarray(v, np.float32)) for k, v in b.items() if isinstance(v, np.ndarray)}
        sc = tr.train_step(bd)
        uni = O.step_uniforms(seed, step, n, 32, 64)
        logs, rets = O.train_step(lv, opt, step, b, uni, cascade_samples=(32, 64), use_depth=False)
        for m in range(2):
            assert np.isfinite(logs[m]['loss']) and np.isfinite(N(sc[m])[:2]).all(), (step, m, logs[m], N(sc[m]))
        close(N(sc[0])[0], logs[0]['loss'], 1e-4, 0)
        close(N(sc[1])[0], logs[1]['loss'], 2e-3, 0)
        assert N(sc[0])[2] == 0 and rets[0][0]['fg_weights'].shape == (n, 32) and rets[1][0]['fg_weights'].shape == (n, 96)
    tr.flush()                                        # the parameter updates run on the trainer's side stream
    for m in range(2):
        now = unflat(N(tr.engines[m].params))
        for k in O.param_order():
            assert np.isfinite(now[k]).all() and np.isfinite(lv[m][k]).all(), (m, k)
            bad = np.abs(now[k] - lv[m][k]) > 2.5e-4
            assert bad.mean() < 0.10, (m, k, bad.mean())


def test_config1_cli_runs(tmp_path):
    from outdoor_nerf_depth_amd import ddp_train_nerf as C
    args = C.config_parser().parse_args(
        ['--expname', 'c1', '--basedir', str(tmp_path), '--synthetic', '--synthetic_hw', '64,64', '--synthetic_frames', '1',
         '--cascade_samples', '32,64', '--world_size', '1', '--N_rand_override', '256', '--N_iters', '3', '--i_weights', '2',
         '--i_print', '1', '--testskip', '1'])
    assert not args.use_depth
    C.validate_args(args)
    args.world_size = 1
    C.ddp_train_nerf(0, args)
    ck = torch.load(tmp_path / 'c1' / 'model_000002.pth', map_location='cpu', weights_only=False)
    assert ck['net_0']['module.nerf_net.fg_net.base_layers.0.0.weight'].shape == (256, 63)
    for net in ('net_0', 'net_1'):                    # a NaN scene used to train (and checkpoint) NaN weights silently
        for k, v in ck[net].items():
            assert bool(torch.isfinite(v).all()), (net, k)
    for opt in ('optim_0', 'optim_1'):
        for st in ck[opt]['state'].values():
            assert bool(torch.isfinite(st['exp_avg']).all()) and bool(torch.isfinite(st['exp_avg_sq']).all())


# ------------------------------------------------------------------------------------------- error behaviour
def test_training_loop_raises_the_unit_sphere_exception(ops):
    """ddp_train_nerf.py:62-63 raises when a camera is outside the unit sphere; the trainer keeps the device-side
    count across steps and raises the same exception wherever the loop synchronises (check_cameras)."""
    from outdoor_nerf_depth_amd.trainer import NerfppTrainer
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    tr = NerfppTrainer(dev(), precision=1, use_depth=False)
    b = SyntheticKitti().random_batch(64, np.random.RandomState(0))
    tr.train_step({k: T(v) for k, v in b.items() if isinstance(v, np.ndarray)})
    tr.check_cameras()                                              # fine
    tr.flush()
    before = [N(e.params).copy() for e in tr.engines]
    mom = [N(m).copy() for m in tr.exp_avg]
    b['ray_o'][5] = np.array([3.0, 0.0, 0.0], np.float32)          # far outside, pointing away
    b['ray_d'][5] = np.array([0.0, 1.0, 0.0], np.float32)
    tr.train_step({k: T(v) for k, v in b.items() if isinstance(v, np.ndarray)})
    good = SyntheticKitti().random_batch(64, np.random.RandomState(1))
    tr.train_step({k: T(v) for k, v in good.items() if isinstance(v, np.ndarray)})   # a later, clean step before the check
    tr.flush()
    # the reference raises BEFORE the step, so neither the poisoned update nor any later one may reach the
    # parameters or the Adam moments until the deferred check has raised (device-side predicate of the Adam kernel)
    for m in range(2):
        np.testing.assert_array_equal(N(tr.engines[m].params), before[m])
        np.testing.assert_array_equal(N(tr.exp_avg[m]), mom[m])
        assert np.isfinite(N(tr.engines[m].params)).all()
    with pytest.raises(Exception, match='bounded by the unit sphere'):
        tr.check_cameras()
    tr.check_cameras()                                              # the counter was reset
    tr.train_step({k: T(v) for k, v in good.items() if isinstance(v, np.ndarray)})
    tr.flush()
    assert not np.array_equal(N(tr.engines[0].params), before[0])   # updates resume after the check


def test_checkpoint_of_a_poisoned_run_is_refused(tmp_path):
    from outdoor_nerf_depth_amd.trainer import NerfppTrainer
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    from outdoor_nerf_depth_amd.ddp_train_nerf import save_checkpoint
    tr = NerfppTrainer(dev(), precision=1, use_depth=False)
    b = SyntheticKitti().random_batch(64, np.random.RandomState(0))
    b['ray_o'][0] = np.array([0.0, 2.5, 0.0], np.float32)
    b['ray_d'][0] = np.array([1.0, 0.0, 0.0], np.float32)
    tr.train_step({k: T(v) for k, v in b.items() if isinstance(v, np.ndarray)})
    with pytest.raises(Exception, match='bounded by the unit sphere'):
        save_checkpoint(str(tmp_path / 'model_000001.pth'), tr, 1)
    assert not (tmp_path / 'model_000001.pth').exists()
