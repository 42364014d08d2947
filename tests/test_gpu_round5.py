"""GPU, round 5: the fp16x2w forward precision (NERFPP_PREC_FP16X2W = 3) and the reference's seed spread.

VERDICT r04 item 3: is there a forward cheaper than three bf16 MFMA passes that still carries north_star's 1e-4 clause?  The
candidate the CPU emulation selected (tools/operand_format_study.py): weights hi + lo in fp16, activations rounded to fp16
once, two passes of v_mfma_f32_32x32x16_f16.  Built as precision 3 of the forward kernels and tested through the C ABI against
the SAME tolerances as the split-bf16 forward (tests/test_gpu_parity.py RET_TOL[2]: rtol 1e-4 + atol 2e-6 on every tensor):

* the imported reference's NerfNet.forward vectors (tests/golden/forward.npz), inference and training mode;
* ragged sizes and the bench sizes (1024 rays x 64 / 192) against the oracle, with the loss head for mse / l1 / kl;
* on TRAINED weights (1000 optimisation steps on the config-1 scene) it is NOT inside 1e-4 any more (rgb 3-4e-4): the test
  pins the measured bounds and the fact, so the mode stays labelled an intermediate precision, not the parity mode;
* its training-mode saves are what the bf16 backward expects: gradients equal to the split_fwd combination's within the
  bf16 noise, trajectory gates of tests/test_gpu_round4.py (EARLY_GATE) for the PREC_FP16_FWD trainer.
"""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, 'golden')
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import nerfpp_oracle as O          # noqa: E402  (the checker)

RET_TOL = dict(rtol=1e-4, atol=2e-6)           # = tests/test_gpu_parity.py RET_TOL[2]


def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def T(x, d=None):
    return torch.from_numpy(np.ascontiguousarray(x)).to(d or dev())


def N(t):
    return t.detach().cpu().numpy()


def flat(level):
    return np.concatenate([level[k].reshape(-1) for k in O.param_order()]).astype(np.float32)


@pytest.fixture(scope='module')
def levels():
    return O.init_params_like_reference(2)


def _tol(k, ref, scale=1.0):
    tol = dict(rtol=RET_TOL['rtol'] * scale, atol=RET_TOL['atol'] * scale)
    if k in ('bg_depth', 'depth'):              # sums of terms up to 1e6 (1 / (z + eps)): scale atol
        tol['atol'] *= max(1.0, float(np.abs(ref).max()))
    return tol


def gate_ratio(got, ref, k):
    tol = _tol(k, ref)
    return float(np.max(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)) / (tol['rtol'] * np.abs(ref) + tol['atol'])))


# ------------------------------------------------------------------------------------------- forward parity
@pytest.mark.parametrize('training', [False, True])
def test_fp16x2w_forward_matches_reference(levels, training):
    """tests/golden/forward.npz = the imported reference's NerfNet.forward at 64 and 192 samples; same gate as split-bf16.
    Measured worst ratio error / allowed: 0.30 (level 0), 0.33 (level 1) -- the CPU emulation had predicted 0.41 / 0.27."""
    from outdoor_nerf_depth_amd import ops, _lib as L
    g = np.load(os.path.join(GOLD, 'forward.npz'))
    for m, (fz, bz) in enumerate((('fg_z0', 'bg_z0'), ('fg_z1', 'bg_z1'))):
        eng = ops.LevelEngine(T(flat(levels[m])), precision=L.PREC_FP16_FWD)
        ret = eng.forward(T(g['ray_o']), T(g['ray_d']), T(g['fg_far']), T(g[fz]), T(g[bz]), training=training)
        assert list(ret.keys()) == list(ops.RET_KEYS)
        worst = 0.0
        for k, v in ret.items():
            ref = g['L%d.%s' % (m, k)]
            np.testing.assert_allclose(N(v), ref, err_msg='L%d.%s' % (m, k), **_tol(k, ref))
            worst = max(worst, gate_ratio(N(v), ref, k))
        assert worst <= 0.6, worst               # margin: a format change that eats it shows up here first


@pytest.mark.parametrize('n_rays,S', [(1, 64), (5, 64), (7, 192), (33, 33), (3, 256)])
def test_fp16x2w_forward_ragged_sizes_match_oracle(levels, n_rays, S):
    from outdoor_nerf_depth_amd import ops, _lib as L
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    b = SyntheticKitti().random_batch(n_rays, np.random.RandomState(n_rays + S))
    rs = np.random.RandomState(S)
    far = O.intersect_sphere(b['ray_o'], b['ray_d'])
    fg, bg = O.coarse_depths(b['min_depth'], far, S)
    fg = O.perturb_samples(fg, rs.rand(n_rays, S).astype(np.float32))
    bg = O.perturb_samples(bg, rs.rand(n_rays, S).astype(np.float32))
    ref = O.nerf_forward(levels[0], b['ray_o'], b['ray_d'], far, fg, bg)
    eng = ops.LevelEngine(T(flat(levels[0])), precision=L.PREC_FP16_FWD)
    for training in (False, True):
        ret = eng.forward(T(b['ray_o']), T(b['ray_d']), T(far), T(fg), T(bg), training=training)
        for k in ('rgb', 'fg_weights', 'bg_weights', 'fg_dists', 'fg_depth', 'bg_lambda'):
            np.testing.assert_allclose(N(ret[k]), ref[k], rtol=2e-4, atol=3e-6, err_msg=k)      # (= the split-bf16 ragged test)


def _full_size_case(ops, levels):
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    b = SyntheticKitti().random_batch(1024, np.random.RandomState(20230804))
    uni = O.step_uniforms(777, 1, 1024, 64, 128)
    ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
    far, fg0, bg0 = ops.sample_coarse(ray_o, ray_d, T(b['min_depth']), 64, T(uni['t_fg']), T(uni['t_bg']))
    e0 = ops.LevelEngine(T(flat(levels[0])), precision=2)
    r0 = e0.forward(ray_o, ray_d, far, fg0, bg0)
    fg1, bg1 = ops.sample_fine_pair(fg0, r0['fg_weights'], bg0, r0['bg_weights'], 128, u_fg=T(uni['u_fg']), u_bg=T(uni['u_bg']))
    return b, far, (fg0, bg0), (fg1, bg1)


@pytest.mark.parametrize('level', [0, 1])
def test_fp16x2w_full_size_forward_and_losses_match_oracle(levels, level):
    """1024 rays x 64 / x 192 -- the sizes bench.py runs -- against the oracle's float32 forward: every returned tensor at
    1e-4, and the logged losses of mse / l1 / kl on those outputs at 1e-4 relative."""
    from outdoor_nerf_depth_amd import ops, _lib as L
    b, far, z0, z1 = _full_size_case(ops, levels)
    fg_z, bg_z = (z0, z1)[level]
    ref = O.nerf_forward(levels[level], b['ray_o'], b['ray_d'], N(far), N(fg_z), N(bg_z))
    eng = ops.LevelEngine(T(flat(levels[level])), precision=L.PREC_FP16_FWD)
    ret = eng.forward(T(b['ray_o']), T(b['ray_d']), far, fg_z, bg_z, training=True)
    for k in ops.RET_KEYS:
        np.testing.assert_allclose(N(ret[k]), ref[k], err_msg=k, **_tol(k, ref[k]))
    for mode in ('mse', 'l1', 'kl'):
        sc = ops.loss_and_grads(ret, T(b['rgb']), T(b['depth_sup']), mode, 0.1, kl_sigma=0.01, fg_z_vals=fg_z, fg_far_depth=far)[0]
        loss, rgb_loss, depth_loss = O.loss_and_grads(ref, N(fg_z), N(far), b['rgb'], b['depth_sup'], True, mode, 0.1, 0.01)[:3]
        np.testing.assert_allclose(N(sc)[:3], [loss, rgb_loss, depth_loss], rtol=1e-4, atol=0)


# ------------------------------------------------------------------------------------------- trained weights
def _train(mode, n_steps, precision):
    import trajectory_common as TC
    from outdoor_nerf_depth_amd.trainer import NerfppTrainer
    d = dev()
    smp = TC.sampler(mode)
    tr = NerfppTrainer(d, precision=precision, cascade_samples=TC.CASCADE, use_depth=True, depth_loss_type=mode,
                       lambda_depth=TC.LAMBDA_DEPTH, depth_sigma=TC.DEPTH_SIGMA, depth_scale=float(smp.get_depth_scale() or 1.0))
    for step in range(1, n_steps + 1):
        b, uni = TC.step_batch(smp, step), TC.step_uniforms(step)
        tr.train_step({k: T(v, d) for k, v in b.items()}, uniforms={k: T(v, d) for k, v in uni.items()})
    tr.flush()
    torch.cuda.synchronize()
    return tr, smp


# Measured on trained weights (tools/probes/p3_trained_check.py, profiles/r05_fp16x2w_trained_weights.json; ratio = max |err| /
# (1e-4 |ref| + atol) against the float32 oracle): split-bf16 0.04-0.7 (1.4-1.6 on single tiny fg_weights after 3000 kl steps),
# fp16x2w 2.8-3.7 on rgb (gt + mse) and up to 35-80 on single per-sample weights (mono_crop + kl: sharp densities), bf16 20-1000.
TRAINED_BOUND = {'rgb': 10.0, 'depth': 20.0, 'weights_rel_l2': 3e-3}


@pytest.mark.parametrize('mode', ['mse', 'kl'])
def test_fp16x2w_forward_on_trained_weights_is_not_the_parity_mode(mode):
    """The finding that closes VERDICT r04 item 3.  At initialisation the fp16x2w forward is inside 1e-4 with 2x margin (tests
    above, CPU emulation); on weights that have been TRAINED -- 1000 steps on the config-1 scene, gt + mse / mono_crop + kl --
    its one fp16 rounding per activation no longer averages out: rendered rgb is 3-4e-4 from the split-bf16 forward (itself
    within 1e-4 of float32 there), single per-sample weights up to 1e-2 relative.  A factor 10 tighter than single-pass bf16, a
    factor 50 looser than split-bf16: an intermediate precision, NOT a carrier of north_star's 1e-4 clause.  Asserted: the
    measured bounds (so a regression shows), and that the 1e-4 gate is in fact exceeded (so nobody re-labels the mode)."""
    import trajectory_common as TC
    from outdoor_nerf_depth_amd import ops, _lib as L
    tr, smp = _train(mode, 1000, L.PREC_SPLIT_BF16)
    d = dev()
    b, uni = TC.step_batch(smp, 5001), TC.step_uniforms(5001)
    ray_o, ray_d = T(b['ray_o'], d), T(b['ray_d'], d)
    far, fg, bg = ops.sample_coarse(ray_o, ray_d, T(b['min_depth'], d), TC.CASCADE[0], T(uni['t_fg'], d), T(uni['t_bg'], d))
    report, worst_all = {}, 0.0
    for m in range(2):
        e2 = tr.engines[m]
        e3 = ops.LevelEngine(e2.params.clone(), precision=L.PREC_FP16_FWD)
        r2 = e2.forward(ray_o, ray_d, far, fg, bg)
        r3 = e3.forward(ray_o, ray_d, far, fg, bg)
        rs = {k: gate_ratio(N(r3[k]), N(r2[k]), k) for k in ops.RET_KEYS}
        report['level%d' % m] = rs
        worst_all = max(worst_all, max(rs.values()))
        for k in ('rgb', 'fg_rgb', 'bg_rgb'):
            assert rs[k] <= TRAINED_BOUND['rgb'], (m, k, rs)
        for k in ('depth', 'fg_depth', 'bg_depth', 'bg_lambda'):
            assert rs[k] <= TRAINED_BOUND['depth'], (m, k, rs)
        for k in ('fg_weights', 'bg_weights'):
            rel = float((r3[k] - r2[k]).norm() / r2[k].norm())
            assert rel <= TRAINED_BOUND['weights_rel_l2'], (m, k, rel)
        if m == 0:
            fg, bg = ops.sample_fine_pair(fg, r2['fg_weights'], bg, r2['bg_weights'], TC.CASCADE[1], u_fg=T(uni['u_fg'], d), u_bg=T(uni['u_bg'], d))
    print('trained-weights report', mode, report)
    assert worst_all > 1.0, report          # (if this ever fails the format DOES carry 1e-4 on trained weights: re-open the question)


# ------------------------------------------------------------------------------------------- what the backward sees
def test_fp16_fwd_saves_feed_the_bf16_backward(levels):
    """Training-mode saves of the fp16x2w forward are single-plane bf16 tensors in precision 1's workspace layout: each is the
    split-bf16 forward's hi plane up to one bf16 ulp on a few per cent of the elements (the value is rounded float32 -> fp16
    -> bf16 instead of float32 -> bf16), and the bf16 backward over them gives the split_fwd combination's gradient up to that."""
    from outdoor_nerf_depth_amd import ops, _lib as L
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    b = SyntheticKitti().random_batch(512, np.random.RandomState(3))
    ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
    far, fg, bg = ops.sample_coarse(ray_o, ray_d, T(b['min_depth']), 64)
    e2 = ops.LevelEngine(T(flat(levels[0])), precision=L.PREC_SPLIT_FWD)
    e3 = ops.LevelEngine(T(flat(levels[0])), precision=L.PREC_FP16_FWD)
    r2 = e2.forward(ray_o, ray_d, far, fg, bg, training=True)
    r3 = e3.forward(ray_o, ray_d, far, fg, bg, training=True)
    for net in (0, 1):
        for t in (0, 2, 3, 4, 5, 6, 7, 8, 10, 11):          # X, H1..H7, G, DIRX (the fp16x2w forward does not materialise H0)
            a2, a3 = e2.saved_tensor(net, t), e3.saved_tensor(net, t)
            if t in (0, 11):          # the encodings: the same float32 value rounded two ways
                # one bf16 ulp is <= 2^-7 of the value; + the last-bit freedom of the point itself (the two kernel instantiations
                # contract its arithmetic differently), which the highest encoding frequency multiplies by 2^9
                ulp = a2.abs() * 2.0 ** -7 + 1e-4
                assert bool(((a2 - a3).abs() <= ulp).all()), (net, t)
            # activations: the two forwards' own 1e-4-level differences on top (a unit next to zero may switch)
            assert float((a2 - a3).norm() / a2.norm()) < 2.0 ** -9, (net, t)
            assert float((a2 != a3).float().mean()) < 0.15, (net, t)
    g_rgb, g_depth = torch.rand_like(r3['rgb']) * 1e-3, torch.rand_like(r3['depth']) * 1e-3
    g2 = e2.backward(g_rgb, g_depth, None).clone()
    g3 = e3.backward(g_rgb, g_depth, None).clone()
    assert float((g3 - g2).norm() / g2.norm()) < 2e-3


def test_fp16x2w_is_a_forward_precision_only(levels):
    """the backward kernels exist in precisions 1 and 2; a split-bf16 backward cannot run over the fp16x2w forward's single plane"""
    import ctypes as C
    from outdoor_nerf_depth_amd import _lib as L
    a = L.BackwardArgs()
    a.n_rays, a.n_samples, a.precision, a.workspace_precision = 8, 64, 3, 3
    assert L.lib().nerfpp_level_backward(None, C.byref(a)) == 1            # NERFPP_ERR_ARG
    a.precision, a.workspace_precision = 2, 3
    assert L.lib().nerfpp_level_backward(None, C.byref(a)) == 1
    assert L.lib().nerfpp_packed_bytes(3) == L.lib().nerfpp_packed_bytes(2)
    assert L.lib().nerfpp_workspace_bytes(64, 64, 3, 1) == L.lib().nerfpp_workspace_bytes(64, 64, 1, 1)


# ------------------------------------------------------------------------------------------- trajectory
@pytest.mark.parametrize('mode', ['rgbonly', 'mse', 'l1', 'kl'])
def test_fp16_fwd_trainer_follows_the_reference_trajectory(mode):
    """PREC_FP16_FWD through tests/golden/trajectory.npz (the imported reference, 200 steps): the logged rgb losses of both
    levels within split_fwd's EARLY_GATE of the float32 reference's (1 % at step 25, 10 % through step 100), and for rgb-only
    and gt + mse the render / tail PSNR within the gate of tests/test_gpu_round4.py."""
    import trajectory_common as TC
    import test_gpu_round4 as R4
    from outdoor_nerf_depth_amd import _lib as L
    g = np.load(os.path.join(GOLD, 'trajectory.npz'))
    rgb_mse, im, mse, ps = R4._trajectory(L.PREC_FP16_FWD, mode)
    logged = rgb_mse[TC.LOG_EVERY - 1::TC.LOG_EVERY]
    for lvl, key in ((1, '.f32.rgb1'), (0, '.f32.rgb0')):
        dev_log = np.abs(logged[:, lvl] / g[mode + key] - 1.0)
        assert dev_log[0] <= R4.EARLY_GATE['split_fwd'][0] and dev_log[:4].max() <= R4.EARLY_GATE['split_fwd'][1], (lvl, dev_log)
    ref_psnr = float(g[mode + '.f32.render_psnr'])
    ref_tail = float(np.mean(TC.psnr(g[mode + '.f32.tail_rgb_mse'][:, 1])))
    tail = float(np.mean(TC.psnr(rgb_mse[-TC.LOG_EVERY:, 1])))
    report = {'mode': mode, 'render_gap_db': ps - ref_psnr, 'tail_gap_db': tail - ref_tail}
    R4._dump('trajectory_fp16_fwd_%s.json' % mode, report)
    if mode in ('rgbonly', 'mse'):
        tol_r, tol_t = R4.psnr_tolerances(g, mode, 'split_fwd')
        assert abs(report['render_gap_db']) <= tol_r and abs(report['tail_gap_db']) <= tol_t, (report, tol_r, tol_t)


# ------------------------------------------------------------------------------------------- the reference's own seeds
@pytest.mark.parametrize('mode', ['mse', 'kl'])
def test_split_bf16_matches_the_reference_seeds(mode):
    """VERDICT r04 item 2a.  tests/golden/trajectory_seeds.npz = the imported float32 reference trained 1000 steps on the config-1
    scene for several seeds of the batch / uniform streams (gt + mse and mono_crop + kl): the reference's OWN seed spread at
    convergence, and a paired reference run for every seed.  The HIP trainer in split-bf16 replays each seed; asserted:
    (i) every run's early part is ON the reference's trajectory (logged rgb loss at steps 25 ... 100 within EARLY_GATE);
    (ii) the paired gaps of the in-loop tail PSNR and of the final render PSNR are inside the reference's own spread:
         |median gap| <= 0.05 dB + 2 SE with SE >= sigma_ref / sqrt(n) (the seed spread of the reference is what n runs of ANY
         faithful implementation scatter by), and no single gap beyond 3 sigma of a DIFFERENCE of two such runs,
         3 sqrt(2) max(sigma_ref, 0.1 dB) (round 6: until then 3 sigma_ref, i.e. 2.1 sigma of what it bounds -- a bound that eight
         chaotic readouts cross by chance every few kernel revisions).
         Measured on the round-6 kernels: gt + mse medians +0.046 (render) / +0.002 dB (tail) against gates 0.164 / 0.178, largest
         single gaps 0.226 / 0.194 against 0.424 / 0.535; mono_crop + kl medians +0.136 / +0.037 against 0.573 / 0.393, largest gaps
         0.608 / 0.267 against 1.62 / 0.905: every margin >= 1.9x.
    This is what validates split-bf16 as the reference's stand-in in the multi-seed precision tests (tests/test_gpu_round4.py)."""
    import trajectory_common as TC
    import test_gpu_round4 as R4
    from outdoor_nerf_depth_amd import _lib as L
    g = np.load(os.path.join(GOLD, 'trajectory_seeds.npz'))
    n_steps = int(g['steps'])
    seeds = sorted(int(k.split('.')[1][1:]) for k in g.files if k.startswith(mode + '.s') and k.endswith('.render_psnr'))
    assert len(seeds) >= 2, seeds
    rows = []
    for seed in seeds:
        rgb_mse, _, _, ps = R4._trajectory(L.PREC_SPLIT_BF16, mode, n_steps=n_steps, seed=seed)
        tag = '%s.s%d' % (mode, seed)
        logged = rgb_mse[TC.LOG_EVERY - 1::TC.LOG_EVERY]
        dev1 = np.abs(logged[:4, 1] / g[tag + '.rgb1'][:4] - 1.0)
        assert dev1[0] <= R4.EARLY_GATE['split_bf16'][0] and dev1.max() <= R4.EARLY_GATE['split_bf16'][1], (tag, dev1)
        ref_tail = float(np.mean(TC.psnr(g[tag + '.tail_rgb_mse'][:, 1])))
        rows.append(dict(seed=seed, ref_render=float(g[tag + '.render_psnr']), ref_tail=ref_tail, render=ps,
                         tail=float(np.mean(TC.psnr(rgb_mse[-TC.LOG_EVERY:, 1])))))
    ref_r, ref_t = np.array([r['ref_render'] for r in rows]), np.array([r['ref_tail'] for r in rows])
    gap_r, gap_t = np.array([r['render'] for r in rows]) - ref_r, np.array([r['tail'] for r in rows]) - ref_t
    n = len(rows)
    sig_r, sig_t = float(ref_r.std(ddof=1)), float(ref_t.std(ddof=1))
    report = {'mode': mode, 'steps': n_steps, 'runs': rows, 'reference_render_psnr_std_over_seeds': sig_r,
              'reference_tail_psnr_std_over_seeds': sig_t, 'render_gap_db': gap_r.tolist(), 'tail_gap_db': gap_t.tolist(),
              'render_gap_median': float(np.median(gap_r)), 'tail_gap_median': float(np.median(gap_t))}
    R4._dump('reference_seeds_%s.json' % mode, report)
    for gap, sig in ((gap_r, sig_r), (gap_t, sig_t)):
        se = max(R4.median_se(gap), sig / np.sqrt(n))
        assert abs(np.median(gap)) <= 0.05 + 2.0 * se, report
        assert np.abs(gap).max() <= 3.0 * np.sqrt(2.0) * max(sig, 0.1), report
