"""GPU, round 4: schedule options of NerfppTrainer change no bit; the reference-generated training trajectory.

* `concurrent_backward` (level 0's backward on its own stream under level 1's forward, trainer.py) against the inline
  schedule: identical parameters, Adam moments and logged scalars after several steps -- in one process and with two
  ranks over gloo sharing cuda:0.
* tests/golden/trajectory.npz (make_golden.py: gen_trajectory, the imported reference on the BASELINE config-1 scene):
  the HIP trainer on the same batches and uniforms, PSNR gates of VERDICT r03 item 3.
"""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden')
if HERE not in sys.path:
    sys.path.insert(0, HERE)            # trajectory_common.py


def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def T(x, d=None):
    return torch.from_numpy(np.ascontiguousarray(x)).to(d or dev())


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batches(rank, n, steps):
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    scene = SyntheticKitti(depth_sup_type='gt')
    rng = np.random.RandomState((rank + 1) * 777)
    return [scene.random_batch(n, rng) for _ in range(steps)]


def _run(concurrent, precision, world=1, rank=0, steps=5, n=96):
    from outdoor_nerf_depth_amd.trainer import NerfppTrainer
    d = torch.device('cuda:0')
    tr = NerfppTrainer(d, precision=precision, use_depth=True, depth_loss_type='mse', lambda_depth=0.1,
                       world_size=world, seed=(rank + 1) * 777)
    tr.concurrent_backward = concurrent
    sc = []
    for b in _batches(rank, n, steps):
        out = tr.train_step({k: T(v, d) for k, v in b.items() if isinstance(v, np.ndarray)})
        sc.append([s.clone() for s in out])
    tr.flush()
    torch.cuda.synchronize()
    return dict(params=np.stack([e.params.cpu().numpy() for e in tr.engines]),
                m=np.stack([x.cpu().numpy() for x in tr.exp_avg]), v=np.stack([x.cpu().numpy() for x in tr.exp_avg_sq]),
                scalars=np.stack([np.stack([s.cpu().numpy() for s in step]) for step in sc]))


@pytest.mark.parametrize('precision', [1, 2])
def test_concurrent_backward_changes_no_bit(precision):
    dev()
    a = _run(True, precision)
    b = _run(False, precision)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert np.isfinite(a['params']).all() and np.isfinite(a['scalars'][..., :2]).all()


def _worker(rank, world, port, out_dir, concurrent):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    r = _run(concurrent, 1, world=world, rank=rank, steps=3, n=64)
    np.savez(os.path.join(out_dir, 'r%d_c%d.npz' % (rank, int(concurrent))), **r)
    dist.barrier()
    dist.destroy_process_group()


def test_concurrent_backward_two_ranks_gloo(tmp_path):
    dev()
    import torch.multiprocessing as mp
    for concurrent in (True, False):
        mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), concurrent), nprocs=2, join=True)
    for rank in (0, 1):
        a = np.load(tmp_path / ('r%d_c1.npz' % rank))
        b = np.load(tmp_path / ('r%d_c0.npz' % rank))
        for k in ('params', 'm', 'v', 'scalars'):
            np.testing.assert_array_equal(a[k], b[k], err_msg='rank %d %s' % (rank, k))
    np.testing.assert_array_equal(np.load(tmp_path / 'r0_c1.npz')['params'], np.load(tmp_path / 'r1_c1.npz')['params'])


# ------------------------------------------------------------------------------------------- training trajectory
def _trajectory(precision, mode, n_steps=None, seed=0, with_depth=False):
    """NerfppTrainer on the fixture's batches and uniforms (seed 0; other seeds shift the numpy streams); returns the logged rgb
    losses per step [n,2] and the final frame rendered by render_single_image (deterministic sampling) with its PSNR against
    the image."""
    import trajectory_common as TC
    from outdoor_nerf_depth_amd.trainer import NerfppTrainer
    from outdoor_nerf_depth_amd.ddp_train_nerf import render_single_image
    d = dev()
    smp = TC.sampler(mode)
    tr = NerfppTrainer(d, precision=precision, cascade_samples=TC.CASCADE, use_depth=(mode != 'rgbonly'),
                       depth_loss_type=(mode if mode != 'rgbonly' else 'mse'), lambda_depth=TC.LAMBDA_DEPTH,
                       depth_sigma=TC.DEPTH_SIGMA, depth_scale=float(smp.get_depth_scale() or 1.0))
    sc_all = []
    for step in range(1, (n_steps or TC.N_STEPS) + 1):
        b, uni = TC.step_batch(smp, step + 100000 * seed), TC.step_uniforms(step + 100000 * seed)
        sc = tr.train_step({k: T(v, d) for k, v in b.items()}, uniforms={k: T(v, d) for k, v in uni.items()})
        sc_all.append(torch.stack([s[1] for s in sc] + [s[2] for s in sc]))
    logged = torch.stack(sc_all).cpu().numpy().astype(np.float64)
    rgb_mse, depth_loss = logged[:, :2], logged[:, 2:]
    tr.check_cameras()
    ret = render_single_image(0, 1, tr, smp, 1024, keep_dists=False)
    im = ret[-1]['rgb'].numpy().astype(np.float64)
    mse = float(np.mean((im - smp.get_img().astype(np.float64)) ** 2))
    if with_depth:
        return rgb_mse, im, mse, float(TC.psnr(mse)), depth_loss
    return rgb_mse, im, mse, float(TC.psnr(mse))


def _dump(name, report):
    import json
    out = os.path.join(os.path.dirname(HERE), 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, name), 'w') as f:
            json.dump(report, f, indent=1)
    print(json.dumps(report))


# relative deviation of the logged rgb losses (both levels) from the float32 reference's: at the first log line (step 25), over
# the first four (steps 25-100).  Measured (profiles/r04_trajectory_*.json): split-bf16 <= 0.0017 / 0.0135, split_fwd <= 0.0024 /
# 0.037, bf16 <= 0.0098 / 0.0675 -- the trajectories coincide with the reference's and then separate exponentially.
EARLY_GATE = {'split_bf16': (5e-3, 4e-2), 'split_fwd': (1e-2, 1e-1), 'bf16': (3e-2, 2e-1)}
# The logged DEPTH loss of both levels at the first two log lines (steps 25 and 50), relative deviation from the float32
# reference's (only the rgb loss was gated).  Measured (profiles/r05_trajectory_depth_dev.json): gt + mse
# -- a mean over the ~13 rays of a batch that carry a prior, the noisiest of the three -- split-bf16 <= 0.0096, split_fwd <= 0.018,
# bf16 <= 0.048; stereo_crop + l1 and mono_crop + kl <= 0.0025 in every mode.  Gate = 2 x the measured maximum.
DEPTH_EARLY_GATE = {'split_bf16': 2e-2, 'split_fwd': 4e-2, 'fp16_fwd': 4e-2, 'bf16': 1e-1}
# PSNR tolerance at the end of the 200-step run, in units of the reference's OWN float64 - float32 spread on the same run (its
# noise floor: a 1e-7 perturbation of the float32 reference moves its final PSNR by that much).  split-bf16 reproduces the
# reference's arithmetic to 1e-5 and is held to 2 x the spread (this gate stays where it was when it was
# introduced); the modes whose GRADIENTS are single-pass bf16 (split_fwd, fp16_fwd, bf16) perturb every step at the 1e-2 level
# of the gradient and get 3 x -- their measured gaps on gt + mse are +0.17 / +0.25 dB (split_fwd) and +0.19 / +0.28 dB (bf16)
# against a spread of 0.10 dB, and re-draw with every change of a kernel's summation order.  Floor: north_star's 0.05 dB.
PSNR_SPREADS = {'split_bf16': 2.0, 'split_fwd': 3.0, 'fp16_fwd': 3.0, 'bf16': 3.0}


def psnr_tolerances(g, mode, name):
    import trajectory_common as TC
    ref_tail = float(np.mean(TC.psnr(g[mode + '.f32.tail_rgb_mse'][:, 1])))
    f64_tail = float(np.mean(TC.psnr(g[mode + '.f64.tail_rgb_mse'][:, 1])))
    f64_gap = abs(float(g[mode + '.f64.render_psnr']) - float(g[mode + '.f32.render_psnr']))
    k = PSNR_SPREADS[name]
    return max(0.05, k * f64_gap), max(0.05, k * abs(f64_tail - ref_tail))


@pytest.mark.parametrize('mode', ['rgbonly', 'mse', 'l1', 'kl'])
def test_training_trajectory_psnr_against_reference(mode):
    """VERDICT r03 item 3.  The imported reference trained 200 steps on the config-1 scene (tests/golden/trajectory.npz): rgb-only
    and with each depth term of the BASELINE configs (gt + mse, stereo_crop + l1, mono_crop + kl); the HIP trainer replays the
    same batches and uniforms in every precision mode.
    Gates, all modes: the logged rgb loss follows the reference's over the first 100 steps (EARLY_GATE), the logged depth loss
    over the first 50 (DEPTH_EARLY_GATE) -- the deterministic part: every mode starts ON the reference's trajectory.
    rgb-only: every precision ends within max(0.05 dB, k x the reference's own float64-float32 spread) of the float32
    reference in render PSNR and in the mean in-loop PSNR of the last 25 steps (north_star's PSNR clause); k = 2 for split-bf16,
    3 for the bf16-gradient modes (PSNR_SPREADS above).  Measured: <= 0.005 dB.
    With a depth term the PSNR at a FIXED STEP of ONE run is a chaotic readout: the reference's own float64 run ends
    0.09-0.14 dB from its float32 run, and every change of a summation order inside a kernel re-draws the HIP gaps -- split-bf16
    on gt + mse read +0.11 / +0.17 dB (round 4), +0.125 / +0.205 (round 5), +0.055 / +0.102 (round 6) against a 2-spread gate of
    0.209 / 0.253: a gate that sits at 40-80 % of itself depending on which bits moved is a coin, not a test.
    stereo_crop + l1 and mono_crop + kl are worse: at step 200 they are in the steep part of training (25 -> 35 dB between
    steps 200 and 1000) and any perturbation moves the fixed-step PSNR by tenths of a dB to 2 dB in either direction.
    So with a depth term the PSNR clause is NOT gated on this single run; it is gated where it can be resolved:
      * split-bf16 against the imported reference's own runs, paired over the 4 seeds of tests/golden/trajectory_seeds.npz at
        1000 steps (tests/test_gpu_round5.py::test_split_bf16_matches_the_reference_seeds: |median gap| <= 0.05 dB + 2 SE, no gap
        beyond 3 sigma of a difference of two reference runs).  Measured on the round-6 kernels, gt + mse: render median +0.046 dB,
        tail median +0.002 dB against gates of 0.164 / 0.178 dB -- margins 3.6x / 100x; largest single gap 0.226 / 0.194 dB against
        0.424 / 0.535 (1.9x / 2.8x);
      * the bf16-gradient modes against split-bf16, paired over 8 seeds at 1000 steps
        (test_bf16_gradient_modes_match_split_bf16_over_seeds below; 32 seeds and BASELINE's shape: tools/probes/traj_seeds*.py).
    What stays here for gt + mse is a SANITY bound of 5 x the reference's float64-float32 spread (0.52 / 0.63 dB; every
    measured gap of every mode and round is <= 0.28 dB: >= 2x margin) -- it catches a broken loss term, it is not the clause."""
    import trajectory_common as TC
    g = np.load(os.path.join(GOLD, 'trajectory.npz'))
    ref_psnr = float(g[mode + '.f32.render_psnr'])
    ref_tail = float(np.mean(TC.psnr(g[mode + '.f32.tail_rgb_mse'][:, 1])))
    f64_gap = abs(float(g[mode + '.f64.render_psnr']) - ref_psnr)             # noise floor of the float32 reference itself
    report = {'mode': mode, 'reference_f32_render_psnr': ref_psnr, 'reference_f64_minus_f32_db': float(g[mode + '.f64.render_psnr']) - ref_psnr,
              'reference_tail_inloop_psnr_L1': ref_tail}
    from outdoor_nerf_depth_amd import _lib as L
    for name, prec in (('split_bf16', L.PREC_SPLIT_BF16), ('split_fwd', L.PREC_SPLIT_FWD), ('bf16', L.PREC_BF16)):
        rgb_mse, im, mse, ps, dl = _trajectory(prec, mode, with_depth=True)
        tail = float(np.mean(TC.psnr(rgb_mse[-TC.LOG_EVERY:, 1])))
        logged = rgb_mse[TC.LOG_EVERY - 1::TC.LOG_EVERY]
        dlog = dl[TC.LOG_EVERY - 1::TC.LOG_EVERY]
        report[name] = {'render_psnr': ps, 'render_gap_db': ps - ref_psnr, 'tail_inloop_psnr_L1': tail,
                        'tail_gap_db': tail - ref_tail,
                        'logged_rgb_mse_rel_dev_max': float(np.max(np.abs(logged[:, 1] / g[mode + '.f32.rgb1'] - 1.0))),
                        'logged_rgb_mse_rel_dev': [float(x) for x in np.abs(logged[:, 1] / g[mode + '.f32.rgb1'] - 1.0)],
                        'logged_rgb0_mse_rel_dev': [float(x) for x in np.abs(logged[:, 0] / g[mode + '.f32.rgb0'] - 1.0)],
                        'logged_depth1_rel_dev': ([float(x) for x in np.abs(dlog[:, 1] / g[mode + '.f32.depth1'] - 1.0)] if mode != 'rgbonly' else None),
                        'logged_depth0_rel_dev': ([float(x) for x in np.abs(dlog[:, 0] / g[mode + '.f32.depth0'] - 1.0)] if mode != 'rgbonly' else None),
                        'image_rms_vs_reference': float(np.sqrt(np.mean((im.reshape(-1, 3) - g[mode + '.f32.render_rgb']) ** 2)))}
    # With a depth term the 200-step trajectory is chaotic at the 0.1 dB level even for the reference -- its own float64 run
    # ends 0.09-0.14 dB from its float32 run (a 1e-7 perturbation): nothing can be pinned to the float32 run tighter than the
    # reference pins itself, and every change of the summation order inside a kernel re-draws these gaps (gt + mse, bf16:
    # +0.09 / +0.13 dB before the remap layer was folded, +0.19 / +0.28 after).  Tolerances: psnr_tolerances above.
    f64_tail = float(np.mean(TC.psnr(g[mode + '.f64.tail_rgb_mse'][:, 1])))
    steep = mode in ('l1', 'kl')
    tols = {name: psnr_tolerances(g, mode, name) for name in ('split_bf16', 'split_fwd', 'bf16')}
    if mode == 'mse':            # a depth term: sanity bound only (docstring); the clause is gated by the paired multi-seed tests
        tols = {name: (max(0.05, 5.0 * f64_gap), max(0.05, 5.0 * abs(f64_tail - ref_tail))) for name in tols}
    report['gate'] = {'tolerances_db_render_tail': tols, 'spreads': PSNR_SPREADS, 'reference_f64_gap_db': f64_gap,
                      'reference_f64_tail_gap_db': f64_tail - ref_tail, 'early': EARLY_GATE,
                      'psnr_gated': [] if steep else ['split_bf16', 'split_fwd', 'bf16'],
                      'psnr_gate_kind': 'none (steep part of training)' if steep else 'sanity bound, 5 x the float64-float32 spread' if mode == 'mse'
                                        else 'clause: k x the float64-float32 spread'}
    _dump('trajectory_%s.json' % mode, report)
    for name in ('split_bf16', 'split_fwd', 'bf16'):
        for key in ('logged_rgb_mse_rel_dev', 'logged_rgb0_mse_rel_dev'):
            dev_log = report[name][key]
            assert dev_log[0] <= EARLY_GATE[name][0] and max(dev_log[:4]) <= EARLY_GATE[name][1], (name, key, dev_log)
        if mode != 'rgbonly':
            for key in ('logged_depth1_rel_dev', 'logged_depth0_rel_dev'):
                assert max(report[name][key][:2]) <= DEPTH_EARLY_GATE[name], (name, key, report[name][key][:2])
        if steep:
            continue
        assert abs(report[name]['render_gap_db']) <= tols[name][0], (name, report[name], tols[name])
        assert abs(report[name]['tail_gap_db']) <= tols[name][1], (name, report[name], tols[name])


def median_se(x, n_boot=2000):
    """bootstrap standard error of the median"""
    x = np.asarray(x, np.float64)
    rs = np.random.RandomState(0)
    return float(np.std([np.median(x[rs.randint(0, len(x), len(x))]) for _ in range(n_boot)]))


@pytest.mark.parametrize('mode', ['mse', 'l1', 'kl'])
def test_bf16_gradient_modes_match_split_bf16_over_seeds(mode):
    """The PSNR clause of north_star for the bf16-gradient modes (split_fwd, fp16_fwd, bf16 = the bench headline) at CONVERGENCE,
    where single trajectories cannot be pinned (see above): 8 seeds of batches / uniforms, 1000 steps each, every precision on
    identical inputs, split-bf16 standing in for the reference (it is pinned to the reference on seed 0, and
    test_split_bf16_matches_the_reference_seeds below pairs it with the imported reference's own 1000-step runs).
    Gate: |median paired gap of the in-loop PSNR (mean of the last 25 steps)| <= 0.05 dB + 2 SE, SE = the
    bootstrap standard error of that median over the seeds.  With 8 seeds SE is 0.03-0.1 dB; the 32-seed run of
    tools/probes/traj_seeds.py (profiles/r05_traj_seeds_32.json) resolves it to 0.014-0.05 dB: medians -0.07 ... +0.05 dB for
    every mode and depth term, every one inside 0.05 + 2 SE."""
    import trajectory_common as TC
    from outdoor_nerf_depth_amd import _lib as L
    n_steps, n_seeds = 1000, 8
    precs = (('split_bf16', L.PREC_SPLIT_BF16), ('split_fwd', L.PREC_SPLIT_FWD), ('fp16_fwd', L.PREC_FP16_FWD), ('bf16', L.PREC_BF16))
    rows = []
    for seed in range(n_seeds):
        r = {}
        for name, prec in precs:
            rgb_mse, _, _, ps = _trajectory(prec, mode, n_steps=n_steps, seed=seed)
            r[name] = (ps, float(np.mean(TC.psnr(rgb_mse[-TC.LOG_EVERY:, 1]))))
        rows.append(r)
    report = {'mode': mode, 'steps': n_steps, 'runs': rows}
    for name in ('split_fwd', 'fp16_fwd', 'bf16'):
        tail = np.array([r[name][1] - r['split_bf16'][1] for r in rows])
        rend = np.array([r[name][0] - r['split_bf16'][0] for r in rows])
        report[name] = {'tail_gap_db': tail.tolist(), 'tail_gap_db_median': float(np.median(tail)), 'tail_gap_db_se_of_median': median_se(tail),
                        'tail_gap_db_std': float(tail.std(ddof=1)), 'render_gap_db': rend.tolist(),
                        'render_gap_db_median': float(np.median(rend)), 'render_gap_db_se_of_median': median_se(rend),
                        'render_gap_db_std': float(rend.std(ddof=1))}
    _dump('trajectory_seeds_%s.json' % mode, report)
    for name in ('split_fwd', 'fp16_fwd', 'bf16'):
        assert abs(report[name]['tail_gap_db_median']) <= 0.05 + 2.0 * report[name]['tail_gap_db_se_of_median'], (name, report[name])


# ------------------------------------------------------------------------------------------- pixel draw of the ray-batch sampler
@pytest.mark.parametrize('n_pixels,k,seed,step', [(465750, 1024, 777, 1), (465750, 1024, 1554, 12345), (4096, 256, 777, 3),
                                                  (768, 128, 5, 2), (300, 150, 9, 7), (8192 * 64, 8192, 777, 1), (33, 1, 1, 1)])
def test_sample_pixels_matches_oracle(n_pixels, k, seed, step):
    """nerfpp_sample_pixels (np.random.choice(H*W, N_rand, replace=False), nerf_sample_ray_split.py:178): bit-exact against the
    oracle's sequential definition -- also where many draws collide (k = n / 2) -- distinct, in range."""
    from outdoor_nerf_depth_amd import ops
    from oracle import nerfpp_oracle as O
    pix = ops.sample_pixels(n_pixels, k, seed, step, dev()).cpu().numpy()
    assert pix.dtype == np.int64 and pix.shape == (k,)
    assert len(set(pix.tolist())) == k and pix.min() >= 0 and pix.max() < n_pixels
    if k <= 1024:
        np.testing.assert_array_equal(pix, O.sample_pixels(n_pixels, k, seed, step))
    else:
        np.testing.assert_array_equal(pix[:64], O.sample_pixels(n_pixels, 64, seed, step))


def test_sample_pixels_is_uniform_and_rejects_bad_sizes():
    from outdoor_nerf_depth_amd import ops, _lib as L
    d = dev()
    n, k = 64, 16
    counts = np.zeros(n, np.int64)
    first = np.zeros(n, np.int64)
    for step in range(1, 2001):
        p = ops.sample_pixels(n, k, 99, step, d).cpu().numpy()
        counts[p] += 1
        first[p[0]] += 1
    # every pixel is in the batch with probability k / n; chi-square over 64 cells, 63 dof: 99.9 % quantile = 103.4
    exp = 2000.0 * k / n
    chi2 = float(((counts - exp) ** 2 / (exp * (1 - k / n))).sum())
    assert chi2 < 110.0, chi2
    chi2_first = float(((first - 2000.0 / n) ** 2 / (2000.0 / n)).sum())
    assert chi2_first < 110.0, chi2_first
    with pytest.raises(L.NerfppError):
        ops.sample_pixels(10, 11, 1, 1, d)
    with pytest.raises(L.NerfppError):
        ops.sample_pixels(1 << 20, 8193, 1, 1, d)


# ------------------------------------------------------------------------------------------- RCCL behind the C ABI
def test_rccl_entry_points_single_rank():
    """nerfpp_rccl_* / nerfpp_allreduce_mean (ddp_train_nerf.py:323): a one-rank communicator built through the
    C ABI, the all-reduce on a side stream, mean semantics for both `prescaled` forms, and a training run whose gradient
    average goes through it -- bit-identical to the run without a communicator (the mean over one rank is the identity)."""
    from outdoor_nerf_depth_amd.dist_utils import RcclComm
    from outdoor_nerf_depth_amd.trainer import NerfppTrainer
    d = dev()
    comm = RcclComm(0, 1)
    g = torch.arange(1202444, device=d, dtype=torch.float32) * 1e-3
    ref = g.clone()
    side = torch.cuda.Stream(device=d)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        comm.allreduce_mean(g, prescaled=True)
        comm.allreduce_mean(g, prescaled=False)
    side.synchronize()
    assert torch.equal(g, ref)
    outs = []
    for c in (None, comm):
        tr = NerfppTrainer(d, precision=1, use_depth=True, depth_loss_type='mse', lambda_depth=0.1, comm=c)
        for b in _batches(0, 64, 3):
            tr.train_step({k: T(v, d) for k, v in b.items() if isinstance(v, np.ndarray)})
        tr.flush()
        torch.cuda.synchronize()
        outs.append(np.stack([e.params.cpu().numpy() for e in tr.engines]))
    np.testing.assert_array_equal(outs[0], outs[1])
    comm.destroy()


def _rccl_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from outdoor_nerf_depth_amd.dist_utils import RcclComm
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(rank)
    comm = RcclComm(rank, world)
    g = torch.full((1202444,), float(rank + 1), device='cuda')
    comm.allreduce_mean(g, prescaled=False)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, 'g%d.npy' % rank), g[:8].cpu().numpy())
    comm.destroy()
    dist.destroy_process_group()


def test_rccl_entry_points_two_ranks(tmp_path):
    """two ranks on two GPUs over xGMI: mean of (1, 2) = 1.5 on both (skipped on a 1-GPU box: RCCL refuses two ranks on one device)"""
    dev()
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    import torch.multiprocessing as mp
    mp.spawn(_rccl_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        np.testing.assert_array_equal(np.load(tmp_path / ('g%d.npy' % r)), np.full(8, 1.5, np.float32))
