#!/usr/bin/env python
"""Is the PSNR gap of the bf16-gradient modes on the config-1 trajectory (tests/test_gpu_round4.py) systematic or the
chaotic spread of a 200-step run?  The same scene and loop as the fixture, several batch / uniform seeds, every precision
on identical inputs; split-bf16 (which follows the float32 reference to 0.003-0.1 dB on the fixture's seed) is the stand-in
for the reference.  Prints per mode and step count the paired gaps (precision - split-bf16) of the final render PSNR.

    python tools/probes/traj_seeds.py --modes l1 kl --seeds 6 --steps 200 1000 --out gpurun_out/x/traj_seeds.json

Round 5: 32 seeds, gt + mse added, the fp16_fwd combination added; per precision the MEDIAN paired gap
with its bootstrap standard error and the verdict `|median| <= 0.05 dB + 2 SE`; with --ref (a
tests/golden/trajectory_seeds.npz written by make_golden.py trajectory_seeds: the imported float32 reference itself on the
same seeds) also the paired gaps HIP - reference for the seeds the fixture holds, split-bf16 included.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import trajectory_common as TC                                            # noqa: E402
from outdoor_nerf_depth_amd import _lib as L                              # noqa: E402
from outdoor_nerf_depth_amd.trainer import NerfppTrainer                  # noqa: E402
from outdoor_nerf_depth_amd.ddp_train_nerf import render_single_image     # noqa: E402


def run(prec, mode, seed, n_steps, dev, lambda_depth):
    smp = TC.sampler(mode)
    tr = NerfppTrainer(dev, precision=prec, cascade_samples=TC.CASCADE, use_depth=(mode != 'rgbonly'),
                       depth_loss_type=(mode if mode != 'rgbonly' else 'mse'), lambda_depth=lambda_depth,
                       depth_sigma=TC.DEPTH_SIGMA, depth_scale=float(smp.get_depth_scale() or 1.0))
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    tail = []
    for step in range(1, n_steps + 1):
        s = step + 100000 * seed
        b, uni = TC.step_batch(smp, s), TC.step_uniforms(s)
        sc = tr.train_step({k: T(v) for k, v in b.items()}, uniforms={k: T(v) for k, v in uni.items()})
        if step > n_steps - 25:
            tail.append(sc[1][1])
    tail_psnr = float(np.mean(TC.psnr(torch.stack(tail).cpu().numpy().astype(np.float64))))
    tr.check_cameras()
    ret = render_single_image(0, 1, tr, smp, 1024, keep_dists=False)
    im = ret[-1]['rgb'].numpy().astype(np.float64)
    mse = float(np.mean((im - smp.get_img().astype(np.float64)) ** 2))
    return float(TC.psnr(mse)), tail_psnr


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--modes', nargs='+', default=['l1', 'kl'])
    p.add_argument('--seeds', type=int, default=6)
    p.add_argument('--steps', nargs='+', type=int, default=[200, 1000])
    p.add_argument('--lambda_depth', type=float, default=TC.LAMBDA_DEPTH)
    p.add_argument('--out', default='traj_seeds.json')
    p.add_argument('--ref', default=None, help='trajectory_seeds.npz of the imported reference (1000 steps)')
    a = p.parse_args()
    ref = np.load(a.ref) if a.ref and os.path.exists(a.ref) else None
    dev = torch.device('cuda:0')
    precs = (('split_bf16', L.PREC_SPLIT_BF16), ('split_fwd', L.PREC_SPLIT_FWD), ('fp16_fwd', L.PREC_FP16_FWD), ('bf16', L.PREC_BF16))

    def summary(x):
        x = np.asarray(x, np.float64)
        n = len(x)
        sd = float(x.std(ddof=1)) if n > 1 else 0.0
        # standard error of the MEDIAN by bootstrap (the normal-theory 1.2533 sd / sqrt(n) is blown up by the rare pair in which
        # one of the two runs falls into the ~21 dB basin: +-15 dB)
        rs = np.random.RandomState(0)
        se_med = float(np.std([np.median(x[rs.randint(0, n, n)]) for _ in range(2000)])) if n > 1 else float('inf')
        return {'n': n, 'median': float(np.median(x)), 'mean': float(x.mean()), 'std': sd, 'se_of_median': float(se_med),
                'within_0p05_plus_2se': bool(abs(np.median(x)) <= 0.05 + 2 * se_med), 'values': [round(float(v), 4) for v in x]}
    rep = {}
    for mode in a.modes:
        for n in a.steps:
            rows = []
            for seed in range(a.seeds):
                r = {name: run(prec, mode, seed, n, dev, a.lambda_depth) for name, prec in precs}
                rows.append(r)
                print(mode, n, seed, {k: (round(v[0], 3), round(v[1], 3)) for k, v in r.items()}, flush=True)
            out = {'runs': rows}
            for name in ('split_fwd', 'fp16_fwd', 'bf16'):
                g = np.array([r[name][0] - r['split_bf16'][0] for r in rows])
                gt = np.array([r[name][1] - r['split_bf16'][1] for r in rows])
                out[name] = {'render_gap_db_mean': float(g.mean()), 'render_gap_db_std': float(g.std(ddof=1)) if len(g) > 1 else 0.0,
                             'render_gap_db': g.tolist(), 'tail_gap_db_mean': float(gt.mean()), 'tail_gap_db': gt.tolist(),
                             'render_vs_split_bf16': summary(g), 'tail_vs_split_bf16': summary(gt)}
                print('== %s, %d steps, %s - split_bf16: render median %+.3f (SE %.3f) dB, in-loop tail median %+.3f (SE %.3f) dB' %
                      (mode, n, name, out[name]['render_vs_split_bf16']['median'], out[name]['render_vs_split_bf16']['se_of_median'],
                       out[name]['tail_vs_split_bf16']['median'], out[name]['tail_vs_split_bf16']['se_of_median']), flush=True)
            if ref is not None and n == int(ref['steps']):
                # the imported float32 reference on the same seeds: its own spread, and every HIP precision paired against it
                have = [sd for sd in range(a.seeds) if '%s.s%d.render_psnr' % (mode, sd) in ref.files]
                if have:
                    rr = np.array([float(ref['%s.s%d.render_psnr' % (mode, sd)]) for sd in have])
                    rt = np.array([float(np.mean(TC.psnr(ref['%s.s%d.tail_rgb_mse' % (mode, sd)][:, 1]))) for sd in have])
                    out['reference'] = {'seeds': have, 'render_psnr': rr.tolist(), 'tail_psnr': rt.tolist(),
                                        'render_psnr_std_over_seeds': float(rr.std(ddof=1)) if len(rr) > 1 else 0.0,
                                        'tail_psnr_std_over_seeds': float(rt.std(ddof=1)) if len(rt) > 1 else 0.0}
                    for name, _ in precs:
                        out[name if name != 'split_bf16' else 'split_bf16_vs_reference'] = dict(
                            out.get(name, {}) if name != 'split_bf16' else {},
                            render_vs_reference=summary([rows[sd][name][0] - rr[i] for i, sd in enumerate(have)]),
                            tail_vs_reference=summary([rows[sd][name][1] - rt[i] for i, sd in enumerate(have)]))
                        print('== %s, %s - REFERENCE (%d seeds): render %s, tail %s' % (
                            mode, name, len(have), [round(rows[sd][name][0] - rr[i], 3) for i, sd in enumerate(have)],
                            [round(rows[sd][name][1] - rt[i], 3) for i, sd in enumerate(have)]), flush=True)
            rep['%s.%d' % (mode, n)] = out
            with open(a.out, 'w') as f:
                json.dump(rep, f, indent=1)


if __name__ == '__main__':
    main()
