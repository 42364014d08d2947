#!/usr/bin/env python
"""Is the PSNR gap of the bf16-gradient modes on the config-1 trajectory (tests/test_gpu_round4.py) systematic or the
chaotic spread of a 200-step run?  The same scene and loop as the fixture, several batch / uniform seeds, every precision
on identical inputs; split-bf16 (which follows the float32 reference to 0.003-0.1 dB on the fixture's seed) is the stand-in
for the reference.  Prints per mode and step count the paired gaps (precision - split-bf16) of the final render PSNR.

    python tools/probes/traj_seeds.py --modes l1 kl --seeds 6 --steps 200 1000 --out gpurun_out/x/traj_seeds.json
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
    a = p.parse_args()
    dev = torch.device('cuda:0')
    precs = (('split_bf16', L.PREC_SPLIT_BF16), ('split_fwd', L.PREC_SPLIT_FWD), ('bf16', L.PREC_BF16))
    rep = {}
    for mode in a.modes:
        for n in a.steps:
            rows = []
            for seed in range(a.seeds):
                r = {name: run(prec, mode, seed, n, dev, a.lambda_depth) for name, prec in precs}
                rows.append(r)
                print(mode, n, seed, {k: (round(v[0], 3), round(v[1], 3)) for k, v in r.items()}, flush=True)
            out = {'runs': rows}
            for name in ('split_fwd', 'bf16'):
                g = np.array([r[name][0] - r['split_bf16'][0] for r in rows])
                gt = np.array([r[name][1] - r['split_bf16'][1] for r in rows])
                out[name] = {'render_gap_db_mean': float(g.mean()), 'render_gap_db_std': float(g.std(ddof=1)) if len(g) > 1 else 0.0,
                             'render_gap_db': g.tolist(), 'tail_gap_db_mean': float(gt.mean()), 'tail_gap_db': gt.tolist()}
                print('== %s, %d steps, %s - split_bf16: render %+.3f +- %.3f dB, in-loop tail %+.3f dB' %
                      (mode, n, name, g.mean(), out[name]['render_gap_db_std'], gt.mean()), flush=True)
            rep['%s.%d' % (mode, n)] = out
            with open(a.out, 'w') as f:
                json.dump(rep, f, indent=1)


if __name__ == '__main__':
    main()
