#!/usr/bin/env python
"""Time the IMPORTED reference (PyTorch CPU) on the training step of the hot path, in the build
container (needs /root/reference; never runs on the GPU box).  Indicative only: the
number bench.py reports beside the GPU figure is the numpy port timed on the GPU box's own host cores.

    python tests/golden/time_reference_cpu.py [--rays 256] [--steps 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G          # noqa: E402  (imports the reference with the stub modules)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--rays', type=int, default=256)
    p.add_argument('--steps', type=int, default=3)
    a = p.parse_args()
    torch.set_num_threads(os.cpu_count())
    nets = G.make_levels(2)
    optims = [torch.optim.Adam(n.parameters(), lr=5e-4) for n in nets]
    scene = G.SyntheticKitti(depth_sup_type='gt')
    times = []
    for step in range(a.steps):
        b = scene.random_batch(a.rays, np.random.RandomState(step))
        bt = {k: G.T(v) for k, v in b.items() if isinstance(v, np.ndarray)}
        t0 = time.perf_counter()
        # the per-level loop of ddp_train_nerf.py:432-498 with the reference's own functions
        far = G.R.intersect_sphere(bt['ray_o'], bt['ray_d'])
        S0 = 64
        step_sz = (far - bt['min_depth']) / (S0 - 1)
        fg = torch.stack([bt['min_depth'] + i * step_sz for i in range(S0)], dim=-1)
        fg = G.R.perturb_samples(fg)
        bg = G.R.perturb_samples(torch.linspace(0., 1., S0).view([1, ] * (far.dim()) + [S0, ]).expand(list(far.shape) + [S0, ]))
        ret = None
        for m in range(2):
            if m > 0:
                w = ret['fg_weights'].clone().detach()
                mids = .5 * (fg[..., 1:] + fg[..., :-1])
                fg, _ = torch.sort(torch.cat((fg, G.R.sample_pdf(bins=mids, weights=w[..., 1:-1], N_samples=128, det=False)), dim=-1))
                w = ret['bg_weights'].clone().detach()
                mids = .5 * (bg[..., 1:] + bg[..., :-1])
                bg, _ = torch.sort(torch.cat((bg, G.R.sample_pdf(bins=mids, weights=w[..., 1:-1], N_samples=128, det=False)), dim=-1))
            optims[m].zero_grad()
            ret, loss, rgb_loss, depth_loss = G.ref_level_step(nets[m], bt, far, fg, bg, 'mse', 0.1, 0.)
            optims[m].step()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:])) if len(times) > 1 else times[0]
    print(json.dumps({'what': 'imported reference (PyTorch %s, CPU), both levels fwd+bwd+Adam, 64+128 samples/ray' % torch.__version__,
                      'rays_per_step': a.rays, 'threads': os.cpu_count(), 's_per_step': t, 'rays_per_s': a.rays / t,
                      'where': 'build container (indicative only)'}))


if __name__ == '__main__':
    main()
