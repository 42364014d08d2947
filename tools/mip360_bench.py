#!/usr/bin/env python
"""MipNeRF-360 (BASELINE config 5) training-step throughput on one MI355X: configs/360.gin shape
(2 x 64 proposal samples through the 4 x 256 PropMLP, 32 samples through the 8 x 1024 NerfMLP), synthetic rays,
depth_loss_type = mse on distance_mean, charb data loss, interlevel + distortion losses, clip + Adam.

    python tools/mip360_bench.py [--rays 4096] [--steps 10] [--warmup 3] [--forward_only]

Dense-layer FLOP per ray (forward): 2 * (64 * 2 * prop_macs + 32 * nerf_macs); training = 3x (fwd + dX + dW)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outdoor_nerf_depth_amd import mip360 as M                                  # noqa: E402


def he_uniform(shapes, rs):
    return [(rs.uniform(-np.sqrt(6.0 / i), np.sqrt(6.0 / i), (i, o)).astype(np.float32), np.zeros(o, np.float32)) for i, o in shapes]


def shapes(cfg):
    W, D = cfg['net_width'], cfg['net_depth']
    out, dim = [], 504
    for i in range(D):
        out.append((dim, W))
        dim = W + (504 if (i % 4 == 0 and i > 0) else 0)
    out.append((dim, 1))
    if not cfg['disable_rgb']:
        out += [(dim, 256), (256 + 27, 128), (128, 3)]
    return out


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--rays', type=int, default=4096)
    p.add_argument('--steps', type=int, default=10)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--forward_only', action='store_true')
    a = p.parse_args()
    dev = torch.device('cuda:0')
    rs = np.random.RandomState(0)
    prop, nerf = he_uniform(shapes(M.PROP_CFG), rs), he_uniform(shapes(M.NERF_CFG), rs)
    n = a.rays
    d = rs.randn(n, 3).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    T = lambda x: torch.from_numpy(x).to(dev)
    rays = dict(origins=T((rs.randn(n, 3) * 0.3).astype(np.float32)), directions=T(d), viewdirs=T(d.copy()),
                radii=T(np.full((n, 1), 2e-3, np.float32)), near=T(np.full((n, 1), 0.2, np.float32)),
                far=T(np.full((n, 1), 1e6, np.float32)))
    gt = T(rs.rand(n, 3).astype(np.float32))
    sup = T(np.where(rs.rand(n) < .5, rs.uniform(1, 6, n), 0).astype(np.float32))
    tr = M.Mip360Trainer(prop, nerf, dev)
    macs = lambda sh: sum(i * o for i, o in sh)
    fwd_flop = 2.0 * (2 * 64 * macs(shapes(M.PROP_CFG)) + 32 * macs(shapes(M.NERF_CFG)))
    step = (lambda: tr.forward(rays, 0.5, None)) if a.forward_only else (lambda: tr.train_step(rays, gt, sup))
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    flop = fwd_flop * (1 if a.forward_only else 3) * n
    print(json.dumps({'workload': 'MipNeRF-360 360.gin, %d rays/step, %s' % (n, 'forward' if a.forward_only else 'train step'),
                      'ms_per_step': 1e3 * dt, 'rays_per_s': n / dt, 'dense_tflops': flop / dt / 1e12,
                      'frac_of_bf16_mfma_peak': flop / dt / 2.5e15, 'fwd_gflop_per_ray': fwd_flop / 1e9}))


if __name__ == '__main__':
    main()
