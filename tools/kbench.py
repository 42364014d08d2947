#!/usr/bin/env python
"""Kernel micro-benchmarks on one GPU: level forward (inference / training mode), backward.

    python tools/kbench.py [--n_rays 1024] [--S 192] [--prec 1] [--iters 20]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outdoor_nerf_depth_amd import ops                       # noqa: E402
from outdoor_nerf_depth_amd.model import init_level_params   # noqa: E402
from outdoor_nerf_depth_amd.synthetic import SyntheticKitti  # noqa: E402
from outdoor_nerf_depth_amd.trainer import ALGO_MACS         # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--n_rays', type=int, default=1024)
    p.add_argument('--S', type=int, default=192)
    p.add_argument('--prec', type=int, default=1)
    p.add_argument('--iters', type=int, default=20)
    p.add_argument('--only', type=str, default='')
    a = p.parse_args()
    dev = torch.device('cuda:0')
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    b = SyntheticKitti().random_batch(a.n_rays, np.random.RandomState(0))
    ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
    far, fg_z, bg_z = ops.sample_coarse(ray_o, ray_d, T(b['min_depth']), a.S)
    eng = ops.LevelEngine(init_level_params(1)[0].to(dev), precision=a.prec)
    rows = a.n_rays * a.S
    fl_f = 2.0 * sum(ALGO_MACS['fwd']) * rows
    fl_b = 2.0 * (sum(ALGO_MACS['dx']) + sum(ALGO_MACS['fwd'])) * rows
    res = {}
    if a.only in ('', 'infer'):
        t = timeit(lambda: eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=False), a.iters)
        res['fwd_infer'] = (t, fl_f / t / 1e9)
    if a.only in ('', 'train', 'bwd'):
        t = timeit(lambda: eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=True), a.iters)
        res['fwd_train'] = (t, fl_f / t / 1e9)
    if a.only in ('', 'bwd'):
        ret = eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=True)
        g_rgb, g_depth = torch.rand_like(ret['rgb']) * 1e-3, torch.rand_like(ret['depth']) * 1e-3
        t = timeit(lambda: eng.backward(g_rgb, g_depth, None), a.iters)
        res['bwd_total'] = (t, fl_b / t / 1e9)
    for k, (t, tf) in res.items():
        print('%-10s n=%d S=%d P=%d : %8.3f ms  %8.1f TFLOP/s (algorithmic)' % (k, a.n_rays, a.S, a.prec, t, tf))


if __name__ == '__main__':
    main()
