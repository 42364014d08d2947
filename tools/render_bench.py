#!/usr/bin/env python
"""Full-frame rendering throughput (render_single_image, deterministic sampling, both
cascade levels, 64 + 128 samples/ray) on one GPU: seconds per 375x1242 frame, rays/s, and the
algorithmic MFMA rate (0.613 GFLOP per ray).

    python tools/render_bench.py [--chunk 8192] [--frames 3] [--precision bf16|split]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outdoor_nerf_depth_amd import _lib as L                                   # noqa: E402
from outdoor_nerf_depth_amd.data_loader_split import synthetic_ray_samplers   # noqa: E402
from outdoor_nerf_depth_amd.ddp_train_nerf import render_single_image         # noqa: E402
from outdoor_nerf_depth_amd.trainer import NerfppTrainer                      # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--chunk', type=int, default=8192)
    p.add_argument('--frames', type=int, default=3)
    p.add_argument('--precision', default='bf16')
    p.add_argument('--keep_dists', action='store_true', help="also return fg_dists [H,W,S] like the reference's dict (0.36 GB D2H per frame)")
    a = p.parse_args()
    dev = torch.device('cuda:0')
    samplers = synthetic_ray_samplers('test', 1, 'gt', 20, 375, 1242)[:1]
    tr = NerfppTrainer(dev, precision=L.PREC_BF16 if a.precision == 'bf16' else L.PREC_SPLIT_BF16, use_depth=False)
    render_single_image(0, 1, tr, samplers[0], a.chunk, keep_dists=a.keep_dists)                   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.frames):
        render_single_image(0, 1, tr, samplers[0], a.chunk, keep_dists=a.keep_dists)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.frames
    n = 375 * 1242
    print(json.dumps({'metric': 'render s/frame 375x1242, 64+128 samples/ray', 's_per_frame': dt, 'rays_per_s': n / dt,
                      'algorithmic_tflops': n * 0.613e9 / dt / 1e12, 'chunk': a.chunk, 'precision': a.precision, 'keep_dists': a.keep_dists}))


if __name__ == '__main__':
    main()
