#!/usr/bin/env python
"""Per-step GPU time over a long run (HIP events at every step boundary): is the start of a run slower than its steady state?
    python tools/probes/step_curve.py [--steps 240]"""
import argparse, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from outdoor_nerf_depth_amd import _lib as L
from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
from outdoor_nerf_depth_amd.trainer import NerfppTrainer, batch_to_device

p = argparse.ArgumentParser(); p.add_argument('--steps', type=int, default=240); a = p.parse_args()
dev = torch.device('cuda:0')
scene = SyntheticKitti(); rng = np.random.RandomState(777)
batches = [batch_to_device(scene.random_batch(1024, rng), dev) for _ in range(32)]
tr = NerfppTrainer(dev, precision=L.PREC_BF16, use_depth=True, depth_loss_type='mse', lambda_depth=0.1, depth_scale=float(scene.depth_scale))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
torch.cuda.synchronize()
ev[0].record()
for i in range(a.steps):
    tr.train_step(batches[i % 32])
    ev[i + 1].record()
tr.flush(); torch.cuda.synchronize()
ms = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps)])
groups = [(0, 1), (1, 3), (3, 8), (8, 28), (28, 60), (60, 120), (120, a.steps)]
print(json.dumps({'ms_per_step_by_step_range': {'%d-%d' % g: round(float(ms[g[0]:g[1]].mean()), 4) for g in groups if g[1] <= a.steps}}))
