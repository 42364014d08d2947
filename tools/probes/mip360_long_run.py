#!/usr/bin/env python
"""MipNeRF-360 trainer, 300 steps on ONE fixed synthetic batch (4096 rays, fixed jitter): the loss terms every 50 steps with the fused
launches of round 5 (default) or without them (--unfused: MIP360_NO_FUSED_PROP / _VIEW / _MULTI_DW / _DEFER_DW / _BATCH_PACK) -- the
two must track each other (same arithmetic up to summation order) and the loss must fall.

    python tools/probes/mip360_long_run.py [--unfused] [--steps 300]
"""
import argparse
import os
import sys

p = argparse.ArgumentParser()
p.add_argument('--unfused', action='store_true')
p.add_argument('--steps', type=int, default=300)
a = p.parse_args()
if a.unfused:
    for k in ('FUSED_PROP', 'FUSED_VIEW', 'MULTI_DW', 'DEFER_DW', 'BATCH_PACK'):
        os.environ['MIP360_NO_' + k] = '1'
import numpy as np   # noqa: E402
import torch   # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from outdoor_nerf_depth_amd import mip360 as M   # noqa: E402

dev = torch.device('cuda:0')
rs_p = np.random.RandomState(0)
he = lambda shapes: [(rs_p.uniform(-np.sqrt(6.0 / i), np.sqrt(6.0 / i), (i, o)).astype(np.float32), np.zeros(o, np.float32)) for i, o in shapes]
prop, nerf = he(M.mlp_shapes(M.PROP_CFG)), he(M.mlp_shapes(M.NERF_CFG))
rs = np.random.RandomState(1)
n = 4096
d = rs.randn(n, 3).astype(np.float32)
d /= np.linalg.norm(d, axis=-1, keepdims=True)
T = lambda x: torch.from_numpy(x).to(dev)
rays = dict(origins=T((rs.randn(n, 3) * 0.3).astype(np.float32)), directions=T(d), viewdirs=T(d.copy()),
            radii=T(np.full((n, 1), 2e-3, np.float32)), near=T(np.full((n, 1), 0.2, np.float32)), far=T(np.full((n, 1), 1e6, np.float32)))
gt = T((0.5 + 0.5 * np.sin(3 * d)).astype(np.float32))              # a smooth function of the direction: learnable
sup = T(np.where(rs.rand(n) < .5, rs.uniform(1, 6, n), 0).astype(np.float32))
jit = [T(rs.rand(n).astype(np.float32)) for _ in range(3)]
tr = M.Mip360Trainer(prop, nerf, dev, max_steps=2000)
for s in range(a.steps + 1):
    sc = tr.train_step(rays, gt, sup, jitter01=jit)
    if s % 50 == 0:
        v = sc.detach().cpu().numpy()
        print('step %4d  %s' % (s, ' '.join('%.6f' % x for x in v[:6])), flush=True)
