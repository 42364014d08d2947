#!/usr/bin/env python
"""What does the side-stream parameter update (slab sum, fix-up, Adam, fold, re-pack: six short launches per level under the next
level's kernels) cost the step beyond being hidden?  One process, alternating blocks: the default trainer, the update inline
on the caller's stream, and NO update at all (timing only: the parameters stay where they are).

    python tools/probes/update_cost_probe.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from outdoor_nerf_depth_amd import _lib as L                               # noqa: E402
from outdoor_nerf_depth_amd.synthetic import SyntheticKitti               # noqa: E402
from outdoor_nerf_depth_amd.trainer import NerfppTrainer, batch_to_device  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    scene = SyntheticKitti()
    rng = np.random.RandomState(777)
    steps, blocks = 60, 6
    batches = [batch_to_device(scene.random_batch(1024, rng), dev) for _ in range(steps)]
    mk = lambda **kw: NerfppTrainer(dev, precision=L.PREC_BF16, use_depth=True, depth_loss_type='mse', lambda_depth=0.1,
                                    depth_scale=float(scene.depth_scale), **kw)
    trainers = {'side_update': mk(), 'inline_update': mk(overlap_allreduce=False), 'no_update': mk(), 'reduce_only': mk()}
    trainers['no_update']._update = lambda m, step: None
    t = trainers['reduce_only']
    t._update = lambda m, step: t.engines[m].reduce_grads()
    # the slab sum right behind the weight-gradient launches, on THEIR stream (level 1: the caller's), the rest of the update
    # (fix-up is part of the C call too; Adam, fold, re-pack) on the side stream as before
    ti = mk()
    for eng in ti.engines:
        def b(*a, _ob=eng.backward, _or=eng.reduce_grads, **kw):
            r = _ob(*a, **kw)
            if kw.get('defer_reduce'):
                _or()
            return r
        eng.backward = b
        eng.reduce_grads = lambda: None
    trainers['reduce_behind_dw'] = ti
    times = {k: [] for k in trainers}
    for blk in range(blocks + 1):
        for k, tr in trainers.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for b in batches:
                tr.train_step(b)
            tr.flush()
            torch.cuda.synchronize()
            if blk:
                times[k].append(1e3 * (time.perf_counter() - t0) / steps)
    base = np.array(times['side_update'])
    for k, v in times.items():
        v = np.array(v)
        print(k, json.dumps({'median': round(float(np.median(v)), 4), 'paired_diff_vs_default_ms': round(float(np.median(v - base)), 4)}))


if __name__ == '__main__':
    main()
