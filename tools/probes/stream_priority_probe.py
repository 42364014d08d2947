#!/usr/bin/env python
"""Does queue priority keep the side-stream kernels (parameter update, level 0's backward) out of the way of the caller's MLP
launches?  One process, alternating blocks: the default trainer on the default stream, and the same trainer driven from a
HIGH-priority stream (its side streams stay at the default priority).

    python tools/probes/stream_priority_probe.py
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
    print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else None)
    scene = SyntheticKitti()
    rng = np.random.RandomState(777)
    steps, blocks = 60, 6
    batches = [batch_to_device(scene.random_batch(1024, rng), dev) for _ in range(steps)]
    mk = lambda **kw: NerfppTrainer(dev, precision=L.PREC_BF16, use_depth=True, depth_loss_type='mse', lambda_depth=0.1,
                                    depth_scale=float(scene.depth_scale), **kw)
    hi = torch.cuda.Stream(device=dev, priority=-1)
    trainers = {'default_stream': (mk(), None), 'caller_high_priority': (mk(), hi)}
    # side streams at LOW priority, caller on the default stream
    try:
        t = mk()
        t.update_stream = torch.cuda.Stream(device=dev, priority=1)
        t.level_streams = [torch.cuda.Stream(device=dev, priority=1) for _ in t.level_streams]
        print('low-priority side streams:', t.update_stream.priority)
        trainers['side_low_priority'] = (t, None)
    except Exception as e:                                                  # noqa: BLE001
        print('priority 1 not available:', e)
    times = {k: [] for k in trainers}
    for blk in range(blocks + 1):
        for k, (tr, st) in trainers.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if st is None:
                for b in batches:
                    tr.train_step(b)
                tr.flush()
            else:
                with torch.cuda.stream(st):
                    for b in batches:
                        tr.train_step(b)
                    tr.flush()
            torch.cuda.synchronize()
            if blk:
                times[k].append(1e3 * (time.perf_counter() - t0) / steps)
    base = np.array(times['default_stream'])
    for k, v in times.items():
        v = np.array(v)
        print(k, json.dumps({'median': round(float(np.median(v)), 4), 'paired_diff_vs_default_ms': round(float(np.median(v - base)), 4)}))


if __name__ == '__main__':
    main()
