#!/usr/bin/env python
"""What do the CUs an RCCL kernel holds cost the step?  No multi-GPU box: a stand-in.

The gradient all-reduce of a level (4.81 MB) runs on the update stream under the next level's kernels.  Every MLP / weight-
gradient workgroup needs a whole CU (159 KiB of LDS), so a CU held by a communication workgroup is a CU a tile cannot use.
Here the all-reduce is replaced by a kernel of n workgroups (64 KiB of LDS each: nothing co-resides with it) that hold their
CUs for t microseconds -- tools/probes/cu_census.hip -- issued exactly where NerfppTrainer issues the collective (the `comm`
hook of the trainer), and the step is timed against the plain single-GPU step in the same process, alternating blocks.

    python tools/probes/rccl_standin_probe.py --out gpurun_out/x/rccl_standin.json

Round 5 also tried holding level 0's collective back until level 1's full weight-gradient launch starts (252 workgroups: the one
launch of a step that leaves 4 CUs idle; a `_release_held` hook in NerfppTrainer, not kept): profiles/r05_rccl_standin_under_dw.json
-- +4.5 ... 13 % against +3 ... 8 % for the plain placement at 2-8 held CUs x 100-200 us, i.e. no gain within the +-2 % noise of
these runs: what a held CU costs is set by when the communication kernel itself gets a CU and by level 1's chain (slab sum ->
collective -> Adam -> re-pack) reaching the next step's level-1 forward in time, not by level 0's collective under level 1's forward.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from outdoor_nerf_depth_amd import _lib as L                               # noqa: E402
from outdoor_nerf_depth_amd.synthetic import SyntheticKitti               # noqa: E402
from outdoor_nerf_depth_amd.trainer import NerfppTrainer, batch_to_device  # noqa: E402

CEN = C.CDLL(os.path.join(ROOT, 'tools', 'probes', 'libcu_census.so'))
CEN.cu_census.restype = C.c_int
CEN.cu_census.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]


class HoldComm(object):
    """stands where dist_utils.RcclComm stands: allreduce_mean() = n workgroups holding their CUs for `cycles` shader cycles"""
    def __init__(self, n_wg, cycles, dev):
        self.n_wg, self.cycles = n_wg, cycles
        self.out = torch.zeros(2 * max(n_wg, 1), dtype=torch.int32, device=dev)

    def allreduce_mean(self, t, prescaled=True):
        if self.n_wg > 0 and t.numel() > 4096:         # (the level's gradient buffer, not the small auto-exposure one)
            CEN.cu_census(C.c_void_p(torch.cuda.current_stream().cuda_stream), self.n_wg, C.c_void_p(self.out.data_ptr()), self.cycles)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--blocks', type=int, default=5)
    p.add_argument('--steps', type=int, default=60)
    p.add_argument('--out', default='rccl_standin.json')
    p.add_argument('--wgs', default='8,16,32', help='held workgroups (= RCCL channels) to try')
    p.add_argument('--usecs', default='150,300')
    a = p.parse_args()
    dev = torch.device('cuda:0')
    scene = SyntheticKitti()
    rng = np.random.RandomState(777)
    batches = [batch_to_device(scene.random_batch(1024, rng), dev) for _ in range(a.steps)]
    GHZ = 2.0                                              # shader cycles per ns, nominal: the hold times below are +-10 %
    variants = {'none': None}
    for n_wg in [int(x) for x in a.wgs.split(',')]:
        for usec in [int(x) for x in a.usecs.split(',')]:
            variants['%dwg_%dus' % (n_wg, usec)] = (n_wg, int(usec * 1e3 * GHZ))
    trainers = {}
    for k, v in variants.items():
        tr = NerfppTrainer(dev, precision=L.PREC_BF16, use_depth=True, depth_loss_type='mse', lambda_depth=0.1,
                           depth_scale=float(scene.depth_scale))
        if v is not None:
            tr.comm = HoldComm(v[0], v[1], dev)
        trainers[k] = tr
    times = {k: [] for k in variants}
    for blk in range(a.blocks + 1):                       # block 0 = warm-up
        for k, tr in trainers.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for b in batches:
                tr.train_step(b)
            tr.flush()
            torch.cuda.synchronize()
            if blk:
                times[k].append(1e3 * (time.perf_counter() - t0) / a.steps)
    base = np.array(times['none'])
    rep = {'steps_per_block': a.steps, 'blocks': a.blocks, 'ms_per_step': {}}
    for k, v in times.items():
        v = np.array(v)
        rep['ms_per_step'][k] = {'median': round(float(np.median(v)), 4), 'paired_diff_vs_none_ms': round(float(np.median(v - base)), 4),
                                 'pct': round(100.0 * float(np.median(v - base)) / float(np.median(base)), 2)}
        print(k, rep['ms_per_step'][k], flush=True)
    with open(a.out, 'w') as f:
        json.dump(rep, f, indent=1)


if __name__ == '__main__':
    main()
