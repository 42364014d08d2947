#!/usr/bin/env python
"""VERDICT r04 item 1, bounding experiment: what would one-layer recompute inside dw_kernel buy?

One process, one NerfppTrainer (bf16, N_rand 1024), blocks of steps alternating between configurations that differ only in
environment switches of the probes build (tools/probes/build_recompute_probe.sh -1; run with NERFPP_HIP_LIB pointing at
csrc/build/variants/skiph_-1.so):
  NERFPP_SKIP_H_RT  bit l: the training forward does not write H_l (the weight-gradient kernel then reads the previous step's)
  NERFPP_DW_DEBUG   4 (+ 8: every full job) (+ k << 4: k-chunks of the H0 job): the full weight-gradient jobs whose input is
                    unsaved issue the recompute's 16 LDS reads + MFMAs + epilogue + tile write + second barrier per 32-row chunk
  NERFPP_DW_RC_K / NERFPP_DW_RC_K0   slices of those jobs (the others share the rest of the 256 workgroups)
Gradients are wrong by construction (stale / garbage operands); the numbers are times.  Per configuration: wall ms per
step over untapped blocks and the level-1 kernel-group times of tapped steps.
"""
import argparse
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

KEYS = ('NERFPP_SKIP_H_RT', 'NERFPP_DW_DEBUG', 'NERFPP_DW_RC_K', 'NERFPP_DW_RC_K0')
CONFIGS = [
    ('baseline', {}),
    ('fwd_skip_H0_H2_H6', {'NERFPP_SKIP_H_RT': 0x45}),
    ('fwd_skip_H0_H2_H4_H6', {'NERFPP_SKIP_H_RT': 0x55}),
    ('dw_emul_3jobs_k16', {'NERFPP_DW_DEBUG': 4}),
    ('skip3+emul_k16', {'NERFPP_SKIP_H_RT': 0x45, 'NERFPP_DW_DEBUG': 4}),
    ('skip3+emul_h0k4', {'NERFPP_SKIP_H_RT': 0x45, 'NERFPP_DW_DEBUG': 4 + (4 << 4)}),
    ('skip3+emul_h0k4_rc23', {'NERFPP_SKIP_H_RT': 0x45, 'NERFPP_DW_DEBUG': 4 + (4 << 4), 'NERFPP_DW_RC_K': 23}),
    ('skip3+emul_h0k4_rc24_k0_18', {'NERFPP_SKIP_H_RT': 0x45, 'NERFPP_DW_DEBUG': 4 + (4 << 4), 'NERFPP_DW_RC_K': 24, 'NERFPP_DW_RC_K0': 18}),
    ('skip3+emul_h0k4_rc26_k0_18', {'NERFPP_SKIP_H_RT': 0x45, 'NERFPP_DW_DEBUG': 4 + (4 << 4), 'NERFPP_DW_RC_K': 26, 'NERFPP_DW_RC_K0': 18}),
    ('dw_emul_all_full_jobs', {'NERFPP_DW_DEBUG': 12}),
    # H0 alone: its job reads the encoded point (4 chunk blocks; 6 for the background net) instead of H0 and recomputes it
    ('skipH0', {'NERFPP_SKIP_H_RT': 1}),
    ('skipH0+emul_k4_short', {'NERFPP_SKIP_H_RT': 1, 'NERFPP_DW_DEBUG': 4 + 512 + (4 << 4)}),
    ('skipH0+emul_k6_short', {'NERFPP_SKIP_H_RT': 1, 'NERFPP_DW_DEBUG': 4 + 512 + (6 << 4)}),
    ('skipH0+emul_k6_short_k0_18', {'NERFPP_SKIP_H_RT': 1, 'NERFPP_DW_DEBUG': 4 + 512 + (6 << 4), 'NERFPP_DW_RC_K': 0, 'NERFPP_DW_RC_K0': 18}),
    ('skipH0+emul_k6_short_k0_13', {'NERFPP_SKIP_H_RT': 1, 'NERFPP_DW_DEBUG': 4 + 512 + (6 << 4), 'NERFPP_DW_RC_K': 0, 'NERFPP_DW_RC_K0': 13}),
]


def set_env(cfg):
    for k in KEYS:
        os.environ.pop(k, None)
    for k, v in cfg.items():
        os.environ[k] = str(v)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--blocks', type=int, default=4)
    p.add_argument('--steps', type=int, default=40)
    p.add_argument('--n_rand', type=int, default=1024)
    p.add_argument('--only', type=str, default='')
    a = p.parse_args()
    dev = torch.device('cuda:0')
    scene = SyntheticKitti()
    rng = np.random.RandomState(777)
    batches = [batch_to_device(scene.random_batch(a.n_rand, rng), dev) for _ in range(a.steps)]
    tr = NerfppTrainer(dev, precision=L.PREC_BF16, use_depth=True, depth_loss_type='mse', lambda_depth=0.1,
                       depth_scale=float(scene.depth_scale))
    cfgs = [c for c in CONFIGS if not a.only or c[0] in a.only.split(',')]
    mk = lambda: torch.cuda.Event(enable_timing=True)
    wall = {c[0]: [] for c in cfgs}
    taps = {c[0]: {'fwd': [], 'bwd': [], 'dw': []} for c in cfgs}
    set_env({})
    for b in batches:                                 # populate every saved tensor with real data
        tr.train_step(b)
    tr.flush()
    for blk in range(a.blocks + 1):                   # block 0 = warm-up
        for name, cfg in cfgs:
            set_env(cfg)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for b in batches:
                tr.train_step(b)
            tr.flush()
            torch.cuda.synchronize()
            if blk:
                wall[name].append(1e3 * (time.perf_counter() - t0) / a.steps)
            for b in batches[:4]:                      # tapped steps: level-1 kernel groups alone on the GPU
                ev = {'fwd': (mk(), mk()), 'bwd': (mk(), mk(), mk(), mk())}
                for e in ev['fwd'] + ev['bwd']:
                    e.record()
                tr.train_step(b, events=[None, ev])
                tr.flush()
                torch.cuda.synchronize()
                if blk:
                    taps[name]['fwd'].append(ev['fwd'][0].elapsed_time(ev['fwd'][1]))
                    taps[name]['bwd'].append(ev['bwd'][0].elapsed_time(ev['bwd'][1]))
                    taps[name]['dw'].append(ev['bwd'][2].elapsed_time(ev['bwd'][3]))
            set_env({})
            for b in batches[:2]:                      # refresh the saved tensors with real data
                tr.train_step(b)
            tr.flush()
    out = {'n_rand': a.n_rand, 'steps_per_block': a.steps, 'blocks': a.blocks, 'configs': {}}
    base = np.array(wall[cfgs[0][0]])
    for name, cfg in cfgs:
        w = np.array(wall[name])
        out['configs'][name] = {'env': cfg, 'ms_per_step_blocks': [round(float(x), 4) for x in w],
                                'ms_per_step_median': round(float(np.median(w)), 4),
                                'paired_diff_vs_first_ms': round(float(np.median(w - base)), 4),
                                'L1_fwd_ms': round(float(np.median(taps[name]['fwd'])), 4),
                                'L1_bwd_ms': round(float(np.median(taps[name]['bwd'])), 4),
                                'L1_dw_ms': round(float(np.median(taps[name]['dw'])), 4)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
