#!/usr/bin/env python
"""Per-block cycle stamps of the fused MLP kernels (probes build, csrc/nerfpp_mlp_probes.h: -DNERFPP_STAMPS=k).

    tools/probes/variant.sh stamps4 "-DNERFPP_STAMPS=4"        # forward, fg net, bf16 training
    NERFPP_HIP_LIB=.../variants/stamps4.so python tools/probes/stamps_probe.py --what fwd --out gpurun_out/x/stamps4

Every wave stamps s_memtime when it ARRIVES at a block boundary (before its waits + the barrier) and when the barrier
RELEASES it.  Printed: per workgroup the cycles per block (release to release), which wave arrives last and how long the
others have waited for it, split by stage.
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from outdoor_nerf_depth_amd import ops, _lib as L            # noqa: E402
from outdoor_nerf_depth_amd.model import init_level_params   # noqa: E402
from outdoor_nerf_depth_amd.synthetic import SyntheticKitti  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--what', default='fwd', choices=['fwd', 'bwd'])
    p.add_argument('--n_rays', type=int, default=1024)
    p.add_argument('--out', default='stamps')
    p.add_argument('--precision', type=int, default=1, help='1 bf16 (stamps builds 2 / 4), 2 split-bf16 (builds 1 / 3 / 5)')
    p.add_argument('--infer', action='store_true', help='the inference forward (stamps build 0 / 1)')
    p.add_argument('--bf', type=int, default=0, help='fragments per weight block (default: 16, 8 in the split-bf16 roles-pipe training kernels)')
    a = p.parse_args()
    train = not a.infer
    dev = torch.device('cuda:0')
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    b = SyntheticKitti().random_batch(a.n_rays, np.random.RandomState(0))
    ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
    far, fg_z, bg_z = ops.sample_coarse(ray_o, ray_d, T(b['min_depth']), 192)
    eng = ops.LevelEngine(init_level_params(1)[0].to(dev), precision=a.precision)
    for _ in range(30):                                          # clocks up
        ret = eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=train)
        if a.what == 'bwd':
            eng.backward(torch.rand_like(ret['rgb']) * 1e-3, torch.rand_like(ret['depth']) * 1e-3, None)
    torch.cuda.synchronize()
    # wall time of the same launches, in this process: HIP events the library records around the MLP kernels
    ms = []
    for _ in range(20):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for e in ev:
            e.record()
        ret = eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=train, events=(ev[0], ev[1]))
        if a.what == 'bwd':
            eng.backward(torch.rand_like(ret['rgb']) * 1e-3, torch.rand_like(ret['depth']) * 1e-3, None, events=ev)
        torch.cuda.synchronize()
        ms.append(ev[0].elapsed_time(ev[1]) if a.what == 'fwd' else ev[0].elapsed_time(ev[1]))
    launch_ms = float(np.median(ms))
    lib = L.lib()
    fn = lib.nerfpp_probe_stamps
    fn.argtypes = [C.c_void_p, C.c_int]
    size = fn(None, 0)                                           # a size mismatch returns the size of the stamp array
    nw = 8 if a.precision == 1 else 4
    buf = np.zeros((8, nw, size // (8 * nw * 8), 2), np.uint32)
    rc = fn(buf.ctypes.data_as(C.c_void_p), buf.nbytes)
    assert rc == 0, 'nerfpp_probe_stamps rc=%d (is this the stamps build?)' % rc
    np.save(a.out + '.npy', buf)
    arr, rel = buf[..., 0].astype(np.int64), buf[..., 1].astype(np.int64)
    # weight blocks per tile (fg net): fwd_frags(0) = 1072, BWD_FRAGS = 992 fragments in blocks of 16 (8 in the split-bf16 training kernels)
    bf = a.bf or (8 if (a.precision == 2 and train) else 16)
    nblk = (1072 if a.what == 'fwd' else 992) // bf
    # blocks per stage, in stream order
    if a.what == 'fwd':
        stages = [('L0', 32)] + [('L%d' % l, 128) for l in range(1, 5)] + [('L5', 160), ('L6', 128), ('L7', 128), ('sigma', 16), ('rgb0', 80), ('rgb1', 16)]
    else:
        stages = [('dG', 16), ('dH7', 80)] + [('dH%d' % l, 128) for l in range(6, -1, -1)]
    first_of_stage, names, o = [], [], 0
    for nm, fr in stages:
        first_of_stage.append(o)
        names.append(nm)
        o += fr // bf
    assert o == nblk
    rep = {'what': a.what, 'precision': a.precision, 'training': train, 'blocks': nblk, 'block_fragments': bf, 'stage_names': names,
           'first_block_of_stage': first_of_stage, 'launch_ms_both_nets': launch_ms, 'workgroups': []}
    print('launch (fg + bg in one launch): %.4f ms' % launch_ms)
    for wg in range(8):
        if rel[wg, 0, 0] == 0:
            continue
        t0 = rel[wg, 0, 0]
        r = (rel[wg, :, :nblk] - t0) % (1 << 32)                              # 32-bit stamps: differences modulo 2^32
        ar = (arr[wg, :, :nblk] - t0 + (1 << 31)) % (1 << 32) - (1 << 31)
        per_blk = np.diff(r[0])                                   # release to release (all waves release together)
        last = ar.argmax(0)                                       # wave that arrived last at each block boundary
        wait = r - ar                                             # cycles each wave waited at the boundary
        # the interval that ENDS at the release of a stage's first block contains the previous stage's epilogue
        fo = np.array([f for f in first_of_stage if 1 <= f < nblk]) - 1
        is_first = np.zeros(nblk - 1, bool)
        is_first[fo] = True
        rep['workgroups'].append({
            'wg_slot': wg, 'total_cycles': int(r[0, -1] - ar[:, 0].min()),
            'cycles_per_block_mean': float(per_blk.mean()), 'cycles_per_block_p10_p50_p90': [float(x) for x in np.percentile(per_blk, [10, 50, 90])],
            'last_arriver_histogram': np.bincount(last, minlength=nw).tolist(),
            'steady_block_cycles_median': float(np.median(per_blk[~is_first])), 'stage_boundary_block_cycles': per_blk[is_first].tolist(),
            'steady_blocks_total': float(per_blk[~is_first].sum()), 'boundary_blocks_total': float(per_blk[is_first].sum()),
            'mean_wait_per_wave': [float(x) for x in wait.mean(1)],
            'per_block_cycles': per_blk.tolist(), 'last_arriver': last.tolist()})
    if rep['workgroups']:
        tot = float(np.mean([w['total_cycles'] for w in rep['workgroups']]))
        rep['tile_cycles_mean'] = tot
        # 768 fg tiles + 768 bg tiles over 256 CUs = 6 rounds of one tile per CU (the bg tile is ~2 % longer); 12 rounds of 128-row tiles
        rounds = 6.0 * (2 if a.precision == 2 else 1) * a.n_rays / 1024
        rep['implied_clock_ghz'] = rounds * tot / (launch_ms * 1e-3) / 1e9
        print('tile: %.0f cycles (mean of %d stamped workgroups); 6 rounds / launch time -> %.2f GHz' % (tot, len(rep['workgroups']), rep['implied_clock_ghz']))
    if rep['workgroups']:
        mfma = bf * (3 if a.precision == 2 else 2) * 32          # matrix-pipe cycles of a full block on one SIMD (two waves per SIMD in bf16)
        sb = float(np.mean([w['steady_block_cycles_median'] for w in rep['workgroups']]))
        st = float(np.mean([w['steady_blocks_total'] for w in rep['workgroups']]))
        bt = float(np.mean([w['boundary_blocks_total'] for w in rep['workgroups']]))
        rep['summary'] = {'matrix_pipe_cycles_per_full_block': mfma, 'steady_block_cycles_median': sb, 'steady_blocks_share_of_tile': st / (st + bt),
                          'stage_boundary_blocks_share_of_tile': bt / (st + bt), 'stage_boundary_blocks': int(is_first.sum()),
                          'mean_boundary_block_cycles': bt / max(1, int(is_first.sum()))}
        print('summary:', json.dumps(rep['summary']))
    with open(a.out + '.json', 'w') as f:
        json.dump(rep, f)
    for w in rep['workgroups']:
        print('wg %d: %d blocks, %.0f cycles/block (p10/50/90 %s), total %d; last arriver by wave %s; mean wait by wave %s'
              % (w['wg_slot'], nblk, w['cycles_per_block_mean'], w['cycles_per_block_p10_p50_p90'], w['total_cycles'],
                 w['last_arriver_histogram'], ['%.0f' % x for x in w['mean_wait_per_wave']]))


if __name__ == '__main__':
    main()
