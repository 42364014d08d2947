#!/usr/bin/env python
"""Per-workgroup timestamps of the weight-gradient launches (probes build: tools/probes/build_probes.sh).

    NERFPP_HIP_LIB=.../variants/libnerfpp_hip_probes.so python tools/probes/dw_stamps_probe.py --out gpurun_out/x/dw_stamps

Every workgroup of dw_kernel<1, true> (256 x 256 jobs) and dw_kernel<1, false> (narrow jobs) records s_memtime at entry, after
its chunk loop and after its slab write, with its job, chunk count and XCD.  Printed per job: slices, chunks per slice, cycles
per chunk (mean / max over the slices), start skew, and the launch's critical workgroup.
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

NARROW_NAMES = {0: 'L0 (dZ0 x X)', 1: 'L5-encoding (dZ5 x X)', 2: '[dS|dG] x H7', 3: 'rgb0-view (dG x DIRX)', 4: 'rgb1 (dP x G)'}


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--n_rays', type=int, default=1024)
    p.add_argument('--out', default='dw_stamps')
    a = p.parse_args()
    dev = torch.device('cuda:0')
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    b = SyntheticKitti().random_batch(a.n_rays, np.random.RandomState(0))
    ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
    far, fg_z, bg_z = ops.sample_coarse(ray_o, ray_d, T(b['min_depth']), 192)
    eng = ops.LevelEngine(init_level_params(1)[0].to(dev), precision=1)
    for _ in range(30):
        ret = eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=True)
        eng.backward(torch.rand_like(ret['rgb']) * 1e-3, torch.rand_like(ret['depth']) * 1e-3, None)
    torch.cuda.synchronize()
    buf = np.zeros((2, 256, 6), np.uint64)
    fn = L.lib().nerfpp_probe_dw_stamps
    fn.argtypes = [C.c_void_p, C.c_int]
    rc = fn(buf.ctypes.data_as(C.c_void_p), buf.nbytes)
    assert rc == 0, 'nerfpp_probe_dw_stamps rc=%d (probes build?)' % rc
    np.save(a.out + '.npy', buf)
    rep = {}
    for which, name in ((0, 'full'), (1, 'narrow')):
        s = buf[which].astype(np.int64)
        s = s[s[:, 0] > 0]
        job, nch, xcc = s[:, 3], s[:, 4], s[:, 5]
        # the cycle counters of the XCDs are not synchronised on every box: time is relative to the first start on the same XCD
        t0 = np.array([s[xcc == x, 0].min() for x in xcc])
        start, loop_end, end = s[:, 0] - t0, s[:, 1] - t0, s[:, 2] - t0
        span = int(end.max())
        crit = int(end.argmax())
        rows = []
        for jid in sorted(set(job.tolist())):
            m = job == jid
            cyc = (loop_end[m] - start[m]) / np.maximum(nch[m], 1)
            rows.append({'job': int(jid), 'slices': int(m.sum()), 'chunks_per_slice': int(nch[m].max()),
                         'cycles_per_chunk_mean': float(cyc.mean()), 'cycles_per_chunk_max': float(cyc.max()),
                         'start_mean': float(start[m].mean()), 'start_max': int(start[m].max()), 'end_max': int(end[m].max()),
                         'epilogue_cycles_mean': float((end[m] - loop_end[m]).mean())})
        rep[name] = {'workgroups': int(len(s)), 'launch_span_cycles': span, 'critical_wg': crit, 'critical_job': int(job[crit]),
                     'start_skew_cycles_p50_p90_max': [float(x) for x in np.percentile(start, [50, 90, 100])],
                     'per_xcd_end_max': {int(x): int(end[xcc == x].max()) for x in sorted(set(xcc.tolist()))}, 'jobs': rows}
        print('== %s: %d workgroups, span %d cycles, critical wg %d (job %d)' % (name, len(s), span, crit, job[crit]))
        for r in rows:
            print('   job %2d: %2d slices x %4d chunks, %6.0f cycles/chunk (max %6.0f), start %6.0f (max %6d), end max %7d, epilogue %5.0f'
                  % (r['job'], r['slices'], r['chunks_per_slice'], r['cycles_per_chunk_mean'], r['cycles_per_chunk_max'],
                     r['start_mean'], r['start_max'], r['end_max'], r['epilogue_cycles_mean']))
    with open(a.out + '.json', 'w') as f:
        json.dump(rep, f, indent=1)


if __name__ == '__main__':
    main()
