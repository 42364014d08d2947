#!/usr/bin/env python
"""Per-tensor gradient error of the HIP path (both precisions) against the float64 run of the
reference (tests/golden/grads_*.npz): relative L2 over the sampled entries."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nerfpp_oracle as O                     # noqa: E402  (test infrastructure)
from outdoor_nerf_depth_amd import ops                    # noqa: E402

dev = torch.device('cuda:0')
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
levels = O.init_params_like_reference(2)
flat = lambda lv: np.concatenate([lv[k].reshape(-1) for k in O.param_order()]).astype(np.float32)
shapes = {}
for net, in_ch in (('fg_net', 63), ('bg_net', 84)):
    for k, s in O.mlp_param_shapes(in_ch, 27).items():
        shapes['%s.%s' % (net, k)] = s
for mode in sys.argv[1:] or ['rgbonly', 'mse']:
    g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'grads_%s.npz' % mode))
    for m in (0, 1):
        res = {}
        for prec in (2, 1):
            eng = ops.LevelEngine(T(flat(levels[m])), precision=prec)
            fz, bz = g['L%d.fg_z' % m], g['L%d.bg_z' % m]
            ret = eng.forward(T(g['ray_o']), T(g['ray_d']), T(g['fg_far']), T(fz), T(bz), training=True)
            sc, g_rgb, g_depth, g_w = ops.loss_and_grads(ret, T(g['rgb_gt']), T(g['depth_sup']), mode,
                                                         float(g['lambda_depth']), kl_sigma=float(g['depth_sigma_scaled']),
                                                         fg_z_vals=T(fz), fg_far_depth=T(g['fg_far']))
            gr = eng.backward(g_rgb, g_depth, g_w).cpu().numpy()
            off = 0
            for k in O.param_order():
                n = int(np.prod(shapes[k]))
                mine = gr[off:off + n][g['L%d.%s.idx' % (m, k)]]
                ref = g['L%d.%s.g64' % (m, k)]
                res.setdefault(k, []).append(np.linalg.norm(mine - ref) / (np.linalg.norm(ref) + 1e-30))
                off += n
        print('== %s level %d   rel-L2 error  [split-bf16, bf16]' % (mode, m))
        for k, v in res.items():
            if 'weight' in k:
                print('  %-36s %.2e  %.2e' % (k, v[0], v[1]))
