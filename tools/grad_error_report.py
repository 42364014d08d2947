#!/usr/bin/env python
"""Measured errors of the HIP path in both precisions against the reference's golden vectors:

  * forward: max relative error of rgb / depth / fg_weights against tests/golden/forward.npz (float32 reference),
  * gradients: per-tensor relative L2 error and max |diff| / RMS against the FLOAT64 run of the reference
    (tests/golden/grads_*.npz; the float32 reference is itself ~1e-1 RMS away from it).

    python tools/grad_error_report.py [--json profiles/r02_bf16_error_report.json] [modes ...]

`measure()` is also what tests/test_gpu_round2.py calls to hold the bf16 kernels to 2x the recorded values.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def measure(modes=('rgbonly', 'mse', 'l1', 'kl')):
    import torch
    from oracle import nerfpp_oracle as O                     # test infrastructure: parameter init + names only
    from outdoor_nerf_depth_amd import ops
    dev = torch.device('cuda:0')
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    levels = O.init_params_like_reference(2)
    flat = lambda lv: np.concatenate([lv[k].reshape(-1) for k in O.param_order()]).astype(np.float32)
    shapes = {}
    for net, in_ch in (('fg_net', 63), ('bg_net', 84)):
        for k, s in O.mlp_param_shapes(in_ch, 27).items():
            shapes['%s.%s' % (net, k)] = s
    out = {'bf16': {}, 'split': {}}
    names = {1: 'bf16', 2: 'split'}
    gf = np.load(os.path.join(GOLDEN, 'forward.npz'))
    for prec in (2, 1):
        for m, (fz, bz) in enumerate((('fg_z0', 'bg_z0'), ('fg_z1', 'bg_z1'))):
            eng = ops.LevelEngine(T(flat(levels[m])), precision=prec)
            ret = eng.forward(T(gf['ray_o']), T(gf['ray_d']), T(gf['fg_far']), T(gf[fz]), T(gf[bz]))
            for k in ('rgb', 'depth', 'fg_weights', 'bg_lambda'):
                ref = gf['L%d.%s' % (m, k)]
                got = ret[k].cpu().numpy()
                out[names[prec]]['fwd.L%d.%s.max_abs_over_max' % (m, k)] = float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))
    for mode in modes:
        g = np.load(os.path.join(GOLDEN, 'grads_%s.npz' % mode))
        for m in (0, 1):
            for prec in (2, 1):
                eng = ops.LevelEngine(T(flat(levels[m])), precision=prec)
                fz, bz = g['L%d.fg_z' % m], g['L%d.bg_z' % m]
                ret = eng.forward(T(g['ray_o']), T(g['ray_d']), T(g['fg_far']), T(fz), T(bz), training=True)
                sc, g_rgb, g_depth, g_w = ops.loss_and_grads(ret, T(g['rgb_gt']), T(g['depth_sup']), mode,
                                                             float(g['lambda_depth']), kl_sigma=float(g['depth_sigma_scaled']),
                                                             fg_z_vals=T(fz), fg_far_depth=T(g['fg_far']))
                gr = eng.backward(g_rgb, g_depth, g_w).cpu().numpy()
                off = 0
                worst_l2, worst_max = 0.0, 0.0
                for k in O.param_order():
                    n = int(np.prod(shapes[k]))
                    mine = gr[off:off + n][g['L%d.%s.idx' % (m, k)]]
                    ref = g['L%d.%s.g64' % (m, k)]
                    rms = g['L%d.%s.norm64' % (m, k)] / np.sqrt(n) + 1e-12
                    if n > 3:                                  # 1- and 3-element tensors are single cancelling sums
                        worst_l2 = max(worst_l2, float(np.linalg.norm(mine - ref) / (np.linalg.norm(ref) + 1e-30)))
                    worst_max = max(worst_max, float(np.abs(mine - ref).max() / rms))
                    off += n
                out[names[prec]]['grad.%s.L%d.worst_rel_l2' % (mode, m)] = worst_l2
                out[names[prec]]['grad.%s.L%d.worst_max_over_rms' % (mode, m)] = worst_max
                out[names[prec]]['loss.%s.L%d.rel' % (mode, m)] = float(abs(float(sc[0]) - float(g['L%d.loss' % m])) /
                                                                       abs(float(g['L%d.loss' % m])))
    return out


if __name__ == '__main__':
    args = sys.argv[1:]
    path = None
    if '--json' in args:
        i = args.index('--json')
        path = args[i + 1]
        del args[i:i + 2]
    res = measure(tuple(args) or ('rgbonly', 'mse', 'l1', 'kl'))
    for prec in ('split', 'bf16'):
        print('== %s' % prec)
        for k, v in sorted(res[prec].items()):
            print('  %-44s %.3e' % (k, v))
    if path:
        res['_doc'] = ('measured on MI355X by tools/grad_error_report.py: forward max|diff|/max|ref| vs the float32 reference, '
                       'gradient worst-tensor relative L2 and max|diff|/RMS vs the float64 run of the reference')
        with open(path, 'w') as f:
            json.dump(res, f, indent=1, sort_keys=True)
