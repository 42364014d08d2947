#!/usr/bin/env python
"""Bit-for-bit comparison of two builds of libnerfpp_hip.so on the split-bf16 path (inference forward, training forward + saved
tensors + sign words as seen by the backward, gradients).

    NERFPP_HIP_LIB=<build A> python tools/probes/split_dump.py --out a.npz
    NERFPP_HIP_LIB=<build B> python tools/probes/split_dump.py --out b.npz
    python tools/probes/split_dump.py --compare a.npz b.npz
"""
import argparse
import os
import sys

import numpy as np


def dump(a):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from outdoor_nerf_depth_amd import ops, _lib as L
    from outdoor_nerf_depth_amd.model import init_level_params
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    dev = torch.device('cuda:0')
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    out = {}
    shapes = [tuple(int(v) for v in x.split('x')) for x in a.shapes.split(',')] if a.shapes else [(a.n_rays, a.S), (37, 64)]
    for n_rays, S in shapes:                                            # default: a full batch and a ragged one (tile tails)
        b = SyntheticKitti().random_batch(n_rays, np.random.RandomState(3))
        ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
        rs = np.random.RandomState(7)                                  # explicit perturbation uniforms: the same inputs in every run
        t_fg, t_bg = rs.rand(n_rays, S).astype(np.float32), rs.rand(n_rays, S).astype(np.float32)
        far, fg_z, bg_z = ops.sample_coarse(ray_o, ray_d, T(b['min_depth']), S, T(t_fg), T(t_bg))
        params = init_level_params(1)[0].to(dev)
        params = params + 0.02 * torch.randn(params.shape, generator=torch.Generator().manual_seed(5)).to(dev)   # biases off zero
        for prec, name in ((L.PREC_SPLIT_BF16, 'split'), (L.PREC_SPLIT_FWD, 'split_fwd')):
            eng = ops.LevelEngine(params.clone(), precision=prec)
            tag = '%s_%dx%d_' % (name, n_rays, S)
            if name == 'split':
                ret = eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=False)
                for k, v in ret.items():
                    out[tag + 'infer_' + k] = v.cpu().numpy()
            ret = eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=True)
            for k, v in ret.items():
                out[tag + 'train_' + k] = v.cpu().numpy()
            for net in (0, 1):
                for t in ([0] + ([] if name == 'split_fwd' else [1]) + list(range(2, 9)) + [10, 11]):     # X, H0..H7, G, DIRX
                    for plane in ((0, 1) if name == 'split' else (0,)):
                        out[tag + 'ws_n%d_t%d_p%d' % (net, t, plane)] = eng.saved_tensor(net, t, plane).cpu().numpy()
            g = torch.Generator(device='cpu').manual_seed(11)
            g_rgb = (torch.rand(ret['rgb'].shape, generator=g) * 1e-3).to(dev)
            g_depth = (torch.rand(ret['depth'].shape, generator=g) * 1e-3).to(dev)
            grads = eng.backward(g_rgb, g_depth, None)
            out[tag + 'grads'] = grads.cpu().numpy()
            if name == 'split':                                     # what the split-bf16 dX chain wrote: dZ0..dZ7, [dS | dG], dP
                for net in (0, 1):
                    for t in list(range(12, 20)) + [21, 23]:
                        for plane in (0, 1):
                            out[tag + 'ws_n%d_t%d_p%d' % (net, t, plane)] = eng.saved_tensor(net, t, plane).cpu().numpy()
    np.savez(a.out, **out)
    print('wrote', a.out, len(out), 'arrays')


def compare(f0, f1):
    a, b = np.load(f0), np.load(f1)
    bad = 0
    for k in a.files:
        x, y = a[k], b[k]
        same = x.shape == y.shape and np.array_equal(x.view(np.uint8), y.view(np.uint8))
        if not same:
            bad += 1
            d = np.abs(x.astype(np.float64) - y.astype(np.float64))
            print('DIFF %-40s max|d| %.3e  mismatching %d / %d  nan %d / %d' % (k, np.nanmax(d), int((x != y).sum()), x.size,
                                                                             int(np.isnan(x).sum()), int(np.isnan(y).sum())))
            if '_ws_' in k and x.ndim == 2:          # saved tensor [rows, ld]: which waves (32-row blocks of a 128-row tile) and chunks
                r, c = np.nonzero(x != y)
                print('     rows %d..%d  wave-of-tile histogram %s  chunk histogram %s' % (r.min(), r.max(), np.bincount((r % 128) // 32, minlength=4).tolist(),
                                                                                           np.bincount(c // 16, minlength=x.shape[1] // 16).tolist()))
    print('%d arrays compared, %d differ' % (len(a.files), bad))
    return bad


if __name__ == '__main__':
    p = argparse.ArgumentParser()
    p.add_argument('--out', default='split_dump.npz')
    p.add_argument('--n_rays', type=int, default=256)
    p.add_argument('--S', type=int, default=192)
    p.add_argument('--shapes', default='', help="comma list of RAYSxSAMPLES, e.g. '1x2,17x31,4096x192' (default: n_rays x S and 37x64)")
    p.add_argument('--compare', nargs=2)
    a = p.parse_args()
    if a.compare:
        sys.exit(1 if compare(*a.compare) else 0)
    dump(a)
