#!/usr/bin/env python
"""Spatial partition experiment: do the HBM-bound weight-gradient GEMMs keep their byte rate on a
SUBSET of the CUs, and what do the MFMA-bound MLP kernels lose on the complement?  Streams with a CU mask
(hipExtStreamCreateWithCUMask) place whole launches on chosen CUs without touching a kernel.

    NERFPP_HIP_LIB=.../libnerfpp_hip_probes.so NERFPP_DEFER_DW=1 python tools/probes/cu_mask_probe.py > out.json

Needs the probes build (tools/probes/build_probes.sh): with NERFPP_DEFER_DW=1 `backward(defer_reduce=True)` stops after the
dX kernels and `reduce_grads()` runs the weight-gradient launches + the slab sum on the stream it is called on.
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from outdoor_nerf_depth_amd import ops                       # noqa: E402
from outdoor_nerf_depth_amd.model import init_level_params   # noqa: E402
from outdoor_nerf_depth_amd.synthetic import SyntheticKitti  # noqa: E402

CEN = C.CDLL(os.path.join(ROOT, 'tools', 'probes', 'libcu_census.so'))
CEN.cu_mask_stream.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_uint32)]
CEN.cu_census.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
N_CU = 256


def mask_stream(bits):
    """bits: iterable of CU-mask bit indices to enable"""
    words = (C.c_uint32 * (N_CU // 32))()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = CEN.cu_mask_stream(C.byref(s), N_CU // 32, words)
    assert rc == 0, 'hipExtStreamCreateWithCUMask rc=%d' % rc
    return torch.cuda.ExternalStream(s.value)


def census(stream, n_blocks=1024):
    out = torch.zeros(2 * n_blocks, dtype=torch.int32, device='cuda')
    with torch.cuda.stream(stream):
        CEN.cu_census(C.c_void_p(stream.cuda_stream), n_blocks, C.c_void_p(out.data_ptr()), 20000)
    stream.synchronize()
    a = out.cpu().numpy().astype(np.uint32).reshape(-1, 2)
    hw, xcc = a[:, 0], a[:, 1] & 15
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    ids = set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_xcc = {}
    for x, _, _, _ in ids:
        per_xcc[int(x)] = per_xcc.get(int(x), 0) + 1
    return {'distinct_cus': len(ids), 'per_xcc': per_xcc}


def layouts(k):
    """two ways of choosing k of the 256 mask bits"""
    return {'low': list(range(k)), 'strided': [int(i * N_CU / k) for i in range(k)]}


def timed(fn, stream, iters=10, warm=3):
    for _ in range(warm):
        fn()
    stream.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        a.record()
        for _ in range(iters):
            fn()
        b.record()
    stream.synchronize()
    return a.elapsed_time(b) / iters


def main():
    assert os.environ.get('NERFPP_DEFER_DW'), 'run with the probes library and NERFPP_DEFER_DW=1'
    dev = torch.device('cuda:0')
    n, S = 1024, 192
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    b = SyntheticKitti().random_batch(n, np.random.RandomState(0))
    ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
    far, fg_z, bg_z = ops.sample_coarse(ray_o, ray_d, T(b['min_depth']), S)
    lv = init_level_params(2)
    engA = ops.LevelEngine(lv[0].to(dev), precision=1)     # holds a finished forward + dX: the weight gradients' operands
    engB = ops.LevelEngine(lv[1].to(dev), precision=1)     # runs forwards / dX chains next to them
    ret = engA.forward(ray_o, ray_d, far, fg_z, bg_z, training=True)
    g_rgb, g_depth = torch.rand_like(ret['rgb']) * 1e-3, torch.rand_like(ret['depth']) * 1e-3
    engB.forward(ray_o, ray_d, far, fg_z, bg_z, training=True)
    torch.cuda.synchronize()

    def dw_on(stream):
        def f():
            with torch.cuda.stream(stream):
                engA.backward(g_rgb, g_depth, None, defer_reduce=True)       # (dX kernels: re-arm the deferred half)
        return f

    res = {'n_rays': n, 'S': S, 'census': {}, 'dw_alone_ms': {}, 'fwd_alone_ms': {}, 'bwd_alone_ms': {}, 'concurrent': []}
    plain = torch.cuda.Stream()
    res['census']['unmasked'] = census(plain)

    def run_dw(stream):
        # the deferred half alone: weight-gradient launches + slab sum (the dX part runs untimed on the plain stream before)
        ts = []
        for _ in range(6):
            with torch.cuda.stream(plain):
                engA.backward(g_rgb, g_depth, None, defer_reduce=True)
            plain.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                a.record()
                engA.reduce_grads()
                e.record()
            stream.synchronize()
            ts.append(a.elapsed_time(e))
        return float(np.median(ts[1:]))

    def fwd_fn(stream):
        def f():
            with torch.cuda.stream(stream):
                engB.forward(ray_o, ray_d, far, fg_z, bg_z, training=True)
        return f

    def bwd_fn(stream):
        def f():
            with torch.cuda.stream(stream):
                engB.backward(g_rgb, g_depth, None, defer_reduce=True)
        return f

    res['dw_alone_ms']['unmasked'] = run_dw(plain)
    res['fwd_alone_ms']['unmasked'] = timed(fwd_fn(plain), plain)
    res['bwd_alone_ms']['unmasked'] = timed(bwd_fn(plain), plain)
    for k in (32, 64, 96, 128, 192):
        for name, bits in layouts(k).items():
            tag = '%d_%s' % (k, name)
            s_dw = mask_stream(bits)
            comp = sorted(set(range(N_CU)) - set(bits))
            s_mlp = mask_stream(comp)
            res['census'][tag] = census(s_dw)
            res['census'][tag + '_complement'] = census(s_mlp)
            res['dw_alone_ms'][tag] = run_dw(s_dw)
            res['fwd_alone_ms']['%d_%s' % (N_CU - k, name)] = timed(fwd_fn(s_mlp), s_mlp)
            res['bwd_alone_ms']['%d_%s' % (N_CU - k, name)] = timed(bwd_fn(s_mlp), s_mlp)
            # concurrent: weight gradients of engine A on k CUs next to `reps` forwards (or dX chains) of engine B on the rest
            for what, fn in (('fwd', fwd_fn), ('bwd', bwd_fn)):
                for masked in (True, False):
                    sa, sb = (s_dw, s_mlp) if masked else (plain, torch.cuda.Stream())
                    walls = []
                    for _ in range(4):
                        with torch.cuda.stream(plain):
                            engA.backward(g_rgb, g_depth, None, defer_reduce=True)
                        torch.cuda.synchronize()
                        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        da, db = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        t0.record(plain)
                        sa.wait_event(t0)
                        sb.wait_event(t0)
                        with torch.cuda.stream(sa):
                            engA.reduce_grads()
                            da.record()
                        f = fn(sb)
                        f(); f()
                        db.record(sb)
                        plain.wait_event(da)
                        plain.wait_event(db)
                        t1.record(plain)
                        torch.cuda.synchronize()
                        walls.append((t0.elapsed_time(t1), t0.elapsed_time(da), t0.elapsed_time(db)))
                    w = np.median(np.array(walls[1:]), 0)
                    res['concurrent'].append({'k_dw': k, 'layout': name, 'with': '2x ' + what, 'masked': masked,
                                              'wall_ms': float(w[0]), 'dw_done_ms': float(w[1]), 'mlp_done_ms': float(w[2])})
            if not os.environ.get('CU_MASK_ALL_LAYOUTS') and name == 'low':
                pass
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
