#!/usr/bin/env python
"""The PropMLP forward of one proposal level (262 144 rows): mip360_prop_mlp_fm against the five launches it replaces.

    python tools/probes/mip360_prop_bench.py [--rows 262144] [--iters 20]      (MIP360_HIP_LIB selects a library variant)
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from outdoor_nerf_depth_amd import mip360 as M   # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--rows', type=int, default=262144)
    p.add_argument('--iters', type=int, default=20)
    a = p.parse_args()
    dev = torch.device('cuda:0')
    rows, W, ld = a.rows, 256, 768
    g = torch.Generator(device='cpu').manual_seed(0)
    bf = lambda t: t.to(dev).to(torch.bfloat16).contiguous()
    enc = torch.zeros(rows * ld, dtype=torch.bfloat16, device=dev)
    M.to_fm(bf(torch.randn(rows, 512, generator=g) * 0.7), out=enc, ld=ld, col0=W)
    ws = [bf(torch.randn(W, 512 if l == 0 else W, generator=g) * np.sqrt(2.0 / (512 if l == 0 else W))) for l in range(4)]
    w_fm = [M.to_fm(w) for w in ws]
    bs = [(torch.randn(W, generator=g) * 0.1).to(dev) for _ in range(4)]
    wd, bd = bf(torch.randn(1, W, generator=g) / 16), torch.zeros(1, device=dev)
    hs = [M.fm_buffer(rows, W, dev) for _ in range(4)]
    masks = [M.fm_mask_buffer(rows, W, dev) for _ in range(4)]
    density = torch.empty(rows, 1, device=dev)
    ldw = [512, W, W, W]

    def fused_train():
        M.prop_mlp_fm(enc, W, ld, rows, w_fm, ldw, bs, wd, bd, density, h=hs, masks=masks)

    def fused_infer():
        M.prop_mlp_fm(enc, W, ld, rows, w_fm, ldw, bs, wd, bd, density)

    def chain():
        x, c0, xl, xk = enc, W, ld, 512
        for l in range(4):
            M.linear_fm(x, w_fm[l], bs[l], 1, rows, W, xk, hs[l], masks[l], lda=xl, ldw=ldw[l], a_col0=c0)
            x, c0, xl, xk = hs[l], 0, W, W
        M._check(M.lib().mip360_rowdot_fm(M._stream(), rows, W, M._fm_ptr(hs[3], 0), W, M._p(wd), M._p(bd), 2, M.DENSITY_BIAS,
                                          M._p(density), 1), 'rowdot')

    gflop = rows * (512 * W + 3 * W * W + W) * 2 / 1e9
    for name, fn in (('fused_train', fused_train), ('fused_infer', fused_infer), ('five_launches', chain)):
        us = timeit(fn, a.iters)
        print('%-14s rows=%d : %8.1f us  %7.1f TFLOP/s' % (name, rows, us, gflop / us * 1e-3))


if __name__ == '__main__':
    main()
