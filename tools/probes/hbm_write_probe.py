#!/usr/bin/env python
"""HBM write / read / copy ceilings with plain torch ops (fill_, sum, copy_) on 1-4 GiB buffers:
the practical rooflines the store-heavy training kernels are compared with in DESIGN.md."""
import torch

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3

for gib in (1, 4):
    n = gib << 28
    x = torch.empty(n, dtype=torch.float32, device='cuda')
    y = torch.empty(n, dtype=torch.float32, device='cuda')
    by = n * 4
    print('%d GiB  fill  %.2f TB/s' % (gib, by / t(lambda: x.fill_(1.0)) / 1e12))
    print('%d GiB  read  %.2f TB/s (sum)' % (gib, by / t(lambda: x.sum()) / 1e12))
    print('%d GiB  copy  %.2f TB/s (read+write bytes)' % (gib, 2 * by / t(lambda: y.copy_(x)) / 1e12))
    xb = x.view(torch.bfloat16)
    print('%d GiB  zero_ %.2f TB/s' % (gib, by / t(lambda: xb.zero_()) / 1e12))
