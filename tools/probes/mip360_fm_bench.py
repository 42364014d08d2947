"""Check and time the fragment-major dense-layer kernel (csrc/mip360_fm.hip) against torch and the row-major kernels.

    python tools/probes/mip360_fm_bench.py [--check-only] [--reps 20]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from outdoor_nerf_depth_amd import mip360 as M                                  # noqa: E402

dev = torch.device('cuda:0')
L = M.lib()
st, p = M._stream, M._p


def to_fm(x, ld=None, col0=0, out=None):
    rows, cols = x.shape
    ld = cols if ld is None else ld
    if out is None:
        out = torch.zeros(rows * ld, dtype=torch.bfloat16, device=dev)
    M._check(L.mip360_to_fm(st(), rows, cols, p(x), x.stride(0), p(out), ld, col0), 'to_fm')
    return out


def from_fm(x, rows, cols, ld=None, col0=0):
    ld = cols if ld is None else ld
    out = torch.empty(rows, cols, dtype=torch.bfloat16, device=dev)
    M._check(L.mip360_from_fm(st(), rows, cols, p(x), ld, col0, p(out), cols), 'from_fm')
    return out


def linear_fm(a, w, bias, act, m, n, k, out, mask, lda=None, ldw=None, ldc=None):
    M._check(L.mip360_linear_fm(st(), m, n, k, p(a), lda or k, p(w), ldw or k, p(bias), act, p(out), ldc or n, p(mask)), 'linear_fm')


def timeit(fn, reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def check(m, n, k, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    a = torch.randn(m, k, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev, generator=g) / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device=dev, generator=g)
    rt = from_fm(to_fm(a), m, k)
    assert torch.equal(rt, a), 'to_fm / from_fm round trip'
    a_fm, w_fm = to_fm(a), to_fm(w)
    out = torch.zeros(m * n, dtype=torch.bfloat16, device=dev)
    mask = torch.zeros(L.mip360_fm_mask_bytes(m, n), dtype=torch.uint8, device=dev)
    ref = a.float() @ w.float().t() + b
    linear_fm(a_fm, w_fm, b, 0, m, n, k, out, mask)
    got = from_fm(out, m, n).float()
    e0 = (got - ref).abs().max().item()
    linear_fm(a_fm, w_fm, b, 1, m, n, k, out, mask)
    got = from_fm(out, m, n).float()
    kept = got != 0                                             # the kernel's own ReLU pattern (|ref| ~ 0 can round either way)
    e1 = (got - torch.relu(ref)).abs().max().item()
    # dX form: C = (A W^T) masked by the bits the relu call wrote
    linear_fm(a_fm, w_fm, None, 2, m, n, k, out, mask)
    got2 = from_fm(out, m, n).float()
    ref2 = (a.float() @ w.float().t()) * kept
    e2 = (got2 - ref2).abs().max().item()
    print('check %6d x %4d x %4d : max abs err  bias %.4f  relu %.4f  masked %.4f' % (m, n, k, e0, e1, e2))
    assert max(e0, e1, e2) < 0.06, (e0, e1, e2)


def main():
    reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 20
    for shape in [(256, 256, 160), (512, 256, 256), (1024, 512, 1024), (4096, 1024, 1536), (131072, 1024, 1024), (262144, 256, 512)]:
        if '--time-only' in sys.argv:                           # (component-removal builds compute garbage)
            break
        check(*shape)
        check(*shape, seed=1)
    if '--check-only' in sys.argv:
        return
    torch.manual_seed(0)
    for (m, n, k) in [(131072, 1024, 1024), (131072, 1024, 1536), (131072, 1024, 512), (262144, 256, 256), (262144, 256, 512)]:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
        b = torch.randn(n, device=dev)
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        mask_rm, ld = M.relu_mask_buffer(m, n, dev)
        a_fm, w_fm = to_fm(a), to_fm(w)
        out_fm = torch.empty(m * n, dtype=torch.bfloat16, device=dev)
        mask = torch.zeros(L.mip360_fm_mask_bytes(m, n), dtype=torch.uint8, device=dev)
        fl = 2.0 * m * n * k
        res = {}
        res['rm relu+mask'] = timeit(lambda: M.linear_relu_mask(a, w, b, out, mask_rm, ld), reps)
        res['rm masked'] = timeit(lambda: M.linear_masked(a, w, out, mask_rm, ld), reps)
        res['fm bias'] = timeit(lambda: linear_fm(a_fm, w_fm, b, 0, m, n, k, out_fm, mask), reps)
        res['fm relu+mask'] = timeit(lambda: linear_fm(a_fm, w_fm, b, 1, m, n, k, out_fm, mask), reps)
        res['fm masked'] = timeit(lambda: linear_fm(a_fm, w_fm, None, 2, m, n, k, out_fm, mask), reps)
        res['torch'] = timeit(lambda: torch.nn.functional.linear(a, w, b.to(torch.bfloat16)), reps)
        print('%7d x %4d x %4d : ' % (m, n, k) + '  '.join('%s %.0f us (%.0f TF/s)' % (key, v, fl / v / 1e6) for key, v in res.items()))




def check_dw(m, n_in, n_out, ksplit, seed=0, time_it=False, reps=10):
    g = torch.Generator(device=dev).manual_seed(seed)
    h = torch.randn(m, n_in, device=dev, generator=g).to(torch.bfloat16)
    dz = torch.randn(m, n_out, device=dev, generator=g).to(torch.bfloat16)
    h_fm, dz_fm = to_fm(h), to_fm(dz)
    slabs = torch.empty(ksplit * (n_in * n_out + n_out), device=dev)
    out, bias = torch.empty(n_in, n_out, device=dev), torch.empty(n_out, device=dev)
    M._check(L.mip360_grad_weight_fm(st(), m, n_in, n_out, p(h_fm), n_in, p(dz_fm), n_out, ksplit, p(slabs), p(out), n_out, 1.0, p(bias)),
             'grad_weight_fm')
    rows = min(m, 16384)
    if rows == m:
        ref = h.float().t() @ dz.float()
        err = ((out - ref).abs().max() / ref.abs().max()).item()
        eb = ((bias - dz.float().sum(0)).abs().max() / dz.float().sum(0).abs().max()).item()
    else:                                                       # against the row-major kernel
        out2, bias2 = torch.empty_like(out), torch.empty_like(bias)
        M._check(L.mip360_grad_weight_bf16(st(), m, n_in, n_out, p(h), n_in, p(dz), n_out, ksplit, p(slabs), p(out2), n_out, 1.0, p(bias2)),
                 'grad_weight_bf16')
        err = ((out - out2).abs().max() / out2.abs().max()).item()
        eb = ((bias - bias2).abs().max() / bias2.abs().max()).item()
    line = 'dW check %6d x %4d x %4d ksplit %3d : rel err kernel %.2e bias %.2e' % (m, n_in, n_out, ksplit, err, eb)
    if time_it:
        t_fm = timeit(lambda: L.mip360_grad_weight_fm(st(), m, n_in, n_out, p(h_fm), n_in, p(dz_fm), n_out, ksplit, p(slabs), None, n_out, 1.0,
                                                      p(bias)), reps)
        t_rm = timeit(lambda: L.mip360_grad_weight_bf16(st(), m, n_in, n_out, p(h), n_in, p(dz), n_out, ksplit, p(slabs), None, n_out, 1.0,
                                                        p(bias)), reps)
        fl = 2.0 * m * n_in * n_out
        line += '   fm %.0f us (%.0f TF/s)  rm %.0f us (%.0f TF/s)' % (t_fm, fl / t_fm / 1e6, t_rm, fl / t_rm / 1e6)
    print(line)
    assert err < 2e-3 and eb < 2e-3, (err, eb)


def main_dw():
    check_dw(256, 256, 256, 1)
    check_dw(4096, 512, 256, 8)
    check_dw(8192, 1024, 1024, 16, seed=1)
    check_dw(131072, 1024, 1024, 16, time_it=True)
    check_dw(131072, 1536, 1024, 8, time_it=True)
    check_dw(131072, 512, 1024, 32, time_it=True)
    check_dw(131072, 1024, 256, 64, time_it=True)
    check_dw(262144, 256, 256, 256, time_it=True)
    check_dw(262144, 512, 256, 128, time_it=True)


if __name__ == '__main__' and '--dw' in sys.argv:
    main_dw()

if __name__ == '__main__' and '--dw' not in sys.argv:
    main()
