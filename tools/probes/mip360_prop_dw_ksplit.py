"""PropMLP weight-gradient GEMM (262 144 x 256 x 256, fm operands): time against the number of row slices (= workgroups; the slab
traffic is ksplit x 256 KiB written + read).  python tools/probes/mip360_prop_dw_ksplit.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from outdoor_nerf_depth_amd import mip360 as M                                  # noqa: E402
dev = torch.device('cuda:0')


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


torch.manual_seed(0)
m, W = 262144, 256
h = M.to_fm(torch.randn(m, W, device=dev).to(torch.bfloat16))
dz = M.to_fm((torch.randn(m, W, device=dev) * 0.1).to(torch.bfloat16))
out, bias = torch.empty(W, W, device=dev), torch.empty(W, device=dev)
for ks in (256, 128, 64, 48, 32, 16):
    buf = torch.empty(ks * (W * W + W), device=dev)

    def gemm():
        M._check(M.lib().mip360_grad_weight_fm(M._stream(), m, W, W, M._p(h), W, M._p(dz), W, ks, M._p(buf), None, W, 1.0, M._p(bias)), 'gw')

    def both():
        gemm()
        M._check(M.lib().mip360_grad_weight_reduce(M._stream(), W, W, W, ks, M._p(buf), M._p(out), W, 1.0, M._p(bias)), 'red')
    print('ksplit %3d (%3d workgroups, %3d chunks per slice): GEMM %6.1f us, + slab sum %6.1f us' % (ks, ks, m // 32 // ks, timeit(gemm), timeit(both)))
