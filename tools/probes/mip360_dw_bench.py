"""Time the MipNeRF-360 weight-gradient GEMM on the NerfMLP / PropMLP layer shapes.  python tools/probes/mip360_dw_bench.py"""
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
for (m, n_in, n_out) in [(131072, 1024, 1024), (131072, 1536, 1024), (131072, 512, 1024), (262144, 256, 256), (262144, 768, 256)]:
    h = torch.randn(m, n_in, device=dev).to(torch.bfloat16)
    dz = (torch.randn(m, n_out, device=dev) * 0.1).to(torch.bfloat16)
    out = torch.empty(n_in, n_out, device=dev)
    bias = torch.empty(n_out, device=dev)
    scratch = [None, None]
    t = timeit(lambda: M._grad_weight(h, dz, n_in, n_out, out, scratch, bias))
    print('%7d x %4d x %4d: %7.1f us  %5.0f TF/s (GEMM + slab sums)' % (m, n_in, n_out, t, 2.0 * m * n_in * n_out / t / 1e6))
