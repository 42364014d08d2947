"""Dense-layer time against K / tile count: separates the per-K-step cost from the per-tile constant."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from outdoor_nerf_depth_amd import mip360 as M                                  # noqa: E402
dev = torch.device('cuda:0')

def timeit(fn, reps=10):
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
for (m, n, k) in [(131072, 1024, 512), (131072, 1024, 1024), (131072, 1024, 2048), (131072, 1024, 4096), (65536, 1024, 1024),
                  (131072, 512, 1024), (16384, 1024, 1024), (16384, 1024, 4096)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device=dev)
    out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    mask, ld = M.relu_mask_buffer(m, n, dev)
    t = timeit(lambda: M.linear_relu_mask(a, w, b, out, mask, ld))
    tiles = (m // 256) * (n // 256)
    print('%7d x %4d x %4d: %7.1f us  %5.0f TF/s  tiles/CU %.2f  us per tile-round %.1f  per 32-k step %.3f' % (
        m, n, k, t, 2.0 * m * n * k / t / 1e6, tiles / 256, t / (tiles / 256), t / (tiles / 256) / (k / 32)))
