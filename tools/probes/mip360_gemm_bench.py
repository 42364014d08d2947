"""Time (and check) the MipNeRF-360 dense-layer kernels on the NerfMLP / PropMLP shapes, next to torch (hipBLASLt).

    python tools/probes/mip360_gemm_bench.py [--check]
"""
import os
import sys

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


def main():
    check = '--check' in sys.argv
    torch.manual_seed(0)
    for (m, n, k) in [(131072, 1024, 1024), (131072, 1024, 1536), (262144, 256, 256), (262144, 256, 512)]:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        if '--zeros' in sys.argv:
            a.zero_()
        if '--small' in sys.argv:
            a.mul_(2.0 ** -20)
        w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
        b = torch.randn(n, device=dev)
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        mask, ld = M.relu_mask_buffer(m, n, dev)
        fl = 2.0 * m * n * k
        res = {}
        res['relu'] = timeit(lambda: M.linear(a, w, b, act=1, out_bf16=out))
        res['relu+mask'] = timeit(lambda: M.linear_relu_mask(a, w, b, out, mask, ld))
        res['masked dX'] = timeit(lambda: M.linear_masked(a, w, out, mask, ld))
        res['torch'] = timeit(lambda: torch.nn.functional.linear(a, w, b.to(torch.bfloat16)))
        print('%7d x %4d x %4d : ' % (m, n, k) + '  '.join('%s %.0f us (%.0f TF/s)' % (key, v, fl / v / 1e6) for key, v in res.items()))
        if check:
            M.linear(a, w, b, act=1, out_bf16=out)
            ref = torch.relu(a[:4096].float() @ w.float().t() + b)
            err = (out[:4096].float() - ref).abs().max().item()
            print('   max abs err vs float32 matmul (first 4096 rows): %.4f' % err)
            assert err < 0.05


if __name__ == '__main__':
    main()
