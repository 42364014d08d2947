"""The view branch's two row-major weight-gradient launches (131 072 rows: h^T d_pre [128 x 3], view_in^T d_hz [288 x 128]) against
the number of row slices.  python tools/probes/mip360_view_dw_ksplit.py"""
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
m = 131072
for (n_in, n_out, ldz) in ((128, 3, 32), (288, 128, 128)):
    h = torch.randn(m, n_in, device=dev).to(torch.bfloat16)
    dz = (torch.randn(m, ldz, device=dev) * 0.1).to(torch.bfloat16)
    out, bias = torch.empty(n_in, n_out, device=dev), torch.empty(n_out, device=dev)
    for ks in (64, 128, 192, 256):
        buf = torch.empty(ks * (n_in * n_out + n_out), device=dev)
        t = timeit(lambda: M._check(M.lib().mip360_grad_weight_bf16(M._stream(), m, n_in, n_out, M._p(h), n_in, M._p(dz), ldz, ks, M._p(buf),
                                                                    M._p(out), n_out, 1.0, M._p(bias)), 'gw'))
        print('%3d x %3d, ksplit %3d: %6.1f us' % (n_in, n_out, ks, t))
