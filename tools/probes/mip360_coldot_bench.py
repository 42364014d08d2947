import os, sys, torch
sys.path.insert(0, '/root/repo')
from outdoor_nerf_depth_amd import mip360 as M
dev = torch.device('cuda:0')
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (m, n_in, ldz) in [(131072, 1024, 320), (262144, 256, 64), (131072, 1024, 8)]:
    h = torch.randn(m, n_in, device=dev).to(torch.bfloat16)
    dzb = (torch.randn(m, ldz, device=dev) * 0.1).to(torch.bfloat16)
    dz = dzb[:, ldz - 8:] if ldz > 8 else dzb
    out = torch.empty(n_in, 1, device=dev); bias = torch.empty(1, device=dev)
    scratch = [None, None]
    t = timeit(lambda: M._grad_weight(h, dz, n_in, 1, out, scratch, bias))
    dens = torch.empty(m, 1, device=dev)
    w = torch.randn(1, n_in, device=dev).to(torch.bfloat16); b = torch.zeros(1, device=dev)
    t2 = timeit(lambda: M.linear(h, w, b, act=2, act_param=-1.0, out_f32=dens, m=m, n=1, k=n_in))
    print('%d x %d ldz %d: coldot+reduce %.1f us (%.2f TB/s)   rowdot %.1f us (%.2f TB/s)' % (m, n_in, ldz, t, m*n_in*2/t/1e6, t2, m*n_in*2/t2/1e6))
