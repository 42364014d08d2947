import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from outdoor_nerf_depth_amd import ops
dev = torch.device('cuda:0')
n, S0, S1 = 1024, 64, 128
z = torch.sort(torch.rand(n, S0, device=dev), dim=1)[0]
w = torch.rand(n, S0, device=dev)
u = torch.rand(n, S1, device=dev)
def timeit(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print('rng      %.1f us' % timeit(lambda: ops.sample_fine_pair(z, w, z, w, S1, rng=(777, 3))))
print('u given  %.1f us' % timeit(lambda: ops.sample_fine_pair(z, w, z, w, S1, u_fg=u, u_bg=u)))
print('det      %.1f us' % timeit(lambda: ops.sample_fine_pair(z, w, z, w, S1, det=True)))
