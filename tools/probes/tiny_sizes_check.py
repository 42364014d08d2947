"""bf16 backward (with its recomputing weight-gradient jobs) against the split-bf16 backward at tiny and odd sizes: relative L2 of
the whole gradient (bf16 grade: a few 1e-2), finiteness."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from outdoor_nerf_depth_amd import ops
from outdoor_nerf_depth_amd.model import init_level_params
from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
dev = torch.device('cuda:0')
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
for n, S in ((1, 2), (1, 64), (2, 33), (3, 256), (17, 31), (129, 64), (1024, 64), (2048, 192)):
    b = SyntheticKitti().random_batch(n, np.random.RandomState(n + S))
    far, fg, bg = ops.sample_coarse(T(b['ray_o']), T(b['ray_d']), T(b['min_depth']), S, rng=(5, 1))
    gs = {}
    for prec in (2, 1, 3, 12):
        eng = ops.LevelEngine(init_level_params(1)[0].to(dev), precision=prec)
        ret = eng.forward(T(b['ray_o']), T(b['ray_d']), far, fg, bg, training=True)
        g = torch.Generator(device=dev); g.manual_seed(1)
        g_rgb = torch.rand(ret['rgb'].shape, device=dev, generator=g) * 1e-3
        g_d = torch.rand(ret['depth'].shape, device=dev, generator=g) * 1e-3
        gs[prec] = eng.backward(g_rgb, g_d, None).clone()
    ref = gs[2]
    print(n, S, {p: (round(float((gs[p] - ref).norm() / ref.norm()), 4), bool(torch.isfinite(gs[p]).all())) for p in (1, 3, 12)}, flush=True)
