"""Dump the flat bf16-mode gradient of one level on a fixed batch (for bit-level A/B of two library builds: run once per
NERFPP_HIP_LIB and compare the files)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from outdoor_nerf_depth_amd import ops
from outdoor_nerf_depth_amd.model import init_level_params
from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
dev = torch.device('cuda:0')
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
out = {}
for n, S in ((1024, 192), (37, 64)):
    b = SyntheticKitti().random_batch(n, np.random.RandomState(n))
    far, fg, bg = ops.sample_coarse(T(b['ray_o']), T(b['ray_d']), T(b['min_depth']), S, rng=(5, 1))
    for prec in (1, 3, 12):
        eng = ops.LevelEngine(init_level_params(1)[0].to(dev), precision=prec)
        ret = eng.forward(T(b['ray_o']), T(b['ray_d']), far, fg, bg, training=True)
        g = torch.Generator(device=dev); g.manual_seed(1)
        g_rgb = torch.rand(ret['rgb'].shape, device=dev, generator=g) * 1e-3
        g_d = torch.rand(ret['depth'].shape, device=dev, generator=g) * 1e-3
        out['g_%d_%d_%d' % (n, S, prec)] = eng.backward(g_rgb, g_d, None).cpu().numpy()
np.savez(sys.argv[1], **out)
