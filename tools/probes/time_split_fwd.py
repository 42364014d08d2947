#!/usr/bin/env python
"""Launch time of the split-bf16 training forward (HIP events over 200 launches after 30 warm-up launches, the bench batch:
1024 rays x 192 samples), for A/B runs of library variants (NERFPP_HIP_LIB):  python tools/probes/time_split_fwd.py [--bwd]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from outdoor_nerf_depth_amd import ops, _lib as L
from outdoor_nerf_depth_amd.model import init_level_params
from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
dev = torch.device('cuda:0')
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
b = SyntheticKitti().random_batch(1024, np.random.RandomState(0))
ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
far, fg_z, bg_z = ops.sample_coarse(ray_o, ray_d, T(b['min_depth']), 192)
eng = ops.LevelEngine(init_level_params(1)[0].to(dev), precision=(L.PREC_BF16 if '--bf16' in sys.argv else L.PREC_SPLIT_BF16))
bwd = '--bwd' in sys.argv
def step():
    ret = eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=True)
    if bwd:
        eng.backward(torch.full_like(ret['rgb'], 1e-3), torch.full_like(ret['depth'], 1e-3), None)
for _ in range(30):
    step()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        step()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 40)
print('%s ms per %s: median %.4f  all %s' % (os.path.basename(os.environ.get('NERFPP_HIP_LIB', 'stock')), 'fwd+bwd' if bwd else 'training forward', np.median(ts), [round(t, 4) for t in ts]))
