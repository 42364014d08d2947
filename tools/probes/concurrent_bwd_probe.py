import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from outdoor_nerf_depth_amd import _lib as L
from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
from outdoor_nerf_depth_amd.trainer import NerfppTrainer, batch_to_device
dev = torch.device('cuda:0'); scene = SyntheticKitti(); rng = np.random.RandomState(777)
steps, blocks = 60, 6
batches = [batch_to_device(scene.random_batch(1024, rng), dev) for _ in range(steps)]
mk = lambda: NerfppTrainer(dev, precision=L.PREC_BF16, use_depth=True, depth_loss_type='mse', lambda_depth=0.1, depth_scale=float(scene.depth_scale))
tr = {'concurrent': mk(), 'inline_bwd': mk()}
tr['inline_bwd'].concurrent_backward = False
times = {k: [] for k in tr}
for blk in range(blocks + 1):
    for k, t in tr.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for b in batches: t.train_step(b)
        t.flush(); torch.cuda.synchronize()
        if blk: times[k].append(1e3 * (time.perf_counter() - t0) / steps)
for k, v in times.items(): print(k, round(float(np.median(v)), 4))
