import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from outdoor_nerf_depth_amd import _lib as L
from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
from outdoor_nerf_depth_amd.trainer import NerfppTrainer, batch_to_device
dev = torch.device('cuda:0')
scene = SyntheticKitti(); rng = np.random.RandomState(777)
batches = [batch_to_device(scene.random_batch(1024, rng), dev) for _ in range(32)]
tr = NerfppTrainer(dev, precision=L.PREC_BF16, use_depth=True, depth_loss_type='mse', lambda_depth=0.1, depth_scale=float(scene.depth_scale))
def run(n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize(); ev[0].record()
    for i in range(n):
        tr.train_step(batches[i % 32]); ev[i + 1].record()
    tr.flush(); torch.cuda.synchronize()
    ms = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(n)])
    return [round(float(ms[a:b].mean()), 3) for a, b in ((0, 1), (1, 3), (3, 8), (8, 28), (28, 40))]
print('first  ', run(40))
print('again  ', run(40))
time.sleep(2.0)
print('sleep2s', run(40))
time.sleep(0.05)
print('sleep50ms', run(40))
x = torch.empty(1 << 28, device=dev)
for _ in range(50): x.fill_(1.0)
torch.cuda.synchronize()
print('after fills', run(40))
