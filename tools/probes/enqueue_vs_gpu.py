import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from outdoor_nerf_depth_amd import _lib as L
from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
from outdoor_nerf_depth_amd.trainer import NerfppTrainer, batch_to_device
dev = torch.device('cuda:0')
scene = SyntheticKitti()
rng = np.random.RandomState(777)
batches = [batch_to_device(scene.random_batch(1024, rng), dev) for _ in range(100)]
tr = NerfppTrainer(dev, precision=L.PREC_BF16, use_depth=True, depth_loss_type='mse', lambda_depth=0.1, depth_scale=float(scene.depth_scale))
for b in batches[:30]:
    tr.train_step(b)
tr.flush(); torch.cuda.synchronize()
t0 = time.perf_counter()
for b in batches:
    tr.train_step(b)
t1 = time.perf_counter()
tr.flush(); torch.cuda.synchronize()
t2 = time.perf_counter()
print('nerfpp: enqueue %.3f ms/step, total %.3f ms/step' % (1e3 * (t1 - t0) / 100, 1e3 * (t2 - t0) / 100))
