import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from outdoor_nerf_depth_amd.trainer import NerfppTrainer, batch_to_device
from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
from outdoor_nerf_depth_amd import _lib as L
dev = torch.device('cuda:0')
scene = SyntheticKitti(depth_sup_type='gt'); rng = np.random.RandomState(1)
bs = [batch_to_device(scene.random_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 1024, rng), dev) for _ in range(8)]
tr = NerfppTrainer(dev, precision=L.PREC_BF16, use_depth=True, depth_loss_type='mse', lambda_depth=0.1)
for i in range(10): tr.train_step(bs[i % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(300): tr.train_step(bs[i % 8])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue %.3f ms/step, total %.3f ms/step' % ((t1 - t0) / 300 * 1e3, (t2 - t0) / 300 * 1e3))
