"""GPU, round 4: schedule options of NerfppTrainer change no bit; the reference-generated training trajectory.

* `concurrent_backward` (level 0's backward on its own stream under level 1's forward, trainer.py) against the inline
  schedule: identical parameters, Adam moments and logged scalars after several steps -- in one process and with two
  ranks over gloo sharing cuda:0.
* tests/golden/trajectory.npz (make_golden.py: gen_trajectory, the imported reference on the BASELINE config-1 scene):
  the HIP trainer on the same batches and uniforms, PSNR gates of VERDICT r03 item 3.
"""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden')


def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def T(x, d=None):
    return torch.from_numpy(np.ascontiguousarray(x)).to(d or dev())


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batches(rank, n, steps):
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    scene = SyntheticKitti(depth_sup_type='gt')
    rng = np.random.RandomState((rank + 1) * 777)
    return [scene.random_batch(n, rng) for _ in range(steps)]


def _run(concurrent, precision, world=1, rank=0, steps=5, n=96):
    from outdoor_nerf_depth_amd.trainer import NerfppTrainer
    d = torch.device('cuda:0')
    tr = NerfppTrainer(d, precision=precision, use_depth=True, depth_loss_type='mse', lambda_depth=0.1,
                       world_size=world, seed=(rank + 1) * 777)
    tr.concurrent_backward = concurrent
    sc = []
    for b in _batches(rank, n, steps):
        out = tr.train_step({k: T(v, d) for k, v in b.items() if isinstance(v, np.ndarray)})
        sc.append([s.clone() for s in out])
    tr.flush()
    torch.cuda.synchronize()
    return dict(params=np.stack([e.params.cpu().numpy() for e in tr.engines]),
                m=np.stack([x.cpu().numpy() for x in tr.exp_avg]), v=np.stack([x.cpu().numpy() for x in tr.exp_avg_sq]),
                scalars=np.stack([np.stack([s.cpu().numpy() for s in step]) for step in sc]))


@pytest.mark.parametrize('precision', [1, 2])
def test_concurrent_backward_changes_no_bit(precision):
    dev()
    a = _run(True, precision)
    b = _run(False, precision)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert np.isfinite(a['params']).all() and np.isfinite(a['scalars'][..., :2]).all()


def _worker(rank, world, port, out_dir, concurrent):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    r = _run(concurrent, 1, world=world, rank=rank, steps=3, n=64)
    np.savez(os.path.join(out_dir, 'r%d_c%d.npz' % (rank, int(concurrent))), **r)
    dist.barrier()
    dist.destroy_process_group()


def test_concurrent_backward_two_ranks_gloo(tmp_path):
    dev()
    import torch.multiprocessing as mp
    for concurrent in (True, False):
        mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), concurrent), nprocs=2, join=True)
    for rank in (0, 1):
        a = np.load(tmp_path / ('r%d_c1.npz' % rank))
        b = np.load(tmp_path / ('r%d_c0.npz' % rank))
        for k in ('params', 'm', 'v', 'scalars'):
            np.testing.assert_array_equal(a[k], b[k], err_msg='rank %d %s' % (rank, k))
    np.testing.assert_array_equal(np.load(tmp_path / 'r0_c1.npz')['params'], np.load(tmp_path / 'r1_c1.npz')['params'])
