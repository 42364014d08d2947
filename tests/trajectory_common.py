"""Deterministic inputs of the training-trajectory fixture (tests/golden/trajectory.npz), shared by the generator
(tests/golden/make_golden.py: gen_trajectory, runs the imported reference in the build container) and by the tests that
replay the same batches and uniforms through the oracle (CPU) and the HIP trainer (GPU).  numpy only: MT19937 streams
are stable across machines and versions, so nothing but seeds has to be stored.

Scene = BASELINE config 1: one 64x64 frame of the KITTI-shaped sequence, --cascade_samples 32,64, N_rand 256
(nerf-methods/nerfplusplus/ddp_train_nerf.py:417-431: one random frame -- there is only one -- and N_rand pixels
without replacement per step).
"""
import numpy as np

H = W = 64
CASCADE = (32, 64)
N_RAND = 256
N_STEPS = 200
LOG_EVERY = 25
LAMBDA_DEPTH = 0.1
MODES = ('rgbonly', 'mse', 'l1', 'kl')
# mode -> depth prior of the scene (BASELINE configs: gt + mse | stereo_crop + l1 | mono_crop + kl); rgbonly ignores it
DEPTH_SUP_TYPE = {'rgbonly': 'gt', 'mse': 'gt', 'l1': 'stereo_crop', 'kl': 'mono_crop'}
DEPTH_SIGMA = 0.01                  # --depth_sigma default; the KL term uses depth_sigma * depth_scale (ddp_train_nerf.py:489)
BATCH_SEED, UNIFORM_SEED = 41000, 52000


def sampler(mode='mse'):
    from outdoor_nerf_depth_amd.data_loader_split import synthetic_ray_samplers
    return synthetic_ray_samplers('train', 1, DEPTH_SUP_TYPE[mode], 1, H, W)[0]


def step_batch(smp, step):
    """ray batch of training step `step` (1-based): N_RAND pixels of the frame without replacement"""
    rs = np.random.RandomState(BATCH_SEED + step)
    idx = rs.choice(H * W, size=(N_RAND,), replace=False)
    b = smp._select(idx)
    return {k: np.ascontiguousarray(v, np.float32) for k, v in b.items() if isinstance(v, np.ndarray)}


def step_uniforms(step):
    """the four uniform tensors of a step in the order the reference consumes torch's RNG:
    rand_like(fg_z), rand_like(bg_z) (perturb_samples, :444,449), sample_pdf u for fg, then bg (:455,463)"""
    rs = np.random.RandomState(UNIFORM_SEED + step)
    S0, S1 = CASCADE
    return dict(t_fg=rs.rand(N_RAND, S0).astype(np.float32), t_bg=rs.rand(N_RAND, S0).astype(np.float32),
                u_fg=rs.rand(N_RAND, S1).astype(np.float32), u_bg=rs.rand(N_RAND, S1).astype(np.float32))


def psnr(mse):
    """utils.py:31 mse2psnr (TINY_NUMBER = 1e-6, utils.py:8)"""
    return -10.0 * np.log(mse + 1e-6) / np.log(10.0)
