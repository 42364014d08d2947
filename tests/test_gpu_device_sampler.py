"""GPU: §8 f-1 -- ray batches generated on the device from GPU-resident frames match the reference's
host-side sampler (get_rays_single_image formula, same dict keys, gathers exact)."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def test_device_batches_match_host_sampler(golden):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from outdoor_nerf_depth_amd.data_loader_split import synthetic_ray_samplers
    from outdoor_nerf_depth_amd.device_sampler import DeviceRaySamplers
    samplers = synthetic_ray_samplers('train', skip=3, depth_sup_type='mono_crop', n_frames=20, H=30, W=44)
    for s_ in samplers:                          # make the ground truth differ from the prior (full_keys check below)
        s_.depth_gt = (s_.depth_sup * np.float32(1.25)).astype(np.float32)
    dev = torch.device('cuda:0')
    ds = DeviceRaySamplers(samplers, dev)
    assert ds.n_frames == len(samplers) and (ds.H, ds.W) == (30, 44)
    rs = np.random.RandomState(0)
    for frame in (0, len(samplers) - 1):
        pix = rs.choice(30 * 44, size=257, replace=False)
        out = ds.gather(frame, torch.from_numpy(pix).to(dev))
        host = samplers[frame]
        np.testing.assert_allclose(out['ray_d'].cpu().numpy(), host.rays_d[pix], rtol=2e-6, atol=1e-7)
        np.testing.assert_array_equal(out['ray_o'].cpu().numpy(), host.rays_o[pix])
        np.testing.assert_array_equal(out['rgb'].cpu().numpy(), host.img[pix])
        np.testing.assert_array_equal(out['depth_sup'].cpu().numpy(), host.depth_sup[pix])
        assert (out['min_depth'].cpu().numpy() == np.float32(1e-4)).all()
    # the reference's own 4x6 vector (tests/golden/rays.npz)
    g = golden('rays')
    from outdoor_nerf_depth_amd.data_loader_split import RaySamplerSingleImage
    one = RaySamplerSingleImage(4, 6, g['K'], g['c2w'], img=np.zeros((24, 3), np.float32))
    d1 = DeviceRaySamplers([one], dev)
    out = d1.gather(0, torch.arange(24, device=dev))
    np.testing.assert_allclose(out['ray_d'].cpu().numpy(), g['rays_d'], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(out['ray_o'].cpu().numpy(), g['rays_o'], rtol=0, atol=0)
    # random_sample: distinct pixels, right shapes and keys
    np.random.seed(3)
    b = ds.random_sample(128)
    assert set(b.keys()) >= {'ray_o', 'ray_d', 'rgb', 'min_depth', 'depth_sup', 'frame'} and 'depth_gt' not in b
    assert b['ray_d'].shape == (128, 3) and 0 <= b['frame'] < ds.n_frames
    # full_keys: depth_gt is the frame's GROUND-TRUTH map (not the mono_crop prior), mask as the host sampler has it
    pix = torch.from_numpy(rs.choice(30 * 44, size=64, replace=False)).to(dev)
    full = ds.gather(1, pix, full_keys=True)
    host = samplers[1]
    np.testing.assert_array_equal(full['depth_gt'].cpu().numpy(), host.depth_gt[pix.cpu().numpy()])
    assert not np.array_equal(host.depth_gt, host.depth_sup) and full['mask'] is None


def test_train_cli_with_device_sampling(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from outdoor_nerf_depth_amd import ddp_train_nerf as T
    args = T.config_parser().parse_args(
        ['--expname', 'dev', '--basedir', str(tmp_path), '--synthetic', '--synthetic_hw', '24,32', '--synthetic_frames',
         '12', '--cascade_samples', '64,128', '--use_depth', '--depth_loss_type', 'l1', '--depth_sup_type', 'stereo_crop',
         '--lambda_depth', '1', '--world_size', '1', '--N_rand_override', '128', '--N_iters', '4', '--i_weights', '3',
         '--i_print', '1', '--device_sampling', '--precision', 'bf16'])
    T.validate_args(args)
    args.world_size = 1
    T.ddp_train_nerf(0, args)
    assert (tmp_path / 'dev' / 'model_000003.pth').exists()
