"""GPU: the drop-in CLI end to end on a tiny synthetic scene -- args.txt, checkpoints with the
reference's key layout, resume, test-set render + metric files."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def test_train_cli_checkpoint_resume_and_eval(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from outdoor_nerf_depth_amd import ddp_train_nerf as T
    base = ['--expname', 'run', '--basedir', str(tmp_path), '--synthetic', '--synthetic_hw', '24,32',
            '--synthetic_frames', '20', '--cascade_samples', '64,128', '--use_depth', '--depth_loss_type', 'kl',
            '--depth_sup_type', 'mono_crop', '--lambda_depth', '0.1', '--sample_every', '2', '--world_size', '1',
            '--N_rand_override', '256', '--i_weights', '5', '--i_test', '5', '--testskip', '1', '--i_print', '1', '--host_sampling']
    args = T.config_parser().parse_args(base + ['--N_iters', '6'])
    T.validate_args(args)
    args.world_size = 1
    T.ddp_train_nerf(0, args)
    exp = tmp_path / 'run'
    assert (exp / 'args.txt').exists() and 'depth_loss_type = kl' in (exp / 'args.txt').read_text()
    ck = torch.load(exp / 'model_000005.pth', map_location='cpu', weights_only=False)
    assert list(ck.keys()) == ['net_0', 'optim_0', 'net_1', 'optim_1']
    assert 'module.nerf_net.bg_net.rgb_layers.2.bias' in ck['net_0']
    assert ck['net_0']['module.nerf_net.fg_net.base_layers.5.0.weight'].shape == (256, 319)
    assert len(ck['optim_1']['state']) == 48 and float(ck['optim_1']['state'][0]['step']) == 6
    rdir = exp / 'render_test_000005'
    assert (rdir / '000000.png').exists() and (rdir / 'psnr_000005.txt').exists() and (rdir / 'rmse_000005.txt').exists()
    psnr = [float(x) for x in (rdir / 'psnr_000005.txt').read_text().split()]
    assert np.isfinite(psnr).all() and len(psnr) == 3          # 2 test frames + mean
    # the per-image artefacts of the in-loop evaluation (ddp_train_nerf.py:549-600), for both test frames, nothing else
    want = {pre + '%06d.png' % i for i in (0, 1) for pre in ('', 'fg_', 'bg_', 'depth_', 'error_rgb_', 'absrel_')}
    want |= {'psnr_000005.txt', 'rmse_000005.txt', 'absrel_000005.txt'}
    assert set(os.listdir(rdir)) == want, set(os.listdir(rdir)) ^ want
    from PIL import Image
    for pre in ('error_rgb_', 'absrel_'):
        m = np.array(Image.open(rdir / (pre + '000000.png')))
        assert m.dtype == np.uint8 and m.ndim == 2 and m.min() == 0 and m.max() == 255      # min-max normalised grayscale
    for f in want:
        os.remove(rdir / f)                                     # the offline entry point below re-creates the directory's contents
    # offline evaluation entry point (ddp_test_nerf.py): same parser, newest checkpoint, metric files
    from outdoor_nerf_depth_amd import ddp_test_nerf as TT
    targs = T.config_parser().parse_args(base + ['--render_splits', 'test'])
    targs.world_size = 1
    TT.ddp_test_nerf(0, targs)
    tdir = exp / 'render_test_000005'
    assert set(os.listdir(tdir)) == want, set(os.listdir(tdir)) ^ want
    assert len((tdir / 'absrel_000005.txt').read_text().split()) == 3
    # "already trained" guard of the reference (:733-735)
    with pytest.raises(SystemExit) as e:
        T.train(base + ['--N_iters', '2'])
    assert e.value.code == 0
    # resume: start = step parsed from the newest checkpoint
    ck_path, start = T.find_latest_checkpoint(args)
    assert start == 5 and ck_path.endswith('model_000005.pth')
    from outdoor_nerf_depth_amd.trainer import NerfppTrainer
    tr = NerfppTrainer(torch.device('cuda:0'), use_depth=True, depth_loss_type='kl')
    T.load_checkpoint(ck_path, tr)
    assert tr.step_count == 6
    got = tr.engines[1].params.cpu()
    from outdoor_nerf_depth_amd.model import state_dict_from_flat
    assert torch.equal(state_dict_from_flat(got)['module.nerf_net.fg_net.sigma_layers.0.weight'],
                       ck['net_1']['module.nerf_net.fg_net.sigma_layers.0.weight'])


def test_train_cli_with_autoexposure(tmp_path):
    """--optim_autoexpo: train_images.json, autoexpo_params.* in net_m, their Adam
    entries after the 48 network tensors, resume."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import json
    from outdoor_nerf_depth_amd import ddp_train_nerf as T
    base = ['--expname', 'ae', '--basedir', str(tmp_path), '--synthetic', '--synthetic_hw', '24,32',
            '--synthetic_frames', '10', '--cascade_samples', '64,128', '--world_size', '1', '--N_rand_override', '128',
            '--i_weights', '4', '--i_print', '1', '--optim_autoexpo', '--lambda_autoexpo', '0.5']
    args = T.config_parser().parse_args(base + ['--N_iters', '5'])
    T.validate_args(args)
    args.world_size = 1
    T.ddp_train_nerf(0, args)
    exp = tmp_path / 'ae'
    names = json.loads((exp / 'train_images.json').read_text())
    ck = torch.load(exp / 'model_000004.pth', map_location='cpu', weights_only=False)
    ae_keys = [k for k in ck['net_1'] if 'autoexpo_params' in k]
    assert len(ae_keys) == len(names) and ae_keys[0] == 'module.autoexpo_params.train/rgb/000000-png'
    assert list(ck['net_0'].keys())[48] == ae_keys[0]                      # after the network tensors
    assert len(ck['optim_0']['param_groups'][0]['params']) == 48 + len(names)
    stepped = [i for i in range(len(names)) if 48 + i in ck['optim_0']['state']]
    assert 1 <= len(stepped) <= 5
    moved = [k for k in ae_keys if not torch.equal(ck['net_0'][k], torch.tensor([0.5, 0.]))]
    assert len(moved) == len(stepped)
    from outdoor_nerf_depth_amd.trainer import NerfppTrainer
    tr = NerfppTrainer(torch.device('cuda:0'), optim_autoexpo=True, img_names=names)
    T.load_checkpoint(str(exp / 'model_000004.pth'), tr)
    assert torch.equal(tr.autoexpo[1].params.cpu()[stepped[0]], ck['net_1'][ae_keys[stepped[0]]])
    assert float(tr.autoexpo[0].steps[stepped[0]]) >= 1
