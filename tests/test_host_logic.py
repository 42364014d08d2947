"""CPU tests of the host side: CLI surface, parameter/checkpoint layout, dataset reader, and the
N > 1 path (world_size 2, gloo)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from oracle import nerfpp_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from outdoor_nerf_depth_amd import model as M
from outdoor_nerf_depth_amd import dist_utils as D
from outdoor_nerf_depth_amd import ddp_train_nerf as T
from outdoor_nerf_depth_amd import data_loader_split as DL

# defaults of the reference parser (nerf-methods/nerfplusplus/ddp_train_nerf.py:657-727)
REF_DEFAULTS = dict(basedir='./logs/', datadir=None, scene=None, testskip=8, trainskip=1, netdepth=8, netwidth=256,
                    use_viewdirs=False, no_reload=False, ckpt_path=None, N_rand=2048, chunk_size=8192,
                    N_iters=250001, render_splits='test', cascade_level=2, cascade_samples='64,64', world_size=-1,
                    optim_autoexpo=False, lambda_autoexpo=1., lrate=5e-4, lrate_decay_factor=0.1,
                    lrate_decay_steps=5000, det=False, max_freq_log2=10, max_freq_log2_viewdirs=4,
                    load_min_depth=False, i_print=100, i_img=500, i_weights=10000, use_depth=False,
                    lambda_depth=1.0, depth_loss_type='mse', depth_sup_type='gt', depth_sigma=0.01, port=12345)


def test_cli_flags_and_defaults_match_reference():
    args = T.config_parser().parse_args(['--expname', 'x'])
    for k, v in REF_DEFAULTS.items():
        assert getattr(args, k) == v, k
    a = T.config_parser().parse_args(['--expname', 'x', '--sample_every', '8', '--use_depth', '--depth_loss_type',
                                      'kl', '--depth_sup_type', 'mono_crop', '--lambda_depth', '0.1'])
    assert a.trainskip == 8 and a.use_depth and a.depth_loss_type == 'kl' and a.depth_sup_type == 'mono_crop'
    with pytest.raises(SystemExit):
        T.config_parser().parse_args(['--depth_loss_type', 'huber'])


def test_config_file_then_command_line_override(tmp_path):
    cfg = tmp_path / 'kitti.txt'
    cfg.write_text('### INPUT\nscene = seq00\ndepth_sup_type = gt\nexpname = debug_only\ntrainskip = 2\n'
                   'config = None\nckpt_path = None\nno_reload = False\nuse_depth = False\nlambda_depth = 1\n'
                   'N_rand = 1024\nlrate = 0.0005\ncascade_samples = 64,128\nuse_viewdirs = True\ni_weights = 10000\n')
    a = T.config_parser().parse_args(['--config', str(cfg), '--trainskip', '4', '--use_depth', '--expname', 'run1'])
    assert a.scene == 'seq00' and a.cascade_samples == '64,128' and a.use_viewdirs and a.N_rand == 1024
    assert a.trainskip == 4 and a.use_depth and a.expname == 'run1' and a.ckpt_path is None
    T.validate_args(a)
    for bad in (['--netwidth', '128'], ['--use_depth', '--depth_loss_type', 'nll']):
        with pytest.raises(SystemExit):
            T.validate_args(T.config_parser().parse_args(['--expname', 'x'] + bad))


def test_parameter_layout_and_init_match_reference(golden):
    g = golden('params_seed777')
    levels = M.init_level_params(2)
    specs = M.level_param_specs()
    assert [k for k, _ in specs] == ['nerf_net.' + k for k in O.param_order()]
    assert sum(int(np.prod(s)) for _, s in specs) == 1202440
    for m, flat in enumerate(levels):
        sd = M.state_dict_from_flat(flat)
        assert list(sd.keys())[0] == 'module.nerf_net.fg_net.base_layers.0.0.weight'      # DDP prefix, :646
        for k in O.param_order():
            v = sd['module.nerf_net.' + k].numpy()
            np.testing.assert_array_equal(v.reshape(-1)[g['L%d.%s.idx' % (m, k)]], g['L%d.%s.val' % (m, k)])


def test_checkpoint_round_trip_with_torch_adam(tmp_path):
    flat = M.init_level_params(1)[0]
    sd = M.state_dict_from_flat(flat)
    # a torch module with the reference's parameter shapes loads our optimiser state
    params = [torch.nn.Parameter(v.clone()) for v in sd.values()]
    opt = torch.optim.Adam(params, lr=5e-4)
    ea, eas = torch.rand_like(flat), torch.rand_like(flat)
    opt.load_state_dict(M.adam_state_dict(ea, eas, step=7))
    assert float(opt.state[params[3]]['step']) == 7
    off = sum(p.numel() for p in params[:3])
    np.testing.assert_array_equal(opt.state[params[3]]['exp_avg'].reshape(-1).numpy(),
                                  ea[off:off + params[3].numel()].numpy())
    ea2, eas2 = torch.zeros_like(flat), torch.zeros_like(flat)
    assert M.load_adam_state_dict(ea2, eas2, opt.state_dict()) == 7
    assert torch.equal(ea, ea2) and torch.equal(eas, eas2)
    flat2 = torch.zeros_like(flat)
    M.load_state_dict_into_flat(flat2, {k[len('module.'):]: v for k, v in sd.items()})     # un-prefixed too
    assert torch.equal(flat, flat2)
    torch.save({'net_0': sd}, tmp_path / 'model_000010.pth')
    M.load_state_dict_into_flat(flat2.zero_(), torch.load(tmp_path / 'model_000010.pth')['net_0'])
    assert torch.equal(flat, flat2)


def test_ray_generation_and_dataset_reader(golden, tmp_path):
    from PIL import Image
    g = golden('rays')
    o, d, depth = DL.get_rays_single_image(4, 6, g['K'], g['c2w'])
    np.testing.assert_allclose(o, g['rays_o'], rtol=1e-6)
    np.testing.assert_allclose(d, g['rays_d'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(depth, g['depth'], rtol=1e-6)
    root = tmp_path / 'data' / 'scene'
    rs = np.random.RandomState(0)
    for split, n in (('train', 3), ('test', 1)):
        for sub in ('rgb', 'pose', 'intrinsics', 'depth', 'depth_mono_crop'):
            os.makedirs(root / split / sub)
        for i in range(n):
            Image.fromarray(rs.randint(0, 255, (4, 6, 3), dtype=np.uint8)).save(root / split / 'rgb' / ('%03d.png' % i))
            dep = rs.randint(0, 80 * 256, (4, 6)).astype(np.uint16)
            Image.fromarray(dep).save(root / split / 'depth' / ('%03d.png' % i))
            Image.fromarray(dep).save(root / split / 'depth_mono_crop' / ('%03d.png' % i))
            np.savetxt(root / split / 'pose' / ('%03d.txt' % i), g['c2w'].reshape(1, 16))
            np.savetxt(root / split / 'intrinsics' / ('%03d.txt' % i), g['K'].reshape(1, 16))
    (root / 'scale').write_text('0.005\n')
    samplers = DL.load_data_split(str(tmp_path / 'data'), 'scene', 'train', skip=2, depth_sup_type='mono_crop')
    assert len(samplers) == 2 and samplers[0].H == 4 and samplers[0].W == 6
    np.random.seed(1)
    b = samplers[0].random_sample(5)
    assert set(b.keys()) >= {'ray_o', 'ray_d', 'depth', 'rgb', 'min_depth', 'img_name', 'depth_gt', 'depth_sup'}
    assert b['ray_d'].shape == (5, 3) and b['rgb'].dtype == np.float32 and np.all(b['min_depth'] == np.float32(1e-4))
    full = samplers[0].get_all()
    np.testing.assert_allclose(full['ray_d'], g['rays_d'], rtol=1e-5, atol=1e-6)
    raw = np.array(Image.open(root / 'train' / 'depth' / '000.png')).astype(np.float32).reshape(-1)
    np.testing.assert_allclose(full['depth_gt'], 0.005 * raw / 256.0, rtol=1e-6)
    assert full['mask'] is None                                   # key present like upstream, no mask/ dir
    # optional mask/ and min_depth/ (+ max_depth.txt): nerf_sample_ray_split.py:81-92, data_loader_split.py:72-75,111-114
    for sub in ('mask', 'min_depth'):
        os.makedirs(root / 'train' / sub)
    md = rs.randint(0, 255, (4, 6), dtype=np.uint8)
    mk = (rs.rand(4, 6) > 0.5).astype(np.uint8) * 255
    for i in range(3):
        Image.fromarray(md).save(root / 'train' / 'min_depth' / ('%03d.png' % i))
        Image.fromarray(mk).save(root / 'train' / 'mask' / ('%03d.png' % i))
    (root / 'train' / 'max_depth.txt').write_text('2.5\n')
    s2 = DL.load_data_split(str(tmp_path / 'data'), 'scene', 'train', skip=1, depth_sup_type='mono_crop')
    full = s2[1].get_all()
    np.testing.assert_allclose(full['min_depth'], md.reshape(-1).astype(np.float32) / 255. * 2.5 + 1e-4, rtol=1e-6)
    np.testing.assert_array_equal(full['mask'], mk.reshape(-1).astype(np.float32) / 255.)
    s3 = DL.load_data_split(str(tmp_path / 'data'), 'scene', 'train', try_load_min_depth=False, depth_sup_type='mono_crop')
    assert np.all(s3[0].get_all()['min_depth'] == np.float32(1e-4))


def test_shard_sizes_ragged_and_reference_behaviour():
    n = 375 * 1242
    assert D.shard_sizes(n, 2) == [n // 2, n // 2]
    s8 = D.shard_sizes(n, 8)
    assert sum(s8) == n and len(set(s8[:-1])) == 1 and s8[-1] - s8[0] == n % 8
    with pytest.raises(Exception, match='not divisible'):
        D.shard_sizes(n, 4, allow_ragged=False)
    assert D.rank_seeds(0) == 777 and D.rank_seeds(3) == 4 * 777


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, golden_path, out):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = np.load(golden_path)
    levels = O.init_params_like_reference(1)
    far = O.intersect_sphere(g['r%d.ray_o' % rank], g['r%d.ray_d' % rank])
    fg, bg = O.coarse_depths(g['r%d.min_depth' % rank], far, 64)
    fg, bg = O.perturb_samples(fg, g['r%d.t_fg' % rank]), O.perturb_samples(bg, g['r%d.t_bg' % rank])
    cache = {}
    ret = O.nerf_forward(levels[0], g['r%d.ray_o' % rank], g['r%d.ray_d' % rank], far, fg, bg, cache=cache)
    _, _, _, g_rgb, g_depth, g_w = O.loss_and_grads(ret, fg, far, g['r%d.rgb' % rank], g['r%d.depth_sup' % rank],
                                                    True, 'mse', 0.1, 0.)
    grads = O.nerf_backward(cache, g_rgb, g_depth, g_w)
    flat = torch.from_numpy(np.concatenate([grads[k].reshape(-1) for k in O.param_order()]))
    flat = flat * (1.0 / world)                              # the HIP backward pre-scales by 1/world
    D.allreduce_mean_(flat, world, prescaled=True)
    # ragged gather of a per-rank shard
    sizes = D.shard_sizes(7, world)
    lo = sum(sizes[:rank])
    shard = torch.arange(lo, lo + sizes[rank], dtype=torch.float32).reshape(-1, 1) * torch.ones(1, 3)
    full = D.gather_ragged(shard, sizes, rank, world)
    if rank == 0:
        np.save(out, flat.numpy())
        assert full.shape == (7, 3) and torch.equal(full[:, 0], torch.arange(7, dtype=torch.float32))
    dist.destroy_process_group()


def test_two_rank_gradient_average_gloo(tmp_path):
    """world_size 2 on CPU (gloo): per-rank gradients, pre-scaled SUM all-reduce == DDP's average,
    checked against the reference's 2-rank emulation (tests/golden/ddp2.npz)."""
    import torch.multiprocessing as mp
    gp = os.path.join(os.path.dirname(__file__), 'golden', 'ddp2.npz')
    out = str(tmp_path / 'avg.npy')
    mp.spawn(_ddp_worker, args=(2, _free_port(), gp, out), nprocs=2, join=True)
    avg = np.load(out)
    g = np.load(gp)
    off = 0
    shapes = {}
    for net, in_ch in (('fg_net', 63), ('bg_net', 84)):
        for k, s in O.mlp_param_shapes(in_ch, 27).items():
            shapes['%s.%s' % (net, k)] = s
    for k in O.param_order():
        n = int(np.prod(shapes[k]))
        mine = avg[off:off + n][g['avg.%s.idx' % k]]
        assert np.abs(mine - g['avg.%s.g' % k]).max() <= 5e-2 * g['avg.%s.rms' % k] + 1e-12, k
        off += n


def test_autoexposure_matches_reference_steps(golden):
    """the per-image (scale, shift) parameters, their part of the loss and their Adam
    update (host-side torch) against 4 steps of the reference (tests/golden/autoexpo.npz).  The HIP loss
    kernel is emulated here by its definition: mse + its gradient on (rgb, gt')."""
    from outdoor_nerf_depth_amd.autoexpo import AutoExposure, remap_name
    g = golden('autoexpo')
    names = [str(x) for x in g['names']]
    assert ['autoexpo_params.' + remap_name(n) for n in names] == [str(k) for k in g['state_keys']]
    lam_d = float(g['lambda_depth'])
    ae = AutoExposure(names, 'cpu', lrate=5e-4, lambda_autoexpo=float(g['lambda_autoexpo']))
    for step in range(1, 5):
        img = int(g['s%d.img' % step])
        idx = ae.lookup('some/prefix/' + names[img])
        assert idx == img
        rgb, gt = torch.from_numpy(g['s%d.ret_rgb' % step]), torch.from_numpy(g['s%d.rgb' % step])
        scale, shift = ae.scale_shift(idx)
        np.testing.assert_allclose(float(scale), g['s%d.scale' % step], rtol=1e-6)
        np.testing.assert_allclose(float(shift), g['s%d.shift' % step], rtol=1e-6, atol=1e-9)
        gtp = ae.target(idx, gt)
        mse = ((rgb - gtp) ** 2).mean()
        depth_loss = float(g['s%d.depth_loss' % step])
        scalars = torch.tensor([float(mse) + lam_d * depth_loss, float(mse), depth_loss, 0.])
        g_rgb = 2.0 * (rgb - gtp) / rgb.numel()
        want_g_rgb = 2.0 * ((rgb - shift) / scale - gt) / rgb.numel() / scale
        grad = ae.finish(idx, rgb, gt, scalars, g_rgb, lam_d)
        np.testing.assert_allclose(g_rgb.numpy(), want_g_rgb.numpy(), rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(float(scalars[1]), g['s%d.rgb_loss' % step], rtol=1e-5)
        np.testing.assert_allclose(float(scalars[0]), g['s%d.loss' % step], rtol=1e-5)
        np.testing.assert_allclose(grad.numpy(), g['s%d.grad' % step], rtol=2e-4, atol=1e-7)
        rows = torch.zeros(len(names), 2)
        rows[idx] = grad
        used = torch.zeros(len(names), dtype=torch.bool)
        used[idx] = True
        ae.apply(rows, used)
        np.testing.assert_allclose(ae.params.numpy(), g['s%d.params_after' % step], rtol=1e-6, atol=1e-9)
    # checkpoint entries: reference key names; never-stepped parameters have no optimiser state
    keys = [k for k, _ in ae.state_dict_entries()]
    assert keys == ['module.' + str(k) for k in g['state_keys']]
    st = ae.adam_entries()
    assert float(st[0]['step']) == 2 and float(st[1]['step']) == 1 and float(st[2]['step']) == 1
    ae2 = AutoExposure(names, 'cpu')
    ae2.load_state_dict_entries(dict(ae.state_dict_entries()))
    ae2.load_adam_entries(st)
    assert torch.equal(ae2.params, ae.params) and torch.equal(ae2.exp_avg_sq, ae.exp_avg_sq)


def test_config_file_syntax_and_unknown_keys(tmp_path):
    """configargparse semantics: `key = value`, `key: value`, `key value`, bare flags; the command line
    wins; a typo in a key is an error instead of a silent default."""
    from outdoor_nerf_depth_amd.ddp_train_nerf import config_parser
    cfg = tmp_path / 'c.txt'
    cfg.write_text('### INPUT\ndatadir = /data/x\nscene: seq00\nexpname run1\nN_iters = 17\nuse_depth = True\n'
                   'no_reload = False\nckpt_path = None\noptim_autoexpo\n# comment\nlambda_depth: 0.1\n')
    a = config_parser().parse_args(['--config', str(cfg), '--N_iters', '5'])
    assert (a.datadir, a.scene, a.expname, a.N_iters) == ('/data/x', 'seq00', 'run1', 5)
    assert a.use_depth and not a.no_reload and a.ckpt_path is None and a.optim_autoexpo and a.lambda_depth == 0.1
    bad = tmp_path / 'bad.txt'
    bad.write_text('lamda_depth = 3\n')
    with pytest.raises(SystemExit):
        config_parser().parse_args(['--config', str(bad)])


def test_bench_refuses_world_size_mismatch():
    """bench.py --gpus N under a launcher environment with a different WORLD_SIZE must fail loudly (CPU-only check:
    the refusal happens before any GPU work)."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0'))
    assert out.returncode != 0 and '--gpus 4 but WORLD_SIZE=2' in (out.stderr + out.stdout)
    # and without a launcher, on this GPU-less container, --gpus 2 refuses instead of running one rank
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env={k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK')})
    if out.returncode == 0:
        pytest.skip('this box has 2+ GPUs')
    assert 'GPU(s) visible' in (out.stderr + out.stdout)


def test_philox_known_answer_and_stream_layout():
    """oracle philox_uniform: Random123's Philox4x32-10 known-answer vector (counter 0, key 0 -> first word
    0x6627e8d5) and the (index, stream, step) counter layout of csrc/nerfpp_common.h."""
    u = O.philox_uniform(0, 0, 0, 4)
    assert u[0] == np.float32(0x6627e8) * np.float32(2.0 ** -24)
    a, b = O.philox_uniform(777, 5, 2, 64), O.philox_uniform(777, 5, 3, 64)
    assert not np.array_equal(a, b) and not np.array_equal(a, O.philox_uniform(777, 6, 2, 64))
    np.testing.assert_array_equal(O.philox_uniform(777, 5, 2, 64)[:10], O.philox_uniform(777, 5, 2, 10))
    uni = O.step_uniforms(777, 1, 8, 64, 128)
    assert uni['t_fg'].shape == (8, 64) and uni['u_bg'].shape == (8, 128) and uni['u_fg'].dtype == np.float32


def test_committed_pmc_profile_matches_kernel_sources():
    """bench.py parses `roofline.traffic` from the rocprofv3 --pmc passes committed under profiles/ (they cannot be collected inside
    the bench process).  The profile records the sha256 of the kernel sources it was measured on; this fails the CPU suite when
    csrc/nerfpp_{mlp,dw}.hip or nerfpp_common.h have changed since -- redo tools/probes/profile_round.sh
    and commit its <tag>_kernel_stats_timeline_hbm.md -- and the bench line then carries traffic = null with the reason."""
    import bench
    t = bench.load_pmc_traffic()
    assert t is not None, 'no PMC profile under profiles/'
    assert 'stale' not in t, t['stale']
    assert t['dw_L1'] > 1e9 and t['mlp_fwd_L1'] > 1e8 and t['mlp_bwd_L1'] > 1e8
    assert bench.kernel_sources_sha256() in t['source']


def test_rccl_env_defaults_single_node_only_and_channel_log_parser(tmp_path, monkeypatch):
    """ADVICE r05 / VERDICT r05 item 5: the channel cap is process-global and tuned for one xGMI node -- applied by default only
    when every rank is local, overridable (--rccl_channels), never over the caller's own NCCL_* settings; and the channel count a
    communicator actually got is read back from RCCL's INIT log."""
    for k in ('NCCL_MAX_NCHANNELS', 'NCCL_MIN_NCHANNELS', 'LOCAL_WORLD_SIZE', 'WORLD_SIZE', 'NCCL_DEBUG', 'NCCL_DEBUG_FILE'):
        monkeypatch.delenv(k, raising=False)
    assert D.single_node(8)                                                   # no launcher environment: this package's own spawn
    assert D.apply_rccl_env_defaults(8) == {'NCCL_MAX_NCHANNELS': '4', 'NCCL_MIN_NCHANNELS': '2'}
    for k in ('NCCL_MAX_NCHANNELS', 'NCCL_MIN_NCHANNELS'):
        monkeypatch.delenv(k)
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')                               # 2 nodes x 8: RCCL keeps its own defaults
    assert not D.single_node(16) and D.apply_rccl_env_defaults(16) == {}
    assert D.apply_rccl_env_defaults(16, channels=6) == {'NCCL_MAX_NCHANNELS': '6', 'NCCL_MIN_NCHANNELS': '2'}
    monkeypatch.setenv('NCCL_MAX_NCHANNELS', '12')                            # the caller's value wins
    assert D.apply_rccl_env_defaults(8, channels=3)['NCCL_MAX_NCHANNELS'] == '12'
    for k in ('NCCL_MAX_NCHANNELS', 'NCCL_MIN_NCHANNELS'):
        monkeypatch.delenv(k, raising=False)
    assert D.apply_rccl_env_defaults(8, channels=0) == {}                     # 0: hands off
    # the INIT log
    env = D.rccl_debug_file_env('t')
    assert env['NCCL_DEBUG'] == 'INFO' and env['NCCL_DEBUG_SUBSYS'] == 'INIT' and '%h' in env['NCCL_DEBUG_FILE'] and '%p' in env['NCCL_DEBUG_FILE']
    assert D.rccl_channels_in_effect() is None                                # no NCCL_DEBUG_FILE
    path = str(tmp_path / 'rccl_%h_%p.log')
    monkeypatch.setenv('NCCL_DEBUG_FILE', path)
    assert D.rccl_channels_in_effect() is None                                # no such file
    real = path.replace('%h', socket.gethostname()).replace('%p', str(os.getpid()))
    with open(real, 'w') as f:
        f.write('host:1:1 [0] NCCL INFO comm 0x1 rank 0 nranks 8 cudaDev 0 busId 1000 - Init START\n'
                'host:1:1 [0] NCCL INFO 4 coll channels, 0 collnet channels, 0 nvls channels, 8 p2p channels, 2 p2p channels per peer\n'
                'host:1:1 [0] NCCL INFO 2 coll channels, 4 p2p channels, 1 p2p channels per peer\n')
    assert D.rccl_channels_in_effect() == [{'coll': 4, 'p2p': 8}, {'coll': 2, 'p2p': 4}]
    monkeypatch.setenv('NCCL_DEBUG', 'WARN')                                  # the caller configured RCCL's logging: not ours to redirect
    assert D.rccl_debug_file_env('t') == {}


def test_bench_power_sampler_is_harmless_without_hwmon(tmp_path):
    """bench.PowerSampler (socket power / shader clock beside the end_to_end loop) never raises: no GPU or no readable hwmon
    files -> summary None; with files (a fake hwmon directory here) -> watts and MHz over the requested window only."""
    import time
    import bench
    with bench.PowerSampler() as ps:
        pass
    assert ps.summary(0.0, float('inf')) is None or isinstance(ps.summary(0.0, float('inf')), dict)
    ps = bench.PowerSampler()
    (tmp_path / 'power1_input').write_text('1366000000\n')
    (tmp_path / 'power1_cap').write_text('1400000000\n')
    (tmp_path / 'freq1_input').write_text('1931000000\n')
    ps.files = {'power_uw': str(tmp_path / 'power1_input'), 'cap_uw': str(tmp_path / 'power1_cap'), 'sclk_hz': str(tmp_path / 'freq1_input'),
                'gone': str(tmp_path / 'missing')}
    import threading
    ps.thread = threading.Thread(target=ps._run, daemon=True)
    t0 = time.perf_counter()
    with ps:
        time.sleep(0.2)
    t1 = time.perf_counter()
    got = ps.summary(t0, t1)
    assert got['socket_w'] == 1366.0 and got['cap_w'] == 1400.0 and got['sclk_mhz'] == 1931.0 and got['samples'] >= 2
    assert ps.summary(t1 + 1.0, t1 + 2.0) is None
