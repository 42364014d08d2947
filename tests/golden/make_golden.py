#!/usr/bin/env python
"""Generate tests/golden/*.npz by importing the REFERENCE's PyTorch modules.

Runs only in the build container (needs /root/reference; the GPU box has none).  The
reference is imported with empty stub modules for cv2 / imageio / tensorboardX /
configargparse (only its visualisation, PNG I/O and CLI parsing touch those; the hot path
never does).  Nothing of the reference is copied: the fixtures hold seeded inputs and the
reference's outputs on them.

    python tests/golden/make_golden.py
"""
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

REF = '/root/reference/nerf-methods/nerfplusplus'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
for name in ('cv2', 'imageio', 'tensorboardX', 'configargparse'):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules['tensorboardX'].SummaryWriter = object
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import ddp_train_nerf as R            # noqa: E402  (reference)
import ddp_model as RM                # noqa: E402
import nerf_network as RN             # noqa: E402
import depth_loss as RL               # noqa: E402
import utils as RU                    # noqa: E402
from nerf_sample_ray_split import get_rays_single_image    # noqa: E402
from outdoor_nerf_depth_amd.synthetic import SyntheticKitti  # noqa: E402

torch.set_num_threads(4)
T = torch.from_numpy


def ref_args(netwidth=256):
    return types.SimpleNamespace(max_freq_log2=10, max_freq_log2_viewdirs=4, netdepth=8,
                                 netwidth=netwidth, use_viewdirs=True)


def make_levels(n_levels=2, netwidth=256):
    """create_nerf's construction order (ddp_train_nerf.py:308-325) without DDP/.to(rank)."""
    torch.manual_seed(777)
    return [RM.NerfNetWithAutoExpo(ref_args(netwidth), optim_autoexpo=False)
            for _ in range(n_levels)]


def state_np(net):
    return OrderedDict((k[len('nerf_net.'):], v.detach().numpy().copy())
                       for k, v in net.state_dict().items())


def sample_idx(numel, n=256, seed=0):
    return np.random.RandomState(seed).choice(numel, size=min(n, numel), replace=False)


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('%-22s %8.1f KB' % (name, os.path.getsize(path) / 1024.))


def batch(n, seed, depth_sup_type='gt'):
    scene = SyntheticKitti(depth_sup_type=depth_sup_type)
    b = scene.random_batch(n, np.random.RandomState(seed))
    return scene, b


# ---------------------------------------------------------------------------------------
def gen_sampling():
    scene, b = batch(48, 1)
    ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
    far = R.intersect_sphere(ray_o, ray_d)
    near = T(b['min_depth'])
    S = 64
    step = (far - near) / (S - 1)
    fg = torch.stack([near + i * step for i in range(S)], dim=-1)
    bg = torch.linspace(0., 1., S).view(1, S).expand(48, S)
    torch.manual_seed(11)
    t_fg = torch.rand_like(fg)
    t_bg = torch.rand_like(bg)
    torch.manual_seed(11)
    fg_p = R.perturb_samples(fg)
    bg_p = R.perturb_samples(bg.contiguous())
    # sample_pdf on synthetic weights: a peaky pdf, a flat one, exact zeros, tiny values
    rs = np.random.RandomState(5)
    w = rs.rand(48, 62).astype(np.float32) ** 4
    w[5] = 0.
    w[6, :30] = 0.
    w[7] = 1e-9
    w[8, 20] = 50.
    bins = (.5 * (fg_p[..., 1:] + fg_p[..., :-1]))
    out = {}
    for tag, det in (('rand', False), ('det', True)):
        torch.manual_seed(23)
        samples = R.sample_pdf(bins=bins, weights=T(w), N_samples=128, det=det)
        torch.manual_seed(23)
        u = torch.linspace(0., 1., 128).view(1, 128).expand(48, 128) if det else torch.rand(48, 128)
        # re-derive the reference's above_inds with its own expressions (ddp_train_nerf.py:92-111)
        ww = T(w) + RU.TINY_NUMBER
        pdf = ww / torch.sum(ww, dim=-1, keepdim=True)
        cdf = torch.cumsum(pdf, dim=-1)
        cdf = torch.cat([torch.zeros_like(cdf[..., 0:1]), cdf], dim=-1)
        above = torch.sum(u.unsqueeze(-1) >= cdf[..., :62].unsqueeze(-2), dim=-1).long()
        margin = (u.unsqueeze(-1) - cdf[..., :62].unsqueeze(-2)).abs().min(-1)[0]
        merged, _ = torch.sort(torch.cat((fg_p, samples), dim=-1))
        out.update({'u_' + tag: u.numpy(), 'samples_' + tag: samples.numpy(),
                    'above_' + tag: above.numpy(), 'margin_' + tag: margin.numpy(),
                    'cdf_' + tag: cdf.numpy(), 'merged_' + tag: merged.numpy()})
    save('sampling', ray_o=b['ray_o'], ray_d=b['ray_d'], min_depth=b['min_depth'],
         fg_far=far.numpy(), fg_coarse=fg.numpy(), bg_coarse=bg.numpy(), t_fg=t_fg.numpy(),
         t_bg=t_bg.numpy(), fg_perturbed=fg_p.numpy(), bg_perturbed=bg_p.numpy(),
         bins=bins.numpy(), weights=w, linspace64=torch.linspace(0., 1., 64).numpy(),
         linspace128=torch.linspace(0., 1., 128).numpy(), **out)


def gen_embed():
    rs = np.random.RandomState(3)
    x3 = (rs.rand(64, 3).astype(np.float32) * 2 - 1)
    x4 = (rs.rand(64, 4).astype(np.float32) * 2 - 1)
    e63 = RN.Embedder(3, 9, 10)(T(x3)).numpy()
    e84 = RN.Embedder(4, 9, 10)(T(x4)).numpy()
    e27 = RN.Embedder(3, 3, 4)(T(x3)).numpy()
    save('embed', x3=x3, x4=x4, e63=e63, e84=e84, e27=e27)


def gen_depth2pts():
    scene, b = batch(32, 2)
    z = torch.linspace(0., 1., 64).view(1, 64).expand(32, 64)
    torch.manual_seed(4)
    z = R.perturb_samples(z.contiguous())
    o = T(b['ray_o']).unsqueeze(-2).expand(32, 64, 3)
    d = T(b['ray_d']).unsqueeze(-2).expand(32, 64, 3)
    pts, depth_real = RM.depth2pts_outside(o, d, z)
    save('depth2pts', ray_o=b['ray_o'], ray_d=b['ray_d'], bg_z=z.numpy(), pts=pts.numpy(),
         depth_real=depth_real.numpy())


def gen_params_and_mlp():
    nets = make_levels(2)
    arrs = {}
    for m, net in enumerate(nets):
        for k, v in state_np(net).items():
            idx = sample_idx(v.size)
            arrs['L%d.%s.idx' % (m, k)] = idx
            arrs['L%d.%s.val' % (m, k)] = v.reshape(-1)[idx]
            arrs['L%d.%s.sum' % (m, k)] = np.float64(v.astype(np.float64).sum())
    save('params_seed777', **arrs)
    rs = np.random.RandomState(9)
    fg_in = (rs.rand(96, 90).astype(np.float32) * 2 - 1)
    bg_in = (rs.rand(96, 111).astype(np.float32) * 2 - 1)
    with torch.no_grad():
        rf = nets[0].nerf_net.fg_net(T(fg_in))
        rb = nets[0].nerf_net.bg_net(T(bg_in))
    save('mlp', fg_in=fg_in, bg_in=bg_in, fg_rgb=rf['rgb'].numpy(), fg_sigma=rf['sigma'].numpy(),
         bg_rgb=rb['rgb'].numpy(), bg_sigma=rb['sigma'].numpy())
    return nets


def level_inputs(b, seed, S0=64, S1=128, net0=None):
    """Sample depths exactly as the training loop does (ddp_train_nerf.py:432-465), returning
    the uniforms so kernels/oracle can replay them."""
    N = b['ray_o'].shape[0]
    ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
    far = R.intersect_sphere(ray_o, ray_d)
    near = T(b['min_depth'])
    step = (far - near) / (S0 - 1)
    fg = torch.stack([near + i * step for i in range(S0)], dim=-1)
    bg = torch.linspace(0., 1., S0).view(1, S0).expand(N, S0).contiguous()
    torch.manual_seed(seed)
    t_fg, t_bg = torch.rand_like(fg), torch.rand_like(bg)
    u_fg, u_bg = torch.rand(N, S1), torch.rand(N, S1)
    torch.manual_seed(seed)
    fg0 = R.perturb_samples(fg)
    bg0 = R.perturb_samples(bg)
    out = dict(far=far, fg0=fg0, bg0=bg0, t_fg=t_fg, t_bg=t_bg, u_fg=u_fg, u_bg=u_bg)
    if net0 is not None:
        with torch.no_grad():
            ret0 = net0(ray_o, ray_d, far, fg0, bg0)
        fgw = ret0['fg_weights'].clone().detach()[..., 1:-1]
        bgw = ret0['bg_weights'].clone().detach()[..., 1:-1]
        fs = R.sample_pdf(bins=.5 * (fg0[..., 1:] + fg0[..., :-1]), weights=fgw, N_samples=S1)
        fg1, _ = torch.sort(torch.cat((fg0, fs), dim=-1))
        bs = R.sample_pdf(bins=.5 * (bg0[..., 1:] + bg0[..., :-1]), weights=bgw, N_samples=S1)
        bg1, _ = torch.sort(torch.cat((bg0, bs), dim=-1))
        out.update(ret0=ret0, fg1=fg1, bg1=bg1)
    return out


def gen_forward(nets):
    scene, b = batch(12, 7)
    li = level_inputs(b, 31, net0=nets[0])
    ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
    arrs = dict(ray_o=b['ray_o'], ray_d=b['ray_d'], min_depth=b['min_depth'],
                fg_far=li['far'].numpy(), fg_z0=li['fg0'].numpy(), bg_z0=li['bg0'].numpy(),
                fg_z1=li['fg1'].numpy(), bg_z1=li['bg1'].numpy(), t_fg=li['t_fg'].numpy(),
                t_bg=li['t_bg'].numpy(), u_fg=li['u_fg'].numpy(), u_bg=li['u_bg'].numpy())
    for k, v in li['ret0'].items():
        arrs['L0.' + k] = v.numpy()
    with torch.no_grad():
        ret1 = nets[1](ray_o, ray_d, li['far'], li['fg1'], li['bg1'])
    for k, v in ret1.items():
        arrs['L1.' + k] = v.numpy()
    save('forward', **arrs)


def gen_losses():
    rs = np.random.RandomState(13)
    N, S = 24, 64
    gt = (rs.rand(N).astype(np.float32) * 0.4) * (rs.rand(N) < 0.6)
    pred = rs.rand(N).astype(np.float32) * 0.4
    w = rs.rand(N, S).astype(np.float32) ** 3 * 0.2
    steps = np.sort(rs.rand(N, S).astype(np.float32) * 0.5, -1)
    lengths = rs.rand(N, S).astype(np.float32) * 0.01
    far = rs.rand(N).astype(np.float32) * 0.3 + 0.15
    sigma = 0.01 * 0.0054
    zero = np.zeros(N, np.float32)
    arrs = dict(gt=gt.astype(np.float32), pred=pred, w=w, steps=steps, lengths=lengths, far=far,
                sigma=np.float64(sigma))
    arrs['mse'] = RL.depth_mse(T(gt.astype(np.float32)), T(pred)).numpy()
    arrs['l1'] = RL.depth_l1(T(gt.astype(np.float32)), T(pred)).numpy()
    arrs['kl'] = RL.depth_kl(T(w), T(gt.astype(np.float32)), T(steps), T(lengths), sigma, T(far)).numpy()
    arrs['kl_nofar'] = RL.depth_kl(T(w), T(gt.astype(np.float32)), T(steps), T(lengths), sigma).numpy()
    arrs['mse_empty'] = RL.depth_mse(T(zero), T(pred)).numpy()
    arrs['l1_empty'] = RL.depth_l1(T(zero), T(pred)).numpy()
    arrs['kl_empty'] = RL.depth_kl(T(w), T(zero), T(steps), T(lengths), sigma, T(far)).numpy()
    x, y = rs.rand(N, 3).astype(np.float32), rs.rand(N, 3).astype(np.float32)
    arrs.update(x=x, y=y, img2mse=RU.img2mse(T(x), T(y)).numpy(),
                psnr=np.float64(RU.mse2psnr(RU.img2mse(T(x), T(y)).item())))
    save('losses', **arrs)


def ref_level_step(net, batch_t, far, fg_z, bg_z, mode, lambda_depth, depth_sigma_scaled,
                   dtype=None):
    """One level of ddp_train_nerf.py:467-497 up to backward(); returns ret, scalars.
    dtype=torch.float64 runs the SAME reference code on a double copy of the net and inputs
    (the float32 reference is itself ~1e-2 noisy on the ill-conditioned depth gradients)."""
    if dtype is not None:
        import copy
        net = copy.deepcopy(net).to(dtype)
        batch_t = {k: v.to(dtype) for k, v in batch_t.items()}
        far, fg_z, bg_z = far.to(dtype), fg_z.to(dtype), bg_z.to(dtype)
        ret, loss, rgb_loss, depth_loss = ref_level_step(net, batch_t, far, fg_z, bg_z, mode,
                                                         lambda_depth, depth_sigma_scaled)
        return net, ret, loss
    net.zero_grad()
    ret = net(batch_t['ray_o'], batch_t['ray_d'], far, fg_z, bg_z)
    rgb_loss = RU.img2mse(ret['rgb'], batch_t['rgb'])
    loss = rgb_loss
    depth_loss = None
    if mode != 'rgbonly':
        if mode == 'kl':
            depth_loss = RL.depth_kl(ret['fg_weights'], batch_t['depth_sup'], fg_z, ret['fg_dists'],
                                     depth_sigma_scaled, far)
        else:
            depth_loss = R.depth_losses_dict[mode](batch_t['depth_sup'], ret['depth'])
        loss = loss + lambda_depth * depth_loss
    loss.backward()
    return ret, loss, rgb_loss, depth_loss


def grads_np(net):
    return OrderedDict((k[len('nerf_net.'):], p.grad.detach().numpy().copy())
                       for k, p in net.named_parameters())


def gen_grads(nets):
    """Level-1-shaped (S=192) and level-0-shaped backward for every loss type."""
    for mode, sup in (('rgbonly', 'gt'), ('mse', 'gt'), ('l1', 'stereo_crop'), ('kl', 'mono_crop')):
        scene, b = batch(10, 17, depth_sup_type=sup)
        if mode in ('mse',):                       # make sure the sparse mask is not empty
            b['depth_sup'][:4] = np.float32(0.05)
        bt = {k: T(v) for k, v in b.items() if isinstance(v, np.ndarray)}
        li = level_inputs(b, 41, net0=nets[0])
        sig = 0.01 * float(scene.depth_scale)
        arrs = dict(depth_sup=b['depth_sup'], rgb_gt=b['rgb'], ray_o=b['ray_o'], ray_d=b['ray_d'],
                    min_depth=b['min_depth'], fg_far=li['far'].numpy(),
                    t_fg=li['t_fg'].numpy(), t_bg=li['t_bg'].numpy(), u_fg=li['u_fg'].numpy(),
                    u_bg=li['u_bg'].numpy(), depth_sigma_scaled=np.float64(sig),
                    lambda_depth=np.float64(0.1))
        for m, (fz, bz) in enumerate(((li['fg0'], li['bg0']), (li['fg1'], li['bg1']))):
            ret, loss, rgb_loss, depth_loss = ref_level_step(nets[m], bt, li['far'], fz, bz, mode,
                                                             0.1, sig)
            arrs['L%d.fg_z' % m] = fz.numpy()
            arrs['L%d.bg_z' % m] = bz.numpy()
            arrs['L%d.loss' % m] = loss.item()
            arrs['L%d.rgb_loss' % m] = rgb_loss.item()
            arrs['L%d.depth_loss' % m] = np.nan if depth_loss is None else depth_loss.item()
            arrs['L%d.rgb' % m] = ret['rgb'].detach().numpy()
            arrs['L%d.depth' % m] = ret['depth'].detach().numpy()
            for k, g in grads_np(nets[m]).items():
                idx = sample_idx(g.size, 192)
                arrs['L%d.%s.idx' % (m, k)] = idx
                arrs['L%d.%s.g' % (m, k)] = g.reshape(-1)[idx]
                arrs['L%d.%s.norm' % (m, k)] = np.float64(np.linalg.norm(g.astype(np.float64)))
            net64, ret64, loss64 = ref_level_step(nets[m], bt, li['far'], fz, bz, mode, 0.1, sig,
                                                  dtype=torch.float64)
            arrs['L%d.loss64' % m] = loss64.item()
            for k, g in grads_np(net64).items():
                arrs['L%d.%s.g64' % (m, k)] = g.reshape(-1)[sample_idx(g.size, 192)]
                arrs['L%d.%s.norm64' % (m, k)] = np.float64(np.linalg.norm(g))
        save('grads_' + mode, **arrs)


def gen_small_net_grads():
    """Reduced-width (W=32) net: full gradients for bit-level debugging of the oracle."""
    torch.manual_seed(777)
    net = RM.NerfNetWithAutoExpo(ref_args(32), optim_autoexpo=False)
    scene, b = batch(8, 19, depth_sup_type='mono_crop')
    bt = {k: T(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    li = level_inputs(b, 43)
    arrs = dict(ray_o=b['ray_o'], ray_d=b['ray_d'], rgb_gt=b['rgb'], depth_sup=b['depth_sup'],
                fg_far=li['far'].numpy(), fg_z=li['fg0'].numpy(), bg_z=li['bg0'].numpy(),
                depth_sigma_scaled=np.float64(0.01 * float(scene.depth_scale)))
    for k, v in state_np(net).items():
        arrs['p.' + k] = v
    for mode in ('rgbonly', 'mse', 'l1', 'kl'):
        ret, loss, rgb_loss, depth_loss = ref_level_step(net, bt, li['far'], li['fg0'], li['bg0'],
                                                         mode, 0.5, arrs['depth_sigma_scaled'])
        arrs[mode + '.loss'] = loss.item()
        for k, g in grads_np(net).items():
            arrs['%s.g.%s' % (mode, k)] = g
    save('small_net_grads', **arrs)


def gen_train_steps():
    """3 full optimisation steps (both levels, Adam) with replayed uniforms + a 2-rank DDP
    emulation (average of per-rank gradients) for step 1."""
    nets = make_levels(2)
    optims = [torch.optim.Adam(n.parameters(), lr=5e-4) for n in nets]
    scene = SyntheticKitti(depth_sup_type='gt')
    arrs = {}
    for step in range(1, 4):
        b = scene.random_batch(16, np.random.RandomState(100 + step))
        b['depth_sup'][:5] = np.float32(0.03 * step)
        bt = {k: T(v) for k, v in b.items() if isinstance(v, np.ndarray)}
        li = level_inputs(b, 1000 + step, net0=nets[0])
        for k in ('ray_o', 'ray_d', 'rgb', 'depth_sup', 'min_depth'):
            arrs['s%d.%s' % (step, k)] = b[k]
        for k in ('t_fg', 't_bg', 'u_fg', 'u_bg'):
            arrs['s%d.%s' % (step, k)] = li[k].numpy()
        for m, (fz, bz) in enumerate(((li['fg0'], li['bg0']), (li['fg1'], li['bg1']))):
            optims[m].zero_grad()
            ret, loss, rgb_loss, depth_loss = ref_level_step(nets[m], bt, li['far'], fz, bz, 'mse',
                                                             0.1, 0.)
            optims[m].step()
            arrs['s%d.L%d.loss' % (step, m)] = loss.item()
            arrs['s%d.L%d.rgb_loss' % (step, m)] = rgb_loss.item()
            arrs['s%d.L%d.depth_loss' % (step, m)] = depth_loss.item()
        # NOTE: the level-1 depths above were drawn from level-0 weights BEFORE level-0's Adam
        # step, exactly like the reference loop (ret of level 0 is computed before optim.step()).
        if step in (1, 3):
            for m in range(2):
                for k, v in state_np(nets[m]).items():
                    idx = sample_idx(v.size, 128)
                    arrs['after%d.L%d.%s.idx' % (step, m, k)] = idx
                    arrs['after%d.L%d.%s.val' % (step, m, k)] = v.reshape(-1)[idx]
    save('train_steps', **arrs)

    # ---- 2-rank emulation: DDP averages gradients (ddp_train_nerf.py:323)
    nets = make_levels(1)
    arrs = {}
    gsum = None
    for r in range(2):
        b = scene.random_batch(8, np.random.RandomState(200 + r))
        b['depth_sup'][:3] = np.float32(0.04)
        bt = {k: T(v) for k, v in b.items() if isinstance(v, np.ndarray)}
        li = level_inputs(b, 2000 + r)
        for k in ('ray_o', 'ray_d', 'rgb', 'depth_sup', 'min_depth'):
            arrs['r%d.%s' % (r, k)] = b[k]
        arrs['r%d.t_fg' % r] = li['t_fg'].numpy()
        arrs['r%d.t_bg' % r] = li['t_bg'].numpy()
        net64, _, _ = ref_level_step(nets[0], bt, li['far'], li['fg0'], li['bg0'], 'mse', 0.1, 0.,
                                     dtype=torch.float64)          # float64 run: see gen_grads
        g = grads_np(net64)
        gsum = g if gsum is None else OrderedDict((k, gsum[k] + g[k]) for k in g)
    for k, g in gsum.items():
        g = g / 2.
        idx = sample_idx(g.size, 128)
        arrs['avg.%s.idx' % k] = idx
        arrs['avg.%s.g' % k] = g.reshape(-1)[idx]
        arrs['avg.%s.rms' % k] = np.float64(np.sqrt((g ** 2).mean()))
    save('ddp2', **arrs)


def gen_adam_unit():
    """torch.optim.Adam (lr 5e-4, defaults) on a single tensor with prescribed gradients."""
    rs = np.random.RandomState(77)
    p0 = rs.randn(512).astype(np.float32) * 0.1
    gs = (rs.randn(4, 512) * np.logspace(-9, 0, 512)[None, :]).astype(np.float32)
    p = torch.nn.Parameter(T(p0.copy()))
    opt = torch.optim.Adam([p], lr=5e-4)
    outs = []
    for i in range(4):
        p.grad = T(gs[i].copy())
        opt.step()
        outs.append(p.detach().numpy().copy())
    st = opt.state[p]
    save('adam_unit', p0=p0, grads=gs, p_after=np.stack(outs), exp_avg=st['exp_avg'].numpy(),
         exp_avg_sq=st['exp_avg_sq'].numpy())


def gen_rays():
    """f-1: get_rays_single_image on a 4x6 image."""
    K = np.eye(4, dtype=np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 5.5, 5.25, 3.1, 1.9
    c2w = np.eye(4, dtype=np.float32)
    a = 0.3
    c2w[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    c2w[:3, 3] = [0.1, -0.05, 0.2]
    o, d, depth = get_rays_single_image(4, 6, K, c2w)
    save('rays', K=K, c2w=c2w, rays_o=o, rays_d=d, depth=depth)


def gen_autoexpo():
    """a11: NerfNetWithAutoExpo with --optim_autoexpo (ddp_model.py:161-192) through 4 steps of the
    level-0 loop of ddp_train_nerf.py:467-498 (rgb loss on the exposure-corrected prediction + the
    lambda_autoexpo regulariser + depth mse; Adam over net.parameters()).  Images 0, 2, 0, 1: image 0's
    parameter is stepped twice, image 1/2 once (per-parameter Adam step counts)."""
    torch.manual_seed(777)
    names = ['scene/train/rgb/000000.png', 'scene/train/rgb/000001.png', 'scene/train/rgb/000002.png']
    net = RM.NerfNetWithAutoExpo(ref_args(), optim_autoexpo=True, img_names=names)
    optim = torch.optim.Adam(net.parameters(), lr=5e-4)
    scene = SyntheticKitti(depth_sup_type='gt')
    lam_ae, lam_d = 0.5, 0.1
    arrs = {'names': np.array(names), 'lambda_autoexpo': lam_ae, 'lambda_depth': lam_d}
    for step, img in enumerate((0, 2, 0, 1), start=1):
        b = scene.random_batch(16, np.random.RandomState(300 + step))
        b['depth_sup'][:5] = np.float32(0.03 * step)
        # a colour cast so that the exposure parameters have something to explain
        b['rgb'] = np.clip(b['rgb'] * np.float32(0.8 + 0.1 * img) + np.float32(0.05 * img), 0, 1).astype(np.float32)
        bt = {k: T(v) for k, v in b.items() if isinstance(v, np.ndarray)}
        li = level_inputs(b, 3000 + step)
        optim.zero_grad()
        ret = net(bt['ray_o'], bt['ray_d'], li['far'], li['fg0'], li['bg0'], img_name=names[img])
        scale, shift = ret['autoexpo']                                       # :472-479
        rgb_pred = (ret['rgb'] - shift) / scale
        rgb_loss = RU.img2mse(rgb_pred, bt['rgb'])
        loss = rgb_loss + lam_ae * (torch.abs(scale - 1.) + torch.abs(shift))
        depth_loss = R.depth_losses_dict['mse'](bt['depth_sup'], ret['depth'])
        loss = loss + lam_d * depth_loss
        loss.backward()
        pname = 'autoexpo_params.' + RM.remap_name(names[img])
        g = dict(net.named_parameters())[pname].grad.numpy().copy()
        scale_v, shift_v = scale.item(), shift.item()         # shift is a view of the parameter: read before the step
        optim.step()
        for k in ('ray_o', 'ray_d', 'rgb', 'depth_sup', 'min_depth'):
            arrs['s%d.%s' % (step, k)] = b[k]
        arrs['s%d.t_fg' % step] = li['t_fg'].numpy()
        arrs['s%d.t_bg' % step] = li['t_bg'].numpy()
        arrs['s%d.img' % step] = img
        arrs['s%d.ret_rgb' % step] = ret['rgb'].detach().numpy()
        arrs['s%d.ret_depth' % step] = ret['depth'].detach().numpy()
        arrs['s%d.loss' % step] = loss.item()
        arrs['s%d.rgb_loss' % step] = rgb_loss.item()
        arrs['s%d.depth_loss' % step] = depth_loss.item()
        arrs['s%d.scale' % step] = scale_v
        arrs['s%d.shift' % step] = shift_v
        arrs['s%d.grad' % step] = g
        arrs['s%d.params_after' % step] = np.stack([p.detach().numpy().copy() for k, p in net.named_parameters()
                                                    if k.startswith('autoexpo_params.')])
    sd = net.state_dict()
    arrs['state_keys'] = np.array([k for k in sd.keys() if k.startswith('autoexpo')])
    w = sd['nerf_net.fg_net.rgb_layers.2.weight'].numpy()
    arrs['final.fg_rgb2_weight'] = w
    save('autoexpo', **arrs)


class _ReplayRand(object):
    """Feed prescribed uniforms to the reference's own perturb_samples / sample_pdf (they call torch.rand_like / torch.rand,
    ddp_train_nerf.py:75,104) in the order the training loop consumes them."""
    def __init__(self, tensors):
        self.q = list(tensors)
    def __enter__(self):
        self.saved = (torch.rand_like, torch.rand)
        def nxt(shape, dtype):
            t = self.q.pop(0)
            assert tuple(t.shape) == tuple(shape), (t.shape, shape)
            return t.to(dtype)
        torch.rand_like = lambda x, **kw: nxt(x.shape, x.dtype)
        torch.rand = lambda *sh, **kw: nxt(sh[0] if len(sh) == 1 and not isinstance(sh[0], int) else sh, torch.get_default_dtype())
        return self
    def __exit__(self, *a):
        torch.rand_like, torch.rand = self.saved
        assert not self.q, 'unconsumed uniforms'


def ref_render_frame(nets, rays, cascade, dtype=torch.float32, chunk=1024):
    """The per-chunk body of render_single_image (ddp_train_nerf.py:156-221: deterministic sampling, no perturbation) with
    the reference's own intersect_sphere / sample_pdf / NerfNet.forward; its rank split and torch.distributed gather are
    left out.  Returns the last level's rgb [n,3] and depth [n]."""
    outs = []
    n = rays['ray_o'].shape[0]
    for s in range(0, n, chunk):
        ray_o, ray_d = T(rays['ray_o'][s:s + chunk]).to(dtype), T(rays['ray_d'][s:s + chunk]).to(dtype)
        min_depth = T(rays['min_depth'][s:s + chunk]).to(dtype)
        N = ray_o.shape[0]
        ret = None
        for m, net in enumerate(nets):
            S = cascade[m]
            if m == 0:
                far = R.intersect_sphere(ray_o, ray_d)
                step = (far - min_depth) / (S - 1)
                fg = torch.stack([min_depth + i * step for i in range(S)], dim=-1)
                bg = torch.linspace(0., 1., S).view(1, S).expand(N, S).to(dtype)
            else:
                fgw = ret['fg_weights'].clone().detach()[..., 1:-1]
                fs = R.sample_pdf(bins=.5 * (fg[..., 1:] + fg[..., :-1]), weights=fgw, N_samples=S, det=True)
                fg, _ = torch.sort(torch.cat((fg, fs.to(dtype)), dim=-1))
                bgw = ret['bg_weights'].clone().detach()[..., 1:-1]
                bs = R.sample_pdf(bins=.5 * (bg[..., 1:] + bg[..., :-1]), weights=bgw, N_samples=S, det=True)
                bg, _ = torch.sort(torch.cat((bg, bs.to(dtype)), dim=-1))
            with torch.no_grad():
                ret = net(ray_o, ray_d, far, fg, bg)
        outs.append((ret['rgb'].double().numpy(), ret['depth'].double().numpy()))
    return np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs])


def run_ref_trajectory(mode, dtype, n_steps, seed=0, tag=None):
    """The imported reference's training loop (ddp_train_nerf.py:417-498) on the BASELINE config-1 scene on replayed batches and
    uniforms (tests/trajectory_common.py: seeds only; seed s shifts the numpy streams by 100000 s, the same rule
    tests/test_gpu_round4.py: _trajectory uses).  Returns the scalars logged every LOG_EVERY steps, the rgb losses of the last
    LOG_EVERY steps, and the frame rendered at the end with deterministic sampling."""
    import time
    import trajectory_common as TC
    smp = TC.sampler(mode)
    full = {k: np.ascontiguousarray(v, np.float32) for k, v in smp.get_all().items() if isinstance(v, np.ndarray)}
    sigma = TC.DEPTH_SIGMA * float(smp.get_depth_scale() or 1.0)
    nets = [n.to(dtype) for n in make_levels(2)]
    optims = [torch.optim.Adam(n.parameters(), lr=5e-4) for n in nets]
    logs = {k: [] for k in ('loss0', 'loss1', 'rgb0', 'rgb1', 'depth0', 'depth1')}
    tail = []
    t0 = time.time()
    for step in range(1, n_steps + 1):
        b, uni = TC.step_batch(smp, step + 100000 * seed), TC.step_uniforms(step + 100000 * seed)
        bt = {k: T(v).to(dtype) for k, v in b.items()}
        N = bt['ray_o'].shape[0]
        row = {}
        with _ReplayRand([T(uni[k]) for k in ('t_fg', 't_bg', 'u_fg', 'u_bg')]):
            for m in range(2):                                      # :432-498
                S = TC.CASCADE[m]
                if m == 0:
                    far = R.intersect_sphere(bt['ray_o'], bt['ray_d'])
                    near = bt['min_depth']
                    stp = (far - near) / (S - 1)
                    fg = torch.stack([near + i * stp for i in range(S)], dim=-1)
                    fg = R.perturb_samples(fg)
                    bg = torch.linspace(0., 1., S).view(1, S).expand(N, S).to(dtype)
                    bg = R.perturb_samples(bg)
                else:
                    fgw = ret['fg_weights'].clone().detach()[..., 1:-1]
                    fs = R.sample_pdf(bins=.5 * (fg[..., 1:] + fg[..., :-1]), weights=fgw, N_samples=S, det=False)
                    fg, _ = torch.sort(torch.cat((fg, fs), dim=-1))
                    bgw = ret['bg_weights'].clone().detach()[..., 1:-1]
                    bs = R.sample_pdf(bins=.5 * (bg[..., 1:] + bg[..., :-1]), weights=bgw, N_samples=S, det=False)
                    bg, _ = torch.sort(torch.cat((bg, bs), dim=-1))
                optims[m].zero_grad()
                ret, loss, rgb_loss, depth_loss = ref_level_step(nets[m], bt, far, fg, bg, mode, TC.LAMBDA_DEPTH, sigma)
                optims[m].step()
                row['loss%d' % m], row['rgb%d' % m] = loss.item(), rgb_loss.item()
                row['depth%d' % m] = depth_loss.item() if depth_loss is not None else 0.0
        if step % TC.LOG_EVERY == 0:
            for k in logs:
                logs[k].append(row[k])
            print(tag, step, row, '%.0f s' % (time.time() - t0), flush=True)
        if step > n_steps - TC.LOG_EVERY:
            tail.append([row['rgb0'], row['rgb1']])
    rgb, depth = ref_render_frame(nets, full, TC.CASCADE, dtype)
    mse = float(np.mean((rgb - full['rgb'].astype(np.float64)) ** 2))
    return logs, np.array(tail, np.float64), rgb, depth, mse


def gen_trajectory(dtypes=(torch.float32, torch.float64), modes=None):
    """the imported reference's training loop (ddp_train_nerf.py:417-498) on the BASELINE config-1
    scene -- one 64x64 frame, --cascade_samples 32,64, N_rand 256, 200 steps -- rgb-only and with the gt / mse depth
    term, on replayed batches and uniforms (tests/trajectory_common.py: seeds only).  Stored: the logged scalars every 25
    steps (level loss, rgb loss, in-loop PSNR = mse2psnr(rgb_loss), utils.py:31), their mean over the last 25 steps, and the
    frame rendered at the end with deterministic sampling (render_single_image) with its PSNR against the image.  The float64 run
    of the same reference code gives the noise floor of the float32 reference itself."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import trajectory_common as TC
    path = os.path.join(HERE, 'trajectory.npz')
    arrs = dict(np.load(path)) if (modes and os.path.exists(path)) else {}      # extend the fixture: earlier modes are kept as they are
    for mode in (modes or TC.MODES):
        for dtype in dtypes:
            tag = '%s.%s' % (mode, 'f32' if dtype == torch.float32 else 'f64')
            logs, tail, rgb, depth, mse = run_ref_trajectory(mode, dtype, TC.N_STEPS, 0, tag)
            for k, v in logs.items():
                arrs['%s.%s' % (tag, k)] = np.array(v, np.float64)
            arrs['%s.tail_rgb_mse' % tag] = tail
            arrs['%s.render_mse' % tag] = np.float64(mse)
            arrs['%s.render_psnr' % tag] = np.float64(RU.mse2psnr(mse))
            if dtype == torch.float32:
                arrs['%s.render_rgb' % tag] = rgb.astype(np.float32)
                arrs['%s.render_depth' % tag] = depth.astype(np.float32)
            print(tag, 'render psnr', arrs['%s.render_psnr' % tag], flush=True)
    save('trajectory', **arrs)


SEED_STEPS = 1000


def gen_trajectory_seeds(mode, seeds, part_dir):
    """the float32 reference trained SEED_STEPS steps on the config-1 scene for several seeds of the batch /
    uniform streams -- the reference's own seed spread at convergence, and paired reference runs for the HIP trainer's replay
    (tests/test_gpu_round5.py).  One part file per (mode, seed) so that several processes can share the ~2 h of CPU;
    `trajectory_seeds merge` collects them into tests/golden/trajectory_seeds.npz.
    The committed fixture: seeds 0-3 (kl) and 0, 2, 3, 4 (mse: seed 1 ran 6x slower than the others -- the torch CPU kernels in
    denormals -- and was replaced), TRAJ_THREADS = 3 (kl 0, 1; mse 0) and 2 (the rest).  torch's CPU reductions depend on the
    thread count in their last bits and a 1000-step run amplifies that to ~0.1 dB, so a regeneration with other thread counts
    yields other -- equally valid -- runs of the reference, not these bits (the 200-step trajectory.npz regenerates bit-identically
    with the 8 threads its branch sets)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.makedirs(part_dir, exist_ok=True)
    for seed in seeds:
        tag = '%s.s%d' % (mode, seed)
        logs, tail, rgb, depth, mse = run_ref_trajectory(mode, torch.float32, SEED_STEPS, seed, tag)
        arrs = {'%s.%s' % (tag, k): np.array(v, np.float64) for k, v in logs.items() if k in ('rgb0', 'rgb1', 'depth1')}
        arrs['%s.tail_rgb_mse' % tag] = tail
        arrs['%s.render_mse' % tag] = np.float64(mse)
        arrs['%s.render_psnr' % tag] = np.float64(RU.mse2psnr(mse))
        np.savez_compressed(os.path.join(part_dir, tag + '.npz'), **arrs)
        print(tag, 'render psnr', arrs['%s.render_psnr' % tag], flush=True)


def merge_trajectory_seeds(part_dir):
    arrs = {}
    for f in sorted(os.listdir(part_dir)):
        if f.endswith('.npz'):
            arrs.update(dict(np.load(os.path.join(part_dir, f))))
    arrs['steps'] = np.int64(SEED_STEPS)
    save('trajectory_seeds', **arrs)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'trajectory':      # ~30 min of CPU: on request only
        torch.set_num_threads(8)
        gen_trajectory(modes=tuple(sys.argv[2:]) or None)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == 'trajectory_seeds':    # ~15 min of CPU per (mode, seed): on request only
        part_dir = os.environ.get('TRAJ_PART_DIR', '/tmp/traj_seed_parts')
        if sys.argv[2] == 'merge':
            merge_trajectory_seeds(part_dir)
        else:                                                       # trajectory_seeds <mode> <seed> [<seed> ...]
            torch.set_num_threads(int(os.environ.get('TRAJ_THREADS', '4')))
            gen_trajectory_seeds(sys.argv[2], [int(x) for x in sys.argv[3:]], part_dir)
        sys.exit(0)
    gen_sampling()
    gen_embed()
    gen_depth2pts()
    nets = gen_params_and_mlp()
    gen_forward(nets)
    gen_losses()
    gen_grads(nets)
    gen_small_net_grads()
    gen_train_steps()
    gen_adam_unit()
    gen_rays()
    gen_autoexpo()
