"""CPU baseline: a PyTorch-CPU restatement of the NeRF++ depth-supervised training step.

THIS IS TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/` and the
`cpu_baseline` leg of `bench.py` may import it.  The reference's own Python cannot travel to the
GPU box, so the CPU number printed beside every GPU number is this module timed on the box's
host cores: the same arithmetic as `oracle/nerfpp_oracle.py` (the pinned numpy oracle), expressed
as torch tensor ops with autograd + torch.optim.Adam -- i.e. the way the reference itself runs on
CPU (multi-threaded sgemm, fused elementwise kernels), which the numpy port is not.

Parity status: pinned THROUGH the numpy oracle: `tests/test_oracle_golden.py::test_torch_cpu_*`
compares this module's forward outputs, losses, gradients and post-Adam parameters with
`nerfpp_oracle` (itself pinned by the reference's golden vectors) on seeded inputs.

Reference lines restated (paths relative to nerf-methods/nerfplusplus/):
  sampling         ddp_train_nerf.py:51-130, 438-465
  Embedder/MLPNet  nerf_network.py:42-60, 120-142
  NerfNet.forward  ddp_model.py:16-45, 74-147
  losses           utils.py:12-16, depth_loss.py:4-44, ddp_train_nerf.py:481-493
  optimiser        ddp_train_nerf.py:324, 497-498
"""
from collections import OrderedDict

import numpy as np
import torch

TINY = 1e-6           # utils.py:8
HUGE = 1e10           # utils.py:7
POS_FREQS, DIR_FREQS = 10, 4


# ------------------------------------------------------------------------------ sampling
def intersect_sphere(o, d):
    dd = (d * d).sum(-1)
    d1 = -(d * o).sum(-1) / dd
    p = o + d1[:, None] * d
    pn = (p * p).sum(-1)
    if bool((pn >= 1.).any()):
        raise Exception('Not all your cameras are bounded by the unit sphere; please make '
                        'sure the cameras are normalized properly!')
    return d1 + torch.sqrt(1. - pn) / torch.sqrt(dd)


def coarse_depths(near, far, S):
    step = (far - near) / (S - 1)
    i = torch.arange(S, dtype=torch.float32)
    fg = near[:, None] + i[None, :] * step[:, None]
    bg = torch.linspace(0., 1., S).expand(fg.shape).contiguous()
    return fg, bg


def perturb(z, t):
    mids = .5 * (z[:, 1:] + z[:, :-1])
    upper = torch.cat([mids, z[:, -1:]], -1)
    lower = torch.cat([z[:, :1], mids], -1)
    return lower + (upper - lower) * t


def sample_pdf(bins, weights, u):
    w = weights + TINY
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    M = weights.shape[-1]
    above = (u[:, :, None] >= cdf[:, None, :M]).sum(-1)
    below = torch.clamp(above - 1, min=0)
    c_lo, c_hi = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b_lo, b_hi = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    den = c_hi - c_lo
    den = torch.where(den < TINY, torch.ones_like(den), den)
    return b_lo + (u - c_lo) / den * (b_hi - b_lo + TINY)


def fine_depths(z_old, weights, u):
    mids = .5 * (z_old[:, 1:] + z_old[:, :-1])
    new = sample_pdf(mids, weights[:, 1:-1], u)
    return torch.sort(torch.cat([z_old, new], -1), -1)[0]


# ------------------------------------------------------------------------------ network
def embed(x, n_freqs):
    parts = [x]
    for k in range(n_freqs):
        parts += [torch.sin(x * float(2 ** k)), torch.cos(x * float(2 ** k))]
    return torch.cat(parts, -1)


def mlp(p, prefix, pts_enc, dir_enc):
    h = pts_enc
    for i in range(8):
        if i == 5:                                            # skip: cat(input, h) before layer 5
            h = torch.cat([pts_enc, h], -1)
        h = torch.relu(torch.addmm(p[prefix + 'base_layers.%d.0.bias' % i], h,
                                   p[prefix + 'base_layers.%d.0.weight' % i].t()))
    sigma = torch.abs(torch.addmm(p[prefix + 'sigma_layers.0.bias'], h, p[prefix + 'sigma_layers.0.weight'].t()))[:, 0]
    remap = torch.addmm(p[prefix + 'base_remap_layers.0.bias'], h, p[prefix + 'base_remap_layers.0.weight'].t())
    g = torch.relu(torch.addmm(p[prefix + 'rgb_layers.0.bias'], torch.cat([remap, dir_enc], -1),
                               p[prefix + 'rgb_layers.0.weight'].t()))
    rgb = torch.sigmoid(torch.addmm(p[prefix + 'rgb_layers.2.bias'], g, p[prefix + 'rgb_layers.2.weight'].t()))
    return rgb, sigma


def depth2pts_outside(o, d, depth):
    o, d = o[:, None, :], d[:, None, :]
    dd = (d * d).sum(-1)
    d1 = -(d * o).sum(-1) / dd
    p_mid = o + d1[..., None] * d
    pmn = torch.norm(p_mid, dim=-1)
    cosd = 1. / torch.sqrt(dd)
    d2 = torch.sqrt(1. - pmn * pmn) * cosd
    ps = o + (d1 + d2)[..., None] * d
    axis = torch.cross(o.expand_as(ps), ps, dim=-1)
    axis = axis / torch.norm(axis, dim=-1, keepdim=True)
    phi, theta = torch.asin(pmn), torch.asin(pmn * depth)
    ang = (phi - theta)[..., None]
    ps_b, axis_b = ps.expand(-1, depth.shape[1], -1), axis.expand(-1, depth.shape[1], -1)
    pn = ps_b * torch.cos(ang) + torch.cross(axis_b, ps_b, dim=-1) * torch.sin(ang) + \
        axis_b * (axis_b * ps_b).sum(-1, keepdim=True) * (1. - torch.cos(ang))
    pn = pn / torch.norm(pn, dim=-1, keepdim=True)
    return torch.cat([pn, depth[..., None]], -1), 1. / (depth + TINY) * torch.cos(theta) * cosd + d1


def nerf_forward(p, o, d, fg_far, fg_z, bg_z):
    N, S = fg_z.shape
    dn = torch.norm(d, dim=-1, keepdim=True)
    dir_enc = embed(d / dn, DIR_FREQS)
    # foreground
    pts = o[:, None, :] + fg_z[..., None] * d[:, None, :]
    rgb_s, sig = mlp(p, 'fg_net.', embed(pts, POS_FREQS).reshape(N * S, -1),
                     dir_enc[:, None, :].expand(N, S, -1).reshape(N * S, -1))
    rgb_s, sig = rgb_s.reshape(N, S, 3), sig.reshape(N, S)
    dists = dn * torch.cat([fg_z[:, 1:] - fg_z[:, :-1], fg_far[:, None] - fg_z[:, -1:]], -1)
    alpha = 1. - torch.exp(-sig * dists)
    T = torch.cumprod(1. - alpha + TINY, -1)
    lam = T[:, -1]
    w = alpha * torch.cat([torch.ones_like(T[:, :1]), T[:, :-1]], -1)
    fg_rgb, fg_depth = (w[..., None] * rgb_s).sum(-2), (w * fg_z).sum(-1)
    # background (inputs flipped along S so that order is near -> far)
    bpts, breal = depth2pts_outside(o, d, bg_z)
    benc = torch.flip(embed(bpts, POS_FREQS), dims=[1])
    bz = torch.flip(bg_z, dims=[1])
    breal = torch.flip(breal, dims=[1])
    brgb_s, bsig = mlp(p, 'bg_net.', benc.reshape(N * S, -1), dir_enc[:, None, :].expand(N, S, -1).reshape(N * S, -1))
    brgb_s, bsig = brgb_s.reshape(N, S, 3), bsig.reshape(N, S)
    bd = torch.cat([bz[:, :-1] - bz[:, 1:], torch.full_like(bz[:, :1], HUGE)], -1)
    ba = 1. - torch.exp(-bsig * bd)
    bT = torch.cumprod(1. - ba + TINY, -1)[:, :-1]
    bw = ba * torch.cat([torch.ones_like(bT[:, :1]), bT], -1)
    bg_rgb = lam[:, None] * (bw[..., None] * brgb_s).sum(-2)
    bg_depth = lam * (bw * breal).sum(-1)
    return OrderedDict([('rgb', fg_rgb + bg_rgb), ('fg_weights', w), ('bg_weights', bw), ('fg_dists', dists),
                        ('fg_rgb', fg_rgb), ('fg_depth', fg_depth), ('bg_rgb', bg_rgb), ('bg_depth', bg_depth),
                        ('bg_lambda', lam), ('depth', fg_depth + bg_depth)])


# ------------------------------------------------------------------------------ losses
def depth_loss(kind, ret, depth_sup, fg_z, fg_far, sigma):
    if kind == 'kl':
        mask = (depth_sup > 0) & (depth_sup < fg_far)
        t = -torch.log(ret['fg_weights'] + 1e-5) * \
            torch.exp(-(fg_z - depth_sup[:, None]) ** 2 / (2 * sigma)) * ret['fg_dists']
        return t[mask].sum(-2).mean()
    mask = depth_sup > 0
    diff = (depth_sup - ret['depth'])[mask]
    return (diff * diff).mean() if kind == 'mse' else diff.abs().mean()


class TorchCpuTrainer(object):
    """Both cascade levels, forward + autograd backward + Adam, float32, all host threads."""

    def __init__(self, level_params, cascade_samples=(64, 128), use_depth=True, depth_loss_type='mse',
                 lambda_depth=0.1, depth_sigma_scaled=0.01, lr=5e-4):
        self.levels = [OrderedDict((k, torch.tensor(np.asarray(v, np.float32), requires_grad=True))
                                   for k, v in lv.items()) for lv in level_params]
        self.optims = [torch.optim.Adam(list(lv.values()), lr=lr) for lv in self.levels]
        self.cascade_samples = cascade_samples
        self.use_depth, self.kind, self.lam, self.sigma = use_depth, depth_loss_type, lambda_depth, depth_sigma_scaled

    def train_step(self, batch, uniforms, z_override=None):
        """z_override: optional {level: (fg_z, bg_z)} replacing that level's sample depths (tests: the fine
        depths are an ill-conditioned function of the coarse weights, so level 1 is compared on equal depths)."""
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32))
        o, d, rgb_gt = T(batch['ray_o']), T(batch['ray_d']), T(batch['rgb'])
        sup = T(batch['depth_sup']) if self.use_depth else None
        far = intersect_sphere(o, d)
        logs, ret = [], None
        for m, S in enumerate(self.cascade_samples):
            if m == 0:
                fg_z, bg_z = coarse_depths(T(batch['min_depth']), far, S)
                fg_z, bg_z = perturb(fg_z, T(uniforms['t_fg'])), perturb(bg_z, T(uniforms['t_bg']))
            else:
                fg_z = fine_depths(fg_z, ret['fg_weights'].detach(), T(uniforms['u_fg']))
                bg_z = fine_depths(bg_z, ret['bg_weights'].detach(), T(uniforms['u_bg']))
            if z_override and m in z_override:
                fg_z, bg_z = T(z_override[m][0]), T(z_override[m][1])
            ret = nerf_forward(self.levels[m], o, d, far, fg_z, bg_z)
            rgb_loss = ((ret['rgb'] - rgb_gt) ** 2).mean()
            loss = rgb_loss
            dl = None
            if self.use_depth:
                dl = depth_loss(self.kind, ret, sup, fg_z, far, self.sigma)
                loss = rgb_loss + self.lam * dl
            self.optims[m].zero_grad()
            loss.backward()
            grads = OrderedDict((k, v.grad.detach().numpy().copy()) for k, v in self.levels[m].items())
            self.optims[m].step()
            logs.append(dict(loss=float(loss.detach()), rgb_loss=float(rgb_loss.detach()),
                             depth_loss=None if dl is None else float(dl.detach()),
                             ret={k: v.detach().numpy() for k, v in ret.items()}, grads=grads,
                             fg_z=fg_z.numpy(), bg_z=bg_z.numpy()))
        return logs

    def params(self, m):
        return OrderedDict((k, v.detach().numpy()) for k, v in self.levels[m].items())
