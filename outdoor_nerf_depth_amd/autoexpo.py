"""Per-image auto-exposure of the reference's NerfNetWithAutoExpo.

Reference behaviour (nerf-methods/nerfplusplus):
  * ddp_model.py:161-192 -- with --optim_autoexpo every cascade level's net owns one parameter
    [0.5, 0.] per training image (keyed by remap_name(img_path), :150-158); forward returns
    ret['autoexpo'] = (scale = |p0| + 0.5, shift = p1) for the image of the batch;
  * ddp_train_nerf.py:472-479 -- rgb_pred = (ret['rgb'] - shift) / scale;
    rgb_loss = img2mse(rgb_pred, rgb_gt); loss = rgb_loss + lambda_autoexpo * (|scale - 1| + |shift|);
  * the parameters sit in the level's optimiser (Adam, lr = --lrate); a parameter without a gradient
    in a step (another image's) is skipped by Adam, so every image has its own step count.

Here this is host-side torch on a handful of scalars (what SURVEY row a11 prescribes): the HIP loss
kernel is reused by handing it gt' = scale * gt + shift -- then (rgb - gt') / scale = rgb_pred - gt, so
  rgb_loss = mse(rgb, gt') / scale^2,   d loss / d rgb = g_rgb(rgb, gt') / scale^2
and the two parameter gradients are small reductions over the [N, 3] batch.
"""
import torch


def remap_name(name):
    """ddp_model.py:150-158: '.' -> '-', keep the last three path components."""
    name = name.replace('.', '-')
    if name[-1] == '/':
        name = name[:-1]
    idx = name.rfind('/')
    for _ in range(2):
        if idx >= 0:
            idx = name[:idx].rfind('/')
    return name[idx + 1:]


class AutoExposure(object):
    """The autoexpo_params of ONE cascade level + their Adam state."""

    def __init__(self, img_names, device, lrate=5e-4, lambda_autoexpo=1.0, world_size=1,
                 betas=(0.9, 0.999), eps=1e-8):
        self.names = [remap_name(x) for x in img_names]
        self.index = {n: i for i, n in enumerate(self.names)}
        n = len(self.names)
        self.device = torch.device(device)
        self.params = torch.tensor([[0.5, 0.]] * n, dtype=torch.float32, device=self.device)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.steps = torch.zeros(n, dtype=torch.float64, device=self.device)     # per-parameter Adam step
        self.lrate, self.lam, self.world_size = lrate, lambda_autoexpo, world_size
        self.betas, self.eps = betas, eps

    def lookup(self, img_name):
        """Row of the batch's image, or None (then the plain rgb loss applies, ddp_model.py:186)."""
        if img_name is None:
            return None
        return self.index.get(remap_name(img_name))

    def scale_shift(self, idx):
        p = self.params[idx]
        return torch.abs(p[0]) + 0.5, p[1]

    def target(self, idx, rgb_gt):
        """gt' = scale * gt + shift: what the HIP loss kernel is given instead of rgb_gt."""
        scale, shift = self.scale_shift(idx)
        return rgb_gt * scale + shift

    def finish(self, idx, rgb, rgb_gt, scalars, g_rgb, lambda_depth):
        """After nerfpp_loss ran on (rgb, gt'): fix up scalars [loss, rgb_loss, depth_loss, n_valid] and
        g_rgb in place and return the gradient of the level's loss w.r.t. this image's parameter [2]
        (already divided by world_size like every other gradient)."""
        scale, shift = self.scale_shift(idx)
        inv2 = 1.0 / (scale * scale)
        g_rgb.mul_(inv2)
        rgb_loss = scalars[1] * inv2
        reg = self.lam * (torch.abs(scale - 1.0) + torch.abs(shift))
        depth_term = scalars[0] - scalars[1]                  # lambda_depth * depth_loss (0 for rgb-only)
        scalars[1] = rgb_loss
        scalars[0] = rgb_loss + reg + depth_term
        pred = (rgb - shift) / scale
        r = (pred - rgb_gt) * (2.0 / rgb.numel())             # d rgb_loss / d pred
        d_scale = -(r * pred).sum() / scale + self.lam * torch.sign(scale - 1.0)
        d_shift = -r.sum() / scale + self.lam * torch.sign(shift)
        g = torch.stack([d_scale * torch.sign(self.params[idx, 0]), d_shift])
        return g / self.world_size

    def apply(self, grad_rows, used):
        """Adam on the rows flagged in `used` (bool [n]); grad_rows [n, 2] (zeros elsewhere).
        torch.optim.Adam's update, with one step counter per parameter."""
        idx = torch.nonzero(used, as_tuple=False).reshape(-1)
        if idx.numel() == 0:
            return
        b1, b2 = self.betas
        g = grad_rows[idx]
        self.steps[idx] += 1
        t = self.steps[idx]
        m = torch.lerp(self.exp_avg[idx], g, 1.0 - b1)
        v = self.exp_avg_sq[idx] * b2 + (1.0 - b2) * g * g
        bias1 = 1.0 - torch.pow(torch.full_like(t, b1), t)
        bias2 = 1.0 - torch.pow(torch.full_like(t, b2), t)
        step_size = (self.lrate / bias1).to(torch.float32).unsqueeze(1)
        denom = v.sqrt() / bias2.sqrt().to(torch.float32).unsqueeze(1) + self.eps
        self.params[idx] = self.params[idx] - step_size * (m / denom)
        self.exp_avg[idx] = m
        self.exp_avg_sq[idx] = v

    # ---- checkpoints: keys / optimiser entries as the reference's state dicts hold them ------------
    def state_dict_entries(self, prefix='module.'):
        return [(prefix + 'autoexpo_params.' + n, self.params[i].clone()) for i, n in enumerate(self.names)]

    def load_state_dict_entries(self, sd):
        for i, n in enumerate(self.names):
            for key in ('module.autoexpo_params.' + n, 'autoexpo_params.' + n):
                if key in sd:
                    self.params[i].copy_(sd[key].to(self.device, torch.float32))

    def adam_entries(self):
        """[(state dict of one parameter) or None if it never had a gradient] in parameter order."""
        out = []
        steps = self.steps.tolist()
        for i in range(len(self.names)):
            out.append(None if steps[i] == 0 else {'step': torch.tensor(float(steps[i])),
                                                   'exp_avg': self.exp_avg[i].clone().cpu(),
                                                   'exp_avg_sq': self.exp_avg_sq[i].clone().cpu()})
        return out

    def load_adam_entries(self, entries):
        for i, st in enumerate(entries):
            if st is not None:
                self.steps[i] = float(st['step'])
                self.exp_avg[i].copy_(st['exp_avg'].to(self.device))
                self.exp_avg_sq[i].copy_(st['exp_avg_sq'].to(self.device))
