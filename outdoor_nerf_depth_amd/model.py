"""Parameter container mirroring the reference's NerfNetWithAutoExpo / NerfNet / MLPNet naming.

The HIP path keeps ONE flat float32 buffer per cascade level, laid out in the order of
`NerfNet.parameters()` (ddp_model.py:48-72, nerf_network.py:88-117): fg_net then bg_net, each
base_layers.{0..7}.0.{weight,bias}, sigma_layers.0.*, base_remap_layers.0.*, rgb_layers.0.*,
rgb_layers.2.*.  Views of that buffer are exposed under the reference's state-dict keys
(`module.nerf_net.fg_net.base_layers.0.0.weight`, ... -- the `module.` prefix comes from DDP,
ddp_train_nerf.py:323,646) so checkpoints round-trip with the reference's.
"""
from collections import OrderedDict

import torch

from . import _lib as L

POS_FREQS, DIR_FREQS = 10, 4            # --max_freq_log2 / --max_freq_log2_viewdirs defaults
FG_IN, BG_IN, DIR_IN = 63, 84, 27
NETDEPTH, NETWIDTH = 8, 256


def _mlp_specs(input_ch):
    """(name, shape) in MLPNet construction == parameters() order (nerf_network.py:88-117)."""
    specs = []
    dim = input_ch
    for i in range(NETDEPTH):
        specs.append(('base_layers.%d.0' % i, (NETWIDTH, dim)))
        dim = NETWIDTH
        if i == 4:                       # skips=[4]: layer 5 sees cat(input_pts, h)
            dim += input_ch
    specs.append(('sigma_layers.0', (1, dim)))
    specs.append(('base_remap_layers.0', (256, dim)))
    specs.append(('rgb_layers.0', (NETWIDTH // 2, 256 + DIR_IN)))
    specs.append(('rgb_layers.2', (3, NETWIDTH // 2)))
    return specs


def level_param_specs():
    """[(key, shape)] for one level, keys as in NerfNetWithAutoExpo.state_dict() (no DDP prefix)."""
    out = []
    for net, in_ch in (('fg_net', FG_IN), ('bg_net', BG_IN)):
        for name, shape in _mlp_specs(in_ch):
            out.append(('nerf_net.%s.%s.weight' % (net, name), shape))
            out.append(('nerf_net.%s.%s.bias' % (net, name), (shape[0],)))
    return out


def init_level_params(n_levels=2, seed=777):
    """Flat CPU tensors initialised exactly like create_nerf (ddp_train_nerf.py:308-325):
    torch.manual_seed(777), then nn.Linear's default init for every layer of net_0, then net_1."""
    torch.manual_seed(seed)
    levels = []
    for _ in range(n_levels):
        chunks = []
        for net, in_ch in (('fg_net', FG_IN), ('bg_net', BG_IN)):
            for name, shape in _mlp_specs(in_ch):
                lin = torch.nn.Linear(shape[1], shape[0])
                chunks += [lin.weight.detach().reshape(-1), lin.bias.detach().reshape(-1)]
        flat = torch.cat(chunks).contiguous()
        assert flat.numel() == L.LEVEL_PARAMS
        levels.append(flat)
    return levels


def state_dict_from_flat(flat, prefix='module.'):
    """OrderedDict of views into `flat` under the reference's state-dict keys."""
    sd, off = OrderedDict(), 0
    for key, shape in level_param_specs():
        n = 1
        for s in shape:
            n *= s
        sd[prefix + key] = flat[off:off + n].view(*shape)
        off += n
    assert off == flat.numel()
    return sd


def load_state_dict_into_flat(flat, sd):
    """Copy a reference-shaped state dict (with or without the DDP `module.` prefix) into flat."""
    views = state_dict_from_flat(flat, prefix='')
    for key, view in views.items():
        src = sd[key] if key in sd else sd['module.' + key]
        view.copy_(src.to(view.device, view.dtype))


def adam_state_dict(exp_avg, exp_avg_sq, step, lr=5e-4):
    """torch.optim.Adam.state_dict() layout for the level's parameters (one entry per tensor, in
    parameters() order) so `optim_m` in a checkpoint loads into the reference's optimiser."""
    state, off = {}, 0
    specs = level_param_specs()
    for i, (key, shape) in enumerate(specs):
        n = 1
        for s in shape:
            n *= s
        state[i] = {'step': torch.tensor(float(step)),
                    'exp_avg': exp_avg[off:off + n].view(*shape).clone(),
                    'exp_avg_sq': exp_avg_sq[off:off + n].view(*shape).clone()}
        off += n
    group = {'lr': lr, 'betas': (0.9, 0.999), 'eps': 1e-08, 'weight_decay': 0, 'amsgrad': False,
             'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False,
             'fused': None, 'params': list(range(len(specs)))}
    return {'state': state, 'param_groups': [group]}


def load_adam_state_dict(exp_avg, exp_avg_sq, sd):
    """Inverse of adam_state_dict; returns the step count."""
    off, step = 0, 0
    for i, (key, shape) in enumerate(level_param_specs()):
        n = 1
        for s in shape:
            n *= s
        st = sd['state'].get(i)
        if st is not None:
            exp_avg[off:off + n].copy_(st['exp_avg'].reshape(-1).to(exp_avg.device))
            exp_avg_sq[off:off + n].copy_(st['exp_avg_sq'].reshape(-1).to(exp_avg.device))
            step = int(float(st['step']))
        off += n
    return step
