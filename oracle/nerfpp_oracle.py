"""CPU oracle: a numpy restatement of the NeRF++ depth-supervised render/train path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it -- as the checker (or as the timed CPU
baseline), never as the thing shipped.  The product path (outdoor_nerf_depth_amd/) calls the
HIP library through the C ABI and fails loudly when that library is missing.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the reference's own PyTorch
modules in the build container (never on the GPU box) and writes `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function here against those vectors.  The reference
ships no tests / golden vectors of its own for this path.

Every function cites the reference lines it restates (paths relative to
nerf-methods/nerfplusplus/ of cwchenwang/outdoor-nerf-depth).

Arithmetic conventions (the documented operation order the HIP kernels mirror):
  * everything is float32 unless stated;
  * `sample_pdf`: sum(w) = sequential float64 accumulation rounded once to float32;
    cdf = sequential float64 cumsum, each prefix rounded to float32 (this is what torch's CPU
    cumsum does: accumulate in double, store float);  `above_inds` = #{k < M : u >= cdf[k]};
  * compositing `cumprod` = sequential float64 product, each prefix rounded to float32
    (torch CPU behaviour);  the HIP kernel uses a float32 wave scan and is compared with a
    tolerance, only the integer bins are bit-exact quantities.
"""
from collections import OrderedDict

import numpy as np

f32 = np.float32
TINY_NUMBER = f32(1e-6)     # utils.py:8
HUGE_NUMBER = f32(1e10)     # utils.py:7


# --------------------------------------------------------------------------------------
# a1  intersect_sphere                                           ddp_train_nerf.py:51-66
# --------------------------------------------------------------------------------------
def _sum3(x):
    """sum over a trailing axis of length 3/4 in left-to-right float32 order."""
    acc = x[..., 0].astype(f32)
    for k in range(1, x.shape[-1]):
        acc = (acc + x[..., k]).astype(f32)
    return acc


def intersect_sphere(ray_o, ray_d):
    ray_o = np.asarray(ray_o, f32)
    ray_d = np.asarray(ray_d, f32)
    d1 = -_sum3(ray_d * ray_o) / _sum3(ray_d * ray_d)
    p = ray_o + d1[..., None] * ray_d
    ray_d_cos = f32(1.) / np.sqrt(_sum3(ray_d * ray_d))
    p_norm_sq = _sum3(p * p)
    if (p_norm_sq >= 1.).any():
        raise Exception('Not all your cameras are bounded by the unit sphere; please make '
                        'sure the cameras are normalized properly!')
    d2 = np.sqrt(f32(1.) - p_norm_sq) * ray_d_cos
    return (d1 + d2).astype(f32)


# --------------------------------------------------------------------------------------
# a2  coarse depths                           ddp_train_nerf.py:438-449 (train), :166-175
# --------------------------------------------------------------------------------------
def torch_linspace(start, end, steps):
    """float32 torch.linspace on CPU (ATen RangeFactories.cpp): step = (end-start)/(steps-1) in
    float32, then a FUSED multiply-add per element -- fma(step, i, start) for the first half,
    fma(-step, steps-1-i, end) for the second half.  The float64 expression below is exact
    before its single rounding, so it equals the fma."""
    start, end = f32(start), f32(end)
    step = np.float64(f32((end - start) / f32(steps - 1)))
    idx = np.arange(steps)
    half = steps // 2
    lo = (np.float64(start) + step * idx).astype(f32)
    hi = (np.float64(end) - step * (steps - idx - 1)).astype(f32)
    return np.where(idx < half, lo, hi).astype(f32)


def coarse_depths(fg_near, fg_far, n_samples):
    """fg_depth[i] = near + i*step (an int*float32 product, not a running add); bg = linspace."""
    fg_near = np.asarray(fg_near, f32)
    fg_far = np.asarray(fg_far, f32)
    step = ((fg_far - fg_near) / f32(n_samples - 1)).astype(f32)
    i = np.arange(n_samples).astype(f32)
    fg_depth = (fg_near[..., None] + i * step[..., None]).astype(f32)
    bg_depth = np.broadcast_to(torch_linspace(0., 1., n_samples), fg_depth.shape).copy()
    return fg_depth, bg_depth


# --------------------------------------------------------------------------------------
# a3  perturb_samples                                            ddp_train_nerf.py:69-78
# --------------------------------------------------------------------------------------
def perturb_samples(z_vals, t_rand):
    """t_rand replaces torch.rand_like(z_vals) so kernels and oracle see the same uniforms."""
    z_vals = np.asarray(z_vals, f32)
    mids = (f32(.5) * (z_vals[..., 1:] + z_vals[..., :-1])).astype(f32)
    upper = np.concatenate([mids, z_vals[..., -1:]], -1)
    lower = np.concatenate([z_vals[..., 0:1], mids], -1)
    return (lower + (upper - lower) * np.asarray(t_rand, f32)).astype(f32)


def philox_uniform(seed, step, stream_id, n):
    """The sampling uniforms of the in-kernel generator (csrc/nerfpp_common.h: philox_uniform):
    Philox4x32-10, key = (seed lo, seed hi), counter = (i lo, i hi, stream_id, step lo); the value is
    (word0 >> 8) * 2^-24.  Stream ids follow the reference's RNG consumption order per step
    (SURVEY 8c): 0 = rand_like(fg_z), 1 = rand_like(bg_z) (ddp_train_nerf.py:71 via :444,449),
    2 = fg sample_pdf u, 3 = bg sample_pdf u (:104 via :455,463).  Returns float32 [n]."""
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    i = np.arange(n, dtype=np.uint64)
    c0, c1 = i & mask, i >> np.uint64(32)
    c2 = np.full(n, stream_id, np.uint64)
    c3 = np.full(n, int(step) & 0xFFFFFFFF, np.uint64)
    k0, k1 = np.uint64(int(seed) & 0xFFFFFFFF), np.uint64((int(seed) >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        h0, l0 = p0 >> np.uint64(32), p0 & mask
        h1, l1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = h1 ^ c1 ^ k0, l1, h0 ^ c3 ^ k1, l0
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return ((c0 >> np.uint64(8)).astype(np.float32) * f32(2.0 ** -24)).astype(f32)


def philox_words(seed, step, stream_id, idx):
    """raw 32-bit word 0 of the same generator at the given 64-bit counters (csrc/nerfpp_common.h: philox_word)"""
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    i = np.asarray(idx, dtype=np.uint64)
    c0, c1 = i & mask, i >> np.uint64(32)
    c2 = np.full(i.shape, stream_id, np.uint64)
    c3 = np.full(i.shape, int(step) & 0xFFFFFFFF, np.uint64)
    k0, k1 = np.uint64(int(seed) & 0xFFFFFFFF), np.uint64((int(seed) >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        h0, l0 = p0 >> np.uint64(32), p0 & mask
        h1, l1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = h1 ^ c1 ^ k0, l1, h0 ^ c3 ^ k1, l0
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return c0


def sample_pixels(n_pixels, n_rays, seed, step):
    """nerf_sample_ray_split.py:178 `np.random.choice(H * W, size=(N_rand,), replace=False)` as the HIP path draws it
    (nerfpp_sample_pixels): element i owns the draws d = 0, 1, ... of Philox stream 4 at counter i + 2^20 d, each mapped to
    [0, n_pixels) by multiply-shift with Lemire's rejection (exactly uniform); x_i = its first draw that differs from every
    x_j, j < i -- i.e. sequential sampling without replacement, uniform over ordered tuples of distinct pixels."""
    n = int(n_pixels)
    thresh = ((1 << 32) - n) % n
    taken, out = set(), []
    for i in range(n_rays):
        d = 0
        while True:
            w = int(philox_words(seed, step, 4, np.array([i + (d << 20)], np.uint64))[0])
            d += 1
            m = w * n
            if (m & 0xFFFFFFFF) < thresh:
                continue
            v = m >> 32
            if v not in taken:
                break
        taken.add(v)
        out.append(v)
    return np.array(out, np.int64)


def step_uniforms(seed, step, n, S0, S1):
    """The four uniform tensors one training step of the HIP trainer draws in its kernels."""
    return dict(t_fg=philox_uniform(seed, step, 0, n * S0).reshape(n, S0),
                t_bg=philox_uniform(seed, step, 1, n * S0).reshape(n, S0),
                u_fg=philox_uniform(seed, step, 2, n * S1).reshape(n, S1),
                u_bg=philox_uniform(seed, step, 3, n * S1).reshape(n, S1))


# --------------------------------------------------------------------------------------
# a4  sample_pdf                                                ddp_train_nerf.py:81-130
# --------------------------------------------------------------------------------------
def sample_pdf_cdf(weights):
    """cdf[..., M+1] with the documented summation order (see module docstring)."""
    w = (np.asarray(weights, f32) + TINY_NUMBER).astype(f32)
    wsum = np.cumsum(w.astype(np.float64), -1)[..., -1:].astype(f32)      # sequential f64
    pdf = (w / wsum).astype(f32)
    cdf = np.cumsum(pdf.astype(np.float64), -1).astype(f32)              # sequential f64
    return np.concatenate([np.zeros_like(cdf[..., :1]), cdf], -1)


def sample_pdf(bins, weights, u):
    """u: the uniforms ([..., N_samples]); det mode passes torch_linspace(0,1,N_samples).
    Returns (samples float32, above_inds int64)."""
    bins = np.asarray(bins, f32)
    u = np.asarray(u, f32)
    M = weights.shape[-1]
    cdf = sample_pdf_cdf(weights)                                          # [..., M+1]
    above = np.sum(u[..., :, None] >= cdf[..., None, :M], -1).astype(np.int64)
    below = np.maximum(above - 1, 0)
    cdf_lo = np.take_along_axis(cdf, below, -1)
    cdf_hi = np.take_along_axis(cdf, above, -1)
    bin_lo = np.take_along_axis(bins, below, -1)
    bin_hi = np.take_along_axis(bins, above, -1)
    denom = (cdf_hi - cdf_lo).astype(f32)
    denom = np.where(denom < TINY_NUMBER, f32(1.), denom).astype(f32)
    t = ((u - cdf_lo) / denom).astype(f32)
    samples = (bin_lo + t * (bin_hi - bin_lo + TINY_NUMBER)).astype(f32)
    return samples, above


def fine_depths(z_old, weights, u):
    """a4+a5: mids -> sample_pdf(weights[..., 1:-1]) -> sort(cat)  ddp_train_nerf.py:450-465.
    `weights` is ret['fg_weights'] / ret['bg_weights'] as returned by the level-0 forward
    (for bg that is the FLIPPED order; the reference pairs it with ascending bins as is)."""
    z_old = np.asarray(z_old, f32)
    mids = (f32(.5) * (z_old[..., 1:] + z_old[..., :-1])).astype(f32)
    samples, above = sample_pdf(mids, np.asarray(weights, f32)[..., 1:-1], u)
    merged = np.sort(np.concatenate([z_old, samples], -1), -1)
    return merged, samples, above


# --------------------------------------------------------------------------------------
# a6  Embedder                                                     nerf_network.py:11-60
# --------------------------------------------------------------------------------------
def embed(x, n_freqs):
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]"""
    x = np.asarray(x, f32)
    out = [x]
    for k in range(n_freqs):
        arg = (x * f32(2. ** k)).astype(f32)
        out.append(np.sin(arg).astype(f32))
        out.append(np.cos(arg).astype(f32))
    return np.concatenate(out, -1)


# --------------------------------------------------------------------------------------
# a9  depth2pts_outside                                               ddp_model.py:16-45
# --------------------------------------------------------------------------------------
def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1).astype(f32)


def depth2pts_outside(ray_o, ray_d, depth):
    """ray_o, ray_d: [N,3]; depth: [N,S] (inverse distance). Returns pts [N,S,4], depth_real [N,S]."""
    ray_o = np.asarray(ray_o, f32)[:, None, :]
    ray_d = np.asarray(ray_d, f32)[:, None, :]
    depth = np.asarray(depth, f32)
    d1 = -_sum3(ray_d * ray_o) / _sum3(ray_d * ray_d)                     # [N,1]
    p_mid = ray_o + d1[..., None] * ray_d
    p_mid_norm = np.sqrt(_sum3(p_mid * p_mid))
    ray_d_cos = f32(1.) / np.sqrt(_sum3(ray_d * ray_d))
    d2 = np.sqrt(f32(1.) - p_mid_norm * p_mid_norm) * ray_d_cos
    p_sphere = ray_o + (d1 + d2)[..., None] * ray_d                       # [N,1,3]
    rot_axis = _cross(ray_o, p_sphere)
    rot_axis = rot_axis / np.sqrt(_sum3(rot_axis * rot_axis))[..., None]
    phi = np.arcsin(p_mid_norm)                                           # [N,1]
    theta = np.arcsin(p_mid_norm * depth)                                 # [N,S]
    rot_angle = (phi - theta)[..., None].astype(f32)                      # [N,S,1]
    cos_a, sin_a = np.cos(rot_angle), np.sin(rot_angle)
    p_new = p_sphere * cos_a + _cross(rot_axis, p_sphere) * sin_a + \
        rot_axis * _sum3(rot_axis * p_sphere)[..., None] * (f32(1.) - cos_a)
    p_new = (p_new / np.sqrt(_sum3(p_new * p_new))[..., None]).astype(f32)
    pts = np.concatenate([p_new, depth[..., None]], -1).astype(f32)
    depth_real = f32(1.) / (depth + TINY_NUMBER) * np.cos(theta) * ray_d_cos + d1
    return pts, depth_real.astype(f32)


# --------------------------------------------------------------------------------------
# a7  MLPNet                                                      nerf_network.py:70-142
# --------------------------------------------------------------------------------------
def mlp_param_names(netdepth=8):
    """Parameter order == the reference's `MLPNet.parameters()` / state_dict order."""
    names = []
    for i in range(netdepth):
        names += ['base_layers.%d.0.weight' % i, 'base_layers.%d.0.bias' % i]
    names += ['sigma_layers.0.weight', 'sigma_layers.0.bias',
              'base_remap_layers.0.weight', 'base_remap_layers.0.bias',
              'rgb_layers.0.weight', 'rgb_layers.0.bias',
              'rgb_layers.2.weight', 'rgb_layers.2.bias']
    return names


def mlp_param_shapes(input_ch, input_ch_viewdirs, netdepth=8, netwidth=256, skips=(4,)):
    shapes = OrderedDict()
    dim = input_ch
    for i in range(netdepth):
        shapes['base_layers.%d.0.weight' % i] = (netwidth, dim)
        shapes['base_layers.%d.0.bias' % i] = (netwidth,)
        dim = netwidth
        if i in skips and i != netdepth - 1:
            dim += input_ch
    shapes['sigma_layers.0.weight'] = (1, dim)
    shapes['sigma_layers.0.bias'] = (1,)
    shapes['base_remap_layers.0.weight'] = (256, dim)
    shapes['base_remap_layers.0.bias'] = (256,)
    shapes['rgb_layers.0.weight'] = (netwidth // 2, 256 + input_ch_viewdirs)
    shapes['rgb_layers.0.bias'] = (netwidth // 2,)
    shapes['rgb_layers.2.weight'] = (3, netwidth // 2)
    shapes['rgb_layers.2.bias'] = (3,)
    return shapes


def round_bf16(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32."""
    u = np.ascontiguousarray(x, f32).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(f32)


def _linear(x, W, b, bf16=False):
    if bf16:
        return (round_bf16(x) @ round_bf16(W).T + b).astype(f32)
    return (x @ W.T + b).astype(f32)


def mlp_forward(p, inp, input_ch, input_ch_viewdirs, netdepth=8, skips=(4,), cache=None, bf16=False):
    """p: dict name -> array.  inp: [R, input_ch + input_ch_viewdirs].  Returns rgb [R,3],
    sigma [R].  If `cache` is a dict the intermediates needed by mlp_backward are stored.
    bf16=True models the single-pass bf16 precision of the HIP kernels (NERFPP_PREC_BF16): both operands of
    every linear layer are rounded to bfloat16, products accumulate in float32, biases are added in float32
    -- a structural check for that mode at ~1e-3 instead of the 3e-2 a float32 comparison allows."""
    if callable(bf16):          # operand-format studies (tools/operand_format_study.py): bf16(x, W) -> x @ W.T in that format
        _lin = lambda x, W, b: (bf16(x, W) + b).astype(f32)
    elif bf16:
        _lin = lambda x, W, b: _linear(x, W, b, True)
    else:
        _lin = _linear
    return _mlp_forward(p, inp, input_ch, input_ch_viewdirs, netdepth, skips, cache, _lin)


def _mlp_forward(p, inp, input_ch, input_ch_viewdirs, netdepth, skips, cache, _linear):
    input_pts = inp[:, :input_ch]
    input_dirs = inp[:, -input_ch_viewdirs:]
    acts_in = []
    pre = []
    base = input_pts
    for i in range(netdepth):
        if i > 0 and (i - 1) in skips:
            base = np.concatenate([input_pts, base], -1)
        acts_in.append(base)
        z = _linear(base, p['base_layers.%d.0.weight' % i], p['base_layers.%d.0.bias' % i])
        pre.append(z)
        base = np.maximum(z, f32(0))
    sigma_raw = _linear(base, p['sigma_layers.0.weight'], p['sigma_layers.0.bias'])[:, 0]
    sigma = np.abs(sigma_raw)
    remap = _linear(base, p['base_remap_layers.0.weight'], p['base_remap_layers.0.bias'])
    rgb_in = np.concatenate([remap, input_dirs], -1)
    g_pre = _linear(rgb_in, p['rgb_layers.0.weight'], p['rgb_layers.0.bias'])
    g = np.maximum(g_pre, f32(0))
    rgb_pre = _linear(g, p['rgb_layers.2.weight'], p['rgb_layers.2.bias'])
    rgb = (f32(1.) / (f32(1.) + np.exp(-rgb_pre))).astype(f32)
    if cache is not None:
        cache.update(acts_in=acts_in, pre=pre, h_last=base, sigma_raw=sigma_raw, rgb_in=rgb_in,
                     g_pre=g_pre, g=g, rgb=rgb, input_ch=input_ch)
    return rgb, sigma.astype(f32)


def mlp_backward(p, cache, d_rgb, d_sigma, netdepth=8, skips=(4,), bf16=False):
    """Closed-form backward of mlp_forward (the reference relies on autograd).
    d_rgb [R,3] (w.r.t. the post-sigmoid colour), d_sigma [R] (w.r.t. sigma=|raw|).
    Returns dict name -> gradient.  bf16=True: every GEMM operand (dZ, weights, saved activations)
    rounded to bfloat16, float32 accumulation (the NERFPP_PREC_BF16 backward)."""
    if bf16:
        class _B(dict):
            def __getitem__(self, k):
                v = dict.__getitem__(self, k)
                return [round_bf16(a) for a in v] if isinstance(v, list) else \
                    (round_bf16(v) if k in ('g', 'rgb_in', 'h_last') else v)
        cache = _B(cache)
        p = {k: (round_bf16(v) if k.endswith('weight') else v) for k, v in p.items()}
        rb = round_bf16
    else:
        rb = lambda a: a
    g = OrderedDict()
    rgb = cache['rgb']
    d_rgb_pre = rb((d_rgb * rgb * (f32(1.) - rgb)).astype(f32))
    g['rgb_layers.2.weight'] = d_rgb_pre.T @ cache['g']
    g['rgb_layers.2.bias'] = d_rgb_pre.sum(0)
    d_g = d_rgb_pre @ p['rgb_layers.2.weight']
    d_g_pre = rb((d_g * (cache['g_pre'] > 0)).astype(f32))
    g['rgb_layers.0.weight'] = d_g_pre.T @ cache['rgb_in']
    g['rgb_layers.0.bias'] = d_g_pre.sum(0)
    d_remap = rb((d_g_pre @ p['rgb_layers.0.weight'])[:, :256])
    g['base_remap_layers.0.weight'] = d_remap.T @ cache['h_last']
    g['base_remap_layers.0.bias'] = d_remap.sum(0)
    d_sigma_raw = rb((d_sigma * np.sign(cache['sigma_raw'])).astype(f32))
    g['sigma_layers.0.weight'] = d_sigma_raw[None, :] @ cache['h_last']
    g['sigma_layers.0.bias'] = d_sigma_raw.sum(keepdims=True)
    d_h = d_remap @ p['base_remap_layers.0.weight'] + \
        d_sigma_raw[:, None] * p['sigma_layers.0.weight']
    input_ch = cache['input_ch']
    for i in reversed(range(netdepth)):
        d_z = rb((d_h * (cache['pre'][i] > 0)).astype(f32))
        g['base_layers.%d.0.weight' % i] = d_z.T @ cache['acts_in'][i]
        g['base_layers.%d.0.bias' % i] = d_z.sum(0)
        if i == 0:
            break
        d_in = d_z @ p['base_layers.%d.0.weight' % i]
        if (i - 1) in skips:
            d_in = d_in[:, input_ch:]          # the raw-input part needs no gradient
        d_h = d_in
    return OrderedDict((k, np.asarray(g[k], f32).reshape(p[k].shape)) for k in p)


# --------------------------------------------------------------------------------------
# a8 + a10  NerfNet.forward                                          ddp_model.py:74-147
# --------------------------------------------------------------------------------------
POS_FREQS = 10      # --max_freq_log2        ddp_train_nerf.py:707
DIR_FREQS = 4       # --max_freq_log2_viewdirs  :709
FG_IN, BG_IN, DIR_IN = 3 + 3 * 2 * POS_FREQS, 4 + 4 * 2 * POS_FREQS, 3 + 3 * 2 * DIR_FREQS


def _cumprod_f64(x):
    return np.cumprod(x.astype(np.float64), -1).astype(f32)


def nerf_forward(params, ray_o, ray_d, fg_z_max, fg_z_vals, bg_z_vals, cache=None, bf16=False):
    """params: {'fg_net.<name>': arr, 'bg_net.<name>': arr}.  Returns the reference's
    OrderedDict (same keys, same order).  bf16: see mlp_forward."""
    ray_o = np.asarray(ray_o, f32)
    ray_d = np.asarray(ray_d, f32)
    fg_z_max = np.asarray(fg_z_max, f32)
    fg_z_vals = np.asarray(fg_z_vals, f32)
    bg_z_vals = np.asarray(bg_z_vals, f32)
    pf = {k[len('fg_net.'):]: v for k, v in params.items() if k.startswith('fg_net.')}
    pb = {k[len('bg_net.'):]: v for k, v in params.items() if k.startswith('bg_net.')}
    N, S = fg_z_vals.shape
    ray_d_norm = np.sqrt(_sum3(ray_d * ray_d))[:, None]                    # [N,1]
    viewdirs = (ray_d / ray_d_norm).astype(f32)
    dir_enc = embed(viewdirs, DIR_FREQS)                                   # [N,27]

    # ---- foreground                                                ddp_model.py:86-105
    fg_pts = ray_o[:, None, :] + fg_z_vals[..., None] * ray_d[:, None, :]
    inp = np.concatenate([embed(fg_pts, POS_FREQS),
                          np.broadcast_to(dir_enc[:, None, :], (N, S, DIR_IN))], -1)
    cf = {} if cache is not None else None
    fg_rgb_s, fg_sigma = mlp_forward(pf, inp.reshape(N * S, -1), FG_IN, DIR_IN, cache=cf, bf16=bf16)
    fg_rgb_s = fg_rgb_s.reshape(N, S, 3)
    fg_sigma = fg_sigma.reshape(N, S)
    fg_dists = fg_z_vals[..., 1:] - fg_z_vals[..., :-1]
    fg_dists = (ray_d_norm * np.concatenate(
        [fg_dists, fg_z_max[:, None] - fg_z_vals[..., -1:]], -1)).astype(f32)
    fg_e = np.exp(-fg_sigma * fg_dists).astype(f32)
    fg_alpha = (f32(1.) - fg_e).astype(f32)
    fg_q = (f32(1.) - fg_alpha + TINY_NUMBER).astype(f32)
    T = _cumprod_f64(fg_q)
    bg_lambda = T[..., -1]
    fg_T = np.concatenate([np.ones_like(T[..., :1]), T[..., :-1]], -1)
    fg_weights = (fg_alpha * fg_T).astype(f32)
    fg_rgb_map = np.sum(fg_weights[..., None] * fg_rgb_s, -2).astype(f32)
    fg_depth_map = np.sum(fg_weights * fg_z_vals, -1).astype(f32)

    # ---- background                                               ddp_model.py:107-128
    S_bg = bg_z_vals.shape[-1]
    bg_pts, bg_depth_real = depth2pts_outside(ray_o, ray_d, bg_z_vals)
    inp = np.concatenate([embed(bg_pts, POS_FREQS),
                          np.broadcast_to(dir_enc[:, None, :], (N, S_bg, DIR_IN))], -1)
    inp = inp[:, ::-1, :]                                                  # flip along S
    bg_z_f = bg_z_vals[:, ::-1]
    bg_dists = bg_z_f[..., :-1] - bg_z_f[..., 1:]
    bg_dists = np.concatenate([bg_dists, np.full_like(bg_dists[..., :1], HUGE_NUMBER)], -1)
    cb = {} if cache is not None else None
    bg_rgb_s, bg_sigma = mlp_forward(pb, np.ascontiguousarray(inp).reshape(N * S_bg, -1),
                                     BG_IN, DIR_IN, cache=cb, bf16=bf16)
    bg_rgb_s = bg_rgb_s.reshape(N, S_bg, 3)
    bg_sigma = bg_sigma.reshape(N, S_bg)
    with np.errstate(over='ignore'):
        bg_e = np.exp(-bg_sigma * bg_dists).astype(f32)
    bg_alpha = (f32(1.) - bg_e).astype(f32)
    bg_q = (f32(1.) - bg_alpha + TINY_NUMBER).astype(f32)
    Tb = _cumprod_f64(bg_q)[..., :-1]
    bg_T = np.concatenate([np.ones_like(Tb[..., :1]), Tb], -1)
    bg_weights = (bg_alpha * bg_T).astype(f32)
    bg_depth_real_f = bg_depth_real[:, ::-1]
    bg_rgb_raw = np.sum(bg_weights[..., None] * bg_rgb_s, -2).astype(f32)
    bg_depth_raw = np.sum(bg_weights * bg_depth_real_f, -1).astype(f32)

    # ---- composite                                                ddp_model.py:130-134
    bg_rgb_map = (bg_lambda[:, None] * bg_rgb_raw).astype(f32)
    bg_depth_map = (bg_lambda * bg_depth_raw).astype(f32)
    rgb_map = (fg_rgb_map + bg_rgb_map).astype(f32)
    depth_map = (fg_depth_map + bg_depth_map).astype(f32)

    if cache is not None:
        cache.update(fg=cf, bg=cb, pf=pf, pb=pb, fg_rgb_s=fg_rgb_s, fg_e=fg_e, fg_alpha=fg_alpha,
                     fg_q=fg_q, fg_T=fg_T, fg_dists=fg_dists, fg_weights=fg_weights,
                     fg_z=fg_z_vals, bg_lambda=bg_lambda, bg_rgb_s=bg_rgb_s, bg_e=bg_e,
                     bg_alpha=bg_alpha, bg_q=bg_q, bg_T=bg_T, bg_dists=bg_dists,
                     bg_weights=bg_weights, bg_depth_real_f=bg_depth_real_f,
                     bg_rgb_raw=bg_rgb_raw, bg_depth_raw=bg_depth_raw,
                     fg_sigma=fg_sigma, bg_sigma=bg_sigma)
    return OrderedDict([('rgb', rgb_map), ('fg_weights', fg_weights), ('bg_weights', bg_weights),
                        ('fg_dists', fg_dists), ('fg_rgb', fg_rgb_map), ('fg_depth', fg_depth_map),
                        ('bg_rgb', bg_rgb_map), ('bg_depth', bg_depth_map),
                        ('bg_lambda', bg_lambda), ('depth', depth_map)])


def _suffix_excl(x):
    """out[i] = sum_{k>i} x[k] along the last axis."""
    c = np.cumsum(x[..., ::-1].astype(np.float64), -1)[..., ::-1]
    return np.concatenate([c[..., 1:], np.zeros_like(c[..., :1])], -1)


def composite_backward(cache, g_rgb, g_depth, g_fg_weights=None):
    """SURVEY.md Appendix A.  Returns per-sample (d_rgb, d_sigma) for fg and bg (bg in the
    FLIPPED order the bg MLP saw its inputs in)."""
    g_rgb = np.asarray(g_rgb, f32)
    g_depth = np.asarray(g_depth, f32)
    lam = cache['bg_lambda']
    # foreground
    w, T, q, e = cache['fg_weights'], cache['fg_T'], cache['fg_q'], cache['fg_e']
    g_w = (cache['fg_rgb_s'] * g_rgb[:, None, :]).sum(-1) + g_depth[:, None] * cache['fg_z']
    if g_fg_weights is not None:
        g_w = g_w + g_fg_weights
    g_lam = (g_rgb * cache['bg_rgb_raw']).sum(-1) + g_depth * cache['bg_depth_raw']
    d_a = g_w * T - (_suffix_excl(g_w * w) + (g_lam * lam)[:, None]) / q
    fg_d_sigma = (d_a * cache['fg_dists'] * e).astype(f32)
    fg_d_rgb = (w[..., None] * g_rgb[:, None, :]).astype(f32)
    # background
    w, T, q, e = cache['bg_weights'], cache['bg_T'], cache['bg_q'], cache['bg_e']
    gC = lam[:, None] * g_rgb
    gD = lam * g_depth
    g_w = (cache['bg_rgb_s'] * gC[:, None, :]).sum(-1) + gD[:, None] * cache['bg_depth_real_f']
    d_a = g_w * T - _suffix_excl(g_w * w) / q
    with np.errstate(invalid='ignore', over='ignore'):
        bg_d_sigma = (d_a * cache['bg_dists'] * e).astype(f32)
    bg_d_rgb = (w[..., None] * gC[:, None, :]).astype(f32)
    return fg_d_rgb, fg_d_sigma, bg_d_rgb, bg_d_sigma


def nerf_backward(cache, g_rgb, g_depth, g_fg_weights=None, bf16=False):
    """Gradient of the loss w.r.t. every parameter, given dL/d rgb [N,3], dL/d depth [N] and
    (KL only) dL/d fg_weights [N,S].  bf16: see mlp_backward."""
    fg_d_rgb, fg_d_sigma, bg_d_rgb, bg_d_sigma = composite_backward(cache, g_rgb, g_depth,
                                                                    g_fg_weights)
    gf = mlp_backward(cache['pf'], cache['fg'], fg_d_rgb.reshape(-1, 3), fg_d_sigma.reshape(-1), bf16=bf16)
    gb = mlp_backward(cache['pb'], cache['bg'], bg_d_rgb.reshape(-1, 3), bg_d_sigma.reshape(-1), bf16=bf16)
    out = OrderedDict()
    for k, v in gf.items():
        out['fg_net.' + k] = v
    for k, v in gb.items():
        out['bg_net.' + k] = v
    return out


# --------------------------------------------------------------------------------------
# a12-a14  losses                      utils.py:12-16,31;  depth_loss.py:4-44
# --------------------------------------------------------------------------------------
def img2mse(x, y):
    d = (np.asarray(x, f32) - np.asarray(y, f32)).astype(f32)
    return f32(np.mean((d * d).astype(np.float64)))


def mse2psnr(x):
    return -10. * np.log(x + float(TINY_NUMBER)) / np.log(10.)


def depth_mse(depth_gt, depth_pred):
    mask = depth_gt > 0.0
    if not mask.any():
        return f32(np.nan)
    d = (depth_gt[mask] - depth_pred[mask]).astype(f32)
    return f32(np.mean((d * d).astype(np.float64)))


def depth_l1(depth_gt, depth_pred):
    mask = depth_gt > 0.0
    if not mask.any():
        return f32(np.nan)
    return f32(np.mean(np.abs(depth_gt[mask] - depth_pred[mask]).astype(np.float64)))


def depth_kl(weights, termination_depth, steps, lengths, sigma, fg_far_depth=None):
    """Quirks preserved: divisor 2*sigma (not 2*sigma^2); `.sum(-2)` sums over the RAY axis,
    then mean over S; an empty mask gives 0."""
    mask = termination_depth > 0
    if fg_far_depth is not None:
        mask = np.logical_and(mask, termination_depth < fg_far_depth)
    loss = -np.log(weights + f32(1e-5)) * \
        np.exp(-((steps - termination_depth[:, None]) ** 2) / f32(2 * sigma)) * lengths
    loss = loss.astype(f32)[mask].astype(np.float64).sum(-2)
    return f32(np.mean(loss))


def loss_and_grads(ret, fg_z_vals, fg_far_depth, rgb_gt, depth_sup, use_depth, depth_loss_type,
                   lambda_depth, depth_sigma_scaled):
    """Loss head of ddp_train_nerf.py:481-493 with its closed-form gradient.
    Returns (loss, rgb_loss, depth_loss|None, g_rgb, g_depth, g_fg_weights|None)."""
    N = ret['rgb'].shape[0]
    rgb_gt = np.asarray(rgb_gt, f32)
    rgb_loss = img2mse(ret['rgb'], rgb_gt)
    g_rgb = (f32(2.) * (ret['rgb'] - rgb_gt) / f32(3 * N)).astype(f32)
    g_depth = np.zeros(N, f32)
    g_w = None
    depth_loss = None
    loss = rgb_loss
    if use_depth:
        depth_sup = np.asarray(depth_sup, f32)
        lam = f32(lambda_depth)
        if depth_loss_type == 'kl':
            depth_loss = depth_kl(ret['fg_weights'], depth_sup, fg_z_vals, ret['fg_dists'],
                                  depth_sigma_scaled, fg_far_depth)
            mask = np.logical_and(depth_sup > 0, depth_sup < fg_far_depth)
            S = fg_z_vals.shape[-1]
            gauss = np.exp(-((fg_z_vals - depth_sup[:, None]) ** 2) / f32(2 * depth_sigma_scaled))
            g_w = (-lam * mask[:, None] * gauss * ret['fg_dists'] /
                   ((ret['fg_weights'] + f32(1e-5)) * f32(S))).astype(f32)
        else:
            mask = depth_sup > 0
            cnt = int(mask.sum())
            pred = ret['depth']
            if depth_loss_type == 'mse':
                depth_loss = depth_mse(depth_sup, pred)
                if cnt:
                    g_depth = (lam * f32(2.) * (pred - depth_sup) * mask / f32(cnt)).astype(f32)
            elif depth_loss_type == 'l1':
                depth_loss = depth_l1(depth_sup, pred)
                if cnt:
                    g_depth = (lam * np.sign(pred - depth_sup) * mask / f32(cnt)).astype(f32)
            else:
                raise ValueError('depth_loss_type %r is dead code in the reference' % depth_loss_type)
        loss = f32(rgb_loss + lam * depth_loss)
    return loss, rgb_loss, depth_loss, g_rgb, g_depth, g_w


# --------------------------------------------------------------------------------------
# a15  Adam (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay)
# --------------------------------------------------------------------------------------
def autoexpo_loss_and_grads(rgb, rgb_gt, p, lambda_autoexpo):
    """--optim_autoexpo (ddp_model.py:186-190, ddp_train_nerf.py:472-479): p = the image's parameter
    [p0, p1]; scale = |p0| + 0.5, shift = p1; rgb_pred = (rgb - shift) / scale;
    rgb_loss = img2mse(rgb_pred, rgb_gt); loss = rgb_loss + lambda * (|scale - 1| + |shift|).
    Returns (loss without the depth term, rgb_loss, scale, shift, d loss / d rgb [N,3], d loss / d p [2])."""
    rgb, rgb_gt = np.asarray(rgb, f32), np.asarray(rgb_gt, f32)
    p = np.asarray(p, f32)
    scale, shift = f32(np.abs(p[0]) + f32(0.5)), p[1]
    pred = (rgb - shift) / scale
    rgb_loss = img2mse(pred, rgb_gt)
    reg = f32(lambda_autoexpo) * (np.abs(scale - f32(1.)) + np.abs(shift))
    r = (pred - rgb_gt) * f32(2.0 / rgb.size)                      # d rgb_loss / d pred
    g_rgb = (r / scale).astype(f32)
    d_scale = -(r * pred).sum(dtype=np.float64) / scale + lambda_autoexpo * np.sign(scale - f32(1.))
    d_shift = -r.sum(dtype=np.float64) / scale + lambda_autoexpo * np.sign(shift)
    g_p = np.array([d_scale * np.sign(p[0]), d_shift], f32)
    return f32(rgb_loss + reg), rgb_loss, scale, shift, g_rgb, g_p


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr=5e-4, beta1=0.9, beta2=0.999, eps=1e-8):
    """In-place torch.optim.Adam single-tensor update, `step` is the 1-based step count."""
    grad = np.asarray(grad, f32)
    exp_avg += (grad - exp_avg) * f32(1 - beta1)                          # lerp_
    exp_avg_sq *= f32(beta2)
    exp_avg_sq += f32(1 - beta2) * grad * grad                            # addcmul_
    bias1 = 1 - beta1 ** step
    bias2 = 1 - beta2 ** step
    step_size = lr / bias1
    denom = (np.sqrt(exp_avg_sq) / f32(np.sqrt(bias2)) + f32(eps)).astype(f32)
    param -= (f32(step_size) * (exp_avg / denom)).astype(f32)


# --------------------------------------------------------------------------------------
# parameter initialisation == nn.Linear default under torch.manual_seed(777)
# (ddp_train_nerf.py:308, nerf_network.py:88-117).  Needs torch (CPU) for its RNG stream.
# --------------------------------------------------------------------------------------
def init_params_like_reference(n_levels=2, seed=777, netdepth=8, netwidth=256):
    import torch
    torch.manual_seed(seed)
    levels = []
    for _ in range(n_levels):
        params = OrderedDict()
        for net, in_ch in (('fg_net', FG_IN), ('bg_net', BG_IN)):
            for name, shape in _creation_order(in_ch, DIR_IN, netdepth, netwidth):
                lin = torch.nn.Linear(shape[1], shape[0])
                params['%s.%s.weight' % (net, name)] = lin.weight.detach().numpy().copy()
                params['%s.%s.bias' % (net, name)] = lin.bias.detach().numpy().copy()
        # re-order to state_dict order (creation order differs: sigma, remap, rgb are created
        # after the trunk in the same order as state_dict, so this is already consistent)
        levels.append(params)
    return levels


def _creation_order(input_ch, input_ch_viewdirs, netdepth, netwidth, skips=(4,)):
    """nn.Linear construction order inside MLPNet.__init__ (nerf_network.py:88-117)."""
    out = []
    dim = input_ch
    for i in range(netdepth):
        out.append(('base_layers.%d.0' % i, (netwidth, dim)))
        dim = netwidth
        if i in skips and i != netdepth - 1:
            dim += input_ch
    out.append(('sigma_layers.0', (1, dim)))
    out.append(('base_remap_layers.0', (256, dim)))
    out.append(('rgb_layers.0', (netwidth // 2, 256 + input_ch_viewdirs)))
    out.append(('rgb_layers.2', (3, netwidth // 2)))
    return out


def param_order():
    """Flat-buffer order of one level-net == NerfNet.parameters() order (fg_net then bg_net)."""
    return ['%s.%s' % (net, n) for net in ('fg_net', 'bg_net') for n in mlp_param_names()]


# --------------------------------------------------------------------------------------
# one full optimisation step of ddp_train_nerf.py:432-498 (both cascade levels)
# --------------------------------------------------------------------------------------
def train_step(levels, opt_state, step, batch, uniforms, cascade_samples=(64, 128),
               use_depth=True, depth_loss_type='mse', lambda_depth=0.1, depth_sigma_scaled=0.01,
               lr=5e-4, grad_hook=None):
    """levels: list of param dicts (updated in place).  opt_state: list of
    {name: (exp_avg, exp_avg_sq)}.  uniforms: dict with t_fg, t_bg [N,S0], u_fg, u_bg [N,S1].
    grad_hook(level, grads) -> grads lets the caller emulate the DDP average (a16).
    Returns per-level dicts of scalars and the per-level forward outputs."""
    ray_o, ray_d = batch['ray_o'], batch['ray_d']
    logs, rets = [], []
    ret = None
    for m, n_samples in enumerate(cascade_samples):
        if m == 0:
            fg_far = intersect_sphere(ray_o, ray_d)
            fg_z, bg_z = coarse_depths(batch['min_depth'], fg_far, n_samples)
            fg_z = perturb_samples(fg_z, uniforms['t_fg'])
            bg_z = perturb_samples(bg_z, uniforms['t_bg'])
        else:
            fg_z, _, _ = fine_depths(fg_z, ret['fg_weights'], uniforms['u_fg'])
            bg_z, _, _ = fine_depths(bg_z, ret['bg_weights'], uniforms['u_bg'])
        cache = {}
        ret = nerf_forward(levels[m], ray_o, ray_d, fg_far, fg_z, bg_z, cache=cache)
        loss, rgb_loss, depth_loss, g_rgb, g_depth, g_w = loss_and_grads(
            ret, fg_z, fg_far, batch['rgb'], batch.get('depth_sup'), use_depth, depth_loss_type,
            lambda_depth, depth_sigma_scaled)
        grads = nerf_backward(cache, g_rgb, g_depth, g_w)
        if grad_hook is not None:
            grads = grad_hook(m, grads)
        for k, p in levels[m].items():
            ea, eas = opt_state[m][k]
            adam_step(p, grads[k], ea, eas, step, lr=lr)
        logs.append(dict(loss=loss, rgb_loss=rgb_loss, depth_loss=depth_loss,
                         psnr=mse2psnr(float(rgb_loss))))
        rets.append((ret, fg_z, bg_z, grads))
    return logs, rets


def new_opt_state(levels):
    return [{k: (np.zeros_like(v), np.zeros_like(v)) for k, v in lv.items()} for lv in levels]
