"""CPU oracle for SURVEY 8 f-4: a numpy restatement of the MipNeRF-360 depth-supervised path
(BASELINE config 5) of cwchenwang/outdoor-nerf-depth.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/` may import it.

Parity status
  * The reference for this path is JAX/Flax (nerf-methods/mipnerf360) and CANNOT be imported in the build
    container (no jax / flax / gin).  This restatement is pinned by the reference project's own unit tests:
    `tests/test_mip360_oracle.py` re-runs the closed-form / brute-force / round-trip properties of
    mipnerf360/tests/{coord,stepfun,render,math}_test.py against the functions below (each test cites the
    reference test it restates), and `generate_basis` -- pure numpy upstream -- against a golden vector captured
    from the imported reference (tests/golden/mip360_basis.npz, tests/golden/make_golden_mip360.py).
  * PARITY UNPINNED: the depth-supervision additions of this fork -- `compute_data_loss`'s depth terms
    (internal/train_utils.py:108-129) and internal/depth_loss.py -- have no tests upstream; they are restated
    line by line and checked only for self-consistency (finite differences of the closed-form gradients).

Every function cites the reference lines it restates (paths relative to nerf-methods/mipnerf360/).
All functions are dtype-generic numpy (float32 in, float32 out; float64 in the gradient checks).
"""
import numpy as np

EPS32 = float(np.finfo(np.float32).eps)


# ======================================================================================== internal/math.py
def safe_sin(x, t=100 * np.pi):
    """math.py:26-38: sin(x mod t) for |x| >= t."""
    x = np.asarray(x)
    return np.sin(np.where(np.abs(x) < t, x, np.mod(x, t)))


def sorted_interp(x, xp, fp):
    """math.py:106-127: piecewise-linear interpolation for sorted xp / fp (brute-force interval search)."""
    mask = x[..., None, :] >= xp[..., :, None]

    def find_interval(v):
        v0 = np.max(np.where(mask, v[..., None], v[..., :1, None]), -2)
        v1 = np.min(np.where(~mask, v[..., None], v[..., -1:, None]), -2)
        return v0, v1

    fp0, fp1 = find_interval(fp)
    xp0, xp1 = find_interval(xp)
    with np.errstate(divide='ignore', invalid='ignore'):
        off = (x - xp0) / (xp1 - xp0)
    off = np.clip(np.nan_to_num(off, nan=0.0), 0, 1)
    return fp0 + off * (fp1 - fp0)


def log_lerp(t, v0, v1):
    """math.py:58-64."""
    lv0, lv1 = np.log(v0), np.log(v1)
    return np.exp(np.clip(t, 0, 1) * (lv1 - lv0) + lv0)


def learning_rate_decay(step, lr_init, lr_final, max_steps, lr_delay_steps=0, lr_delay_mult=1):
    """math.py:67-97."""
    if lr_delay_steps > 0:
        delay = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
    else:
        delay = 1.
    return delay * log_lerp(step / max_steps, lr_init, lr_final)


# ======================================================================================== internal/geopoly.py
def generate_basis(base_shape='icosahedron', angular_tesselation=2, eps=1e-4):
    """geopoly.py:74-124 (with :33-71): unit vectors of a tesselated icosahedron, antipodal duplicates removed,
    coordinate order reversed.  Returns [n, 3] like upstream (n = 21 for the MLP default: icosahedron, 2
    subdivisions); the MLPs use its transpose (models.py:387-389), see pos_basis_t()."""
    if base_shape != 'icosahedron':
        raise ValueError('only the icosahedron basis (the MLP default, models.py:381) is restated')
    a = (np.sqrt(5) + 1) / 2
    verts = np.array([(-1, 0, a), (1, 0, a), (-1, 0, -a), (1, 0, -a), (0, a, 1), (0, a, -1), (0, -a, 1),
                      (0, -a, -1), (a, 1, 0), (-a, 1, 0), (a, -1, 0), (-a, -1, 0)]) / np.sqrt(a + 2)
    faces = [(0, 4, 1), (0, 9, 4), (9, 5, 4), (4, 5, 8), (4, 8, 1), (8, 10, 1), (8, 3, 10), (5, 3, 8), (5, 2, 3),
             (2, 7, 3), (7, 10, 3), (7, 6, 10), (7, 11, 6), (11, 0, 6), (0, 1, 6), (6, 1, 10), (9, 0, 11),
             (9, 11, 2), (9, 2, 5), (7, 2, 11)]
    v = int(angular_tesselation)
    bary = np.array([(i, j, v - i - j) for i in range(v + 1) for j in range(v + 1 - i)], np.float64) / v
    pts = []
    for f in faces:
        p = bary @ verts[list(f), :]
        pts.append(p / np.sqrt(np.sum(p ** 2, 1, keepdims=True)))
    pts = np.concatenate(pts, 0)
    sq = lambda A, B: np.maximum(0, np.sum(A ** 2, 1)[:, None] + np.sum(B ** 2, 1)[None, :] - 2 * A @ B.T)
    first = np.array([np.min(np.argwhere(d <= eps)) for d in sq(pts, pts)])      # first occurrence of every vertex
    pts = pts[np.unique(first), :]
    match = sq(pts, -pts) < eps
    pts = pts[np.any(np.triu(match), 1), :]
    return pts[:, ::-1].copy()


def pos_basis_t(base_shape='icosahedron', subdivisions=2):
    """models.py:387-389: the [3, n] matrix the IPE lifts means / covariances onto."""
    return generate_basis(base_shape, subdivisions).T.astype(np.float32)


# ======================================================================================== internal/coord.py
def contract(x):
    """coord.py:21-27 (Eq. 10 of arXiv:2111.12077): identity inside the unit ball, (2 - 1/|x|) x/|x| outside."""
    x = np.asarray(x)
    m2 = np.maximum(np.finfo(np.float32).eps, np.sum(x ** 2, -1, keepdims=True))
    with np.errstate(invalid='ignore', divide='ignore'):
        z = np.where(m2 <= 1, x, ((2 * np.sqrt(m2) - 1) / m2) * x)
    return z


def inv_contract(z):
    """coord.py:30-36."""
    z = np.asarray(z)
    m2 = np.maximum(np.finfo(np.float32).eps, np.sum(z ** 2, -1, keepdims=True))
    with np.errstate(invalid='ignore', divide='ignore'):
        x = np.where(m2 <= 1, z, z / (2 * np.sqrt(m2) - m2))
    return x


def contract_jacobian(x):
    """d contract / d x, closed form (upstream gets it from jax.linearize, coord.py:58):
    |x|^2 <= 1: I;  else s I + (ds/dm2) 2 x x^T with s = (2 sqrt(m2) - 1) / m2,
    ds/dm2 = (1 - sqrt(m2)) / m2^2."""
    x = np.asarray(x)
    m2 = np.maximum(np.finfo(np.float32).eps, np.sum(x ** 2, -1, keepdims=True))[..., None]
    eye = np.broadcast_to(np.eye(x.shape[-1], dtype=x.dtype), x.shape + (x.shape[-1],))
    r = np.sqrt(m2)
    s = (2 * r - 1) / m2
    ds = (1 - r) / (m2 * m2)
    outer = x[..., :, None] * x[..., None, :]
    return np.where(m2 <= 1, eye, s * eye + 2 * ds * outer)


def track_linearize_contract(mean, cov):
    """coord.py:39-60 specialised to fn = contract (PropMLP.warp_fn / NerfMLP.warp_fn, configs/360.gin:12,17):
    (contract(mean), J cov J^T)."""
    J = contract_jacobian(mean)
    return contract(mean), J @ cov @ np.swapaxes(J, -1, -2)


def track_linearize_affine(a_mat, b, mean, cov):
    """coord.py:39-60 for an affine fn(x) = A x + b (the case coord_test.py:137-170 checks)."""
    return mean @ a_mat.T + b, a_mat @ cov @ a_mat.T


def construct_ray_warps(fn, t_near, t_far):
    """coord.py:63-100.  fn: None | 'reciprocal' | 'log' | 'sqrt' | 'piecewise'.  Returns (t_to_s, s_to_t)."""
    if fn is None:
        fwd, inv = (lambda x: x), (lambda x: x)
    elif fn == 'piecewise':
        fwd = lambda x: np.where(x < 1, .5 * x, 1 - .5 / x)
        inv = lambda x: np.where(x < .5, 2 * x, .5 / (1 - x))
    else:
        fwd, inv = {'reciprocal': (np.reciprocal, np.reciprocal), 'log': (np.log, np.exp),
                    'sqrt': (np.sqrt, np.square), 'exp': (np.exp, np.log), 'square': (np.square, np.sqrt)}[fn]
    def _f(v):
        v = np.asarray(v)
        return v if v.dtype.kind == 'f' else v.astype(np.float64)
    with np.errstate(divide='ignore'):
        s_near, s_far = fwd(_f(t_near)), fwd(_f(t_far))
    t_to_s = lambda t: (fwd(t) - s_near) / (s_far - s_near)
    s_to_t = lambda s: inv(s * s_far + (1 - s) * s_near)
    return t_to_s, s_to_t


def expected_sin(mean, var):
    """coord.py:103-105."""
    return np.exp(-0.5 * var) * safe_sin(mean)


def integrated_pos_enc(mean, var, min_deg, max_deg):
    """coord.py:108-128: [sin(2^k mu) e^{-4^k var / 2} for all k, dims | the same with cos]."""
    mean, var = np.asarray(mean), np.asarray(var)
    scales = (2.0 ** np.arange(min_deg, max_deg)).astype(mean.dtype)
    shape = mean.shape[:-1] + (-1,)
    sm = np.reshape(mean[..., None, :] * scales[:, None], shape)
    sv = np.reshape(var[..., None, :] * scales[:, None] ** 2, shape)
    return expected_sin(np.concatenate([sm, sm + 0.5 * np.pi], -1), np.concatenate([sv, sv], -1))


def lift_and_diagonalize(mean, cov, basis):
    """coord.py:131-135.  basis [3, n]."""
    return mean @ basis, np.sum(basis * (cov @ basis), -2)


def pos_enc(x, min_deg, max_deg, append_identity=True):
    """coord.py:138-147."""
    x = np.asarray(x)
    scales = (2.0 ** np.arange(min_deg, max_deg)).astype(x.dtype)
    sx = np.reshape(x[..., None, :] * scales[:, None], x.shape[:-1] + (-1,))
    four = np.sin(np.concatenate([sx, sx + 0.5 * np.pi], -1))
    return np.concatenate([x, four], -1) if append_identity else four


# ======================================================================================== internal/stepfun.py
def searchsorted(a, v):
    """stepfun.py:30-53: (idx_lo, idx_hi) with a[idx_lo] <= v < a[idx_hi]; both clamp to the first / last index
    when v is out of range."""
    i = np.arange(a.shape[-1])
    ge = v[..., None, :] >= a[..., :, None]
    lo = np.max(np.where(ge, i[:, None], i[:1, None]), -2)
    hi = np.min(np.where(~ge, i[:, None], i[-1:, None]), -2)
    return lo, hi


def query(tq, t, y, outside_value=0):
    """stepfun.py:56-61."""
    lo, hi = searchsorted(t, tq)
    # (idx_lo can be the last edge index = one past the last bin when tq is beyond the domain; jnp's gather does
    # not raise there and the value is discarded by the where)
    return np.where(lo == hi, outside_value, np.take_along_axis(y, np.minimum(lo, y.shape[-1] - 1), -1))


def inner_outer(t0, t1, y1):
    """stepfun.py:64-78."""
    cy1 = np.concatenate([np.zeros_like(y1[..., :1]), np.cumsum(y1, -1)], -1)
    lo, hi = searchsorted(t1, t0)
    c_lo, c_hi = np.take_along_axis(cy1, lo, -1), np.take_along_axis(cy1, hi, -1)
    outer = c_hi[..., 1:] - c_lo[..., :-1]
    inner = np.where(hi[..., :-1] <= lo[..., 1:], c_lo[..., 1:] - c_hi[..., :-1], 0)
    return inner, outer


def lossfun_outer(t, w, t_env, w_env, eps=EPS32):
    """stepfun.py:81-87: max(0, w - w_outer)^2 / (w + eps) per interval of (t, w)."""
    _, w_outer = inner_outer(t, t_env, w_env)
    return np.maximum(0, w - w_outer) ** 2 / (w + eps)


def lossfun_outer_grad_w_env(t, w, t_env, w_env, eps=EPS32):
    """d sum(lossfun_outer) / d w_env (closed form; upstream: autograd).  w_outer[i] = cy1[hi[i+1]] - cy1[lo[i]]
    is the sum of w_env over bins [lo[i], hi[i+1]), so the gradient of interval i's term
    -2 max(0, w_i - w_outer_i) / (w_i + eps) is scattered to those bins."""
    lo, hi = searchsorted(t_env, t)
    _, w_outer = inner_outer(t, t_env, w_env)
    g_outer = -2 * np.maximum(0, w - w_outer) / (w + eps)                       # [..., n]
    j = np.arange(w_env.shape[-1])
    cover = (j >= lo[..., :-1, None]) & (j < hi[..., 1:, None])                  # [..., n, m]
    return np.sum(g_outer[..., None] * cover, -2)


def weight_to_pdf(t, w, eps=EPS32 ** 2):
    """stepfun.py:90-92."""
    return w / np.maximum(eps, t[..., 1:] - t[..., :-1])


def pdf_to_weight(t, p):
    """stepfun.py:95-97."""
    return p * (t[..., 1:] - t[..., :-1])


def max_dilate(t, w, dilation, domain=(-np.inf, np.inf)):
    """stepfun.py:100-115."""
    t0 = t[..., :-1] - dilation
    t1 = t[..., 1:] + dilation
    td = np.sort(np.concatenate([t, t0, t1], -1), -1)
    td = np.clip(td, *domain)
    wd = np.max(np.where((t0[..., None, :] <= td[..., None]) & (t1[..., None, :] > td[..., None]),
                         w[..., None, :], 0), -1)[..., :-1]
    return td, wd


def max_dilate_weights(t, w, dilation, domain=(-np.inf, np.inf), renormalize=False, eps=EPS32 ** 2):
    """stepfun.py:118-130."""
    p = weight_to_pdf(t, w)
    td, pd = max_dilate(t, p, dilation, domain=domain)
    wd = pdf_to_weight(td, pd)
    if renormalize:
        wd = wd / np.maximum(eps, np.sum(wd, -1, keepdims=True))
    return td, wd


def integrate_weights(w):
    """stepfun.py:133-152."""
    cw = np.minimum(1, np.cumsum(w[..., :-1], -1))
    shape = cw.shape[:-1] + (1,)
    return np.concatenate([np.zeros(shape, w.dtype), cw, np.ones(shape, w.dtype)], -1)


def softmax(x):
    with np.errstate(invalid='ignore'):
        m = np.max(x, -1, keepdims=True)
        e = np.exp(x - m)
    return e / np.sum(e, -1, keepdims=True)


def invert_cdf(u, t, w_logits):
    """stepfun.py:155-163 (sorted_interp branch)."""
    w = softmax(w_logits)
    return sorted_interp(u, integrate_weights(w), t)


def sample_u(shape_prefix, num_samples, jitter01=None, single_jitter=False, deterministic_center=False,
             dtype=np.float32):
    """The uniform positions of stepfun.py:195-213.  jitter01 replaces jax.random.uniform: values in [0, 1) of shape
    prefix + (1 if single_jitter else num_samples,), scaled by max_jitter here."""
    eps = np.finfo(np.float32).eps
    if jitter01 is None:
        if deterministic_center:
            pad = 1 / (2 * num_samples)
            u = np.linspace(pad, 1. - pad - eps, num_samples)
        else:
            u = np.linspace(0, 1. - eps, num_samples)
        return np.broadcast_to(u.astype(dtype), tuple(shape_prefix) + (num_samples,))
    u_max = eps + (1 - eps) / num_samples
    max_jitter = (1 - u_max) / (num_samples - 1) - eps
    return (np.linspace(0, 1 - u_max, num_samples) + np.asarray(jitter01) * max_jitter).astype(dtype)


def sample(t, w_logits, num_samples, jitter01=None, single_jitter=False, deterministic_center=False):
    """stepfun.py:166-215."""
    u = sample_u(t.shape[:-1], num_samples, jitter01, single_jitter, deterministic_center, t.dtype)
    return invert_cdf(u, t, w_logits)


def sample_intervals(t, w_logits, num_samples, jitter01=None, single_jitter=False, domain=(-np.inf, np.inf)):
    """stepfun.py:218-270."""
    if num_samples <= 1:
        raise ValueError('num_samples must be > 1, is %d.' % num_samples)
    centers = sample(t, w_logits, num_samples, jitter01, single_jitter, deterministic_center=True)
    mid = (centers[..., 1:] + centers[..., :-1]) / 2
    lo, hi = domain
    first = np.maximum(lo, 2 * centers[..., :1] - mid[..., :1])
    last = np.minimum(hi, 2 * centers[..., -1:] - mid[..., -1:])
    return np.concatenate([first, mid, last], -1)


def lossfun_distortion(t, w):
    """stepfun.py:273-283."""
    ut = (t[..., 1:] + t[..., :-1]) / 2
    dut = np.abs(ut[..., :, None] - ut[..., None, :])
    inter = np.sum(w * np.sum(w[..., None, :] * dut, -1), -1)
    intra = np.sum(w ** 2 * (t[..., 1:] - t[..., :-1]), -1) / 3
    return inter + intra


def lossfun_distortion_grad_w(t, w):
    """d lossfun_distortion / d w (closed form): 2 sum_j w_j |ut_i - ut_j| + (2/3) w_i (t_{i+1} - t_i)."""
    ut = (t[..., 1:] + t[..., :-1]) / 2
    dut = np.abs(ut[..., :, None] - ut[..., None, :])
    return 2 * np.sum(w[..., None, :] * dut, -1) + 2 * w * (t[..., 1:] - t[..., :-1]) / 3


def interval_distortion(t0_lo, t0_hi, t1_lo, t1_hi):
    """stepfun.py:286-303."""
    d_disjoint = np.abs((t1_lo + t1_hi) / 2 - (t0_lo + t0_hi) / 2)
    d_overlap = (2 * (np.minimum(t0_hi, t1_hi) ** 3 - np.maximum(t0_lo, t1_lo) ** 3) +
                 3 * (t1_hi * t0_hi * np.abs(t1_hi - t0_hi) + t1_lo * t0_lo * np.abs(t1_lo - t0_lo) +
                      t1_hi * t0_lo * (t0_lo - t1_hi) + t1_lo * t0_hi * (t1_lo - t0_hi))) / \
        (6 * (t0_hi - t0_lo) * (t1_hi - t1_lo))
    return np.where((t0_lo > t1_hi) | (t1_lo > t0_hi), d_disjoint, d_overlap)


def weighted_percentile(t, w, ps):
    """stepfun.py:306-317."""
    cw = integrate_weights(w)
    cw2, t2 = cw.reshape(-1, cw.shape[-1]), t.reshape(-1, t.shape[-1])
    out = np.stack([np.interp(np.asarray(ps) / 100, c, tt) for c, tt in zip(cw2, t2)], 0)
    return out.reshape(cw.shape[:-1] + (len(ps),))


def resample(t, tp, vp, use_avg=False, eps=EPS32):
    """stepfun.py:320-354."""
    if use_avg:
        wp = np.diff(tp, axis=-1)
        return resample(t, tp, vp * wp) / np.maximum(eps, resample(t, tp, wp))
    acc0 = np.concatenate([np.zeros(vp.shape[:-1] + (1,), vp.dtype), np.cumsum(vp, -1)], -1)
    t2, tp2, a2 = np.broadcast_arrays(t[..., :, None], tp[..., None, :], acc0[..., None, :])
    flat_t, flat_tp, flat_a = t.reshape(-1, t.shape[-1]), \
        np.broadcast_to(tp, t.shape[:-1] + tp.shape[-1:]).reshape(-1, tp.shape[-1]), \
        np.broadcast_to(acc0, t.shape[:-1] + acc0.shape[-1:]).reshape(-1, acc0.shape[-1])
    res = np.stack([np.interp(a, b, c) for a, b, c in zip(flat_t, flat_tp, flat_a)], 0).reshape(t.shape)
    return np.diff(res, axis=-1)


# ======================================================================================== internal/render.py
def lift_gaussian(d, t_mean, t_var, r_var, diag):
    """render.py:21-42."""
    mean = d[..., None, :] * t_mean[..., None]
    d_mag_sq = np.maximum(1e-10, np.sum(d ** 2, -1, keepdims=True))
    if diag:
        d_outer_diag = d ** 2
        null_outer_diag = 1 - d_outer_diag / d_mag_sq
        return mean, t_var[..., None] * d_outer_diag[..., None, :] + r_var[..., None] * null_outer_diag[..., None, :]
    d_outer = d[..., :, None] * d[..., None, :]
    null_outer = np.eye(d.shape[-1], dtype=d.dtype) - d[..., :, None] * (d / d_mag_sq)[..., None, :]
    return mean, t_var[..., None, None] * d_outer[..., None, :, :] + r_var[..., None, None] * null_outer[..., None, :, :]


def conical_frustum_to_gaussian(d, t0, t1, base_radius, diag, stable=True):
    """render.py:45-82 (Eq. 7 of arXiv:2103.13415 when stable)."""
    if stable:
        mu, hw = (t0 + t1) / 2, (t1 - t0) / 2
        eps = np.finfo(np.float32).eps
        denom = np.maximum(eps, 3 * mu ** 2 + hw ** 2)
        t_mean = mu + (2 * mu * hw ** 2) / denom
        t_var = (hw ** 2) / 3 - (4 / 15) * hw ** 4 * (12 * mu ** 2 - hw ** 2) / denom ** 2
        r_var = (mu ** 2) / 4 + (5 / 12) * hw ** 2 - (4 / 15) * (hw ** 4) / denom
    else:
        t_mean = (3 * (t1 ** 4 - t0 ** 4)) / (4 * (t1 ** 3 - t0 ** 3))
        r_var = 3 / 20 * (t1 ** 5 - t0 ** 5) / (t1 ** 3 - t0 ** 3)
        t_var = 3 / 5 * (t1 ** 5 - t0 ** 5) / (t1 ** 3 - t0 ** 3) - t_mean ** 2
    return lift_gaussian(d, t_mean, t_var, r_var * base_radius ** 2, diag)


def cylinder_to_gaussian(d, t0, t1, radius, diag):
    """render.py:85-105."""
    return lift_gaussian(d, (t0 + t1) / 2, (t1 - t0) ** 2 / 12, radius ** 2 / 4 * np.ones_like(t0), diag)


def cast_rays(tdist, origins, directions, radii, ray_shape='cone', diag=True):
    """render.py:108-133."""
    fn = {'cone': conical_frustum_to_gaussian, 'cylinder': cylinder_to_gaussian}[ray_shape]
    means, covs = fn(directions, tdist[..., :-1], tdist[..., 1:], radii, diag)
    return means + origins[..., None, :], covs


def compute_alpha_weights(density, tdist, dirs, opaque_background=False):
    """render.py:136-158."""
    delta = (tdist[..., 1:] - tdist[..., :-1]) * np.linalg.norm(dirs[..., None, :], axis=-1)
    dd = density * delta
    if opaque_background:
        dd = np.concatenate([dd[..., :-1], np.full_like(dd[..., -1:], np.inf)], -1)
    alpha = 1 - np.exp(-dd)
    trans = np.exp(-np.concatenate([np.zeros_like(dd[..., :1]), np.cumsum(dd[..., :-1], -1)], -1))
    return alpha * trans, alpha, trans


def alpha_weights_backward(density, tdist, dirs, g_weights, opaque_background=False):
    """d L / d density given d L / d weights (closed form; upstream: autograd).  With x_i = density_i delta_i,
    w_i = (1 - e^{-x_i}) T_i, T_i = exp(-sum_{j<i} x_j):  dL/dx_i = g_i e^{-x_i} T_i - sum_{k>i} g_k w_k."""
    delta = (tdist[..., 1:] - tdist[..., :-1]) * np.linalg.norm(dirs[..., None, :], axis=-1)
    w, alpha, trans = compute_alpha_weights(density, tdist, dirs, opaque_background)
    gw = g_weights * w
    suffix = np.cumsum(gw[..., ::-1], -1)[..., ::-1] - gw                       # sum_{k>i} g_k w_k
    g_x = g_weights * (1 - alpha) * trans - suffix
    if opaque_background:
        g_x = np.concatenate([g_x[..., :-1], np.zeros_like(g_x[..., -1:])], -1)   # x_last = inf: no gradient
    return g_x * delta


def volumetric_rendering(rgbs, weights, tdist, bg_rgbs, t_far, compute_extras=True):
    """render.py:161-216 (extras other than the distance statistics are not restated)."""
    eps = np.finfo(np.float32).eps
    out = {}
    acc = weights.sum(-1)
    bg_w = np.maximum(0, 1 - acc[..., None])
    out['rgb'] = (weights[..., None] * rgbs).sum(-2) + bg_w * bg_rgbs
    if compute_extras:
        out['acc'] = acc
        t_mids = 0.5 * (tdist[..., :-1] + tdist[..., 1:])
        with np.errstate(divide='ignore', invalid='ignore'):
            e = np.exp((weights * np.log(t_mids)).sum(-1) / np.maximum(eps, acc))
        out['distance_mean'] = np.clip(np.nan_to_num(e, nan=np.inf), tdist[..., 0], tdist[..., -1])
        out['depth'] = np.clip(np.nan_to_num((weights * t_mids).sum(-1), nan=np.inf), tdist[..., 0], tdist[..., -1])
        t_aug = np.concatenate([tdist, t_far], -1)
        w_aug = np.concatenate([weights, bg_w], -1)
        pct = weighted_percentile(t_aug, w_aug, [5, 50, 95])
        for i, p in enumerate([5, 50, 95]):
            out['distance_' + ('median' if p == 50 else 'percentile_%d' % p)] = pct[..., i]
    return out


# ======================================================================================== internal/models.py
def softplus(x):
    return np.logaddexp(x, 0)


PROP_CFG = dict(net_depth=4, net_width=256, disable_rgb=True)                  # configs/360.gin:12-16
NERF_CFG = dict(net_depth=8, net_width=1024, disable_rgb=False)                # configs/360.gin:17-20
MLP_DEFAULTS = dict(bottleneck_width=256, net_width_viewdirs=128, min_deg_point=0, max_deg_point=12, skip_layer=4,
                    deg_view=4, density_bias=-1., rgb_padding=0.001)            # models.py:344-381


def mlp_param_shapes(cfg):
    """flax nn.Dense parameter shapes in construction order of MLP.__call__ (models.py:436-606) for
    disable_density_normals=True, net_depth_viewdirs=1: Dense_0..Dense_{D-1} trunk, Dense_D density,
    Dense_{D+1} bottleneck, Dense_{D+2} view layer, Dense_{D+3} rgb.  kernel [in, out] like flax."""
    c = dict(MLP_DEFAULTS, **cfg)
    n_basis = 21
    in_dim = n_basis * 2 * (c['max_deg_point'] - c['min_deg_point'])
    shapes, dim = [], in_dim
    for i in range(c['net_depth']):
        shapes.append((dim, c['net_width']))
        dim = c['net_width']
        if i % c['skip_layer'] == 0 and i > 0:
            dim += in_dim
    shapes.append((dim, 1))
    if not c['disable_rgb']:
        shapes.append((dim, c['bottleneck_width']))
        dir_dim = 3 + 3 * 2 * c['deg_view']
        shapes.append((c['bottleneck_width'] + dir_dim, c['net_width_viewdirs']))
        shapes.append((c['net_width_viewdirs'], 3))
    return shapes


def init_mlp_params(cfg, rng):
    """he_uniform kernels (jax.nn.initializers.he_uniform: U(+-sqrt(6 / fan_in))), zero biases (flax default)."""
    ps = []
    for fi, fo in mlp_param_shapes(cfg):
        lim = np.sqrt(6.0 / fi)
        ps.append((rng.uniform(-lim, lim, (fi, fo)).astype(np.float32), np.zeros(fo, np.float32)))
    return ps


def mlp_forward(params, cfg, means, covs, viewdirs, basis, cache=None, q=None):
    """models.py:436-606 for the 360.gin configuration (warp_fn = contract, disable_density_normals).
    means [..., n, 3], covs [..., n, 3, 3], viewdirs [..., 3].  Returns dict(density [..., n], rgb [..., n, 3]).
    cache: optional dict that receives what mlp_backward needs.
    q: optional rounding of every GEMM operand (weights, layer inputs) -- identity upstream; the GPU tests pass a
    round-to-bfloat16 to model where a bf16-MFMA implementation rounds (test infrastructure, not upstream behaviour)."""
    c = dict(MLP_DEFAULTS, **cfg)
    q = (lambda a: a) if q is None else q
    params = [(q(W), b) for W, b in params]
    m, cv = track_linearize_contract(means, covs)
    lm, lv = lift_and_diagonalize(m, cv, basis)
    x = q(integrated_pos_enc(lm, lv, c['min_deg_point'], c['max_deg_point']))
    inputs = x
    k = 0
    acts_in, pre = [], []
    for i in range(c['net_depth']):
        W, b = params[k]; k += 1
        acts_in.append(x)
        z = x @ W + b
        pre.append(z)
        x = q(np.maximum(z, 0))
        if i % c['skip_layer'] == 0 and i > 0:
            x = np.concatenate([x, inputs], -1)
    W, b = params[k]; k += 1
    raw_density = (x @ W + b)[..., 0]
    density = softplus(raw_density + c['density_bias'])
    if cache is not None:
        cache.update(acts_in=acts_in, pre=pre, trunk_out=x, raw_density=raw_density, cfg=c)
    if c['disable_rgb']:
        return dict(density=density, rgb=np.zeros_like(means))
    W, b = params[k]; k += 1
    bott = q(x @ W + b)
    de = q(pos_enc(viewdirs, 0, c['deg_view'], append_identity=True))
    de = np.broadcast_to(de[..., None, :], bott.shape[:-1] + (de.shape[-1],))
    view_in = np.concatenate([bott, de], -1)
    W, b = params[k]; k += 1
    hz = view_in @ W + b
    h = q(np.maximum(hz, 0))
    W, b = params[k]; k += 1
    s = 1 / (1 + np.exp(-(h @ W + b)))
    rgb = s * (1 + 2 * c['rgb_padding']) - c['rgb_padding']
    if cache is not None:
        cache.update(view_in=view_in, hz=hz, h=h, sig=s)
    return dict(density=density, rgb=rgb)


def mlp_backward(params, cache, g_density, g_rgb=None, q=None):
    """Closed-form backward of mlp_forward w.r.t. the parameters (upstream: jax.grad; positions get no gradient --
    models.py:203-204 stop_level_grad and the inputs are not learned).  g_density [..., n], g_rgb [..., n, 3].
    Returns a list of (d kernel, d bias) in the parameter order.  q: see mlp_forward (rounds the dZ operands too)."""
    c = cache['cfg']
    D = c['net_depth']
    q = (lambda a: a) if q is None else q
    params = [(q(W), b) for W, b in params]
    flat = lambda a, w: np.reshape(a, (-1, w))
    grads = [None] * len(params)
    x = flat(cache['trunk_out'], cache['trunk_out'].shape[-1])
    sig = 1 / (1 + np.exp(-(cache['raw_density'] + c['density_bias'])))          # softplus'
    d_raw = q(np.reshape(g_density * sig, (-1, 1)))
    W_d, _ = params[D]
    grads[D] = (x.T @ d_raw, d_raw.sum(0))
    d_x = d_raw @ W_d.T
    if not c['disable_rgb']:
        s = flat(cache['sig'], 3)
        d_pre = q(flat(g_rgb, 3) * (1 + 2 * c['rgb_padding']) * s * (1 - s))
        h = flat(cache['h'], cache['h'].shape[-1])
        W3, _ = params[D + 3]
        grads[D + 3] = (h.T @ d_pre, d_pre.sum(0))
        d_hz = q((d_pre @ W3.T) * (flat(cache['hz'], h.shape[-1]) > 0))
        vin = flat(cache['view_in'], cache['view_in'].shape[-1])
        W2, _ = params[D + 2]
        grads[D + 2] = (vin.T @ d_hz, d_hz.sum(0))
        d_bott = q((d_hz @ W2.T)[:, :c['bottleneck_width']])
        W1, _ = params[D + 1]
        grads[D + 1] = (x.T @ d_bott, d_bott.sum(0))
        d_x = d_x + d_bott @ W1.T
    for i in reversed(range(D)):
        if i % c['skip_layer'] == 0 and i > 0:
            d_x = d_x[:, :c['net_width']]                                        # the encoding part needs no gradient
        z = flat(cache['pre'][i], c['net_width'])
        d_z = q(d_x * (z > 0))
        a_in = flat(cache['acts_in'][i], cache['acts_in'][i].shape[-1])
        grads[i] = (a_in.T @ d_z, d_z.sum(0))
        if i > 0:
            d_x = d_z @ params[i][0].T
    return grads


def model_forward(prop_params, nerf_params, rays, train_frac=1.0, jitter01=None, basis=None, num_prop_samples=64,
                  num_nerf_samples=32, num_levels=3, anneal_slope=10., dilation_multiplier=0.5, dilation_bias=0.0025,
                  raydist_fn='reciprocal', opaque_background=True, single_jitter=True, resample_padding=0.0,
                  bg_rgb=1.0, caches=None, q=None, sdist_override=None):
    """Model.__call__ (models.py:76-303) for configs/360.gin: 2 proposal levels + 1 NeRF level.
    rays: dict origins, directions, viewdirs [N,3], radii, near, far [N,1].  jitter01: None (deterministic) or a
    list of num_levels arrays [N,1] in [0,1) replacing the per-level jax.random.uniform of stepfun.sample.
    sdist_override (test infrastructure): per-level sample positions to use instead of the re-sampled ones, so that an
    implementation whose earlier levels differ by rounding can be compared level by level on identical intervals.
    Returns (renderings, ray_history) like upstream."""
    basis = pos_basis_t() if basis is None else basis
    _, s_to_t = construct_ray_warps(raydist_fn, rays['near'], rays['far'])
    s_near, s_far = 0., 1.
    sdist = np.concatenate([np.full_like(rays['near'], s_near), np.full_like(rays['far'], s_far)], -1)
    weights = np.ones_like(rays['near'])
    prod = 1
    renderings, history = [], []
    for lvl in range(num_levels):
        is_prop = lvl < num_levels - 1
        ns = num_prop_samples if is_prop else num_nerf_samples
        dilation = dilation_bias + dilation_multiplier * (s_far - s_near) / prod
        prod *= ns
        if lvl > 0 and (dilation_bias > 0 or dilation_multiplier > 0):
            sdist, weights = max_dilate_weights(sdist, weights, dilation, domain=(s_near, s_far), renormalize=True)
            sdist, weights = sdist[..., 1:-1], weights[..., 1:-1]
        anneal = (anneal_slope * train_frac) / ((anneal_slope - 1) * train_frac + 1) if anneal_slope > 0 else 1.
        with np.errstate(divide='ignore'):
            logits = np.where(sdist[..., 1:] > sdist[..., :-1], anneal * np.log(weights + resample_padding), -np.inf)
        sdist = sample_intervals(sdist, logits, ns, None if jitter01 is None else jitter01[lvl], single_jitter,
                                 domain=(s_near, s_far))
        if sdist_override is not None:
            sdist = np.asarray(sdist_override[lvl], sdist.dtype)
        tdist = s_to_t(sdist)
        means, covs = cast_rays(tdist, rays['origins'], rays['directions'], rays['radii'], 'cone', diag=False)
        cache = None
        if caches is not None:                       # train_step: one cache per level for mlp_backward
            cache = {}
            caches.append(cache)
        res = mlp_forward(prop_params if is_prop else nerf_params, PROP_CFG if is_prop else NERF_CFG, means, covs,
                          rays['viewdirs'], basis, cache=cache, q=q)
        weights = compute_alpha_weights(res['density'], tdist, rays['directions'], opaque_background)[0]
        renderings.append(volumetric_rendering(res['rgb'], weights, tdist, bg_rgb, rays['far']))
        res.update(sdist=sdist, tdist=tdist, weights=weights)
        history.append(res)
    return renderings, history


# ======================================================================================== internal/train_utils.py + depth_loss.py
URF_SIGMA_SCALE_FACTOR = 3.0


def ds_nerf_depth_loss(weights, termination_depth, steps, lengths, sigma):
    """depth_loss.py:5-29.  Quirks kept: log(w + 1e-7); divisor 2*sigma; `.sum(-2)` sums over the RAY axis
    ([N,S] -> [S]) and `* depth_mask` then broadcasts a [N] mask against [S] -- upstream that needs N == S or
    N == 1; like upstream this raises otherwise."""
    mask = termination_depth > 0
    loss = -np.log(weights + 1e-7) * np.exp(-((steps - termination_depth[:, None]) ** 2) / (2 * sigma)) * lengths
    return np.mean(loss.sum(-2) * mask)


def urban_radiance_field_depth_loss(weights, termination_depth, predicted_depth, steps, sigma):
    """depth_loss.py:31-65 (same `.sum(-2)` quirk)."""
    mask = termination_depth > 0
    expected = (termination_depth - predicted_depth) ** 2
    scale = sigma / URF_SIGMA_SCALE_FACTOR
    logp = lambda v: -(v ** 2) / (2 * scale ** 2) - np.log(scale) - np.log(np.sqrt(2 * np.pi))
    td = termination_depth[:, None]
    near = np.logical_and(steps <= td + sigma, steps >= td - sigma)
    l_near = (near * (weights - np.exp(logp(steps - td))) ** 2).sum(-2)
    l_empty = ((steps < td - sigma) * weights ** 2).sum(-2)
    return np.mean((expected + l_near + l_empty) * mask)


def depth_loss(weights, tdist, termination_depth, predicted_depth, sigma, dirs, depth_loss_type):
    """depth_loss.py:67-102."""
    steps = 0.5 * (tdist[..., :-1] + tdist[..., 1:])
    if depth_loss_type == 'kl':
        lengths = (tdist[..., 1:] - tdist[..., :-1]) * np.linalg.norm(dirs[..., None, :], axis=-1)
        return ds_nerf_depth_loss(weights, termination_depth, steps, lengths, sigma)
    if depth_loss_type == 'urf':
        return urban_radiance_field_depth_loss(weights, termination_depth, predicted_depth, steps, sigma)
    return None


def compute_data_loss(rgb_gt, disps_sup, renderings, ray_history, directions, lossmult=None, data_loss_type='charb',
                      charb_padding=0.001, data_coarse_loss_mult=0., data_loss_mult=1., compute_disp_metrics=True,
                      depth_loss_type='mse', lambda_depth=0.1, depth_sigma=0.01, depth_scale=1.0):
    """train_utils.py:72-146.  Depth terms (:108-129) as written upstream: 'mse' / 'l1' compare the rendering's
    `distance_mean` with the supervision on the masked rays but average over ALL rays (`.mean()` of the masked
    difference)."""
    lossmult = np.ones_like(rgb_gt) if lossmult is None else np.broadcast_to(lossmult, rgb_gt.shape)
    data_losses, depth_losses, mses = [], [], []
    for i, r in enumerate(renderings):
        resid_sq = (r['rgb'] - rgb_gt) ** 2
        denom = lossmult.sum()
        mses.append((lossmult * resid_sq).sum() / denom)
        dl = resid_sq if data_loss_type == 'mse' else np.sqrt(resid_sq + charb_padding ** 2)
        data_losses.append((lossmult * dl).sum() / denom)
        if compute_disp_metrics:
            m = disps_sup > 0
            if depth_loss_type == 'mse':
                depth_losses.append(((m * r['distance_mean'] - m * disps_sup) ** 2).mean())
            elif depth_loss_type == 'l1':
                depth_losses.append(np.abs(m * r['distance_mean'] - m * disps_sup).mean())
            else:
                depth_losses.append(depth_loss(ray_history[i]['weights'], ray_history[i]['tdist'], disps_sup,
                                               r['distance_mean'], depth_sigma * depth_scale, directions, depth_loss_type))
    data_losses = np.array(data_losses)
    if compute_disp_metrics:
        depth_losses = np.array(depth_losses)
        loss = data_coarse_loss_mult * (data_losses[:-1].sum() + lambda_depth * depth_losses[:-1].sum()) + \
            data_loss_mult * (data_losses[-1] + lambda_depth * depth_losses[-1])
    else:
        loss = data_coarse_loss_mult * data_losses[:-1].sum() + data_loss_mult * data_losses[-1]
    return loss, dict(mses=np.array(mses), depth_losses=depth_losses)


def interlevel_loss(ray_history, interlevel_loss_mult=1.0):
    """train_utils.py:149-160."""
    c, w = ray_history[-1]['sdist'], ray_history[-1]['weights']
    total = 0.
    for h in ray_history[:-1]:
        total += np.mean(lossfun_outer(c, w, h['sdist'], h['weights']))
    return interlevel_loss_mult * total


def distortion_loss(ray_history, distortion_loss_mult=0.01):
    """train_utils.py:163-169."""
    return distortion_loss_mult * np.mean(lossfun_distortion(ray_history[-1]['sdist'], ray_history[-1]['weights']))


# ======================================================================================== train_utils.create_train_step
def volumetric_rendering_backward(rgbs, weights, tdist, bg_rgbs, g_rgb=None, g_distance_mean=None):
    """Closed-form gradient of volumetric_rendering's `rgb` and `distance_mean` (render.py:161-216; upstream: autograd)
    w.r.t. the weights and the per-sample colours.  rgb = sum w c + max(0, 1 - acc) bg;
    distance_mean = clip(exp(sum w log t_mid / max(eps, acc)), t_0, t_N) (zero gradient where the clip is active)."""
    eps = np.finfo(np.float32).eps
    acc = weights.sum(-1)
    g_w = np.zeros_like(weights)
    g_rgbs = None
    if g_rgb is not None:
        bg_on = ((1 - acc) > 0)[..., None]
        g_w = g_w + (rgbs * g_rgb[..., None, :]).sum(-1) - bg_on * np.sum(bg_rgbs * g_rgb, -1, keepdims=True)
        g_rgbs = weights[..., None] * g_rgb[..., None, :]
    if g_distance_mean is not None:
        logt = np.log(0.5 * (tdist[..., :-1] + tdist[..., 1:]))
        A = (weights * logt).sum(-1)
        B = np.maximum(eps, acc)
        e = np.exp(A / B)
        inside = (e > tdist[..., 0]) & (e < tdist[..., -1])
        dB = (acc > eps)
        g_w = g_w + (g_distance_mean * inside * e)[..., None] * (logt / B[..., None] - (A / B ** 2 * dB)[..., None])
    return g_w, g_rgbs


def tree_norm(grads):
    """train_utils.py:45-57 over one MLP's (kernel, bias) list."""
    return np.sqrt(sum(float(np.sum(np.square(k.astype(np.float64))) + np.sum(np.square(b.astype(np.float64)))) for k, b in grads))


def new_train_state(prop_params, nerf_params):
    """TrainState.create with optax.adam (train_utils.py:371-395): parameters + zero first / second moments + count."""
    zeros = lambda ps: [(np.zeros_like(k), np.zeros_like(b)) for k, b in ps]
    return dict(count=0, prop=[(k.copy(), b.copy()) for k, b in prop_params], nerf=[(k.copy(), b.copy()) for k, b in nerf_params],
                mu=dict(prop=zeros(prop_params), nerf=zeros(nerf_params)), nu=dict(prop=zeros(prop_params), nerf=zeros(nerf_params)))


def apply_gradients(state, grads, max_steps=250000, lr_init=2e-3, lr_final=2e-5, lr_delay_steps=512, lr_delay_mult=0.01,
                    grad_max_norm=0.001, adam_b1=0.9, adam_b2=0.999, adam_eps=1e-6):
    """train_utils.py:344-347: clip_gradients (per MLP, by global norm, :215-236) -> nan_to_num -> optax.adam with the
    learning-rate schedule evaluated at the PRE-increment count (optax scale_by_schedule) and bias correction at the
    post-increment count (optax.scale_by_adam).  grads: dict(prop=[(dk, db)], nerf=[...]).  Returns the clip multipliers."""
    lr = learning_rate_decay(state['count'], lr_init, lr_final, max_steps, lr_delay_steps, lr_delay_mult)
    t = state['count'] + 1
    mults = {}
    for name in ('prop', 'nerf'):
        g = grads[name]
        mult = 1.0
        if grad_max_norm > 0:
            ratio = grad_max_norm / (EPS32 + tree_norm(g))
            mult = ratio if np.isnan(ratio) else min(1.0, ratio)              # jnp.minimum propagates NaN
        mults[name] = mult
        new_p, new_mu, new_nu = [], [], []
        for (p_k, p_b), (g_k, g_b), (m_k, m_b), (v_k, v_b) in zip(state[name], g, state['mu'][name], state['nu'][name]):
            out = []
            for p_, g_, m_, v_ in ((p_k, g_k, m_k, v_k), (p_b, g_b, m_b, v_b)):
                with np.errstate(invalid='ignore', over='ignore'):
                    gg = np.nan_to_num((mult * g_).astype(p_.dtype))           # jnp.nan_to_num: nan -> 0, inf -> finfo.max
                m2 = adam_b1 * m_ + (1 - adam_b1) * gg
                v2 = adam_b2 * v_ + (1 - adam_b2) * gg * gg
                m_hat = m2 / (1 - adam_b1 ** t)
                v_hat = v2 / (1 - adam_b2 ** t)
                out.append((p_ - lr * m_hat / (np.sqrt(v_hat) + adam_eps), m2, v2))
            new_p.append((out[0][0], out[1][0]))
            new_mu.append((out[0][1], out[1][1]))
            new_nu.append((out[0][2], out[1][2]))
        state[name], state['mu'][name], state['nu'][name] = new_p, new_mu, new_nu
    state['count'] = t
    return mults


def loss_and_grads(prop_params, nerf_params, rays, rgb_gt, disps_sup, train_frac=1.0, jitter01=None, basis=None,
                   data_loss_type='charb', charb_padding=0.001, data_loss_mult=1.0, depth_loss_type='mse', lambda_depth=0.1,
                   depth_sigma=0.01, depth_scale=1.0, interlevel_loss_mult=1.0, distortion_loss_mult=0.01, bg_rgb=1.0, q=None,
                   **model_kw):
    """loss_fn + jax.value_and_grad of train_utils.py:258-333 for configs/360.gin (data_coarse_loss_mult = 0,
    compute_disp_metrics = True): the total of :297 = data (incl. data_loss_mult * lambda * depth[-1], :136-139) +
    stats['loss_disp_mse'] (lambda * sum of ALL levels' depth terms, :143, :268-269) + interlevel + distortion, and its
    gradient w.r.t. both MLPs by the closed forms of this module (sample positions carry no gradient: models.py:203-204).
    Returns (stats dict, grads dict(prop=[(dk, db)], nerf=[...]))."""
    caches = []
    renderings, history = model_forward(prop_params, nerf_params, rays, train_frac, jitter01, basis, bg_rgb=bg_rgb,
                                        caches=caches, q=q, **model_kw)
    n = rgb_gt.shape[0]
    data, st = compute_data_loss(rgb_gt, disps_sup, renderings, history, rays['directions'], data_loss_type=data_loss_type,
                                 charb_padding=charb_padding, data_loss_mult=data_loss_mult, depth_loss_type=depth_loss_type,
                                 lambda_depth=lambda_depth, depth_sigma=depth_sigma, depth_scale=depth_scale)
    dep = st['depth_losses']
    disp = lambda_depth * dep.sum()
    inter = interlevel_loss(history, interlevel_loss_mult)
    dist = distortion_loss(history, distortion_loss_mult)
    stats = dict(loss=data + disp + inter + dist, data=data, loss_disp_mse=disp, interlevel=inter, distortion=dist,
                 depth_losses=dep, mses=st['mses'])
    L = len(history)
    # ---- d total / d (rgb, distance_mean, weights) per level
    g_w = [np.zeros_like(h['weights']) for h in history]
    g_dm = [np.zeros(n, h['weights'].dtype) for h in history]
    resid = renderings[-1]['rgb'] - rgb_gt
    if data_loss_type == 'charb':
        g_rgb = data_loss_mult * resid / np.sqrt(resid ** 2 + charb_padding ** 2) / (3 * n)
    else:
        g_rgb = data_loss_mult * 2 * resid / (3 * n)
    m = (disps_sup > 0).astype(g_rgb.dtype)
    sigma = depth_sigma * depth_scale
    for i in range(L):
        k = lambda_depth * (1.0 + (data_loss_mult if i == L - 1 else 0.0))            # :136-143
        if depth_loss_type in ('mse', 'l1'):
            diff = m * renderings[i]['distance_mean'] - m * disps_sup
            g_dm[i] = k * (2 * diff if depth_loss_type == 'mse' else np.sign(diff)) * m / n
        elif depth_loss_type in ('kl', 'urf'):
            gw, gd = depth_loss_grads(history[i]['weights'], history[i]['tdist'], disps_sup, renderings[i]['distance_mean'],
                                      sigma, rays['directions'], depth_loss_type)
            g_w[i] = g_w[i] + k * gw
            g_dm[i] = g_dm[i] + k * gd
    c, w = history[-1]['sdist'], history[-1]['weights']
    g_w[-1] = g_w[-1] + distortion_loss_mult * lossfun_distortion_grad_w(c, w) / n
    for i in range(L - 1):
        g_w[i] = g_w[i] + interlevel_loss_mult * lossfun_outer_grad_w_env(c, w, history[i]['sdist'], history[i]['weights']) / \
            (n * w.shape[-1])
    # ---- back through the compositing and the MLPs
    grads = dict(prop=None, nerf=None)
    for i in range(L):
        is_nerf = i == L - 1
        h = history[i]
        gw_r, g_rgbs = volumetric_rendering_backward(h['rgb'], h['weights'], h['tdist'], bg_rgb, g_rgb if is_nerf else None, g_dm[i])
        g_density = alpha_weights_backward(h['density'], h['tdist'], rays['directions'], g_w[i] + gw_r, True)
        gl = mlp_backward(nerf_params if is_nerf else prop_params, caches[i], g_density, g_rgbs if is_nerf else None, q=q)
        key = 'nerf' if is_nerf else 'prop'
        grads[key] = gl if grads[key] is None else [(a + c_, b + d_) for (a, b), (c_, d_) in zip(grads[key], gl)]
    return stats, grads


def depth_loss_grads(weights, tdist, termination_depth, predicted_depth, sigma, dirs, depth_loss_type):
    """Closed-form gradient of depth_loss() w.r.t. (weights, predicted_depth) with upstream's `.sum(-2)` / mask broadcast
    (n == S: column s meets the mask and expected term of ray s; n == 1: of ray 0)."""
    n, S = weights.shape
    if n != S and n != 1:
        raise ValueError('operands could not be broadcast together with shapes (%d,) (%d,)' % (S, n))
    steps = 0.5 * (tdist[..., :-1] + tdist[..., 1:])
    mask = (termination_depth > 0).astype(weights.dtype)
    mcol = np.broadcast_to(mask, (S,))                                            # [S]: mask met by column s
    td = termination_depth[:, None]
    g_dm = np.zeros(n, weights.dtype)
    if depth_loss_type == 'kl':
        lengths = (tdist[..., 1:] - tdist[..., :-1]) * np.linalg.norm(dirs[..., None, :], axis=-1)
        g_w = -np.exp(-((steps - td) ** 2) / (2 * sigma)) * lengths / (weights + 1e-7) * mcol[None, :] / S
    else:
        scale = sigma / URF_SIGMA_SCALE_FACTOR
        pdf = np.exp(-((steps - td) ** 2) / (2 * scale ** 2) - np.log(scale) - np.log(np.sqrt(2 * np.pi)))
        near = np.logical_and(steps <= td + sigma, steps >= td - sigma)
        empty = steps < td - sigma
        g_w = (near * 2 * (weights - pdf) + empty * 2 * weights) * mcol[None, :] / S
        e = -2 * (termination_depth - predicted_depth) * mask                     # d expected / d pred, masked by the own ray
        g_dm = e / S if n == S else e * 1.0                                        # n == 1: S columns x 1/S
    return g_w, g_dm


def train_step(state, rays, rgb_gt, disps_sup, train_frac=None, jitter01=None, max_steps=250000, grad_max_norm=0.001,
               adam_eps=1e-6, **loss_kw):
    """train_utils.create_train_step's train_step (:259-364) on one device: loss_fn, gradients, [pmean is the identity],
    clip, nan_to_num, Adam update of both MLPs.  train_frac defaults to count / (max_steps - 1) like train.py feeds it.
    Mutates `state`; returns (stats, grads before clipping, clip multipliers)."""
    if train_frac is None:
        train_frac = float(np.clip(state['count'] / (max_steps - 1), 0, 1))
    stats, grads = loss_and_grads(state['prop'], state['nerf'], rays, rgb_gt, disps_sup, train_frac, jitter01, **loss_kw)
    mults = apply_gradients(state, grads, max_steps=max_steps, grad_max_norm=grad_max_norm, adam_eps=adam_eps)
    return stats, grads, mults
