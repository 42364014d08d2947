"""Host-side mirror of the reference's call face, backed by libnerfpp_hip.so.

Same names and argument meaning as nerf-methods/nerfplusplus/ddp_train_nerf.py
(`intersect_sphere`, `perturb_samples`, `sample_pdf`) and depth_loss.py / utils.py; tensors are
torch CUDA(=HIP) float32 tensors, kernels run on torch's current stream.  PyTorch is only the
device-memory / stream plumbing here -- every arithmetic step is a HIP kernel behind the C ABI.
"""
import ctypes as C
from collections import OrderedDict

import torch

from . import _lib as L


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32(t, shape=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise L.NerfppError('expected a CUDA/HIP tensor (the NeRF++ hot path has no CPU fallback)')
    t = t.contiguous()
    if t.dtype != torch.float32:
        t = t.float()
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise L.NerfppError('bad tensor shape %s, expected %s' % (tuple(t.shape), tuple(shape)))
    return t


# ------------------------------------------------------------------------------------- sampling
def intersect_sphere(ray_o, ray_d):
    """ddp_train_nerf.py:51-66.  Raises like the reference if a camera is outside the unit sphere."""
    ray_o, ray_d = _f32(ray_o), _f32(ray_d)
    n = ray_o.shape[0]
    far = torch.empty(n, device=ray_o.device)
    bad = torch.zeros(1, dtype=torch.int32, device=ray_o.device)
    L.check(L.lib().nerfpp_intersect_sphere(_stream(), n, _p(ray_o), _p(ray_d), _p(far), _p(bad)),
            'nerfpp_intersect_sphere')
    if int(bad.item()) != 0:
        raise Exception(CAMERA_ERROR)
    return far


CAMERA_ERROR = ('Not all your cameras are bounded by the unit sphere; please make sure the '
                'cameras are normalized properly!')                      # ddp_train_nerf.py:62-63


def rng_uniform(seed, step, stream_id, shape, device):
    """The uniforms the in-kernel generator yields for (seed, step, stream_id): stream 0 / 1 = the
    stratified jitter of the fg / bg depths, 2 / 3 = the fg / bg sample_pdf draws."""
    out = torch.empty(shape, device=device)
    L.check(L.lib().nerfpp_rng_uniform(_stream(), int(seed), int(step), int(stream_id), out.numel(), _p(out)),
            'nerfpp_rng_uniform')
    return out


def sample_pixels(n_pixels, n_rays, seed, step, device):
    """np.random.choice(H * W, size=(N_rand,), replace=False) of nerf_sample_ray_split.py:178 on the device: n_rays distinct
    flat pixel indices (int64), uniform over ordered tuples of distinct values, from Philox stream 4 of (seed, step)."""
    pix = torch.empty(n_rays, dtype=torch.int64, device=device)
    L.check(L.lib().nerfpp_sample_pixels(_stream(), int(seed), int(step), int(n_pixels), int(n_rays), _p(pix)),
            'nerfpp_sample_pixels')
    return pix


def sample_coarse(ray_o, ray_d, min_depth, n_samples, t_rand_fg=None, t_rand_bg=None, perturb=True,
                  check=True, rng=None, bad=None):
    """Level-0 depths of ddp_train_nerf.py:438-449 (perturb=True, training) / :166-175 (render).
    rng = (seed, step): draw the jitter inside the kernel (Philox) instead of from t_rand_* / torch.rand.
    bad: optional persistent int32[1] device counter of rays outside the unit sphere (not zeroed here; the
    caller reads it where it synchronises anyway and raises the reference's exception).
    Returns fg_far [n], fg_z [n,S], bg_z [n,S]."""
    ray_o, ray_d, min_depth = _f32(ray_o), _f32(ray_d), _f32(min_depth)
    n = ray_o.shape[0]
    dev = ray_o.device
    if rng is not None and perturb:
        far = torch.empty(n, device=dev)
        fg_z = torch.empty(n, n_samples, device=dev)
        bg_z = torch.empty(n, n_samples, device=dev)
        if bad is None:
            bad = torch.zeros(1, dtype=torch.int32, device=dev)
        L.check(L.lib().nerfpp_sample_coarse_rng(_stream(), n, n_samples, _p(ray_o), _p(ray_d), _p(min_depth),
                                                 int(rng[0]), int(rng[1]), _p(far), _p(fg_z), _p(bg_z), _p(bad)),
                'nerfpp_sample_coarse_rng')
        if check and int(bad.item()) != 0:
            raise Exception(CAMERA_ERROR)
        return far, fg_z, bg_z
    if perturb:
        # RNG call order of the reference: fg rand_like, then bg rand_like
        if t_rand_fg is None:
            t_rand_fg = torch.rand(n, n_samples, device=dev)
        if t_rand_bg is None:
            t_rand_bg = torch.rand(n, n_samples, device=dev)
        t_rand_fg, t_rand_bg = _f32(t_rand_fg, (n, n_samples)), _f32(t_rand_bg, (n, n_samples))
    else:
        t_rand_fg = t_rand_bg = None
    far = torch.empty(n, device=dev)
    fg_z = torch.empty(n, n_samples, device=dev)
    bg_z = torch.empty(n, n_samples, device=dev)
    if bad is None:
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
    L.check(L.lib().nerfpp_sample_coarse(_stream(), n, n_samples, _p(ray_o), _p(ray_d), _p(min_depth),
                                         _p(t_rand_fg), _p(t_rand_bg), _p(far), _p(fg_z), _p(bg_z), _p(bad)),
            'nerfpp_sample_coarse')
    if check and int(bad.item()) != 0:
        raise Exception(CAMERA_ERROR)
    return far, fg_z, bg_z


def perturb_samples(z_vals, t_rand=None):
    """ddp_train_nerf.py:69-78; t_rand defaults to torch.rand_like(z_vals) like the reference."""
    z_vals = _f32(z_vals)
    if t_rand is None:
        t_rand = torch.rand_like(z_vals)
    t_rand = _f32(t_rand, z_vals.shape)
    out = torch.empty_like(z_vals)
    n, S = z_vals.shape
    L.check(L.lib().nerfpp_perturb_samples(_stream(), n, S, _p(z_vals), _p(t_rand), _p(out)),
            'nerfpp_perturb_samples')
    return out


def sample_pdf(bins, weights, N_samples, det=False, u=None, return_inds=False):
    """ddp_train_nerf.py:81-130.  bins [n,M+1], weights [n,M] -> samples [n,N_samples]
    (and the int64 `above_inds` when return_inds)."""
    bins, weights = _f32(bins), _f32(weights)
    n, M = weights.shape
    if bins.shape != (n, M + 1):
        raise L.NerfppError('bins must be [n, M+1]')
    if not det and u is None:
        u = torch.rand(n, N_samples, device=bins.device)
    u = None if det else _f32(u, (n, N_samples))
    samples = torch.empty(n, N_samples, device=bins.device)
    above = torch.empty(n, N_samples, dtype=torch.int64, device=bins.device) if return_inds else None
    L.check(L.lib().nerfpp_sample_pdf(_stream(), n, M, N_samples, _p(bins), _p(weights), _p(u), _p(samples),
                                      _p(above)), 'nerfpp_sample_pdf')
    return (samples, above) if return_inds else samples


def sample_fine(z_old, weights, N_samples, det=False, u=None, return_all=False):
    """ddp_train_nerf.py:450-465 for one volume: mids -> sample_pdf(weights[..., 1:-1]) ->
    sort(cat(z_old, samples)).  `weights` is ret['fg_weights'] or ret['bg_weights'] as returned."""
    z_old, weights = _f32(z_old), _f32(weights)
    n, S_old = z_old.shape
    if weights.shape != (n, S_old):
        raise L.NerfppError('weights must be [n, S_old]')
    if not det and u is None:
        u = torch.rand(n, N_samples, device=z_old.device)
    u = None if det else _f32(u, (n, N_samples))
    merged = torch.empty(n, S_old + N_samples, device=z_old.device)
    samples = torch.empty(n, N_samples, device=z_old.device) if return_all else None
    above = torch.empty(n, N_samples, dtype=torch.int64, device=z_old.device) if return_all else None
    L.check(L.lib().nerfpp_sample_fine(_stream(), n, S_old, N_samples, _p(z_old), _p(weights), _p(u), _p(merged),
                                       _p(samples), _p(above)), 'nerfpp_sample_fine')
    return (merged, samples, above) if return_all else merged


def sample_fine_pair(fg_z, fg_weights, bg_z, bg_weights, N_samples, det=False, u_fg=None, u_bg=None, rng=None):
    """Both volumes of a level in one launch (same arithmetic as two sample_fine calls; the uniforms are
    drawn fg first, then bg, like the reference's two sample_pdf calls).  rng = (seed, step): draw them
    inside the kernel (Philox streams 2 and 3)."""
    fg_z, fg_weights, bg_z, bg_weights = _f32(fg_z), _f32(fg_weights), _f32(bg_z), _f32(bg_weights)
    n, S_old = fg_z.shape
    if fg_weights.shape != (n, S_old) or bg_z.shape != (n, S_old) or bg_weights.shape != (n, S_old):
        raise L.NerfppError('z / weights must all be [n, S_old]')
    if rng is not None and not det:
        fg_m = torch.empty(n, S_old + N_samples, device=fg_z.device)
        bg_m = torch.empty(n, S_old + N_samples, device=fg_z.device)
        L.check(L.lib().nerfpp_sample_fine_pair_rng(_stream(), n, S_old, N_samples, _p(fg_z), _p(fg_weights), _p(fg_m),
                                                    _p(bg_z), _p(bg_weights), _p(bg_m), int(rng[0]), int(rng[1])),
                'nerfpp_sample_fine_pair_rng')
        return fg_m, bg_m
    if det:
        u_fg = u_bg = None
    else:
        u_fg = _f32(u_fg, (n, N_samples)) if u_fg is not None else torch.rand(n, N_samples, device=fg_z.device)
        u_bg = _f32(u_bg, (n, N_samples)) if u_bg is not None else torch.rand(n, N_samples, device=fg_z.device)
    fg_m = torch.empty(n, S_old + N_samples, device=fg_z.device)
    bg_m = torch.empty(n, S_old + N_samples, device=fg_z.device)
    L.check(L.lib().nerfpp_sample_fine_pair(_stream(), n, S_old, N_samples, _p(fg_z), _p(fg_weights), _p(u_fg),
                                            _p(fg_m), _p(bg_z), _p(bg_weights), _p(u_bg), _p(bg_m)),
            'nerfpp_sample_fine_pair')
    return fg_m, bg_m


# ------------------------------------------------------------------------------------- one level
_TABLES = {}


def level_tables(device):
    """Device copy of the index tables (built once per device by the library's host code)."""
    key = str(device)
    if key not in _TABLES:
        _TABLES[key] = torch.from_numpy(L.build_level_tables()).to(device)
    return _TABLES[key]


RET_KEYS = ('rgb', 'fg_weights', 'bg_weights', 'fg_dists', 'fg_rgb', 'fg_depth', 'bg_rgb', 'bg_depth',
            'bg_lambda', 'depth')            # ddp_model.py:136-146 order


class LevelEngine(object):
    """One cascade level's NerfNet (fg_net + bg_net) on the HIP path: flat float32 parameters in
    NerfNet.parameters() order, their packed MFMA streams, and the forward / backward calls."""

    def __init__(self, params_flat, precision=L.PREC_SPLIT_BF16):
        """precision: PREC_BF16, PREC_SPLIT_BF16, or a combination with the single-pass bf16 backward over the bf16 planes the
        forward saved: PREC_SPLIT_FWD = split-bf16 forward (rendered outputs and loss within 1e-4 of float32), PREC_FP16_FWD =
        fp16x2w forward (NERFPP_PREC_FP16X2W, two MFMA passes: 1e-4 at initialisation, 3-4e-4 on trained weights)."""
        if params_flat.numel() != L.LEVEL_PARAMS:
            raise L.NerfppError('expected %d parameters, got %d' % (L.LEVEL_PARAMS, params_flat.numel()))
        self.params = _f32(params_flat)
        self.device = self.params.device
        precision = int(precision)
        if precision not in (L.PREC_BF16, L.PREC_SPLIT_BF16, L.PREC_FP16_FWD, L.PREC_SPLIT_FWD):
            raise L.NerfppError('unknown precision %r' % (precision,))
        self.precision = L.PREC_SPLIT_BF16 if precision == L.PREC_SPLIT_FWD else precision        # forward
        self.bwd_precision = L.PREC_BF16 if precision in (L.PREC_SPLIT_FWD, L.PREC_FP16_FWD) else precision
        self.tables = level_tables(self.device)
        self.packed = torch.empty(L.lib().nerfpp_packed_bytes(self.precision), dtype=torch.uint8,
                                  device=self.device)
        self.packed_bwd = self.packed if self.bwd_precision == self.precision else \
            torch.empty(L.lib().nerfpp_packed_bytes(self.bwd_precision), dtype=torch.uint8, device=self.device)
        self.workspace = None
        self._fwd = None
        self.repack()

    def repack(self):
        """Re-derive the bf16 weight streams from the float32 master parameters (after Adam)."""
        L.check(L.lib().nerfpp_pack_level(_stream(), self.precision, _p(self.params), _p(self.tables),
                                          _p(self.packed)), 'nerfpp_pack_level')
        if self.packed_bwd is not self.packed:
            L.check(L.lib().nerfpp_pack_level(_stream(), self.bwd_precision, _p(self.params), _p(self.tables),
                                              _p(self.packed_bwd)), 'nerfpp_pack_level')

    def _workspace(self, n, S, training):
        need = L.lib().nerfpp_workspace_bytes(n, S, self.precision, int(training))
        if need < 0:
            raise L.NerfppError('unsupported sizes n_rays=%d n_samples=%d' % (n, S))
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self.workspace

    def forward(self, ray_o, ray_d, fg_z_max, fg_z_vals, bg_z_vals, training=False, events=None):
        """ret = net(ray_o, ray_d, fg_z_max, fg_z_vals, bg_z_vals)      ddp_model.py:74-147"""
        ray_o, ray_d = _f32(ray_o), _f32(ray_d)
        n = ray_o.shape[0]
        fg_z_vals = _f32(fg_z_vals)
        S = fg_z_vals.shape[1]
        bg_z_vals, fg_z_max = _f32(bg_z_vals, (n, S)), _f32(fg_z_max, (n,))
        dev = self.device
        out = OrderedDict()
        for k in RET_KEYS:
            shape = (n, 3) if k in ('rgb', 'fg_rgb', 'bg_rgb') else \
                (n, S) if k in ('fg_weights', 'bg_weights', 'fg_dists') else (n,)
            out[k] = torch.empty(shape, device=dev)
        a = L.ForwardArgs()
        # training = 2: split-bf16 forward whose backward is single-pass bf16 (PREC_SPLIT_FWD) -- the kernels save hi planes only
        hi_only = training and self.bwd_precision != self.precision
        a.n_rays, a.n_samples, a.precision, a.training = n, S, self.precision, (2 if hi_only else int(training))
        a.ray_o, a.ray_d, a.fg_far, a.fg_z, a.bg_z = [t.data_ptr() for t in
                                                      (ray_o, ray_d, fg_z_max, fg_z_vals, bg_z_vals)]
        a.packed = self.packed.data_ptr()
        a.workspace = self._workspace(n, S, training).data_ptr()
        for k in RET_KEYS:
            setattr(a, k, out[k].data_ptr())
        if events is not None:          # (begin, end) torch.cuda.Event pair around the fg MLP kernel
            a.ev_mlp_begin, a.ev_mlp_end = events[0].cuda_event, events[1].cuda_event
        L.check(L.lib().nerfpp_level_forward(_stream(), C.byref(a)), 'nerfpp_level_forward')
        if training:
            self._fwd = (n, S, ray_d, fg_z_max, fg_z_vals, bg_z_vals)
        return out

    def saved_tensor(self, net, tensor, plane=0):
        """A saved tensor of the last training-mode forward / backward as a float32 [rows, ld] torch tensor (inspection /
        tests only): un-does the fragment-major layout of include/nerfpp_hip.h (nerfpp_workspace_tensor)."""
        if self._fwd is None:
            raise L.NerfppError('saved_tensor() needs a preceding forward(training=True)')
        n, S = self._fwd[0], self._fwd[1]
        off, ld, pb = C.c_int64(), C.c_int32(), C.c_int64()
        L.check(L.lib().nerfpp_workspace_tensor(n, S, self.precision, int(net), int(tensor), C.byref(off), C.byref(ld),
                                                C.byref(pb)), 'nerfpp_workspace_tensor')
        rows, ld = n * S, ld.value
        rows_p = (rows + 31) // 32 * 32
        nblk = rows_p // 32 * (ld // 16)
        raw = self.workspace[off.value + plane * pb.value: off.value + plane * pb.value + nblk * 1024]
        blk = raw.view(torch.bfloat16).view(rows_p // 32, ld // 16, 32, 2, 8)      # [tile, chunk, j, hi, slot t]
        # slot t of lane-half hi = column 8 (t >> 2) + 4 hi + (t & 3) of the chunk
        blk = blk.view(rows_p // 32, ld // 16, 32, 2, 2, 4).permute(0, 2, 1, 4, 3, 5)   # [tile, j, chunk, t >> 2, hi, t & 3]
        return blk.reshape(rows_p, ld)[:rows].float()

    def backward(self, g_rgb, g_depth, g_fg_weights=None, grad_scale=1.0, out=None, events=None, defer_reduce=False,
                 fused_loss=None, bad_count=None):
        """Gradient of the loss w.r.t. the flat parameters given dL/d rgb, dL/d depth and (KL)
        dL/d fg_weights, for the last training-mode forward.  defer_reduce: stop after the weight-gradient
        GEMMs; `reduce_grads()` (on any stream ordered after this call) then fills the returned tensor.
        fused_loss: instead of g_* (pass None), dict(loss_type, lambda_depth, kl_sigma, ret, rgb_gt, depth_sup): the
        loss head of ddp_train_nerf.py:481-493 is differentiated inside the compositing backward (same arithmetic as
        loss_and_grads, which then only serves the logged scalars and can run off the critical path).
        bad_count: optional int32 device tensor [1] (the `bad` counter of sample_coarse); `out` must then have LEVEL_PARAMS + 1
        elements and the slab-sum launch writes float(bad_count) behind the gradients (nerfpp_backward_args.bad_count)."""
        if self._fwd is None:
            raise L.NerfppError('backward() needs a preceding forward(training=True)')
        n, S, ray_d, fg_far, fg_z, bg_z = self._fwd
        keep = None
        if fused_loss is None:
            g_rgb, g_depth = _f32(g_rgb, (n, 3)), _f32(g_depth, (n,))
            g_fg_weights = _f32(g_fg_weights, (n, S)) if g_fg_weights is not None else None
        else:
            g_rgb = g_depth = g_fg_weights = None
            f = fused_loss
            t = L.LOSS_TYPES[f['loss_type']]
            keep = (_f32(f['ret']['rgb'], (n, 3)), _f32(f['ret']['depth'], (n,)), _f32(f['rgb_gt'], (n, 3)),
                    _f32(f['depth_sup'], (n,)) if t != L.LOSS_RGB_ONLY else None)
        n_out = L.LEVEL_PARAMS + (1 if bad_count is not None else 0)
        grads = out if out is not None else torch.empty(n_out, device=self.device)
        if grads.numel() < n_out or not grads.is_contiguous() or grads.dtype != torch.float32:
            raise L.NerfppError('backward(): `out` needs %d contiguous float32 elements' % n_out)
        a = L.BackwardArgs()
        a.n_rays, a.n_samples, a.precision = n, S, self.bwd_precision
        a.workspace_precision = self.precision
        a.ray_d, a.fg_far, a.fg_z, a.bg_z = [t.data_ptr() for t in (ray_d, fg_far, fg_z, bg_z)]
        a.packed, a.workspace, a.tables = self.packed_bwd.data_ptr(), self.workspace.data_ptr(), self.tables.data_ptr()
        if fused_loss is None:
            a.g_rgb, a.g_depth = g_rgb.data_ptr(), g_depth.data_ptr()
            a.g_fg_weights = g_fg_weights.data_ptr() if g_fg_weights is not None else None
        else:
            a.fused_loss, a.loss_type = 1, t
            a.lambda_depth, a.kl_sigma = float(fused_loss.get('lambda_depth', 1.0)), float(fused_loss.get('kl_sigma', 0.01))
            a.rgb, a.depth, a.rgb_gt = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()
            a.depth_sup = keep[3].data_ptr() if keep[3] is not None else None
        a.grad_scale = float(grad_scale)
        a.grads = grads.data_ptr()
        a.params = self.params.data_ptr()
        if events is not None:          # (bwd begin, bwd end, dw begin, dw end)
            a.ev_bwd_begin, a.ev_bwd_end, a.ev_dw_begin, a.ev_dw_end = [e.cuda_event for e in events]
        a.defer_reduce = int(bool(defer_reduce))
        if bad_count is not None:
            if bad_count.dtype != torch.int32 or bad_count.numel() != 1 or bad_count.device != self.device:
                raise L.NerfppError('bad_count: one int32 on the engine\'s device')
            a.bad_count = bad_count.data_ptr()
            keep = (keep, bad_count)
        L.check(L.lib().nerfpp_level_backward(_stream(), C.byref(a)), 'nerfpp_level_backward')
        # the reduction reads nothing of the batch: keep only what nerfpp_level_reduce_grads looks at alive
        self._bwd_args = (a, grads) if defer_reduce else None
        return grads

    def reduce_grads(self):
        """Second half of backward(defer_reduce=True): split-K slabs -> gradient tensor, on the current stream."""
        if getattr(self, '_bwd_args', None) is None:
            raise L.NerfppError('reduce_grads() needs a preceding backward(defer_reduce=True)')
        a, grads = self._bwd_args
        self._bwd_args = None
        L.check(L.lib().nerfpp_level_reduce_grads(_stream(), C.byref(a)), 'nerfpp_level_reduce_grads')
        return grads


def loss_and_grads(ret, rgb_gt, depth_sup=None, loss_type='rgbonly', lambda_depth=1.0, kl_sigma=0.01,
                   fg_z_vals=None, fg_far_depth=None):
    """Loss head of ddp_train_nerf.py:481-493 fused with its gradient.
    Returns (scalars [4] = loss, rgb_loss, depth_loss, n_valid; g_rgb, g_depth, g_fg_weights|None)."""
    rgb = _f32(ret['rgb'])
    n = rgb.shape[0]
    S = ret['fg_weights'].shape[1]
    dev = rgb.device
    t = L.LOSS_TYPES[loss_type]
    scalars = torch.empty(4, device=dev)
    g_rgb = torch.empty(n, 3, device=dev)
    g_depth = torch.empty(n, device=dev)
    g_w = torch.empty(n, S, device=dev) if t == L.LOSS_KL else None
    rgb_gt = _f32(rgb_gt, (n, 3))
    depth_sup = _f32(depth_sup, (n,)) if depth_sup is not None else None
    fg_z_vals = _f32(fg_z_vals) if fg_z_vals is not None else None
    fg_far_depth = _f32(fg_far_depth) if fg_far_depth is not None else None
    L.check(L.lib().nerfpp_loss(_stream(), n, S, t, float(lambda_depth), float(kl_sigma), _p(rgb), _p(rgb_gt),
                                _p(_f32(ret['depth'])), _p(depth_sup), _p(_f32(ret['fg_weights'])), _p(fg_z_vals),
                                _p(_f32(ret['fg_dists'])), _p(fg_far_depth), _p(scalars), _p(g_rgb), _p(g_depth),
                                _p(g_w)), 'nerfpp_loss')
    return scalars, g_rgb, g_depth, g_w


def adam_step(params, grads, exp_avg, exp_avg_sq, step, lr=5e-4, beta1=0.9, beta2=0.999, eps=1e-8, skip=None):
    """torch.optim.Adam(lr) single step on flat tensors, in place (ddp_train_nerf.py:324,498).
    skip: optional float32 device tensor [1]; the update is dropped on the device when it is non-zero."""
    if skip is not None and (skip.dtype != torch.float32 or not skip.is_cuda):
        raise L.NerfppError('adam_step: skip must be a float32 device tensor')
    L.check(L.lib().nerfpp_adam_step(_stream(), _p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq),
                                     params.numel(), int(step), lr, beta1, beta2, eps, _p(skip)), 'nerfpp_adam_step')
