"""Host-side mirror of the MipNeRF-360 path (BASELINE config 5) on libmip360_hip.so.

Upstream (nerf-methods/mipnerf360, JAX) has no FFI; this module mirrors its call face -- `Model.__call__`
(internal/models.py:76-303) as `Mip360Model.forward`, `MLP.__call__` (:436-606) as `mlp_forward`, the loss terms of
`train_utils.py:72-169` as `losses` -- with the arithmetic in HIP kernels behind the C ABI of include/mip360_hip.h.
PyTorch is device memory and streams only.  There is no CPU fallback: `lib()` raises if the library is missing.

Built: the whole training step of train_utils.create_train_step (:239-370) -- forward pass of the three sampling levels
(resampling, cone casting + contraction + IPE, the PropMLP / NerfMLP dense layers on the matrix cores, compositing), the
loss terms with their gradients (charb / mse data term; mse / l1 / kl / urf depth terms; interlevel; distortion), the
compositing and MLP backward (dX, dW), per-MLP gradient clipping + nan_to_num, Adam with the upstream learning-rate schedule
and the pmean data-parallel step -- parity-tested link by link and end to end against oracle/mip360_oracle.py.
"""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MIP360_HIP_LIB') or os.path.join(_HERE, 'libmip360_hip.so')
ABI_VERSION = 8
N_BASIS, IPE_DIM, IPE_LD = 21, 504, 512
_fp = C.c_void_p
_fpp = C.POINTER(C.c_void_p)

class PackDesc(C.Structure):
    """include/mip360_hip.h: mip360_pack_desc"""
    _fields_ = [('kernel', C.c_void_p), ('n_in', C.c_int32), ('n_out', C.c_int32), ('fwd_bf16', C.c_void_p), ('bwd_bf16', C.c_void_p),
                ('fwd_fm', C.c_void_p), ('bwd_fm', C.c_void_p), ('ld_fwd', C.c_int32), ('ld_bwd', C.c_int32), ('ld_fwd_fm', C.c_int32),
                ('ld_bwd_fm', C.c_int32), ('bwd_rows', C.c_int32), ('bwd_col0', C.c_int32)]


SYMBOLS = {
    'mip360_last_error': (C.c_char_p, []),
    'mip360_abi_version': (C.c_int, []),
    'mip360_resample': (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, C.c_float, C.c_float, C.c_float, C.c_int, _fp, C.c_float,
                                  C.c_float, _fp, _fp, _fp, _fp]),
    'mip360_cast_encode': (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int]),
    'mip360_render_level': (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, C.c_int, C.c_float, _fp, _fp, _fp, _fp, _fp]),
    'mip360_render_level_backward': (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, C.c_int, C.c_float, _fp, _fp, _fp,
                                               _fp, _fp]),
    'mip360_losses': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, _fpp, _fpp, C.c_int,
                                C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _fp, _fp, _fp, _fp,
                                _fpp, _fp, C.c_float, _fpp, _fpp]),
    'mip360_depth_loss_klurf': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, C.c_float, C.c_float, _fp, _fp,
                                          _fp, _fp]),
    'mip360_linear_bf16': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_float, _fp,
                                     C.c_int, _fp, C.c_int, _fp, C.c_int]),
    'mip360_relu_mask_bytes': (C.c_int64, [C.c_int, C.c_int, _fp]),
    'mip360_linear_relu_mask_bf16': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, _fp, C.c_int, _fp,
                                               C.c_int]),
    'mip360_linear_masked_bf16': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int]),
    'mip360_to_fm': (C.c_int, [_fp, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int]),
    'mip360_from_fm': (C.c_int, [_fp, C.c_int, C.c_int, _fp, C.c_int, C.c_int, _fp, C.c_int]),
    'mip360_fm_mask_bytes': (C.c_int64, [C.c_int, C.c_int]),
    'mip360_linear_fm': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp]),
    'mip360_grad_weight_fm': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, _fp, _fp, C.c_int,
                                        C.c_float, _fp]),
    'mip360_grad_weight_fm_multi': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), _fpp, C.POINTER(C.c_int),
                                              _fpp, C.POINTER(C.c_int), _fpp]),
    'mip360_rowdot_fm': (C.c_int, [_fp, C.c_int, C.c_int, _fp, C.c_int, _fp, _fp, C.c_int, C.c_float, _fp, C.c_int]),
    'mip360_grad_weight_col_fm': (C.c_int, [_fp, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_float, _fp]),
    'mip360_outer_masked_fm': (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, C.c_int]),
    'mip360_prop_mlp_fm': (C.c_int, [_fp, C.c_int, _fp, C.c_int, C.c_int, _fpp, C.POINTER(C.c_int), _fpp, _fpp, _fpp, _fp, _fp,
                                     C.c_float, _fp]),
    'mip360_prop_mlp_bwd_fm': (C.c_int, [_fp, C.c_int, _fp, _fp, _fpp, _fpp, C.POINTER(C.c_int), _fpp]),
    'mip360_view_branch_bwd_fm': (C.c_int, [_fp, C.c_int, _fp, _fp, _fp, _fp, C.c_float, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, _fp,
                                            C.c_int, _fp]),
    'mip360_view_branch_fm': (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, _fp, C.c_int, _fp, _fp, C.c_int, _fp, C.c_float, _fp, C.c_int,
                                        _fp, C.c_int, _fp]),
    'mip360_pack_weight_fm': (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int]),
    'mip360_pack_weights_fm_batch': (C.c_int, [_fp, C.c_int, _fp]),
    'mip360_grad_weight_tile': (C.c_int, [C.c_int] * 5),
    'mip360_grad_weight_bf16': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, _fp, _fp, C.c_int,
                                          C.c_float, _fp]),
    'mip360_grad_weight_reduce': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_int, C.c_float, _fp]),
    'mip360_grad_bias_bf16': (C.c_int, [_fp, C.c_int, C.c_int, _fp, C.c_int, C.c_int, _fp, _fp, C.c_float]),
    'mip360_head_backward': (C.c_int, [_fp, C.c_int64, _fp, _fp, _fp, _fp, C.c_float, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    'mip360_sum_squares': (C.c_int, [_fp, C.c_int64, _fp, _fp, C.c_int]),
    'mip360_clip_multiplier': (C.c_int, [_fp, C.c_int, _fp, C.c_float, _fp]),
    'mip360_adam_step': (C.c_int, [_fp, C.c_int64, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_double, C.c_double, C.c_double,
                                   C.c_double]),
    'mip360_pack_weight': (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, C.c_int, _fp, C.c_int]),
    'mip360_dir_encode': (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int]),
}
_lib = None


class Mip360Error(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Mip360Error('libmip360_hip.so not found at %s -- build it with `python -c "import __graft_entry__ as g; '
                              'g.build()"`. There is no CPU fallback for the MipNeRF-360 path.' % LIB_PATH)
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        if h.mip360_abi_version() != ABI_VERSION:
            raise Mip360Error('libmip360_hip.so ABI version mismatch')
        _lib = h
    return _lib


def _check(rc, what):
    if rc != 0:
        raise Mip360Error('%s failed (code %d): %s' % (what, rc, lib().mip360_last_error().decode('utf-8', 'replace')))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32(t):
    if not t.is_cuda:
        raise Mip360Error('expected a CUDA/HIP tensor (no CPU fallback)')
    return t.contiguous().float()


def pos_basis_t():
    """models.py:387-389: transpose of geopoly.generate_basis('icosahedron', 2) -- [3, 21] float32 (host numpy).
    The 21 directions are the distinct-up-to-sign vertices of the twice-tesselated icosahedron."""
    a = (np.sqrt(5) + 1) / 2
    verts = np.array([(-1, 0, a), (1, 0, a), (-1, 0, -a), (1, 0, -a), (0, a, 1), (0, a, -1), (0, -a, 1), (0, -a, -1),
                      (a, 1, 0), (-a, 1, 0), (a, -1, 0), (-a, -1, 0)]) / np.sqrt(a + 2)
    faces = [(0, 4, 1), (0, 9, 4), (9, 5, 4), (4, 5, 8), (4, 8, 1), (8, 10, 1), (8, 3, 10), (5, 3, 8), (5, 2, 3), (2, 7, 3),
             (7, 10, 3), (7, 6, 10), (7, 11, 6), (11, 0, 6), (0, 1, 6), (6, 1, 10), (9, 0, 11), (9, 11, 2), (9, 2, 5), (7, 2, 11)]
    bary = np.array([(i, j, 2 - i - j) for i in range(3) for j in range(3 - i)], np.float64) / 2
    pts = np.concatenate([bary @ verts[list(f)] for f in faces], 0)
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    keep = []
    for i, p in enumerate(pts):                       # first occurrence of every vertex, in order of appearance
        if not any(np.sum((p - pts[k]) ** 2) <= 1e-4 for k in keep):
            keep.append(i)
    pts = pts[keep]
    out = []
    for i, p in enumerate(pts):                       # drop the later member of every antipodal pair
        if not any(np.sum((p + q) ** 2) < 1e-4 for q in pts[:i]):
            out.append(p)
    return np.ascontiguousarray(np.array(out)[:, ::-1].T, np.float32)


# ----------------------------------------------------------------------------------------------------- kernels
def resample(sdist, weights, dilation, anneal, num_samples, t_near, t_far, jitter01=None, resample_padding=0.0,
             domain=(0.0, 1.0)):
    """models.py:158-208 for one level.  Returns (sdist', tdist') [n, num_samples + 1]."""
    sdist, weights = _f32(sdist), _f32(weights)
    n, m = weights.shape
    so = torch.empty(n, num_samples + 1, device=sdist.device)
    to = torch.empty_like(so)
    jit = _f32(jitter01).reshape(-1) if jitter01 is not None else None
    _check(lib().mip360_resample(_stream(), n, m, _p(sdist), _p(weights), float(dilation), float(anneal), float(resample_padding),
                                 int(num_samples), _p(jit), float(domain[0]), float(domain[1]), _p(_f32(t_near).reshape(-1)),
                                 _p(_f32(t_far).reshape(-1)), _p(so), _p(to)), 'mip360_resample')
    return so, to


def cast_encode(tdist, origins, directions, radii, basis_t, out=None, bf16=True, ld=IPE_LD):
    """render.cast_rays + contract + lift + IPE -> [n*S, ld] (bf16 by default; `out` may be a column view base)."""
    tdist = _f32(tdist)
    n, S = tdist.shape[0], tdist.shape[1] - 1
    if out is None:
        out = torch.empty(n * S, ld, dtype=torch.bfloat16 if bf16 else torch.float32, device=tdist.device)
    _check(lib().mip360_cast_encode(_stream(), n, S, _p(tdist), _p(_f32(origins)), _p(_f32(directions)),
                                    _p(_f32(radii).reshape(-1)), _p(basis_t), _p(out), int(out.dtype == torch.bfloat16), ld),
           'mip360_cast_encode')
    return out


def cast_encode_fm(tdist, origins, directions, radii, basis_t, out, col0, ld):
    """cast_encode straight into columns [col0, col0 + 512) of the fm tensor `out` (ld columns)"""
    tdist = _f32(tdist)
    n, S = tdist.shape[0], tdist.shape[1] - 1
    _check(lib().mip360_cast_encode(_stream(), n, S, _p(tdist), _p(_f32(origins)), _p(_f32(directions)),
                                    _p(_f32(radii).reshape(-1)), _p(basis_t), C.c_void_p(out.data_ptr() + (col0 // 16) * 1024), 2, ld),
           'mip360_cast_encode')
    return out


def linear(a, w, bias, act=0, act_param=0.0, out_bf16=None, out_f32=None, m=None, n=None, k=None, aux=None):
    """act(A W^T + b): a [M, lda] bf16 (k leading columns used), w [N, ldw] bf16.  act 4: multiply by (aux > 0)."""
    m = a.shape[0] if m is None else m
    n = w.shape[0] if n is None else n
    k = w.shape[1] if k is None else k
    ld = lambda t: 0 if t is None else (t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0)))   # size-1 dims have free strides
    _check(lib().mip360_linear_bf16(_stream(), m, n, k, _p(a), ld(a), _p(w), ld(w), _p(bias), int(act), float(act_param),
                                    _p(out_bf16), ld(out_bf16), _p(out_f32), ld(out_f32), _p(aux), ld(aux)), 'mip360_linear_bf16')


def relu_mask_buffer(m, n, device):
    """(uint8 buffer, ldmask) for the ReLU bit mask of an [m, n] layer output (include/mip360_hip.h)."""
    ld = C.c_int(0)
    nbytes = lib().mip360_relu_mask_bytes(int(m), int(n), C.byref(ld))
    return torch.empty(nbytes, dtype=torch.uint8, device=device), ld.value


def linear_relu_mask(a, w, bias, out_bf16, mask, ldmask, m=None, n=None, k=None):
    """relu(A W^T + b) -> out_bf16, bit (out > 0) -> mask."""
    m = a.shape[0] if m is None else m
    n = w.shape[0] if n is None else n
    k = w.shape[1] if k is None else k
    ld = lambda t: t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    _check(lib().mip360_linear_relu_mask_bf16(_stream(), m, n, k, _p(a), ld(a), _p(w), ld(w), _p(bias), _p(out_bf16), ld(out_bf16),
                                              _p(mask), int(ldmask)), 'mip360_linear_relu_mask_bf16')


def linear_masked(a, w, out_bf16, mask, ldmask, m=None, n=None, k=None):
    """(A W^T) * mask bits -> out_bf16: one step of the dX chain."""
    m = a.shape[0] if m is None else m
    n = w.shape[0] if n is None else n
    k = w.shape[1] if k is None else k
    ld = lambda t: t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    _check(lib().mip360_linear_masked_bf16(_stream(), m, n, k, _p(a), ld(a), _p(w), ld(w), _p(out_bf16), ld(out_bf16), _p(mask),
                                           int(ldmask)), 'mip360_linear_masked_bf16')


# ---- fragment-major ("fm") activations of the wide layers (include/mip360_hip.h, csrc/mip360_fm.hip) ------------------
# MIP360_NO_FM=1 keeps every layer on the row-major kernels (A/B runs); shapes the fm kernels do not take fall back too.
USE_FM = os.environ.get('MIP360_NO_FM') is None


def fm_ok(rows, width):
    return USE_FM and rows % 256 == 0 and width % 256 == 0


def _fm_ptr(buf, col0=0):
    """column col0 (a multiple of 16) of an fm tensor: its blocks follow the row block's earlier columns"""
    return C.c_void_p(buf.data_ptr() + (col0 // 16) * 1024)


def fm_buffer(rows, ld, device):
    return torch.empty(rows * ld, dtype=torch.bfloat16, device=device)


def to_fm(x, out=None, ld=None, col0=0, rows=None, cols=None):
    """row-major bf16 [rows, cols] -> columns [col0, col0 + cols) of an fm tensor with ld columns"""
    rows = x.shape[0] if rows is None else rows
    cols = x.shape[1] if cols is None else cols
    ld = cols if ld is None else ld
    if out is None:
        out = fm_buffer(rows, ld, x.device)
    _check(lib().mip360_to_fm(_stream(), rows, cols, _p(x), x.stride(0), _p(out), ld, col0), 'mip360_to_fm')
    return out


def from_fm(buf, rows, cols, ld=None, col0=0, out=None):
    ld = cols if ld is None else ld
    if out is None:
        out = torch.empty(rows, cols, dtype=torch.bfloat16, device=buf.device)
    _check(lib().mip360_from_fm(_stream(), rows, cols, _p(buf), ld, col0, _p(out), out.stride(0)), 'mip360_from_fm')
    return out


def fm_mask_buffer(m, n, device):
    return torch.empty(lib().mip360_fm_mask_bytes(int(m), int(n)), dtype=torch.uint8, device=device)


def linear_fm(a, w, bias, act, m, n, k, out, mask, lda=None, ldw=None, ldc=None, a_col0=0, out_col0=0):
    """act 0: A W^T + b; 1: relu(.) and the bit mask; 2: (A W^T) masked -- all operands fm."""
    _check(lib().mip360_linear_fm(_stream(), m, n, k, _fm_ptr(a, a_col0), lda or k, _p(w), ldw or k, _p(bias), int(act),
                                  _fm_ptr(out, out_col0), ldc or n, _p(mask)), 'mip360_linear_fm')


# MIP360_NO_BATCH_PACK=1: one mip360_pack_weight_fm launch per parameter tensor instead of one per MLP (A/B runs)
USE_BATCH_PACK = os.environ.get('MIP360_NO_BATCH_PACK') is None
# MIP360_NO_FUSED_PROP=1 keeps the PropMLP forward on four mip360_linear_fm + mip360_rowdot_fm launches (A/B runs)
USE_FUSED_PROP = os.environ.get('MIP360_NO_FUSED_PROP') is None


def fused_prop_ok(cfg, rows):
    """the shapes csrc/mip360_prop.hip takes: the 4 x 256 density-only MLP of configs/360.gin on whole 256-row tiles"""
    return (USE_FM and USE_FUSED_PROP and cfg['disable_rgb'] and cfg['net_width'] == 256 and cfg['net_depth'] == 4
            and rows % 256 == 0)


def prop_mlp_fm(enc_buf, x_col0, ldx, rows, w_fm, ldw, bias, wd, bd, density, h=None, masks=None):
    """The PropMLP forward as one launch (include/mip360_hip.h: mip360_prop_mlp_fm).  h / masks: 4 fm buffers / mask buffers
    (training) or None."""
    arr = lambda ts: (C.c_void_p * 4)(*[_p(t) for t in ts])
    keep = [arr(w_fm), (C.c_int * 4)(*[int(v) for v in ldw]), arr(bias)]
    hp = arr([h_[0] if isinstance(h_, tuple) else h_ for h_ in h]) if h is not None else None
    mp = arr(masks) if masks is not None else None
    _check(lib().mip360_prop_mlp_fm(_stream(), int(rows), _fm_ptr(enc_buf, 0), int(ldx), int(x_col0), keep[0], keep[1], keep[2], hp, mp,
                                    _p(wd), _p(bd), DENSITY_BIAS, _p(density)), 'mip360_prop_mlp_fm')


# MIP360_NO_DEFER_DW=1: the NerfMLP's trunk weight gradients stay one launch per layer between the dX layers (A/B runs)
DEFER_DW_MAX_BYTES = 8 << 30         # dZ bytes the deferred weight-gradient launch may keep alive (mlp_backward_fm)
USE_DEFER_DW = os.environ.get('MIP360_NO_DEFER_DW') is None
# MIP360_NO_MULTI_DW=1: the PropMLP's four weight-gradient GEMMs stay four launches (A/B runs)
USE_MULTI_DW = os.environ.get('MIP360_NO_MULTI_DW') is None
# MIP360_NO_FUSED_VIEW=1 keeps the view branch forward on from_fm + dir_encode + two row-major GEMMs (A/B runs)
USE_FUSED_VIEW = os.environ.get('MIP360_NO_FUSED_VIEW') is None


def view_branch_fm(pk, depth, bott, rows, n_samples, viewdirs, view_in, h, rgb):
    """include/mip360_hip.h: mip360_view_branch_fm on the fm copies pk.w_fm[depth + 2] / [depth + 3]; view_in / h may be None"""
    n_rays = rows // n_samples
    table = torch.empty(n_rays, DIR_LD, dtype=torch.bfloat16, device=bott.device)      # the rays' direction features, once per ray
    _check(lib().mip360_dir_encode(_stream(), n_rays, 1, _p(_f32(viewdirs)), _p(table), DIR_LD, 0, DIR_LD), 'mip360_dir_encode')
    _check(lib().mip360_view_branch_fm(_stream(), int(rows), int(n_samples), _p(bott), _p(table), _p(pk.w_fm[depth + 2]),
                                       BOTTLENECK + DIR_LD, _p(pk.b[depth + 2]), _p(pk.w_fm[depth + 3]), VIEW_WIDTH, _p(pk.b[depth + 3]),
                                       RGB_PADDING, _p(view_in), view_in.stride(0) if view_in is not None else 0, _p(h),
                                       h.stride(0) if h is not None else 0, _p(rgb)), 'mip360_view_branch_fm')


def fused_view_ok(pk, depth, rows):
    return USE_FUSED_VIEW and rows % 256 == 0 and (depth + 3) in pk.w_fm


def prop_mlp_bwd_fm(rows, z, wd, masks, wb_fm, ldwb, dz):
    """The PropMLP dX chain as one launch (include/mip360_hip.h: mip360_prop_mlp_bwd_fm): z bf16 [rows], masks / dz: 4 buffers,
    wb_fm / ldwb: entries 1..3 (entry 0 ignored)."""
    arr = lambda ts: (C.c_void_p * 4)(*[_p(t) for t in ts])
    _check(lib().mip360_prop_mlp_bwd_fm(_stream(), int(rows), _p(z), _p(wd), arr(masks), arr(wb_fm), (C.c_int * 4)(*[int(v) for v in ldwb]),
                                        arr(dz)), 'mip360_prop_mlp_bwd_fm')


def render_level(density, rgb_samples, tdist, directions, opaque_background=True, bg_rgb=1.0):
    density, tdist = _f32(density), _f32(tdist)
    n, S = density.shape
    dev = density.device
    w = torch.empty(n, S, device=dev)
    rgb = torch.empty(n, 3, device=dev) if rgb_samples is not None else None
    acc, dm, depth = torch.empty(n, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev)
    _check(lib().mip360_render_level(_stream(), n, S, _p(density), _p(rgb_samples), _p(tdist), _p(_f32(directions)),
                                     int(opaque_background), float(bg_rgb), _p(w), _p(rgb), _p(acc), _p(dm), _p(depth)),
           'mip360_render_level')
    return dict(weights=w, rgb=rgb, acc=acc, distance_mean=dm, depth=depth)


def render_level_backward(density, rgb_samples, tdist, directions, g_weights=None, g_rgb=None, g_distance_mean=None,
                          opaque_background=True, bg_rgb=1.0):
    density = _f32(density)
    n, S = density.shape
    g_density = torch.empty_like(density)
    g_rgbs = torch.empty(n, S, 3, device=density.device) if rgb_samples is not None else None
    _check(lib().mip360_render_level_backward(_stream(), n, S, _p(density), _p(rgb_samples), _p(_f32(tdist)),
                                              _p(_f32(directions)), int(opaque_background), float(bg_rgb), _p(g_weights),
                                              _p(g_rgb), _p(g_distance_mean), _p(g_density), _p(g_rgbs)),
           'mip360_render_level_backward')
    return g_density, g_rgbs


DEPTH_TYPES = {None: 0, 'none': 0, 'mse': 1, 'l1': 2, 'kl': 3, 'urf': 4}


def depth_loss_klurf(depth_loss_type, weights, tdist, depth_sup, distance_mean, directions, sigma, scale=1.0, g_weights=None,
                     g_distance_mean=None):
    """depth_loss.depth_loss (internal/depth_loss.py:67-102) for 'kl' / 'urf' of one level: returns the value (device
    scalar); g_weights [n,S] / g_distance_mean [n], when given, are ACCUMULATED with scale * gradient.  Shapes that
    upstream's `loss.sum(-2) * depth_mask` cannot broadcast (n != S and n != 1) raise, like JAX does."""
    n, S = weights.shape
    out = torch.empty(1, device=weights.device)
    _check(lib().mip360_depth_loss_klurf(_stream(), DEPTH_TYPES[depth_loss_type], n, S, _p(_f32(weights)), _p(_f32(tdist)),
                                         _p(_f32(depth_sup)), _p(_f32(distance_mean)) if distance_mean is not None else None,
                                         _p(_f32(directions)) if directions is not None else None, float(sigma), float(scale),
                                         _p(out), _p(g_weights), _p(g_distance_mean), None), 'mip360_depth_loss_klurf')
    return out


def losses(rgb, rgb_gt, distance_mean, depth_sup, sdist_nerf, w_nerf, sdist_prop, w_prop, data_loss_type='charb',
           charb_padding=0.001, data_loss_mult=1.0, depth_loss_type='mse', lambda_depth=0.1, depth_weight=2.0,
           interlevel_loss_mult=1.0, distortion_loss_mult=0.01, dm_prop=None, prop_depth_weight=1.0, tdist_nerf=None,
           tdist_prop=None, directions=None, depth_sigma=0.01):
    """train_utils.py:72-169 + loss_fn :258-300.  depth_weight = 2 / prop_depth_weight = 1 are the reference's
    effective weights: its total adds stats['loss_disp_mse'] = lambda * sum over ALL levels (:143, :268-269) on top of
    the data_loss_mult * lambda * depth[nerf] already inside the data loss.  dm_prop: the proposal levels' distance_mean
    [n] each.  depth_loss_type 'kl' / 'urf' (internal/depth_loss.py, dispatch train_utils.py:121-128) also need every
    level's tdist, the ray directions and depth_sigma (= config.depth_sigma * config.depth_scale).
    Returns (scalars[6], g_rgb, g_distance_mean, g_w_nerf, [g_w_prop], [g_dm_prop])."""
    n, Sn = w_nerf.shape
    Sp = w_prop[0].shape[1] if w_prop else 1
    dev = w_nerf.device
    scalars = torch.empty(6, device=dev)
    g_rgb, g_dm, g_wn = torch.empty(n, 3, device=dev), torch.empty(n, device=dev), torch.empty(n, Sn, device=dev)
    g_wp = [torch.empty(n, Sp, device=dev) for _ in w_prop]
    ws = torch.empty((4 + len(w_prop)) * n, device=dev)
    dm_prop = list(dm_prop) if dm_prop is not None else []
    g_dmp = [torch.empty(n, device=dev) for _ in dm_prop]
    arr = lambda ts: (C.c_void_p * max(1, len(ts)))(*[t.data_ptr() for t in ts])
    dtype = DEPTH_TYPES[depth_loss_type]
    klurf = dtype >= 3
    if klurf:
        if tdist_nerf is None or tdist_prop is None or (dtype == 3 and directions is None):
            raise Mip360Error("depth_loss_type %r needs tdist_nerf, tdist_prop and the ray directions" % depth_loss_type)
        if dtype == 4 and len(dm_prop) != len(w_prop):
            raise Mip360Error("depth_loss_type 'urf' needs the proposal levels' distance_mean (dm_prop)")
    _check(lib().mip360_losses(_stream(), n, Sn, Sp, len(w_prop), _p(_f32(rgb)), _p(_f32(rgb_gt)),
                               _p(distance_mean), _p(depth_sup), _p(_f32(sdist_nerf)), _p(_f32(w_nerf)), arr(sdist_prop),
                               arr(w_prop), int(data_loss_type == 'charb'), float(charb_padding), float(data_loss_mult),
                               0 if klurf else dtype, float(lambda_depth), float(depth_weight), float(interlevel_loss_mult),
                               float(distortion_loss_mult), _p(scalars), _p(g_rgb), _p(g_dm), _p(g_wn), arr(g_wp), _p(ws),
                               float(prop_depth_weight), arr(dm_prop) if dm_prop else None, arr(g_dmp) if dm_prop else None),
           'mip360_losses')
    if klurf:
        # per-level depth_loss.depth_loss values with the weights of train_utils.py:136-143: the NeRF level enters the
        # total as data_loss_mult * lambda (inside `data`) + (depth_weight - 1) * lambda (loss_disp_mse), a proposal
        # level as prop_depth_weight * lambda; their gradients are added to what mip360_losses left in g_w_* / g_d*
        k_nerf = (data_loss_mult + (depth_weight - 1.0)) * lambda_depth
        v = depth_loss_klurf(depth_loss_type, w_nerf, tdist_nerf, depth_sup, distance_mean, directions, depth_sigma, k_nerf,
                             g_wn, g_dm if dtype == 4 else None)
        scalars[2:3] = v
        scalars[0:1] += k_nerf * v
        scalars[5:6] = 0.0
        for k in range(len(w_prop)):
            kp = prop_depth_weight * lambda_depth
            v = depth_loss_klurf(depth_loss_type, w_prop[k], tdist_prop[k], depth_sup, dm_prop[k] if dm_prop else None,
                                 directions, depth_sigma, kp, g_wp[k], g_dmp[k] if (dtype == 4 and g_dmp) else None)
            scalars[5:6] += v
            scalars[0:1] += kp * v
    return scalars, g_rgb, g_dm, g_wn, g_wp, g_dmp


# ----------------------------------------------------------------------------------------------------- MLPs
PROP_CFG = dict(net_depth=4, net_width=256, disable_rgb=True)          # configs/360.gin:12-16
NERF_CFG = dict(net_depth=8, net_width=1024, disable_rgb=False)        # configs/360.gin:17-20
SKIP_LAYER, BOTTLENECK, VIEW_WIDTH, DIR_DIM, DIR_LD = 4, 256, 128, 27, 32
DENSITY_BIAS, RGB_PADDING = -1.0, 0.001


class PackedMLP(object):
    """bf16 device copies of one MLP's flax parameters (list of (kernel [in, out], bias [out]) in construction order,
    models.py:436-606), transposed to [out, in_padded] (K contiguous, K padded to a multiple of 32 with zero columns;
    the skip layer's input is [x (width) | encoding (504 -> 512)], the view layer's [bottleneck (256) | dirs (27 -> 32)])."""

    def __init__(self, params, cfg, device):
        self.cfg = cfg
        self.device = torch.device(device)
        W, depth = cfg['net_width'], cfg['net_depth']
        self.w, self.b = [], []
        for i, (k, b) in enumerate(params):
            k = np.asarray(k, np.float32)
            if i < depth and i > 0 and (i - 1) % SKIP_LAYER == 0 and (i - 1) > 0:      # input = cat([x, enc])
                k = np.concatenate([k[:W], k[W:], np.zeros((IPE_LD - IPE_DIM, k.shape[1]), np.float32)], 0)
            elif i == 0:
                k = np.concatenate([k, np.zeros((IPE_LD - IPE_DIM, k.shape[1]), np.float32)], 0)
            elif not cfg['disable_rgb'] and i == depth + 2:                             # view layer
                k = np.concatenate([k, np.zeros((BOTTLENECK + DIR_LD - k.shape[0], k.shape[1]), np.float32)], 0)
            assert k.shape[0] % 32 == 0, (i, k.shape)
            self.w.append(torch.from_numpy(np.ascontiguousarray(k.T)).to(self.device).to(torch.bfloat16).contiguous())
            self.b.append(torch.from_numpy(np.asarray(b, np.float32)).to(self.device))
        # fm copies of the wide layers' operands (trunk, bottleneck head) for mlp_forward_fm
        self.w_fm = {}
        if USE_FM and W % 256 == 0:
            for i in list(range(depth)) + ([] if cfg['disable_rgb'] else [depth + 1]):
                self.w_fm[i] = to_fm(self.w[i])
            if not cfg['disable_rgb']:                     # the view branch as one launch (mip360_view_branch_fm)
                self.w_fm[depth + 2] = to_fm(self.w[depth + 2])
                w3 = torch.zeros(32, VIEW_WIDTH, dtype=torch.bfloat16, device=self.device)
                w3[:3] = self.w[depth + 3]
                self.w_fm[depth + 3] = to_fm(w3)
        self.mask_scratch = None


def mlp_forward(pk, enc_buf, rows, viewdirs=None, n_rays=None, n_samples=None):
    """MLP.__call__ (models.py:436-606) for 360.gin.  enc_buf: bf16 [rows, W + 512] whose columns [W, W + 512) hold
    the IPE features (cast_encode wrote them there) -- layer 0 reads them in place and the skip layer finds them next to
    the hidden state without a concat.  Returns (density [rows] f32, rgb [rows, 3] f32 or None)."""
    cfg = pk.cfg
    W, depth = cfg['net_width'], cfg['net_depth']
    dev = enc_buf.device
    enc = enc_buf[:, W:]                                           # [rows, 512] view, stride W + 512
    ping = [torch.empty(rows, W, dtype=torch.bfloat16, device=dev) for _ in range(2)]
    x, x_k, nxt = enc, IPE_LD, 0
    for i in range(depth):
        skip_out = (i % SKIP_LAYER == 0 and i > 0)                 # this layer's output is concatenated with the encoding
        out = enc_buf[:, :W] if skip_out else ping[nxt]
        linear(x, pk.w[i], pk.b[i], act=1, out_bf16=out, m=rows, n=W, k=x_k)
        if skip_out:
            x, x_k = enc_buf, W + IPE_LD
        else:
            x, x_k, nxt = out, W, nxt ^ 1
    density = torch.empty(rows, 1, device=dev)
    linear(x, pk.w[depth], pk.b[depth], act=2, act_param=DENSITY_BIAS, out_f32=density, m=rows, n=1, k=x_k)
    if cfg['disable_rgb']:
        return density[:, 0], None
    view_in = torch.empty(rows, BOTTLENECK + DIR_LD, dtype=torch.bfloat16, device=dev)
    linear(x, pk.w[depth + 1], pk.b[depth + 1], act=0, out_bf16=view_in, m=rows, n=BOTTLENECK, k=x_k)
    _check(lib().mip360_dir_encode(_stream(), n_rays, n_samples, _p(_f32(viewdirs)), _p(view_in), view_in.stride(0), BOTTLENECK,
                                   DIR_LD), 'mip360_dir_encode')
    h = torch.empty(rows, VIEW_WIDTH, dtype=torch.bfloat16, device=dev)
    linear(view_in, pk.w[depth + 2], pk.b[depth + 2], act=1, out_bf16=h, m=rows, n=VIEW_WIDTH, k=BOTTLENECK + DIR_LD)
    rgb = torch.empty(rows, 3, device=dev)
    linear(h, pk.w[depth + 3], pk.b[depth + 3], act=3, act_param=RGB_PADDING, out_f32=rgb, m=rows, n=3, k=VIEW_WIDTH)
    return density[:, 0], rgb


def mlp_forward_fm(pk, enc_buf, rows, viewdirs=None, n_rays=None, n_samples=None):
    """mlp_forward with the trunk in the fm layout (enc_buf: fm tensor [rows, W + 512], columns [W, W + 512) from
    cast_encode_fm); the ReLU bit masks of the layers go to one scratch buffer nobody reads."""
    cfg = pk.cfg
    W, depth = cfg['net_width'], cfg['net_depth']
    dev = enc_buf.device
    ld_enc = W + IPE_LD
    if fused_prop_ok(cfg, rows):
        density = torch.empty(rows, 1, device=dev)
        prop_mlp_fm(enc_buf, W, ld_enc, rows, [pk.w_fm[i] for i in range(depth)], [pk.w[i].shape[1] for i in range(depth)],
                    [pk.b[i] for i in range(depth)], pk.w[depth], pk.b[depth], density)
        return density[:, 0], None
    nbytes = lib().mip360_fm_mask_bytes(int(rows), int(W))
    if pk.mask_scratch is None or pk.mask_scratch.numel() < nbytes:
        pk.mask_scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ping = [fm_buffer(rows, W, dev) for _ in range(2)]
    x, x_col0, x_ld, x_k, nxt = enc_buf, W, ld_enc, IPE_LD, 0
    for i in range(depth):
        skip_out = (i % SKIP_LAYER == 0 and i > 0)
        out, out_ld = (enc_buf, ld_enc) if skip_out else (ping[nxt], W)
        linear_fm(x, pk.w_fm[i], pk.b[i], 1, rows, W, x_k, out, pk.mask_scratch, lda=x_ld, ldw=pk.w[i].shape[1], ldc=out_ld, a_col0=x_col0)
        if skip_out:
            x, x_col0, x_ld, x_k = enc_buf, 0, ld_enc, W + IPE_LD
        else:
            x, x_col0, x_ld, x_k, nxt = out, 0, W, W, nxt ^ 1
    density = torch.empty(rows, 1, device=dev)
    _check(lib().mip360_rowdot_fm(_stream(), rows, x_k, _fm_ptr(x, x_col0), x_ld, _p(pk.w[depth]), _p(pk.b[depth]), 2, DENSITY_BIAS,
                                  _p(density), 1), 'mip360_rowdot_fm')
    if cfg['disable_rgb']:
        return density[:, 0], None
    bott = fm_buffer(rows, BOTTLENECK, dev)
    linear_fm(x, pk.w_fm[depth + 1], pk.b[depth + 1], 0, rows, BOTTLENECK, x_k, bott, None, lda=x_ld, ldw=x_k, a_col0=x_col0)
    if fused_view_ok(pk, depth, rows):
        rgb = torch.empty(rows, 3, device=dev)
        view_branch_fm(pk, depth, bott, rows, n_samples, viewdirs, None, None, rgb)
        return density[:, 0], rgb
    view_in = torch.empty(rows, BOTTLENECK + DIR_LD, dtype=torch.bfloat16, device=dev)
    from_fm(bott, rows, BOTTLENECK, out=view_in)
    _check(lib().mip360_dir_encode(_stream(), n_rays, n_samples, _p(_f32(viewdirs)), _p(view_in), view_in.stride(0), BOTTLENECK,
                                   DIR_LD), 'mip360_dir_encode')
    h = torch.empty(rows, VIEW_WIDTH, dtype=torch.bfloat16, device=dev)
    linear(view_in, pk.w[depth + 2], pk.b[depth + 2], act=1, out_bf16=h, m=rows, n=VIEW_WIDTH, k=BOTTLENECK + DIR_LD)
    rgb = torch.empty(rows, 3, device=dev)
    linear(h, pk.w[depth + 3], pk.b[depth + 3], act=3, act_param=RGB_PADDING, out_f32=rgb, m=rows, n=3, k=VIEW_WIDTH)
    return density[:, 0], rgb


class Mip360Model(object):
    """Model.__call__ (models.py:76-303) for configs/360.gin: two proposal levels (64 samples, PropMLP) and one NeRF
    level (32 samples, NerfMLP), stop-gradient between levels, dilation + annealed resampling, opaque background."""

    def __init__(self, prop_params, nerf_params, device, num_prop_samples=64, num_nerf_samples=32, num_levels=3,
                 anneal_slope=10., dilation_multiplier=0.5, dilation_bias=0.0025, bg_rgb=1.0):
        self.device = torch.device(device)
        self.prop = PackedMLP(prop_params, PROP_CFG, device)
        self.nerf = PackedMLP(nerf_params, NERF_CFG, device)
        self.basis_t = torch.from_numpy(pos_basis_t()).to(self.device)
        self.cfg = dict(num_prop_samples=num_prop_samples, num_nerf_samples=num_nerf_samples, num_levels=num_levels,
                        anneal_slope=anneal_slope, dilation_multiplier=dilation_multiplier, dilation_bias=dilation_bias,
                        bg_rgb=bg_rgb)

    def forward(self, rays, train_frac=1.0, jitter01=None):
        """rays: dict of device tensors origins, directions, viewdirs [n,3], radii, near, far [n,1].
        jitter01: None (deterministic) or a list of per-level [n] tensors in [0,1).  Returns (renderings, ray_history)."""
        c = self.cfg
        n = rays['origins'].shape[0]
        dev = self.device
        sdist = torch.tensor([[0., 1.]], device=dev).repeat(n, 1)
        weights = torch.ones(n, 1, device=dev)
        prod = 1
        renderings, history = [], []
        for lvl in range(c['num_levels']):
            is_prop = lvl < c['num_levels'] - 1
            ns = c['num_prop_samples'] if is_prop else c['num_nerf_samples']
            dilation = c['dilation_bias'] + c['dilation_multiplier'] * 1.0 / prod
            prod *= ns
            s = c['anneal_slope']
            anneal = (s * train_frac) / ((s - 1) * train_frac + 1) if s > 0 else 1.
            sdist, tdist = resample(sdist, weights, dilation if lvl > 0 else 0.0, anneal, ns, rays['near'], rays['far'],
                                    None if jitter01 is None else jitter01[lvl])
            pk = self.prop if is_prop else self.nerf
            W = pk.cfg['net_width']
            rows = n * ns
            if fm_ok(rows, W) and pk.w_fm:
                enc_buf = fm_buffer(rows, W + IPE_LD, dev)
                cast_encode_fm(tdist, rays['origins'], rays['directions'], rays['radii'], self.basis_t, enc_buf, W, W + IPE_LD)
                density, rgb = mlp_forward_fm(pk, enc_buf, rows, rays['viewdirs'], n, ns)
            else:
                enc_buf = torch.empty(rows, W + IPE_LD, dtype=torch.bfloat16, device=dev)
                cast_encode(tdist, rays['origins'], rays['directions'], rays['radii'], self.basis_t, out=enc_buf[:, W:],
                            ld=W + IPE_LD)
                density, rgb = mlp_forward(pk, enc_buf, rows, rays['viewdirs'], n, ns)
            density = density.reshape(n, ns)
            rgb_s = rgb.reshape(n, ns, 3) if rgb is not None else None
            r = render_level(density, rgb_s, tdist, rays['directions'], True, c['bg_rgb'])
            weights = r['weights']
            renderings.append(r)
            history.append(dict(sdist=sdist, tdist=tdist, weights=weights, density=density, rgb=rgb_s))
        return renderings, history


# ----------------------------------------------------------------------------------------------------- training
def learning_rate(step, lr_init=2e-3, lr_final=2e-5, max_steps=250000, lr_delay_steps=512, lr_delay_mult=0.01):
    """math.learning_rate_decay (internal/math.py:67-97) with the Config defaults (configs.py:91,118-121)."""
    delay = 1.0
    if lr_delay_steps > 0:
        delay = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
    t = np.clip(step / max_steps, 0, 1)
    return float(delay * np.exp(t * (np.log(lr_final) - np.log(lr_init)) + np.log(lr_init)))


class TrainableMLP(object):
    """One MLP of the model for training: float32 master parameters (flax layout, one flat buffer with per-tensor
    views), Adam moments, gradients, and the bf16 operand copies the dense-layer kernels read -- forward
    [out, in_padded] and backward [in, out_padded] (the density and bottleneck heads share one stacked backward
    operand so that dH_trunk = [d bottleneck | d raw_density] * [K_bottleneck | K_density]^T is a single GEMM)."""

    def __init__(self, params, cfg, device):
        self.cfg, self.device = cfg, torch.device(device)
        self.W, self.depth = cfg['net_width'], cfg['net_depth']
        self.shapes = [tuple(np.asarray(k).shape) for k, _ in params]
        sizes = []
        for (i, o) in self.shapes:
            sizes += [i * o, o]
        self.offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        n = int(self.offsets[-1])
        self.flat = torch.empty(n, device=self.device)
        for t, (k, b) in enumerate(params):
            self.kernel(t).copy_(torch.from_numpy(np.asarray(k, np.float32)))
            self.bias(t).copy_(torch.from_numpy(np.asarray(b, np.float32)))
        self.grads = torch.zeros_like(self.flat)
        self.slabs = {}                                  # per-layer split-K slab buffers of the deferred weight-gradient sums
        self.mu, self.nu = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        D, W = self.depth, self.W
        self.in_pad = []
        for t, (i, o) in enumerate(self.shapes):
            if t == 0:
                self.in_pad.append(IPE_LD)
            elif t < D and i == W + IPE_DIM:
                self.in_pad.append(W + IPE_LD)
            elif not cfg['disable_rgb'] and t == D + 2:
                self.in_pad.append(BOTTLENECK + DIR_LD)
            else:
                self.in_pad.append(i)
        bf = lambda *sh: torch.zeros(*sh, dtype=torch.bfloat16, device=self.device)
        self.w = [bf(o, self.in_pad[t]) for t, (i, o) in enumerate(self.shapes)]           # forward operands
        self.b = [self.bias(t) for t in range(len(self.shapes))]
        self.wb = {}                                                                         # backward operands
        for t in range(1, D):
            self.wb[t] = bf(self.in_pad[t], W)
        # a multiple of 64 (the persistent row-major GEMM)
        self.head_k = (BOTTLENECK + 64) if not cfg['disable_rgb'] else 64
        self.wb['heads'] = bf(W, self.head_k)                    # [K_bottleneck | K_density | 0], or [K_density | 0]
        if not cfg['disable_rgb']:
            self.wb[D + 2] = bf(BOTTLENECK + DIR_LD, VIEW_WIDTH)
            self.wb[D + 3] = bf(VIEW_WIDTH, 32)
        # fm copies of the wide layers' operands: forward [W, in_pad], backward [W (first W inputs), W], heads [W, head_k],
        # and the bottleneck head's forward operand [256, W]
        self.w_fm, self.wb_fm, self.rm_stale = {}, {}, False
        if USE_FM and W % 256 == 0:
            fmz = lambda r, c: torch.zeros(r * c, dtype=torch.bfloat16, device=self.device)     # (the packing never writes padding)
            for t in range(D):
                self.w_fm[t] = fmz(W, self.in_pad[t])
            for t in range(1, D):
                self.wb_fm[t] = fmz(W, W)
            self.wb_fm['heads'] = fmz(W, self.head_k)
            if not cfg['disable_rgb']:
                self.w_fm[D + 1] = fmz(BOTTLENECK, W)
                self.w_fm[D + 2] = fmz(VIEW_WIDTH, BOTTLENECK + DIR_LD)       # the view branch as one launch (mip360_view_branch_fm)
                self.w_fm[D + 3] = fmz(32, VIEW_WIDTH)                        # (3 live rows)
                self.wb_fm[D + 2] = fmz(BOTTLENECK + DIR_LD, VIEW_WIDTH)      # ... and its backward (mip360_view_branch_bwd_fm)
                self.wb_fm[D + 3] = fmz(VIEW_WIDTH, 32)
        self.repack()

    def kernel(self, t, buf=None):
        i, o = self.shapes[t]
        a = int(self.offsets[2 * t])
        return (self.flat if buf is None else buf)[a:a + i * o].view(i, o)

    def bias(self, t, buf=None):
        a = int(self.offsets[2 * t + 1])
        return (self.flat if buf is None else buf)[a:a + self.shapes[t][1]]

    def ensure_rm(self):
        """the row-major copies of the wide layers (skipped by repack(lazy=True)) before a row-major kernel reads them"""
        if self.rm_stale:
            self.repack()

    def repack(self, lazy=False):
        """float32 master -> bf16 operand copies (after every Adam step).  lazy: with the fm copies in use, the row-major
        ones of the wide layers (nobody reads them on the fm path) are left stale until ensure_rm()."""
        D = self.depth
        L = lib()
        skip_rm = bool(lazy and self.w_fm)
        self.rm_stale = skip_rm
        descs = []
        for t, (i, o) in enumerate(self.shapes):
            k = self.kernel(t)
            bwd, ldb = None, 0
            if 1 <= t < D:
                bwd, ldb = self.wb[t], self.W
            elif t == D:                                                      # density head -> column 256 (or 0)
                col = BOTTLENECK if not self.cfg['disable_rgb'] else 0
                bwd, ldb = self.wb['heads'][:, col:], self.head_k
            elif t == D + 1:
                bwd, ldb = self.wb['heads'], self.head_k
            elif t in (D + 2, D + 3):
                bwd, ldb = self.wb[t], self.wb[t].shape[1]
            fwd_fm, bwd_fm, ld_bwd_fm, bwd_rows, bwd_col0 = self.w_fm.get(t), None, 0, 0, 0
            if self.wb_fm:
                if 1 <= t < D:
                    bwd_fm, ld_bwd_fm, bwd_rows = self.wb_fm[t], self.W, self.W
                elif t == D:
                    bwd_fm, ld_bwd_fm, bwd_rows = self.wb_fm['heads'], self.head_k, self.W
                    bwd_col0 = BOTTLENECK if not self.cfg['disable_rgb'] else 0
                elif t == D + 1:
                    bwd_fm, ld_bwd_fm, bwd_rows = self.wb_fm['heads'], self.head_k, self.W
                elif t in (D + 2, D + 3) and t in self.wb_fm:
                    bwd_fm, ld_bwd_fm, bwd_rows = self.wb_fm[t], self.wb[t].shape[1], self.wb[t].shape[0]
            fwd = self.w[t]
            if skip_rm and (t < D or t == D + 1):                # (the density head's vector and the view branch stay current)
                fwd, bwd = None, None
            elif skip_rm and t == D:
                bwd = None
            ptr = lambda x: None if x is None else x.data_ptr()
            descs.append(PackDesc(ptr(k), i, o, ptr(fwd), ptr(bwd), ptr(fwd_fm), ptr(bwd_fm), self.in_pad[t], ldb,
                                  self.in_pad[t] if fwd_fm is not None else 0, ld_bwd_fm, bwd_rows, bwd_col0))
        # one launch for all tensors (twelve small launches on the update stream otherwise run next to the next step's first kernels)
        if not USE_BATCH_PACK:                              # (A/B runs: one launch per tensor)
            for d in descs:
                _check(L.mip360_pack_weight_fm(_stream(), d.n_in, d.n_out, d.kernel, d.fwd_bf16, d.ld_fwd, d.bwd_bf16, d.ld_bwd, d.fwd_fm,
                                               d.ld_fwd_fm, d.bwd_fm, d.ld_bwd_fm, d.bwd_rows, d.bwd_col0), 'mip360_pack_weight_fm')
            return
        for a0 in range(0, len(descs), 16):
            chunk = descs[a0:a0 + 16]
            arr = (PackDesc * len(chunk))(*chunk)
            _check(L.mip360_pack_weights_fm_batch(_stream(), len(chunk), C.cast(arr, C.c_void_p)), 'mip360_pack_weights_fm_batch')

    def state(self):
        return [(self.kernel(t).clone(), self.bias(t).clone()) for t in range(len(self.shapes))]


def _grad_weight(h, dz, n_in, n_out, out, scratch, bias_out=None, rows_out=None, side=None, slabs=None):
    """d kernel = H^T dZ into `out` [rows_out <= n_in, n_out]; bias_out [n_out] = column sums of dZ from the same pass.
    side = (stream, persistent slab dict, key): the split-K slabs are summed on that stream (under the GEMMs that follow on
    the current one) out of a buffer that belongs to this layer alone."""
    m = h.shape[0]
    ld = lambda t: t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    tile = lib().mip360_grad_weight_tile(m, n_in, n_out, ld(h), ld(dz))
    tiles = ((n_in + tile - 1) // tile) * ((n_out + tile - 1) // tile)
    # (small outputs -- the view branch's [128, 3] and [288, 128] -- have small slabs: as many row slices as fill the chip;
    #  tools/probes/mip360_view_dw_ksplit.py: 48 -> 30 us and 50 -> 33 us at 131 072 rows from 64 to 256 / 85 slices)
    cap = 256 if (tile == 256 or n_in * n_out <= 288 * 128) else 64
    ksplit = int(max(1, min(cap, (256 + tiles - 1) // tiles, (m + 31) // 32)))
    if ksplit >= 8 or m >= 8 * 256:
        ksplit = max(8, (ksplit // 8) * 8)            # multiples of 8: one or more whole row slices per XCD
    if n_out == 1 and n_in % 8 == 0 and m >= 256 * 64:
        ksplit = 256                                  # the column-dot kernel: (n_in / 256) x 256 blocks of whole 512-byte row pieces
    need = ksplit * (n_in * n_out + n_out)
    rows_out = n_in if rows_out is None else rows_out
    if side is None:
        if scratch[0] is None or scratch[0].numel() < need:
            scratch[0] = torch.empty(need, device=h.device)
        buf = scratch[0]
        if rows_out == n_in:
            _check(lib().mip360_grad_weight_bf16(_stream(), m, n_in, n_out, _p(h), ld(h), _p(dz), ld(dz), ksplit, _p(buf), _p(out),
                                                 n_out, 1.0, _p(bias_out)), 'mip360_grad_weight_bf16')
            return
        _check(lib().mip360_grad_weight_bf16(_stream(), m, n_in, n_out, _p(h), ld(h), _p(dz), ld(dz), ksplit, _p(buf), None,
                                             n_out, 1.0, _p(bias_out)), 'mip360_grad_weight_bf16')
        _check(lib().mip360_grad_weight_reduce(_stream(), rows_out, n_in, n_out, ksplit, _p(buf), _p(out), n_out, 1.0,
                                               _p(bias_out)), 'mip360_grad_weight_reduce')
        return
    stream, store, key = side
    buf = store.get(key)
    if buf is None or buf.numel() < need:
        buf = store[key] = torch.empty(need, device=h.device)
    _check(lib().mip360_grad_weight_bf16(_stream(), m, n_in, n_out, _p(h), ld(h), _p(dz), ld(dz), ksplit, _p(buf), None, n_out, 1.0,
                                         _p(bias_out)), 'mip360_grad_weight_bf16')
    ev = torch.cuda.Event()
    ev.record()
    stream.wait_event(ev)
    with torch.cuda.stream(stream):
        _check(lib().mip360_grad_weight_reduce(_stream(), rows_out, n_in, n_out, ksplit, _p(buf), _p(out), n_out, 1.0,
                                               _p(bias_out)), 'mip360_grad_weight_reduce')


def _grad_bias(dz, n_out, out, scratch):
    m = dz.shape[0]
    nslice = int(max(1, min(256, m // 256)))
    if scratch[1] is None or scratch[1].numel() < nslice * n_out:
        scratch[1] = torch.empty(nslice * max(n_out, 1024), device=dz.device)
    ld = dz.stride(0) if dz.shape[0] > 1 else max(dz.shape[1], dz.stride(0))
    _check(lib().mip360_grad_bias_bf16(_stream(), m, n_out, _p(dz), ld, nslice, _p(scratch[1]), _p(out), 1.0),
           'mip360_grad_bias_bf16')


def mlp_forward_train(tm, enc_buf, rows, viewdirs, n_rays, n_samples):
    """mlp_forward keeping what the backward needs (every layer's bf16 output, the view-branch input and hidden state)."""
    tm.ensure_rm()
    W, D = tm.W, tm.depth
    dev = enc_buf.device
    bf = lambda c: torch.empty(rows, c, dtype=torch.bfloat16, device=dev)
    enc = enc_buf[:, W:]
    saved = dict(enc_buf=enc_buf, H=[], inputs=[], masks=[])
    x, x_k = enc, IPE_LD
    for i in range(D):
        skip_out = (i % SKIP_LAYER == 0 and i > 0)
        out = enc_buf[:, :W] if skip_out else bf(W)
        mask = relu_mask_buffer(rows, W, dev)                          # 1 bit per element for the dX chain
        linear_relu_mask(x, tm.w[i], tm.b[i], out, mask[0], mask[1], m=rows, n=W, k=x_k)
        saved['inputs'].append((x, x_k))
        saved['H'].append(out)
        saved['masks'].append(mask)
        x, x_k = (enc_buf, W + IPE_LD) if skip_out else (out, W)
    saved['trunk'] = (x, x_k)
    density = torch.empty(rows, 1, device=dev)
    linear(x, tm.w[D], tm.b[D], act=2, act_param=DENSITY_BIAS, out_f32=density, m=rows, n=1, k=x_k)
    saved['density'] = density
    rgb = None
    if not tm.cfg['disable_rgb']:
        view_in = bf(BOTTLENECK + DIR_LD)
        linear(x, tm.w[D + 1], tm.b[D + 1], act=0, out_bf16=view_in, m=rows, n=BOTTLENECK, k=x_k)
        _check(lib().mip360_dir_encode(_stream(), n_rays, n_samples, _p(_f32(viewdirs)), _p(view_in), view_in.stride(0),
                                       BOTTLENECK, DIR_LD), 'mip360_dir_encode')
        h = bf(VIEW_WIDTH)
        linear(view_in, tm.w[D + 2], tm.b[D + 2], act=1, out_bf16=h, m=rows, n=VIEW_WIDTH, k=BOTTLENECK + DIR_LD)
        rgb = torch.empty(rows, 3, device=dev)
        linear(h, tm.w[D + 3], tm.b[D + 3], act=3, act_param=RGB_PADDING, out_f32=rgb, m=rows, n=3, k=VIEW_WIDTH)
        saved.update(view_in=view_in, h=h, rgb=rgb)
    return density[:, 0], rgb, saved


def mlp_forward_train_fm(tm, enc_buf, rows, viewdirs, n_rays, n_samples):
    """mlp_forward_train with the wide layers in the fm layout: enc_buf is an fm tensor [rows, W + 512] whose columns
    [W, W + 512) cast_encode_fm filled; every trunk activation stays fm (the next layer's DMA copies its blocks), the view
    branch (27- / 128-column operands) runs on the row-major kernels behind one from_fm of the 256-column bottleneck."""
    W, D = tm.W, tm.depth
    dev = enc_buf.device
    ld_enc = W + IPE_LD
    saved = dict(fm=True, enc_buf=enc_buf, H=[], inputs=[], masks=[])
    if fused_prop_ok(tm.cfg, rows):
        # one launch for the four layers and the head; it leaves exactly what the loop below would (H_l, masks, density)
        hs = [fm_buffer(rows, W, dev) for _ in range(D)]
        masks = [fm_mask_buffer(rows, W, dev) for _ in range(D)]
        density = torch.empty(rows, 1, device=dev)
        prop_mlp_fm(enc_buf, W, ld_enc, rows, [tm.w_fm[i] for i in range(D)], [tm.in_pad[i] for i in range(D)],
                    [tm.b[i] for i in range(D)], tm.w[D], tm.b[D], density, h=hs, masks=masks)
        saved['inputs'] = [(enc_buf, W, ld_enc, IPE_LD)] + [(hs[i], 0, W, W) for i in range(D - 1)]
        saved['H'] = [(hs[i], W) for i in range(D)]
        saved['masks'] = masks
        saved['trunk'] = (hs[D - 1], 0, W, W)
        saved['density'] = density
        return density[:, 0], None, saved
    x, x_col0, x_ld, x_k = enc_buf, W, ld_enc, IPE_LD
    for i in range(D):
        skip_out = (i % SKIP_LAYER == 0 and i > 0)
        out, out_ld = (enc_buf, ld_enc) if skip_out else (fm_buffer(rows, W, dev), W)
        mask = fm_mask_buffer(rows, W, dev)
        linear_fm(x, tm.w_fm[i], tm.b[i], 1, rows, W, x_k, out, mask, lda=x_ld, ldw=tm.in_pad[i], ldc=out_ld, a_col0=x_col0)
        saved['inputs'].append((x, x_col0, x_ld, x_k))
        saved['H'].append((out, out_ld))
        saved['masks'].append(mask)
        x, x_col0, x_ld, x_k = (enc_buf, 0, ld_enc, W + IPE_LD) if skip_out else (out, 0, W, W)
    saved['trunk'] = (x, x_col0, x_ld, x_k)
    density = torch.empty(rows, 1, device=dev)
    _check(lib().mip360_rowdot_fm(_stream(), rows, x_k, _fm_ptr(x, x_col0), x_ld, _p(tm.w[D]), _p(tm.b[D]), 2, DENSITY_BIAS,
                                  _p(density), 1), 'mip360_rowdot_fm')
    saved['density'] = density
    rgb = None
    if not tm.cfg['disable_rgb']:
        bott = fm_buffer(rows, BOTTLENECK, dev)
        linear_fm(x, tm.w_fm[D + 1], tm.b[D + 1], 0, rows, BOTTLENECK, x_k, bott, None, lda=x_ld, ldw=x_k, a_col0=x_col0)
        view_in = torch.empty(rows, BOTTLENECK + DIR_LD, dtype=torch.bfloat16, device=dev)
        h = torch.empty(rows, VIEW_WIDTH, dtype=torch.bfloat16, device=dev)
        rgb = torch.empty(rows, 3, device=dev)
        if fused_view_ok(tm, D, rows):
            view_branch_fm(tm, D, bott, rows, n_samples, viewdirs, view_in, h, rgb)
        else:
            from_fm(bott, rows, BOTTLENECK, out=view_in)
            _check(lib().mip360_dir_encode(_stream(), n_rays, n_samples, _p(_f32(viewdirs)), _p(view_in), view_in.stride(0),
                                           BOTTLENECK, DIR_LD), 'mip360_dir_encode')
            linear(view_in, tm.w[D + 2], tm.b[D + 2], act=1, out_bf16=h, m=rows, n=VIEW_WIDTH, k=BOTTLENECK + DIR_LD)
            linear(h, tm.w[D + 3], tm.b[D + 3], act=3, act_param=RGB_PADDING, out_f32=rgb, m=rows, n=3, k=VIEW_WIDTH)
        saved.update(view_in=view_in, h=h, rgb=rgb)
    return density[:, 0], rgb, saved


def _grad_weight_fm(h, h_col0, ldh, dz, lddz, m, n_in, n_out, out, scratch, bias_out, rows_out=None):
    """d kernel = H^T dZ from fm operands (n_in, n_out multiples of 256) into out [rows_out <= n_in, n_out]"""
    tiles = (n_in // 256) * (n_out // 256)
    ksplit = int(max(1, min(256, (256 + tiles - 1) // tiles, m // 32)))
    if ksplit >= 8:
        ksplit = (ksplit // 8) * 8
    need = ksplit * (n_in * n_out + n_out)
    if scratch[0] is None or scratch[0].numel() < need:
        scratch[0] = torch.empty(need, device=h.device)
    buf = scratch[0]
    rows_out = n_in if rows_out is None else rows_out
    _check(lib().mip360_grad_weight_fm(_stream(), m, n_in, n_out, _fm_ptr(h, h_col0), ldh, _p(dz), lddz, ksplit, _p(buf), None, n_out,
                                       1.0, _p(bias_out)), 'mip360_grad_weight_fm')
    _check(lib().mip360_grad_weight_reduce(_stream(), rows_out, n_in, n_out, ksplit, _p(buf), _p(out), n_out, 1.0, _p(bias_out)),
           'mip360_grad_weight_reduce')


def _grad_weight_fm_multi(tm, items, rows, scratch):
    """The weight-gradient GEMMs of several trunk layers as ONE launch + one slab sum per layer (include/mip360_hip.h:
    mip360_grad_weight_fm_multi).  items: (layer t, x, x_col0, x_ld, x_k, dz) with dz [rows, W] fm.  With all their 256 x 256 tiles
    on the chip together the layers need ksplit ~ 256 / tiles row slices: 48 for the PropMLP's 5 tiles, 2 for the NerfMLP's 128."""
    W, G, n = tm.W, tm.grads, len(items)
    tiles = sum((it[4] // 256) * (W // 256) for it in items)
    # row slices per tile: enough to fill the 256 CUs once; more than 256 tiles fill them by themselves (one slice: the tiles
    # then take several rounds), and a slice is at least one 32-row chunk
    ks = 1 if tiles >= 256 else max(1, min(256 // tiles, rows // 32))
    while ks > 1 and (tiles * ks) % 8:                       # whole XCD rounds (the launch deals workgroups XCD-major)
        ks -= 1
    sizes = [ks * (it[4] * W + W) for it in items]
    if scratch[1] is None or scratch[1].numel() < sum(sizes):
        scratch[1] = torch.empty(sum(sizes), device=tm.device)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    slabs = [scratch[1][int(offs[j]):int(offs[j + 1])] for j in range(n)]
    ci = lambda v: (C.c_int * n)(*[int(x) for x in v])
    cp = lambda ps: (C.c_void_p * n)(*ps)
    _check(lib().mip360_grad_weight_fm_multi(
        _stream(), n, rows, ks, ci([it[4] for it in items]), ci([W] * n), cp([_fm_ptr(it[1], it[2]) for it in items]),
        ci([it[3] for it in items]), cp([_p(it[5]) for it in items]), ci([W] * n), cp([_p(t) for t in slabs])),
        'mip360_grad_weight_fm_multi')
    for j, it in enumerate(items):
        t = it[0]
        _check(lib().mip360_grad_weight_reduce(_stream(), tm.shapes[t][0], it[4], W, ks, _p(slabs[j]), _p(tm.kernel(t, G)), W, 1.0,
                                               _p(tm.bias(t, G))), 'mip360_grad_weight_reduce')


def mlp_backward_fm(tm, saved, rows, g_density, g_rgb, scratch):
    """mlp_backward on the fm tensors mlp_forward_train_fm saved."""
    W, D = tm.W, tm.depth
    dev = tm.device
    G = tm.grads
    bf = lambda c: torch.empty(rows, c, dtype=torch.bfloat16, device=dev)
    nerf = not tm.cfg['disable_rgb']
    trunk, t_col0, t_ld, trunk_k = saved['trunk']
    ks = min(256, rows // 32)                                        # row slices of the density head's column-dot kernel
    if scratch[0] is None or scratch[0].numel() < ks * (trunk_k + 1):
        scratch[0] = torch.empty(ks * (trunk_k + 1), device=dev)
    dz = fm_buffer(rows, W, dev)
    if not nerf:
        # the density column is the only head: d_raw as a plain bf16 vector, dZ of the last trunk layer as a masked outer product
        d_raw = torch.empty(rows, dtype=torch.bfloat16, device=dev)
        _check(lib().mip360_head_backward(_stream(), rows, _p(saved['density']), _p(_f32(g_density).reshape(-1)), None, None,
                                          RGB_PADDING, _p(d_raw), 1, 0, 1, None), 'mip360_head_backward')
        _check(lib().mip360_grad_weight_col_fm(_stream(), rows, trunk_k, _fm_ptr(trunk, t_col0), t_ld, _p(d_raw), 1, 0, ks, _p(scratch[0]),
                                               _p(tm.kernel(D, G)), 1.0, _p(tm.bias(D, G))), 'mip360_grad_weight_col_fm')
        if fused_prop_ok(tm.cfg, rows):
            # the whole dX chain in one launch, then the four weight-gradient GEMMs on what it wrote
            dzs = [fm_buffer(rows, W, dev) for _ in range(D - 1)] + [dz]
            prop_mlp_bwd_fm(rows, d_raw, tm.w[D], saved['masks'], [None] + [tm.wb_fm[i] for i in range(1, D)], [0] + [W] * (D - 1), dzs)
            if USE_MULTI_DW:
                _grad_weight_fm_multi(tm, [(i,) + tuple(saved['inputs'][i]) + (dzs[i],) for i in range(D)], rows, scratch)
                return
            for i in reversed(range(D)):
                x, x_col0, x_ld, x_k = saved['inputs'][i]
                _grad_weight_fm(x, x_col0, x_ld, dzs[i], W, rows, x_k, W, tm.kernel(i, G), scratch, tm.bias(i, G), rows_out=tm.shapes[i][0])
            return
        _check(lib().mip360_outer_masked_fm(_stream(), rows, W, _p(d_raw), _p(tm.w[D]), _p(saved['masks'][D - 1]), _p(dz), W),
               'mip360_outer_masked_fm')
    else:
        raw_col = BOTTLENECK
        d_pre = bf(32)
        d_hz = bf(VIEW_WIDTH)
        h, view_in = saved['h'], saved['view_in']
        if fused_view_ok(tm, D, rows) and (D + 3) in tm.wb_fm and tm.head_k == BOTTLENECK + 64:
            # head gradients, the two small dX layers and the conversion to fm in one launch (mip360_view_branch_bwd_fm)
            heads_fm = fm_buffer(rows, tm.head_k, dev)
            _check(lib().mip360_view_branch_bwd_fm(_stream(), rows, _p(saved['density']), _p(_f32(g_density).reshape(-1)), _p(saved['rgb']),
                                                   _p(_f32(g_rgb).reshape(-1, 3)), RGB_PADDING, _p(h), h.stride(0), _p(tm.wb_fm[D + 3]), 32,
                                                   _p(tm.wb_fm[D + 2]), VIEW_WIDTH, _p(d_pre), _p(d_hz), d_hz.stride(0), _p(heads_fm)),
                   'mip360_view_branch_bwd_fm')
            _grad_weight(h, d_pre, VIEW_WIDTH, 3, tm.kernel(D + 3, G), scratch, tm.bias(D + 3, G))
            _grad_weight(view_in, d_hz, BOTTLENECK + DIR_LD, VIEW_WIDTH, tm.kernel(D + 2, G), scratch, tm.bias(D + 2, G),
                         rows_out=BOTTLENECK + DIR_DIM)
        else:
            heads = bf(tm.head_k)                                    # row-major: written by the head / view-branch kernels
            _check(lib().mip360_head_backward(_stream(), rows, _p(saved['density']), _p(_f32(g_density).reshape(-1)),
                                              _p(saved.get('rgb')), _p(_f32(g_rgb).reshape(-1, 3)), RGB_PADDING,
                                              _p(heads), tm.head_k, raw_col, tm.head_k, _p(d_pre)), 'mip360_head_backward')
            _grad_weight(h, d_pre, VIEW_WIDTH, 3, tm.kernel(D + 3, G), scratch, tm.bias(D + 3, G))
            linear(d_pre, tm.wb[D + 3], None, act=4, out_bf16=d_hz, m=rows, n=VIEW_WIDTH, k=32, aux=h)
            _grad_weight(view_in, d_hz, BOTTLENECK + DIR_LD, VIEW_WIDTH, tm.kernel(D + 2, G), scratch, tm.bias(D + 2, G),
                         rows_out=BOTTLENECK + DIR_DIM)
            linear(d_hz, tm.wb[D + 2], None, act=0, out_bf16=heads, m=rows, n=BOTTLENECK, k=VIEW_WIDTH)     # -> heads[:, :256]
            heads_fm = to_fm(heads)                                  # [rows, head_k]
        _grad_weight_fm(trunk, t_col0, t_ld, heads_fm, tm.head_k, rows, trunk_k, BOTTLENECK, tm.kernel(D + 1, G), scratch,
                        tm.bias(D + 1, G))
        if scratch[0].numel() < ks * (trunk_k + 1):
            scratch[0] = torch.empty(ks * (trunk_k + 1), device=dev)
        # density head: d kernel[i] = sum_m trunk[m][i] d_raw[m]
        _check(lib().mip360_grad_weight_col_fm(_stream(), rows, trunk_k, _fm_ptr(trunk, t_col0), t_ld, _p(heads_fm), tm.head_k, raw_col, ks,
                                               _p(scratch[0]), _p(tm.kernel(D, G)), 1.0, _p(tm.bias(D, G))), 'mip360_grad_weight_col_fm')
        # dZ of the last trunk layer: both heads in one GEMM, masked by relu'(H_{D-1})
        linear_fm(heads_fm, tm.wb_fm['heads'], None, 2, rows, W, tm.head_k, dz, saved['masks'][D - 1])
    # The deferred form keeps all D dZ tensors alive until its one launch (D x rows x W x 2 B: 2.1 GB at 131 072 rows of the
    # 1024-wide NerfMLP) next to slab scratch for all layers: only while that is a small share of the free memory (ADVICE r05:
    # a larger batch or width must fall back to the per-layer launches instead of raising peak memory sharply)
    defer = USE_DEFER_DW and D <= 8 and all(saved['inputs'][i][3] % 256 == 0 for i in range(D))
    if defer:
        retained = (D - 1) * rows * W * 2
        defer = retained <= DEFER_DW_MAX_BYTES and retained <= torch.cuda.mem_get_info(dev)[0] // 4
    pending = []
    for i in reversed(range(D)):
        x, x_col0, x_ld, x_k = saved['inputs'][i]
        if defer:
            pending.append((i, x, x_col0, x_ld, x_k, dz))
        else:
            _grad_weight_fm(x, x_col0, x_ld, dz, W, rows, x_k, W, tm.kernel(i, G), scratch, tm.bias(i, G), rows_out=tm.shapes[i][0])
        if i > 0:
            nxt = fm_buffer(rows, W, dev)
            linear_fm(dz, tm.wb_fm[i], None, 2, rows, W, W, nxt, saved['masks'][i - 1])
            dz = nxt
    if defer:
        # every trunk layer's weight gradient in ONE launch behind the dX chain (all dZ_l kept: 8 x 268 MB): 128 tiles x 2 row
        # slices fill the chip, where a layer alone needs 16 slices -- an eighth of the split-K slab traffic, no launch boundaries
        _grad_weight_fm_multi(tm, pending, rows, scratch)


def mlp_backward(tm, saved, rows, g_density, g_rgb, scratch, side_stream=None):
    """Parameter gradients of one MLP into tm.grads (oracle: mip360_oracle.mlp_backward).  g_density [rows] f32,
    g_rgb [rows, 3] f32 or None."""
    if saved.get('fm'):
        return mlp_backward_fm(tm, saved, rows, g_density, g_rgb, scratch)
    tm.ensure_rm()
    W, D = tm.W, tm.depth
    dev = tm.device
    G = tm.grads
    bf = lambda c: torch.empty(rows, c, dtype=torch.bfloat16, device=dev)
    nerf = not tm.cfg['disable_rgb']
    # side_stream: the slab sums of the weight gradients run there (persistent per-layer slab buffers in tm.slabs); the
    # caller orders whatever consumes tm.grads after that stream.  Measured on the trainer: no gain (-3 %) -- the GEMMs
    # that follow occupy every CU's LDS and registers, so the short reduction kernels cannot run beside them.
    side = (lambda key: (side_stream, tm.slabs, key)) if side_stream is not None else (lambda key: None)
    trunk, trunk_k = saved['trunk']
    heads = bf(tm.head_k)                                            # [d bottleneck (256) | d raw | 0] or [d raw | 0]
    raw_col = BOTTLENECK if nerf else 0
    d_pre = bf(32) if nerf else None
    _check(lib().mip360_head_backward(_stream(), rows, _p(saved['density']), _p(_f32(g_density).reshape(-1)),
                                      _p(saved.get('rgb')), _p(_f32(g_rgb).reshape(-1, 3)) if nerf else None, RGB_PADDING,
                                      _p(heads), tm.head_k, raw_col, tm.head_k, _p(d_pre)), 'mip360_head_backward')
    if nerf:
        h, view_in = saved['h'], saved['view_in']
        _grad_weight(h, d_pre, VIEW_WIDTH, 3, tm.kernel(D + 3, G), scratch, tm.bias(D + 3, G), side=side(D + 3))
        d_hz = bf(VIEW_WIDTH)
        linear(d_pre, tm.wb[D + 3], None, act=4, out_bf16=d_hz, m=rows, n=VIEW_WIDTH, k=32, aux=h)
        _grad_weight(view_in, d_hz, BOTTLENECK + DIR_LD, VIEW_WIDTH, tm.kernel(D + 2, G), scratch, tm.bias(D + 2, G),
                     rows_out=BOTTLENECK + DIR_DIM, side=side(D + 2))
        linear(d_hz, tm.wb[D + 2], None, act=0, out_bf16=heads, m=rows, n=BOTTLENECK, k=VIEW_WIDTH)     # -> heads[:, :256]
        _grad_weight(trunk, heads, trunk_k, BOTTLENECK, tm.kernel(D + 1, G), scratch, tm.bias(D + 1, G), side=side(D + 1))
    d_raw = heads[:, raw_col:]
    _grad_weight(trunk, d_raw, trunk_k, 1, tm.kernel(D, G), scratch, tm.bias(D, G), side=side(D))
    # dZ of the last trunk layer: both heads in one GEMM, masked by relu'(H_{D-1})
    dz = bf(W)
    linear_masked(heads, tm.wb['heads'], dz, *saved['masks'][D - 1], m=rows, n=W, k=tm.head_k)
    for i in reversed(range(D)):
        x, x_k = saved['inputs'][i]
        # (padded input, 504 -> 512 encoding columns: the slab sum drops the padding rows)
        _grad_weight(x, dz, x_k, W, tm.kernel(i, G), scratch, tm.bias(i, G), rows_out=tm.shapes[i][0], side=side(i))
        if i > 0:
            nxt = bf(W)
            linear_masked(dz, tm.wb[i], nxt, *saved['masks'][i - 1], m=rows, n=W, k=W)
            dz = nxt


class Mip360Trainer(object):
    """One optimisation step of train_utils.create_train_step (:239-370) for configs/360.gin on the HIP kernels:
    model forward (3 levels), loss terms (charb data + depth on distance_mean + interlevel + distortion), backward
    through the compositing and the MLPs, per-MLP gradient-norm clipping, Adam with the log-decayed learning rate.
    Data parallel: the flat gradient buffers are averaged over ranks (jax.lax.pmean, :340-342) with one all-reduce
    per MLP (torch.distributed, backend nccl = RCCL)."""

    def __init__(self, prop_params, nerf_params, device, max_steps=250000, lambda_depth=0.1, depth_loss_type='mse',
                 world_size=1, grad_max_norm=0.001, adam_eps=1e-6, depth_sigma=0.01, depth_scale=1.0, **model_kw):
        self.device = torch.device(device)
        self.prop = TrainableMLP(prop_params, PROP_CFG, device)
        self.nerf = TrainableMLP(nerf_params, NERF_CFG, device)
        self.basis_t = torch.from_numpy(pos_basis_t()).to(self.device)
        self.cfg = dict(num_prop_samples=64, num_nerf_samples=32, num_levels=3, anneal_slope=10., dilation_multiplier=0.5,
                        dilation_bias=0.0025, bg_rgb=1.0)
        self.cfg.update(model_kw)
        if depth_loss_type not in DEPTH_TYPES:
            raise ValueError('depth_loss_type %r: mse / l1 (train_utils.py:108-119) or kl / urf (internal/depth_loss.py)' % depth_loss_type)
        self.max_steps, self.lambda_depth, self.depth_loss_type = max_steps, lambda_depth, depth_loss_type
        if depth_loss_type in ('kl', 'urf'):
            # upstream's `loss.sum(-2) * depth_mask` (internal/depth_loss.py:27,64) sums over RAYS and then broadcasts a
            # [n_samples] vector against the [n_rays] mask: it only type-checks for n_rays == n_samples on every level (or one
            # ray).  Say so when the trainer is built instead of on the first step.
            counts = {self.cfg['num_prop_samples'], self.cfg['num_nerf_samples']}
            if len(counts) != 1:
                raise Mip360Error(
                    "depth_loss_type %r with num_prop_samples=%d / num_nerf_samples=%d: upstream's loss.sum(-2) * depth_mask "
                    "(internal/depth_loss.py:27,64) broadcasts [n_samples] against [n_rays], so it needs the same sample count "
                    "on every level and batches of exactly that many rays (with configs/360.gin's 64 / 64 / 32 the reference "
                    "itself raises for every batch size but 1).  Use mse / l1, or equal sample counts."
                    % (depth_loss_type, self.cfg['num_prop_samples'], self.cfg['num_nerf_samples']))
            self._klurf_rays = counts.pop()
        else:
            self._klurf_rays = None
        self.depth_sigma = depth_sigma * depth_scale                      # train_utils.py:123
        self.world_size, self.grad_max_norm, self.adam_eps = world_size, grad_max_norm, adam_eps
        self.step = 0
        self.overlap_update = True
        self._update_stream = torch.cuda.Stream(device=self.device)
        self._prop_stream = torch.cuda.Stream(device=self.device)
        self.concurrent_prop_backward = True       # proposal levels' backward on its own stream (False: after the NeRF level's)
        # defer_update: train_step returns with the parameter updates still running on their streams -- the NerfMLP's under the
        # next step's proposal levels (which read the PropMLP only).  The next step orders itself behind them; anything else that
        # reads parameters, moments or gradients calls flush() first (as with NerfppTrainer).  Off: every step ends joined.
        self.defer_update = False
        self._pending = {}                         # 'prop' / 'nerf' -> event after that MLP's update
        self._keep_alive = None                    # the previous step's tensors the side streams may still be reading
        self.scratch = [None, None]
        self.scratch_prop = [None, None]
        self.partials = torch.empty(2, 256, device=self.device)
        self.clip = torch.empty(2, 2, device=self.device)

    def _join(self, which):
        ev = self._pending.pop(which, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def flush(self):
        """Order the caller's stream behind the parameter updates of the last step (defer_update)."""
        self._join('prop')
        self._join('nerf')
        self._keep_alive = None

    def forward(self, rays, train_frac, jitter01, training=True):
        c = self.cfg
        n = rays['origins'].shape[0]
        dev = self.device
        sdist = torch.tensor([[0., 1.]], device=dev).repeat(n, 1)
        weights = torch.ones(n, 1, device=dev)
        prod = 1
        levels = []
        for lvl in range(c['num_levels']):
            is_prop = lvl < c['num_levels'] - 1
            ns = c['num_prop_samples'] if is_prop else c['num_nerf_samples']
            dilation = c['dilation_bias'] + c['dilation_multiplier'] / prod
            prod *= ns
            s = c['anneal_slope']
            anneal = (s * train_frac) / ((s - 1) * train_frac + 1) if s > 0 else 1.
            sdist, tdist = resample(sdist, weights, dilation if lvl > 0 else 0.0, anneal, ns, rays['near'], rays['far'],
                                    None if jitter01 is None else jitter01[lvl])
            tm = self.prop if is_prop else self.nerf
            self._join('prop' if is_prop else 'nerf')            # (deferred update of the previous step)
            rows = n * ns
            if fm_ok(rows, tm.W) and tm.w_fm:
                enc_buf = fm_buffer(rows, tm.W + IPE_LD, dev)
                cast_encode_fm(tdist, rays['origins'], rays['directions'], rays['radii'], self.basis_t, enc_buf, tm.W, tm.W + IPE_LD)
                density, rgb, saved = mlp_forward_train_fm(tm, enc_buf, rows, rays['viewdirs'], n, ns)
            else:
                enc_buf = torch.empty(rows, tm.W + IPE_LD, dtype=torch.bfloat16, device=dev)
                cast_encode(tdist, rays['origins'], rays['directions'], rays['radii'], self.basis_t, out=enc_buf[:, tm.W:],
                            ld=tm.W + IPE_LD)
                density, rgb, saved = mlp_forward_train(tm, enc_buf, rows, rays['viewdirs'], n, ns)
            density = density.reshape(n, ns)
            rgb_s = rgb.reshape(n, ns, 3) if rgb is not None else None
            r = render_level(density, rgb_s, tdist, rays['directions'], True, c['bg_rgb'])
            weights = r['weights']
            levels.append(dict(sdist=sdist, tdist=tdist, density=density, rgb_s=rgb_s, saved=saved, rows=rows, ns=ns, **r))
        return levels

    def _apply_one(self, k, tm):
        """train_utils.py:340-364 on the flat gradient buffer of one MLP: mean over ranks (jax.lax.pmean; SUM all-reduce
        over RCCL, then / world_size), global-norm clipping (per MLP), Adam with the log-decayed learning rate, re-pack
        of the bf16 weight copies."""
        # optax.adam(learning_rate=lr_fn) evaluates the schedule at the PRE-increment count: lr_fn(0) on the first update
        # (the bias correction below uses the post-increment count, like optax.scale_by_adam)
        lr = learning_rate(self.step - 1, max_steps=self.max_steps)
        L = lib()
        if self.world_size > 1:
            import torch.distributed as dist
            dist.all_reduce(tm.grads)
            tm.grads.div_(self.world_size)
        n = tm.grads.numel()
        _check(L.mip360_sum_squares(_stream(), n, _p(tm.grads), _p(self.partials[k]), 256), 'mip360_sum_squares')
        _check(L.mip360_clip_multiplier(_stream(), 256, _p(self.partials[k]), float(self.grad_max_norm), _p(self.clip[k])),
               'mip360_clip_multiplier')
        _check(L.mip360_adam_step(_stream(), n, _p(tm.flat), _p(tm.grads), _p(tm.mu), _p(tm.nu), _p(self.clip[k]), self.step,
                                  lr, 0.9, 0.999, self.adam_eps), 'mip360_adam_step')
        tm.repack(lazy=True)

    def apply_gradients(self):
        """Both MLPs, on the caller's stream (NerfMLP first, like the pmean order of the step)."""
        self._apply_one(0, self.nerf)
        self._apply_one(1, self.prop)

    def train_step(self, rays, rgb_gt, depth_sup, jitter01=None):
        """rays / rgb_gt [n,3] / depth_sup [n] on the device.  Returns the scalars tensor of mip360_losses."""
        if self._klurf_rays is not None and rays['origins'].shape[0] not in (1, self._klurf_rays):
            raise Mip360Error("depth_loss_type %r: batches must hold exactly %d rays (= the per-level sample count) or 1 -- "
                              "upstream's loss.sum(-2) * depth_mask broadcast (internal/depth_loss.py:27,64); got %d"
                              % (self.depth_loss_type, self._klurf_rays, rays['origins'].shape[0]))
        self.step += 1
        self._join('prop')                         # the previous step's proposal backward has left its tensors
        self._keep_alive = None
        train_frac = float(np.clip((self.step - 1) / (self.max_steps - 1), 0, 1))          # train.py: step / max_steps
        if jitter01 is None:
            n = rays['origins'].shape[0]
            jitter01 = [torch.rand(n, device=self.device) for _ in range(self.cfg['num_levels'])]
        lv = self.forward(rays, train_frac, jitter01)
        props, nerf = lv[:-1], lv[-1]
        sc, g_rgb, g_dm, g_wn, g_wp, g_dmp = losses(
            nerf['rgb'], rgb_gt, nerf['distance_mean'], depth_sup, nerf['sdist'], nerf['weights'], [p['sdist'] for p in props],
            [p['weights'] for p in props], depth_loss_type=self.depth_loss_type, lambda_depth=self.lambda_depth,
            dm_prop=[p['distance_mean'] for p in props] if self.depth_loss_type else None, tdist_nerf=nerf['tdist'],
            tdist_prop=[p['tdist'] for p in props], directions=rays['directions'], depth_sigma=self.depth_sigma)
        def prop_backward():
            # proposal levels share the PropMLP: gradients add up
            acc = None
            for k, p in enumerate(props):
                gd, _ = render_level_backward(p['density'], None, p['tdist'], rays['directions'], g_wp[k], None,
                                              g_dmp[k] if g_dmp else None, True, self.cfg['bg_rgb'])
                mlp_backward(self.prop, p['saved'], p['rows'], gd, None, self.scratch_prop)
                acc = self.prop.grads.clone() if acc is None else acc.add_(self.prop.grads)
            self.prop.grads.copy_(acc)

        def nerf_backward():
            gd, grgbs = render_level_backward(nerf['density'], nerf['rgb_s'], nerf['tdist'], rays['directions'], g_wn, g_rgb, g_dm,
                                              True, self.cfg['bg_rgb'])
            mlp_backward(self.nerf, nerf['saved'], nerf['rows'], gd, grgbs, self.scratch)

        if self.concurrent_prop_backward:
            # The proposal levels' backward depends on the losses only (stop-gradient between the levels, models.py:210-214), not
            # on the NeRF level's: it runs on its own stream next to the NerfMLP's backward GEMMs and fills their tails.
            main = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record()
            self._prop_stream.wait_event(ev)
            with torch.cuda.stream(self._prop_stream):
                prop_backward()
                self._apply_one(1, self.prop)
                prop_done = torch.cuda.Event()
                prop_done.record()
            nerf_backward()
            if self.defer_update:
                ev = torch.cuda.Event()
                ev.record()
                self._update_stream.wait_event(ev)
                with torch.cuda.stream(self._update_stream):
                    self._apply_one(0, self.nerf)
                    nerf_done = torch.cuda.Event()
                    nerf_done.record()
                self._pending = {'prop': prop_done, 'nerf': nerf_done}
                self._keep_alive = (lv, g_wp, g_dmp, rays, jitter01)
                return sc
            self._apply_one(0, self.nerf)
            main.wait_event(prop_done)
            return sc
        nerf_backward()
        # the NerfMLP's update (all-reduce, norm, clip, Adam, 12 re-pack launches: 0.2 ms of short kernels) runs on a
        # side stream under the proposal levels' backward GEMMs; joined at the end of the step
        side = self._update_stream if self.overlap_update else None
        if side is not None:
            ev = torch.cuda.Event()
            ev.record()
            side.wait_event(ev)
            with torch.cuda.stream(side):
                self._apply_one(0, self.nerf)
                done = torch.cuda.Event()
                done.record()
        prop_backward()
        if side is not None:
            self._apply_one(1, self.prop)
            torch.cuda.current_stream().wait_event(done)
        else:
            self.apply_gradients()
        return sc


# ------------------------------------------------------------------------------------------------- measurement
def mlp_shapes(cfg):
    """(fan_in, fan_out) of every dense layer of MLP.__call__ (models.py:436-606) for a PROP_CFG / NERF_CFG dict."""
    W, D = cfg['net_width'], cfg['net_depth']
    out, dim = [], 504
    for i in range(D):
        out.append((dim, W))
        dim = W + (504 if (i % SKIP_LAYER == 0 and i > 0) else 0)
    out.append((dim, 1))
    if not cfg['disable_rgb']:
        out += [(dim, BOTTLENECK), (BOTTLENECK + DIR_DIM, VIEW_WIDTH), (VIEW_WIDTH, 3)]
    return out


def benchmark_step(device, n_rays=4096, steps=10, warmup=3, forward_only=False, seed=0, world_size=1):
    """configs/360.gin shape on synthetic rays (2 x 64 proposal samples through the 4 x 256 PropMLP, 32 samples through the
    8 x 1024 NerfMLP, he_uniform weights): whole training steps (forward, losses, backward, clip, Adam, re-pack) or
    forwards only.  Dense-layer FLOP per ray (forward) = 2 * (2 * 64 * prop MACs + 32 * nerf MACs); training = 3x.
    world_size > 1 (torch.distributed initialised by the caller): every rank runs n_rays rays of its own (seed = its
    rank), gradients are averaged over ranks (train_utils.py:340-342), the timed region is bracketed by barriers and the
    slowest rank's time counts; value = world_size * n_rays / that time."""
    import time
    rs_p = np.random.RandomState(0)                                   # identical parameters on every rank
    he = lambda shapes: [(rs_p.uniform(-np.sqrt(6.0 / i), np.sqrt(6.0 / i), (i, o)).astype(np.float32), np.zeros(o, np.float32))
                         for i, o in shapes]
    prop, nerf = he(mlp_shapes(PROP_CFG)), he(mlp_shapes(NERF_CFG))
    rs = np.random.RandomState(seed)
    n = n_rays
    d = rs.randn(n, 3).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    T = lambda x: torch.from_numpy(x).to(device)
    rays = dict(origins=T((rs.randn(n, 3) * 0.3).astype(np.float32)), directions=T(d), viewdirs=T(d.copy()),
                radii=T(np.full((n, 1), 2e-3, np.float32)), near=T(np.full((n, 1), 0.2, np.float32)),
                far=T(np.full((n, 1), 1e6, np.float32)))
    gt = T(rs.rand(n, 3).astype(np.float32))
    sup = T(np.where(rs.rand(n) < .5, rs.uniform(1, 6, n), 0).astype(np.float32))
    tr = Mip360Trainer(prop, nerf, device, world_size=world_size)
    macs = lambda sh: sum(i * o for i, o in sh)
    fwd_flop = 2.0 * (2 * 64 * macs(mlp_shapes(PROP_CFG)) + 32 * macs(mlp_shapes(NERF_CFG)))
    step = (lambda: tr.forward(rays, 0.5, None)) if forward_only else (lambda: tr.train_step(rays, gt, sup))

    def timed(defer):
        # defer = False: Mip360Trainer's default (every step ends joined) -- the headline `value`; True: the parameter
        # updates pipelined under the next step (tr.flush() belongs to the timed region), reported beside it and labelled
        tr.defer_update = defer
        for _ in range(warmup):
            step()
        tr.flush()
        torch.cuda.synchronize(device)
        if world_size > 1:
            import torch.distributed as dist
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        tr.flush()
        torch.cuda.synchronize(device)
        if world_size > 1:
            dist.barrier()
        dt_ = (time.perf_counter() - t0) / steps
        if world_size > 1:
            t = torch.tensor([dt_], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_
    dt = timed(False)
    dt_deferred = None if forward_only else timed(True)
    tr.defer_update = False
    n = n_rays * world_size
    flop = fwd_flop * (1 if forward_only else 3) * n
    return {'workload': 'MipNeRF-360 configs/360.gin shape, %d rays/step, %s, depth_loss_type=mse on distance_mean, synthetic rays'
                        % (n_rays, 'forward' if forward_only else 'train step'),
            'n_gpus': world_size, 'rays_per_gpu': n_rays, 'value': n / dt, 'unit': 'rays/s', 'ms_per_step': 1e3 * dt, 'steps': steps, 'dense_tflops': flop / dt / 1e12,
            'frac_of_bf16_mfma_peak': flop / dt / 2.5e15 / world_size, 'fwd_gflop_per_ray': fwd_flop / 1e9,
            'update_mode': 'joined at the end of every step (Mip360Trainer default)',
            'deferred_updates': None if dt_deferred is None else {
                'value': n / dt_deferred, 'ms_per_step': 1e3 * dt_deferred,
                'note': 'defer_update=True: updates pipelined under the next step, flush() inside the timed region (opt-in mode)'},
            'dtype': 'bf16 MFMA operands, f32 accumulate / f32 master weights'}
