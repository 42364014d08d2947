"""ctypes binding of libnerfpp_hip.so (include/nerfpp_hip.h).

The HIP library IS the product path: there is no CPU / PyTorch fallback.  If the shared object is
missing this module raises at import of the symbol table (`lib()`), loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NERFPP_HIP_LIB') or os.path.join(_HERE, 'libnerfpp_hip.so')   # override: diagnostic builds

OK = 0
ABI_VERSION = 8
PREC_BF16, PREC_SPLIT_BF16 = 1, 2
# NERFPP_PREC_FP16X2W: a FORWARD precision (weights hi + lo in fp16, activations rounded to fp16 once, two MFMA passes): an
# intermediate one -- outputs within 1e-4 of float32 at initialisation, 3-4e-4 on trained weights (tests/test_gpu_round5.py).
# As a LevelEngine / trainer precision it means: that forward, single-pass bf16 backward.
PREC_FP16_FWD = 3
PREC_SPLIT_FWD = 12       # host-side combination: split-bf16 forward, bf16 backward (ops.LevelEngine)
LOSS_RGB_ONLY, LOSS_MSE, LOSS_L1, LOSS_KL = 0, 1, 2, 3
LOSS_TYPES = {'rgbonly': LOSS_RGB_ONLY, 'mse': LOSS_MSE, 'l1': LOSS_L1, 'kl': LOSS_KL}
FG_PARAMS, BG_PARAMS, LEVEL_PARAMS = 595844, 606596, 1202440
MAX_SAMPLES = 256

_fp = C.c_void_p      # device pointers travel as integers
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)


class ForwardArgs(C.Structure):
    _fields_ = [('n_rays', C.c_int32), ('n_samples', C.c_int32), ('precision', C.c_int32),
                ('training', C.c_int32)] + \
               [(k, _fp) for k in ('ray_o', 'ray_d', 'fg_far', 'fg_z', 'bg_z', 'packed', 'workspace',
                                   'rgb', 'depth', 'fg_weights', 'bg_weights', 'fg_dists', 'fg_rgb',
                                   'fg_depth', 'bg_rgb', 'bg_depth', 'bg_lambda', 'ev_mlp_begin',
                                   'ev_mlp_end')]


class BackwardArgs(C.Structure):
    _fields_ = [('n_rays', C.c_int32), ('n_samples', C.c_int32), ('precision', C.c_int32),
                ('workspace_precision', C.c_int32)] + \
               [(k, _fp) for k in ('ray_d', 'fg_far', 'fg_z', 'bg_z', 'packed', 'workspace', 'tables',
                                   'g_rgb', 'g_depth', 'g_fg_weights')] + \
               [('grad_scale', C.c_float), ('grads', _fp)] + \
               [(k, _fp) for k in ('ev_bwd_begin', 'ev_bwd_end', 'ev_dw_begin', 'ev_dw_end', 'params')] + \
               [('defer_reduce', C.c_int32), ('fused_loss', C.c_int32), ('loss_type', C.c_int32),
                ('lambda_depth', C.c_float), ('kl_sigma', C.c_float)] + \
               [(k, _fp) for k in ('rgb', 'depth', 'rgb_gt', 'depth_sup', 'bad_count')]


# every symbol include/nerfpp_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    'nerfpp_last_error': (C.c_char_p, []),
    'nerfpp_abi_version': (C.c_int, []),
    'nerfpp_intersect_sphere': (C.c_int, [_fp, C.c_int, _fp, _fp, _fp, _fp]),
    'nerfpp_sample_coarse': (C.c_int, [_fp, C.c_int, C.c_int] + [_fp] * 9),
    'nerfpp_sample_coarse_rng': (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, _fp, C.c_uint64, C.c_uint64, _fp, _fp, _fp, _fp]),
    'nerfpp_rng_uniform': (C.c_int, [_fp, C.c_uint64, C.c_uint64, C.c_int, C.c_int64, _fp]),
    'nerfpp_sample_pixels': (C.c_int, [_fp, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, _fp]),
    'nerfpp_perturb_samples': (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, _fp]),
    'nerfpp_sample_pdf': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp]),
    'nerfpp_sample_fine': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp]),
    'nerfpp_sample_fine_pair': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    'nerfpp_sample_fine_pair_rng': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, C.c_uint64,
                                              C.c_uint64]),
    'nerfpp_gather_rays': (C.c_int, [_fp, C.c_int, C.c_int] + [_fp] * 9),
    'nerfpp_table_sizes': (C.c_int, [C.c_int, _i64p, _i64p, _i64p, _i64p, _i64p]),
    'nerfpp_build_tables': (C.c_int, [C.c_int, _i32p, _i32p, _i32p, _i32p]),
    'nerfpp_level_tables_elems': (C.c_int64, []),
    'nerfpp_build_level_tables': (C.c_int, [_i32p]),
    'nerfpp_dw_plan': (C.c_int, [C.c_int64, C.c_int, _i32p, _i32p]),
    'nerfpp_packed_bytes': (C.c_int64, [C.c_int]),
    'nerfpp_pack_level': (C.c_int, [_fp, C.c_int, _fp, _fp, _fp]),
    'nerfpp_workspace_bytes': (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'nerfpp_workspace_tensor': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i64p, _i32p, _i64p]),
    'nerfpp_level_forward': (C.c_int, [_fp, C.POINTER(ForwardArgs)]),
    'nerfpp_loss': (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float] + [_fp] * 12),
    'nerfpp_level_backward': (C.c_int, [_fp, C.POINTER(BackwardArgs)]),
    'nerfpp_level_reduce_grads': (C.c_int, [_fp, C.POINTER(BackwardArgs)]),
    'nerfpp_adam_step': (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int64, C.c_int, C.c_double, C.c_double,
                                   C.c_double, C.c_double, _fp]),
    'nerfpp_comm_last_error': (C.c_char_p, []),
    'nerfpp_rccl_unique_id': (C.c_int, [C.c_char_p]),
    'nerfpp_rccl_comm_init': (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_int]),
    'nerfpp_rccl_comm_destroy': (C.c_int, [_fp]),
    'nerfpp_allreduce_mean': (C.c_int, [_fp, _fp, _fp, C.c_int64, C.c_int, C.c_int]),
}

_lib = None


class NerfppError(RuntimeError):
    pass


def lib():
    """The loaded library with typed prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NerfppError(
                'libnerfpp_hip.so not found at %s -- build it with `python -c "import '
                '__graft_entry__ as g; g.build()"` (hipcc --offload-arch=gfx950). There is no '
                'CPU fallback for the NeRF++ hot path.' % LIB_PATH)
        # PyTorch-ROCm ships its own libamdhip64; device pointers and streams only make sense inside ONE
        # HIP runtime, so torch's must be the copy already in the process when our library resolves
        # its libamdhip64 dependency (whichever is loaded first wins the SONAME).
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)         # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        if handle.nerfpp_abi_version() != ABI_VERSION:
            raise NerfppError('libnerfpp_hip.so ABI version mismatch')
        _lib = handle
    return _lib


def check(rc, what=''):
    if rc != OK:
        msg = lib().nerfpp_last_error().decode('utf-8', 'replace')
        raise NerfppError('%s failed (code %d): %s' % (what or 'nerfpp call', rc, msg))


def build_level_tables():
    """Host int32 numpy array with all index tables of one cascade level."""
    import numpy as np
    n = lib().nerfpp_level_tables_elems()
    out = np.empty(n, np.int32)
    check(lib().nerfpp_build_level_tables(out.ctypes.data_as(_i32p)), 'nerfpp_build_level_tables')
    return out


def build_net_tables(net):
    """(fwd_tbl, bias_tbl, bwd_tbl, unpack_tbl, slab_floats) numpy arrays of one net."""
    import numpy as np
    sizes = [C.c_int64() for _ in range(5)]
    check(lib().nerfpp_table_sizes(net, *[C.byref(s) for s in sizes]), 'nerfpp_table_sizes')
    fwd, bias, bwd, slab, npar = [s.value for s in sizes]
    arrs = [np.empty(n, np.int32) for n in (fwd, bias, bwd, npar)]
    check(lib().nerfpp_build_tables(net, *[a.ctypes.data_as(_i32p) for a in arrs]), 'nerfpp_build_tables')
    return arrs[0], arrs[1], arrs[2], arrs[3], slab
