"""GPU, round 6: the unit-pipelined split-bf16 kernels (csrc/nerfpp_mlp_split.h) and ABI 8.

The split-bf16 forward (inference and training) and dX chain were rebuilt around units of 12 MFMAs with their LDS reads one unit
ahead, a lazily converted epilogue and -- in training -- no wave roles (2-slot ring of 16-fragment blocks, full vmcnt drain per
block).  Every existing 1e-4 / gradient / trajectory test runs on them unchanged (tests/test_gpu_parity.py, test_gpu_round2-5.py);
the build-against-build comparison with the stage-at-a-time kernels is tools/probes/split_dump.py (192 arrays bit-identical,
profiles/r06_split_stamps.md).  Here, what those do not cover:

* RACES.  The new bodies keep more in flight (LDS reads across barriers, a ring that is refilled one step after it was read, stores
  and weight DMA on one counter): repeated launches of one input must agree bit for bit -- outputs, every plane of every saved tensor,
  the dZ tensors, the gradients -- at the bench shape (many tiles per CU) and at a ragged one (tile tails).
* the saved tensors of the split-bf16 training forward against the float32 oracle at ragged sizes: hi + lo plane = the activation to
  ~2^-16 (the packed order, the lazy per-chunk stores and the tail zeroing all have to be right for that).
* ABI 8: nerfpp_backward_args.bad_count rides the slab sum; dZ7 / H0 are refused by nerfpp_workspace_tensor at single-plane precisions.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import nerfpp_oracle as O          # noqa: E402  (the checker)


def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def T(x, d=None):
    return torch.from_numpy(np.ascontiguousarray(x)).to(d or dev())


def N(t):
    return t.detach().cpu().numpy()


def _inputs(n_rays, S, seed):
    from outdoor_nerf_depth_amd import ops
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    b = SyntheticKitti().random_batch(n_rays, np.random.RandomState(seed))
    rs = np.random.RandomState(seed + 1)
    t_fg, t_bg = rs.rand(n_rays, S).astype(np.float32), rs.rand(n_rays, S).astype(np.float32)
    ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
    far, fg_z, bg_z = ops.sample_coarse(ray_o, ray_d, T(b['min_depth']), S, T(t_fg), T(t_bg))
    return b, ray_o, ray_d, far, fg_z, bg_z


def _params(seed=5):
    from outdoor_nerf_depth_amd.model import init_level_params
    p = init_level_params(1)[0]
    return p + 0.02 * torch.randn(p.shape, generator=torch.Generator().manual_seed(seed))      # biases off zero


def _run(eng, inp, precision_name, with_saved=True):
    _, ray_o, ray_d, far, fg_z, bg_z = inp
    out = {}
    ret = eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=False)
    out.update({'infer_' + k: N(v) for k, v in ret.items()})
    ret = eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=True)
    out.update({'train_' + k: N(v) for k, v in ret.items()})
    planes = (0, 1) if precision_name == 'split' else (0,)
    if with_saved:
        for net in (0, 1):
            for t in [0] + ([1] if precision_name == 'split' else []) + list(range(2, 9)) + [10, 11]:
                for pl in planes:
                    out['ws_n%d_t%d_p%d' % (net, t, pl)] = N(eng.saved_tensor(net, t, pl))
    g = torch.Generator().manual_seed(11)
    g_rgb = (torch.rand(ret['rgb'].shape, generator=g) * 1e-3).to(ray_o.device)
    g_depth = (torch.rand(ret['depth'].shape, generator=g) * 1e-3).to(ray_o.device)
    out['grads'] = N(eng.backward(g_rgb, g_depth, None))
    if with_saved and precision_name == 'split':
        for net in (0, 1):
            for t in list(range(12, 20)) + [21, 23]:
                for pl in (0, 1):
                    out['ws_n%d_t%d_p%d' % (net, t, pl)] = N(eng.saved_tensor(net, t, pl))
    return out


@pytest.mark.parametrize('shape', [(1024, 192), (37, 64)])
@pytest.mark.parametrize('precision_name', ['split', 'split_fwd'])
def test_split_kernels_repeat_bit_for_bit(shape, precision_name):
    from outdoor_nerf_depth_amd import ops, _lib as L
    d = dev()
    inp = _inputs(shape[0], shape[1], 3)
    prec = {'split': L.PREC_SPLIT_BF16, 'split_fwd': L.PREC_SPLIT_FWD}[precision_name]
    eng = ops.LevelEngine(_params().to(d), precision=prec)
    big = shape[0] * shape[1] > 100000
    first = _run(eng, inp, precision_name, with_saved=not big)
    # other work in between (a bf16 engine on the same device: different LDS / L2 / clock state for the repeats)
    other = ops.LevelEngine(_params(6).to(d), precision=L.PREC_BF16)
    for rep in range(3 if big else 2):
        other.forward(*inp[1:], training=True)
        again = _run(eng, inp, precision_name, with_saved=not big)
        for k in first:
            assert first[k].shape == again[k].shape and np.array_equal(first[k].view(np.uint8), again[k].view(np.uint8)), (k, rep)
    assert all(np.isfinite(v).all() for v in first.values())


def test_split_training_saves_match_oracle_ragged():
    """hi + lo plane of every saved activation of the split-bf16 training forward against the float32 oracle at a ragged size
    (37 rays x 64 samples: 18.5 tiles per net, one partly and one completely invalid wave in the last tile)."""
    from outdoor_nerf_depth_amd import ops, _lib as L
    d = dev()
    n, S = 37, 64
    level = O.init_params_like_reference(1)[0]
    flat = np.concatenate([level[k].reshape(-1) for k in O.param_order()]).astype(np.float32)
    b, ray_o, ray_d, far, fg_z, bg_z = _inputs(n, S, 9)
    eng = ops.LevelEngine(T(flat), precision=L.PREC_SPLIT_BF16)
    eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=True)
    cache = {}
    O.nerf_forward(level, b['ray_o'], b['ray_d'], N(far), N(fg_z), N(bg_z), cache=cache, bf16=False)
    tol = dict(rtol=3e-5, atol=3e-5)                     # = tests/test_gpu_round3.py (48 x 64: whole tiles in split-bf16)
    for net, key in ((0, 'fg'), (1, 'bg')):
        c = cache[key]
        # (the oracle flips the background network's rows along S, ddp_model.py:116-117; the kernels keep bg_z order)
        order = (lambda x: x.reshape(n, S, -1)[:, ::-1].reshape(n * S, -1)) if net == 1 else (lambda x: x)
        for l in range(8):
            hi_, lo = N(eng.saved_tensor(net, 1 + l)), N(eng.saved_tensor(net, 1 + l, plane=1))
            assert np.all(np.abs(lo) <= np.abs(hi_) * 2.0 ** -7 + 1e-30)                               # lo = the bf16 residual of hi
            np.testing.assert_allclose(hi_ + lo, order(np.maximum(c['pre'][l], 0)), err_msg='net %d H%d' % (net, l), **tol)
        g = N(eng.saved_tensor(net, 10)) + N(eng.saved_tensor(net, 10, plane=1))
        np.testing.assert_allclose(g, order(c['g']), err_msg='net %d G' % net, **tol)


def test_bad_count_rides_the_slab_sum():
    from outdoor_nerf_depth_amd import ops, _lib as L
    d = dev()
    inp = _inputs(64, 64, 4)
    for prec in (L.PREC_BF16, L.PREC_SPLIT_BF16):
        eng = ops.LevelEngine(_params().to(d), precision=prec)
        ret = eng.forward(*inp[1:], training=True)
        g_rgb, g_depth = torch.full_like(ret['rgb'], 1e-3), torch.full_like(ret['depth'], 1e-3)
        plain = N(eng.backward(g_rgb, g_depth, None))
        eng.forward(*inp[1:], training=True)
        out = torch.full((L.LEVEL_PARAMS + 1,), -1.0, device=d)
        bad = torch.tensor([7], dtype=torch.int32, device=d)
        got = eng.backward(g_rgb, g_depth, None, out=out, bad_count=bad)
        assert got.data_ptr() == out.data_ptr()
        np.testing.assert_array_equal(N(out[:L.LEVEL_PARAMS]), plain)
        assert float(out[L.LEVEL_PARAMS]) == 7.0
        # deferred form: the flag is written by nerfpp_level_reduce_grads, with the value the counter has THEN
        eng.forward(*inp[1:], training=True)
        out.fill_(-1.0)
        eng.backward(g_rgb, g_depth, None, out=out, bad_count=bad, defer_reduce=True)
        bad.fill_(3)
        eng.reduce_grads()
        np.testing.assert_array_equal(N(out[:L.LEVEL_PARAMS]), plain)
        assert float(out[L.LEVEL_PARAMS]) == 3.0
        with pytest.raises(L.NerfppError):
            eng.backward(g_rgb, g_depth, None, out=torch.empty(L.LEVEL_PARAMS, device=d), bad_count=bad)


def test_workspace_tensor_refuses_recomputed_tensors():
    from outdoor_nerf_depth_amd import _lib as L
    lib = L.lib()
    off, ld, pb = C.c_int64(), C.c_int32(), C.c_int64()
    q = lambda prec, t: lib.nerfpp_workspace_tensor(64, 64, prec, 0, t, C.byref(off), C.byref(ld), C.byref(pb))
    for prec in (L.PREC_BF16, L.PREC_FP16_FWD):
        assert q(prec, 1) != L.OK and q(prec, 19) != L.OK                 # H0 and dZ7: recomputed by their weight-gradient jobs
        assert q(prec, 18) == L.OK and q(prec, 2) == L.OK
    assert q(L.PREC_SPLIT_BF16, 1) == L.OK and q(L.PREC_SPLIT_BF16, 19) == L.OK
    # ... and dZ7 is not allocated there either: [dS | dG] (tensor 21) follows dZ6 (18) directly in a single-plane workspace, two
    # planes of dZ6 and two of dZ7 lie between them in a split-bf16 one
    def offset(prec, t):
        assert q(prec, t) == L.OK
        return off.value
    rows_padded = 64 * 64
    assert offset(L.PREC_BF16, 21) - offset(L.PREC_BF16, 18) == rows_padded * 512
    assert offset(L.PREC_SPLIT_BF16, 21) - offset(L.PREC_SPLIT_BF16, 18) == 4 * rows_padded * 512
