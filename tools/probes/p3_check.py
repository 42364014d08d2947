"""Forward precision 3 (fp16x2w) against the reference golden (gate ratios), against the split-bf16 forward (outputs, saved
tensors, gradients through the bf16 backward), and level-1 forward timings of the three precisions."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from outdoor_nerf_depth_amd import ops, _lib as L
from oracle import nerfpp_oracle as O
dev = torch.device('cuda:0')
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
N = lambda t: t.detach().cpu().numpy()
levels = O.init_params_like_reference(2)
flat = lambda lv: np.concatenate([lv[k].reshape(-1) for k in O.param_order()]).astype(np.float32)
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'forward.npz'))
def ratio(got, ref, k):
    atol = 2e-6 * (max(1.0, float(np.abs(ref).max())) if k in ('bg_depth', 'depth') else 1.0)
    return float(np.max(np.abs(got.astype(np.float64) - ref) / (1e-4 * np.abs(ref) + atol)))
for prec in (2, 3):
    for m, (fz, bz) in enumerate((('fg_z0', 'bg_z0'), ('fg_z1', 'bg_z1'))):
        eng = ops.LevelEngine(T(flat(levels[m])), precision=prec)
        for training in (False, True):
            ret = eng.forward(T(g['ray_o']), T(g['ray_d']), T(g['fg_far']), T(g[fz]), T(g[bz]), training=training)
            rs = {k: ratio(N(v), g['L%d.%s' % (m, k)], k) for k, v in ret.items()}
            w = max(rs, key=rs.get)
            print('prec', prec, 'L%d' % m, 'train' if training else 'infer', 'worst %.3f (%s) rgb %.3f depth %.3f fgw %.3f' % (rs[w], w, rs['rgb'], rs['depth'], rs['fg_weights']))
# saved tensors of P=3 vs P=2 hi plane
from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
b = SyntheticKitti().random_batch(1024, np.random.RandomState(0))
far, fg, bg = ops.sample_coarse(T(b['ray_o']), T(b['ray_d']), T(b['min_depth']), 192)
e2 = ops.LevelEngine(T(flat(levels[1])), precision=2); e3 = ops.LevelEngine(T(flat(levels[1])), precision=3)
r2 = e2.forward(T(b['ray_o']), T(b['ray_d']), far, fg, bg, training=True)
r3 = e3.forward(T(b['ray_o']), T(b['ray_d']), far, fg, bg, training=True)
for k in r2:
    print(k, ratio(N(r3[k]), N(r2[k]).astype(np.float64), k))
for net in (0, 1):
    for t in (4, 8, 10, 11):        # (X and H0 are not materialised at precision 3)
        a2, a3 = e2.saved_tensor(net, t), e3.saved_tensor(net, t)
        print('saved net', net, 'tensor', t, 'max abs diff', float((a2 - a3).abs().max()), 'max', float(a2.abs().max()), 'frac differing', float((a2 != a3).float().mean()))
# backward through P=3 forward
g_rgb = torch.rand_like(r3['rgb']) * 1e-3; g_d = torch.rand_like(r3['depth']) * 1e-3
gr3 = e3.backward(g_rgb, g_d, None).clone()
e12 = ops.LevelEngine(T(flat(levels[1])), precision=L.PREC_SPLIT_FWD)
e12.forward(T(b['ray_o']), T(b['ray_d']), far, fg, bg, training=True)
gr12 = e12.backward(g_rgb, g_d, None).clone()
print('grad rel L2 p3 vs split_fwd', float((gr3 - gr12).norm() / gr12.norm()))
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    c.record(); torch.cuda.synchronize(); return a.elapsed_time(c) / it
for prec in (1, 2, 3):
    e = ops.LevelEngine(T(flat(levels[1])), precision=prec)
    for tr in (False, True):
        print('prec', prec, 'train' if tr else 'infer', 'L1 fwd ms %.4f' % timeit(lambda: e.forward(T(b['ray_o']), T(b['ray_d']), far, fg, bg, training=tr)))
