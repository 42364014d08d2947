#!/usr/bin/env python
"""is there a cheaper forward than three bf16 MFMA passes that still holds north_star's 1e-4?

CPU emulation (numpy, no GPU): the oracle's NerfNet forward with every dense layer's operands rounded to a candidate MFMA
operand format, products accumulated in float32 (as v_mfma_f32_32x32x16_{bf16,f16} do), measured against the float32 reference
on (a) tests/golden/forward.npz (the imported reference's own outputs, 12 rays x 64 / 192) and (b) a 256-ray x 192-sample
batch against the oracle's float32 forward -- with the gate of tests/test_gpu_parity.py (rtol 1e-4 + atol 2e-6 elementwise on
every returned tensor; depth tensors' atol scaled by their magnitude) and the worst ratio error / allowed per tensor.
Passing needs ratio <= 1; the verdict asks for a 2x margin (ratio <= 0.5).

Formats (passes = MFMA issues per product; v_mfma_f32_32x32x16_f16 runs at the bf16 rate):
  bf16x1 / fp16x1          both operands rounded once
  bf16x3 / fp16x3          x = hi + lo on both sides, hi*hi + hi*lo + lo*hi   (bf16x3 = the shipped split-bf16 mode)
  fp16x2a / bf16x2a        activations split hi + lo, weights rounded once:  Wh*Ah + Wh*Al
  fp16x2w / bf16x2w        weights split, activations rounded once:          Wh*Ah + Wl*Ah
  fp16w_bf16x2 ...         mixed: see FORMATS
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nerfpp_oracle as O                      # noqa: E402

f32 = np.float32


def r_bf16(x):
    return O.round_bf16(x)


def r_fp16(x):
    return np.asarray(x, f32).astype(np.float16).astype(f32)


def split(x, r):
    hi = r(x)
    return hi, r((np.asarray(x, f32) - hi).astype(f32))


def mm(a, w):
    return (a @ w.T).astype(f32)


def fmt_x1(r):
    return lambda x, W: mm(r(x), r(W))


def fmt_x3(r):
    def f(x, W):
        ah, al = split(x, r)
        wh, wl = split(W, r)
        return (mm(ah, wh) + mm(al, wh) + mm(ah, wl)).astype(f32)
    return f


def fmt_x2a(r, rw=None):
    rw = rw or r

    def f(x, W):
        ah, al = split(x, r)
        wh = rw(W)
        return (mm(ah, wh) + mm(al, wh)).astype(f32)
    return f


def fmt_x2w(r, ra=None):
    ra = ra or r

    def f(x, W):
        wh, wl = split(W, r)
        ah = ra(x)
        return (mm(ah, wh) + mm(ah, wl)).astype(f32)
    return f


def fmt_by_layer(f_pe, f_rest):
    """f_pe for the two layers that see the encoded point (in_ch 63 / 84 and, with the skip, 319 / 340), f_rest elsewhere"""
    return lambda x, W: (f_pe if W.shape[1] in (63, 84, 319, 340) else f_rest)(x, W)


def fmt_trunk(f_trunk, f_heads):
    """f_trunk for the eight 256-wide trunk layers, f_heads for sigma / remap / colour layers"""
    return lambda x, W: (f_trunk if (W.shape[0] == 256 and W.shape[1] in (63, 84, 256, 319, 340)) else f_heads)(x, W)


FORMATS = [
    ('bf16x1', 1, fmt_x1(r_bf16)),
    ('fp16x1', 1, fmt_x1(r_fp16)),
    ('bf16x2a', 2, fmt_x2a(r_bf16)),
    ('bf16x2w', 2, fmt_x2w(r_bf16)),
    ('fp16x2a', 2, fmt_x2a(r_fp16)),
    ('fp16x2w', 2, fmt_x2w(r_fp16)),
    ('fp16: x2w on the encoded-point layers, x1 elsewhere', 1.2, fmt_by_layer(fmt_x2w(r_fp16), fmt_x1(r_fp16))),
    ('fp16: x2w trunk, x1 heads', 1.9, fmt_trunk(fmt_x2w(r_fp16), fmt_x1(r_fp16))),
    ('fp16: x1 trunk, x2w heads', 1.1, fmt_trunk(fmt_x1(r_fp16), fmt_x2w(r_fp16))),
    ('fp16: x3 on the encoded-point layers, x2w elsewhere', 2.2, fmt_by_layer(fmt_x3(r_fp16), fmt_x2w(r_fp16))),
    ('fp16: x2w trunk, x3 heads', 2.1, fmt_trunk(fmt_x2w(r_fp16), fmt_x3(r_fp16))),
    ('fp16: x3 trunk, x2w heads', 2.9, fmt_trunk(fmt_x3(r_fp16), fmt_x2w(r_fp16))),
    ('bf16x3', 3, fmt_x3(r_bf16)),
    ('fp16x3', 3, fmt_x3(r_fp16)),
]


def gate_ratio(got, ref, key):
    atol = 2e-6
    if key in ('bg_depth', 'depth'):
        atol *= max(1.0, float(np.abs(ref).max()))
    allowed = 1e-4 * np.abs(ref) + atol
    return float(np.max(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)) / allowed))


def main():
    levels = O.init_params_like_reference(2)          # manual_seed(777), as the parity tests do
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'forward.npz'))
    cases = []
    for m, (fz, bz) in enumerate((('fg_z0', 'bg_z0'), ('fg_z1', 'bg_z1'))):
        ref = {k[len('L%d.' % m):]: g[k] for k in g.files if k.startswith('L%d.' % m)}
        cases.append(('reference golden L%d (12 rays x %d)' % (m, g[fz].shape[1]), levels[m],
                      (g['ray_o'], g['ray_d'], g['fg_far'], g[fz], g[bz]), ref))
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    n, S = 256, 192
    b = SyntheticKitti().random_batch(n, np.random.RandomState(5))
    rs = np.random.RandomState(6)
    far = O.intersect_sphere(b['ray_o'], b['ray_d'])
    fg, bg = O.coarse_depths(b['min_depth'], far, S)
    fg = O.perturb_samples(fg, rs.rand(n, S).astype(f32))
    bg = O.perturb_samples(bg, rs.rand(n, S).astype(f32))
    ref = O.nerf_forward(levels[1], b['ray_o'], b['ray_d'], far, fg, bg)
    cases.append(('oracle float32, %d rays x %d' % (n, S), levels[1], (b['ray_o'], b['ray_d'], far, fg, bg), dict(ref)))
    if len(sys.argv) > 2 and sys.argv[1] == '--trained':
        # weights after N optimisation steps of the float32 CPU restatement (oracle/nerfpp_torch_cpu.py) on the config-1 scene,
        # gt + mse: sharper densities and larger weights than at initialisation
        import torch
        torch.set_num_threads(2)
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import trajectory_common as TC
        from oracle import nerfpp_torch_cpu as TCPU
        n_steps = int(sys.argv[2])
        smp = TC.sampler('mse')
        tc = TCPU.TorchCpuTrainer(O.init_params_like_reference(2), cascade_samples=TC.CASCADE, use_depth=True, depth_loss_type='mse',
                                  lambda_depth=TC.LAMBDA_DEPTH, depth_sigma_scaled=TC.DEPTH_SIGMA * float(smp.get_depth_scale() or 1.0))
        for step in range(1, n_steps + 1):
            logs = tc.train_step(TC.step_batch(smp, step), TC.step_uniforms(step))
            if step % 50 == 0:
                print('trained', step, logs[1]['loss'], logs[1]['rgb_loss'], flush=True)
        bt = TC.step_batch(smp, n_steps + 1)
        logs = tc.train_step(bt, TC.step_uniforms(n_steps + 1))       # its level-1 depths; the parameters move once more, so:
        far = O.intersect_sphere(bt['ray_o'], bt['ray_d'])
        cases = []
        for m in range(2):
            pm = {k: np.asarray(v, f32) for k, v in tc.params(m).items()}
            inp = (bt['ray_o'], bt['ray_d'], far, logs[m]['fg_z'], logs[m]['bg_z'])
            cases.append(('after %d steps, level %d (256 rays x %d)' % (n_steps + 1, m, logs[m]['fg_z'].shape[1]), pm, inp,
                          dict(O.nerf_forward(pm, *inp))))
    if len(sys.argv) > 2 and sys.argv[1] == '--params':
        # parameters trained on the GPU (tools/probes/dump_trained.py: 1000 split-bf16 steps on the config-1 scene) with one
        # training batch and its depths
        cases = []
        for path in sys.argv[2:]:
            d = np.load(path)
            shapes = {}
            for net, in_ch in (('fg_net', O.FG_IN), ('bg_net', O.BG_IN)):
                for k, sh in O.mlp_param_shapes(in_ch, O.DIR_IN).items():
                    shapes['%s.%s' % (net, k)] = sh
            for m in range(2):
                vec, lv, off = d['p%d' % m], {}, 0
                for k in O.param_order():
                    n = int(np.prod(shapes[k])); lv[k] = vec[off:off + n].reshape(shapes[k]); off += n
                inp = (d['ray_o'], d['ray_d'], d['far'], d['fg%d' % m], d['bg%d' % m])
                cases.append(('%s level %d' % (os.path.basename(path)[:-4], m), lv, inp, dict(O.nerf_forward(lv, *inp))))
    report = {'gate': 'rtol 1e-4 + atol 2e-6 per element (tests/test_gpu_parity.py RET_TOL[2]); ratio = max |err| / allowed', 'cases': {}}
    for name, lvl, inp, ref in cases:
        rows = {}
        for fname, passes, f in FORMATS:
            ret = O.nerf_forward(lvl, *inp, bf16=f)
            ratios = {k: gate_ratio(ret[k], ref[k], k) for k in ref}
            worst = max(ratios, key=ratios.get)
            rows[fname] = {'passes': passes, 'worst_ratio': round(ratios[worst], 3), 'worst_tensor': worst,
                           'rgb_ratio': round(ratios['rgb'], 3), 'depth_ratio': round(ratios['depth'], 3),
                           'fg_weights_ratio': round(ratios['fg_weights'], 3), 'passes_gate': bool(ratios[worst] <= 1.0),
                           'passes_with_2x_margin': bool(ratios[worst] <= 0.5)}
            print('%-40s %-52s passes=%.1f worst %8.3f (%s) rgb %7.3f depth %7.3f fg_w %8.3f' % (
                name, fname, passes, ratios[worst], worst, ratios['rgb'], ratios['depth'], ratios['fg_weights']), flush=True)
        report['cases'][name] = rows
    out = os.path.join(ROOT, 'profiles', 'r05_operand_format_study%s.json' % ('_trained' if len(sys.argv) > 2 and sys.argv[1] == '--trained' else '_gpu_trained' if len(sys.argv) > 2 else ''))
    with open(out, 'w') as fjs:
        json.dump(report, fjs, indent=1)
    print('written', out)


if __name__ == '__main__':
    main()
