"""fp16x2w / split-bf16 forward on TRAINED weights against the float32 oracle (numpy): gate ratios (max |err| / (1e-4 |ref| + atol))
per returned tensor after n optimisation steps on the config-1 scene."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from outdoor_nerf_depth_amd import ops, _lib as L
from oracle import nerfpp_oracle as O
import trajectory_common as TC
import test_gpu_round5 as R5
dev = torch.device('cuda:0')
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
N = lambda t: t.detach().cpu().numpy()
out = {}
for mode in ('mse', 'kl'):
    for n_steps in (0, 200, 1000, 3000):
        tr, smp = R5._train(mode, n_steps, L.PREC_SPLIT_BF16)
        b, uni = TC.step_batch(smp, 5001), TC.step_uniforms(5001)
        ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
        far, fg, bg = ops.sample_coarse(ray_o, ray_d, T(b['min_depth']), TC.CASCADE[0], T(uni['t_fg']), T(uni['t_bg']))
        for m in range(2):
            e2 = tr.engines[m]
            pm = R5_unflat = None
            vec = N(e2.params)
            # parameters as the oracle's dict
            shapes = {}
            for net, in_ch in (('fg_net', O.FG_IN), ('bg_net', O.BG_IN)):
                for k, sh in O.mlp_param_shapes(in_ch, O.DIR_IN).items():
                    shapes['%s.%s' % (net, k)] = sh
            lv, off = {}, 0
            for k in O.param_order():
                n = int(np.prod(shapes[k])); lv[k] = vec[off:off + n].reshape(shapes[k]); off += n
            ref = O.nerf_forward(lv, b['ray_o'], b['ray_d'], N(far), N(fg), N(bg))
            row = {}
            for name, prec in (('split_bf16', 2), ('fp16x2w', 3), ('bf16', 1)):
                e = ops.LevelEngine(e2.params.clone(), precision=prec)
                r = e.forward(ray_o, ray_d, far, fg, bg)
                rs = {k: R5.gate_ratio(N(r[k]), ref[k], k) for k in ops.RET_KEYS}
                w = max(rs, key=rs.get)
                row[name] = {'worst': round(rs[w], 3), 'worst_tensor': w, 'rgb': round(rs['rgb'], 3), 'depth': round(rs['depth'], 3), 'fg_weights': round(rs['fg_weights'], 3)}
            row['max_abs_param'] = float(np.abs(vec).max()); row['rms_param'] = float(np.sqrt(np.mean(vec ** 2)))
            out['%s/%d steps/level %d' % (mode, n_steps, m)] = row
            print(mode, n_steps, m, json.dumps(row), flush=True)
            if m == 0:
                r2 = e2.forward(ray_o, ray_d, far, fg, bg)
                fg, bg = ops.sample_fine_pair(fg, r2['fg_weights'], bg, r2['bg_weights'], TC.CASCADE[1], u_fg=T(uni['u_fg']), u_bg=T(uni['u_bg']))
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'p3_trained_check.json'), 'w'), indent=1)
