"""Train the config-1 scene for n steps (split-bf16) and dump both levels' parameters + one training batch with its depths, for
the CPU operand-format study (tools/operand_format_study.py --params)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from outdoor_nerf_depth_amd import ops, _lib as L
import trajectory_common as TC
import test_gpu_round5 as R5
dev = torch.device('cuda:0')
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
N = lambda t: t.detach().cpu().numpy()
for mode, n_steps in (('mse', 1000), ('kl', 1000)):
    tr, smp = R5._train(mode, n_steps, L.PREC_SPLIT_BF16)
    b, uni = TC.step_batch(smp, 5001), TC.step_uniforms(5001)
    ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
    far, fg0, bg0 = ops.sample_coarse(ray_o, ray_d, T(b['min_depth']), TC.CASCADE[0], T(uni['t_fg']), T(uni['t_bg']))
    r = tr.engines[0].forward(ray_o, ray_d, far, fg0, bg0)
    fg1, bg1 = ops.sample_fine_pair(fg0, r['fg_weights'], bg0, r['bg_weights'], TC.CASCADE[1], u_fg=T(uni['u_fg']), u_bg=T(uni['u_bg']))
    np.savez_compressed(os.path.join(ROOT, 'gpurun_out', 'trained_%s_%d.npz' % (mode, n_steps)),
                        p0=N(tr.engines[0].params).astype(np.float16 if False else np.float32), p1=N(tr.engines[1].params),
                        ray_o=b['ray_o'], ray_d=b['ray_d'], far=N(far), fg0=N(fg0), bg0=N(bg0), fg1=N(fg1), bg1=N(bg1))
