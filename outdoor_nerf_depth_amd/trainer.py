"""Per-rank optimisation step of the NeRF++ path on the HIP library.

Mirrors the per-cascade-level loop of nerf-methods/nerfplusplus/ddp_train_nerf.py:432-498:
level 0 draws stratified depths, level 1 re-samples from level 0's (detached) weights; each level
has its own net, its own Adam state and its own gradient all-reduce (DDP averages gradients,
:323); the depth term is added when --use_depth (:486-493).
"""
import numpy as np
import torch

from . import _lib as L
from . import ops
from .model import init_level_params

ALGO_MACS = {            # dense MACs per sample (SURVEY.md 8a / BASELINE.md section 2)
    'fwd': (593408, 604160),         # fg, bg forward (= weight-gradient MACs)
    'dx': (557696, 557696),          # backward dX chain (no dX for L0, raw-input part of L5, dirs)
}


class NerfppTrainer(object):
    def __init__(self, device, precision=L.PREC_SPLIT_BF16, cascade_samples=(64, 128), lrate=5e-4,
                 use_depth=True, depth_loss_type='mse', lambda_depth=0.1, depth_sigma=0.01, depth_scale=1.0,
                 world_size=1, level_params=None, overlap_allreduce=True, optim_autoexpo=False, img_names=None,
                 lambda_autoexpo=1.0, seed=777, torch_rng=False):
        """seed: key of the in-kernel sampling RNG (the CLI passes (rank+1)*777 like ddp_train_nerf.py:406-408);
        torch_rng=True draws the four uniform tensors with torch.rand in the reference's call order instead
        (4 extra launches per step)."""
        self.device = torch.device(device)
        self.precision = precision
        self.cascade_samples = tuple(cascade_samples)
        self.lrate = lrate
        self.loss_type = depth_loss_type if use_depth else 'rgbonly'
        if self.loss_type not in L.LOSS_TYPES:
            raise ValueError("depth_loss_type %r: only mse / l1 / kl exist in the reference "
                             "('los' and 'nll' are dead code there)" % depth_loss_type)
        self.lambda_depth = lambda_depth
        self.kl_sigma = depth_sigma * depth_scale           # ddp_train_nerf.py:489
        self.world_size = world_size
        if level_params is None:
            level_params = init_level_params(len(self.cascade_samples))   # manual_seed(777), :308
        self.engines = [ops.LevelEngine(p.to(self.device), precision) for p in level_params]
        self.exp_avg = [torch.zeros_like(e.params) for e in self.engines]
        self.exp_avg_sq = [torch.zeros_like(e.params) for e in self.engines]
        self.grads = [torch.empty_like(e.params) for e in self.engines]
        self.step_count = 0
        self.seed, self.torch_rng = int(seed), bool(torch_rng)
        self.rng_step = 0                 # counter of the in-kernel RNG (not reset by checkpoint reloads of step_count)
        # rays whose closest point to the origin lies outside the unit sphere, summed over all steps since
        # the last check_cameras() (the reference raises on the spot, ddp_train_nerf.py:62-63; here the
        # counter is read wherever the caller synchronises anyway)
        self.bad_cameras = torch.zeros(1, dtype=torch.int32, device=self.device)
        # per-image auto-exposure parameters, one set per level's net (ddp_model.py:161-192)
        self.autoexpo = None
        if optim_autoexpo:
            if not img_names:
                raise ValueError('optim_autoexpo needs the training image names (ddp_model.py:168)')
            from .autoexpo import AutoExposure
            self.img_names = list(img_names)
            self.autoexpo = [AutoExposure(img_names, self.device, lrate, lambda_autoexpo, world_size)
                             for _ in self.engines]
        self._ae_grad = [None] * len(self.engines)
        self.last_autoexpo = [None] * len(self.engines)
        self.comm_stream = torch.cuda.Stream(device=self.device) if (world_size > 1 and overlap_allreduce) else None
        self._pending = {}
        self._late = None                 # last level whose all-reduce + Adam are finished at its next use

    # -- distributed -------------------------------------------------------------------------------
    def _allreduce_begin(self, m):
        """Average gradients over ranks (DDP semantics).  grads were pre-scaled by 1/world_size in
        the backward kernel, so a SUM all-reduce yields the mean.  Runs on a side stream so level
        0's reduction overlaps level 1's sampling + forward."""
        if self.world_size <= 1:
            return
        import torch.distributed as dist
        if self.autoexpo is not None:
            # [n_img, 3]: grads | used flag; a few hundred bytes.  EVERY rank issues this collective every
            # step: a rank whose image has no auto-exposure entry (ddp_model.py:186 falls back to the plain
            # rgb loss) contributes zeros, otherwise the ranks' collective sequences would diverge
            if self._ae_grad[m] is None:
                self._ae_grad[m] = torch.zeros(len(self.autoexpo[m].names), 3, device=self.device)
            dist.all_reduce(self._ae_grad[m])
        if self.comm_stream is None:
            dist.all_reduce(self.grads[m])
            return
        ev = torch.cuda.Event()
        ev.record()
        self.comm_stream.wait_event(ev)
        with torch.cuda.stream(self.comm_stream):
            dist.all_reduce(self.grads[m])
            done = torch.cuda.Event()
            done.record()
        self._pending[m] = done

    def _allreduce_end(self, m):
        done = self._pending.pop(m, None)
        if done is not None:
            torch.cuda.current_stream().wait_event(done)

    def flush(self):
        """Finish the parameter update a multi-GPU step left in flight (the last level's all-reduce is
        overlapped with the NEXT step's level-0 work).  Call before reading parameters: checkpoints,
        rendering, end of a timed region."""
        if self._late is not None:
            self._apply(*self._late)
            self._late = None

    def check_cameras(self):
        """Raise the reference's exception (ddp_train_nerf.py:62-63) if any ray of any step since the last
        call left the unit sphere.  Synchronises (one 4-byte D2H): call it where the loop syncs anyway --
        the log line, checkpoints, evaluation, the end of training."""
        if int(self.bad_cameras.item()) != 0:
            self.bad_cameras.zero_()
            raise Exception(ops.CAMERA_ERROR)

    def _apply(self, m, step=None):
        self._allreduce_end(m)
        eng = self.engines[m]
        ops.adam_step(eng.params, self.grads[m], self.exp_avg[m], self.exp_avg_sq[m],
                      self.step_count if step is None else step, lr=self.lrate)
        eng.repack()
        if self._ae_grad[m] is not None:
            self.autoexpo[m].apply(self._ae_grad[m][:, :2], self._ae_grad[m][:, 2] > 0)
            self._ae_grad[m] = None

    # -- one optimisation step ----------------------------------------------------------------------
    def train_step(self, batch, uniforms=None, events=None):
        """batch: dict of device tensors ray_o, ray_d, rgb, min_depth, [depth_sup].
        uniforms: optional dict t_fg, t_bg [N,S0], u_fg, u_bg [N,S1] (replay); else torch.rand in the
        reference's call order.  events: optional per-level dict of torch.cuda.Event taps.
        Returns per-level scalar tensors [loss, rgb_loss, depth_loss, n_valid] (device, no sync)."""
        self.step_count += 1
        ray_o, ray_d = batch['ray_o'], batch['ray_d']
        n = ray_o.shape[0]
        S0, S1 = self.cascade_samples[0], self.cascade_samples[1] if len(self.cascade_samples) > 1 else 0
        dev = self.device
        u = uniforms or {}
        self.rng_step += 1
        rng = None if (uniforms is not None or self.torch_rng) else (self.seed, self.rng_step)
        if rng is not None:
            far, fg_z, bg_z = ops.sample_coarse(ray_o, ray_d, batch['min_depth'], S0, check=False, rng=rng,
                                                bad=self.bad_cameras)
        else:
            t_fg = u['t_fg'] if 't_fg' in u else torch.rand(n, S0, device=dev)
            t_bg = u['t_bg'] if 't_bg' in u else torch.rand(n, S0, device=dev)
            far, fg_z, bg_z = ops.sample_coarse(ray_o, ray_d, batch['min_depth'], S0, t_fg, t_bg, check=False,
                                                bad=self.bad_cameras)
        depth_sup = batch.get('depth_sup') if self.loss_type != 'rgbonly' else None
        scalars = []
        ret = None
        deferred = None
        ae_idx = None
        if self.autoexpo is not None:
            name = batch.get('img_name')
            if name is None and 'frame' in batch:
                name = self.img_names[int(batch['frame'])]
            ae_idx = self.autoexpo[0].lookup(name)
        for m, eng in enumerate(self.engines):
            if self._late is not None and self._late[0] == m:     # this level's update from the previous step
                self._apply(*self._late)
                self._late = None
            if m > 0 and rng is not None:
                fg_z, bg_z = ops.sample_fine_pair(fg_z, ret['fg_weights'], bg_z, ret['bg_weights'], S1, rng=rng)
            elif m > 0:
                u_fg = u['u_fg'] if 'u_fg' in u else torch.rand(n, S1, device=dev)
                u_bg = u['u_bg'] if 'u_bg' in u else torch.rand(n, S1, device=dev)
                fg_z, bg_z = ops.sample_fine_pair(fg_z, ret['fg_weights'], bg_z, ret['bg_weights'], S1,
                                                  u_fg=u_fg, u_bg=u_bg)
            ev = events[m] if events is not None else None
            ret = eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=True,
                              events=ev['fwd'] if ev else None)
            ae = self.autoexpo[m] if ae_idx is not None else None
            rgb_gt = ae.target(ae_idx, batch['rgb']) if ae is not None else batch['rgb']
            sc, g_rgb, g_depth, g_w = ops.loss_and_grads(ret, rgb_gt, depth_sup, self.loss_type,
                                                         self.lambda_depth, self.kl_sigma, fg_z, far)
            if ae is not None:                    # ddp_train_nerf.py:472-479
                g_ae = ae.finish(ae_idx, ret['rgb'], batch['rgb'], sc, g_rgb, self.lambda_depth)
                rows = torch.zeros(len(ae.names), 3, device=dev)
                rows[ae_idx, :2] = g_ae
                rows[ae_idx, 2] = 1.0
                self._ae_grad[m] = rows
                self.last_autoexpo[m] = ae.scale_shift(ae_idx)
            eng.backward(g_rgb, g_depth, g_w, grad_scale=1.0 / self.world_size, out=self.grads[m],
                         events=ev['bwd'] if ev else None)
            scalars.append(sc)
            if deferred is not None:              # level m-1's Adam after level m's work was queued
                self._apply(deferred)
                deferred = None
            self._allreduce_begin(m)
            if self.world_size > 1 and self.comm_stream is not None:
                if m + 1 < len(self.engines):
                    deferred = m                  # overlap this level's all-reduce with the next level
                else:
                    self._late = (m, self.step_count)   # ... and the last level's with the next step's level 0
            else:
                self._apply(m)
        return scalars


def batch_to_device(b, device):
    out = {}
    for k, v in b.items():
        if isinstance(v, np.ndarray):
            out[k] = torch.from_numpy(np.ascontiguousarray(v)).to(device)
    return out
