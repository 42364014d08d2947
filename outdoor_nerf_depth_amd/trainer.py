"""Per-rank optimisation step of the NeRF++ path on the HIP library.

Mirrors the per-cascade-level loop of nerf-methods/nerfplusplus/ddp_train_nerf.py:432-498:
level 0 draws stratified depths, level 1 re-samples from level 0's (detached) weights; each level
has its own net, its own Adam state and its own gradient all-reduce (DDP averages gradients,
:323); the depth term is added when --use_depth (:486-493).
"""
import os

import numpy as np
import torch

from . import _lib as L
from . import ops
from .model import init_level_params

# (what the reference computes.  The MLP kernels execute 65 536 fewer per net in the forward and in the dX chain: the remap layer
# is folded into the colour head, csrc/nerfpp_common.h.)
ALGO_MACS = {            # dense MACs per sample (SURVEY.md 8a / BASELINE.md section 2)
    'fwd': (593408, 604160),         # fg, bg forward (= weight-gradient MACs)
    'dx': (557696, 557696),          # backward dX chain (no dX for L0, raw-input part of L5, dirs)
}


class NerfppTrainer(object):
    def __init__(self, device, precision=L.PREC_SPLIT_BF16, cascade_samples=(64, 128), lrate=5e-4,
                 use_depth=True, depth_loss_type='mse', lambda_depth=0.1, depth_sigma=0.01, depth_scale=1.0,
                 world_size=1, level_params=None, overlap_allreduce=True, optim_autoexpo=False, img_names=None,
                 lambda_autoexpo=1.0, seed=777, torch_rng=False, fuse_loss=False, comm=None):
        """seed: key of the in-kernel sampling RNG (the CLI passes (rank+1)*777 like ddp_train_nerf.py:406-408);
        torch_rng=True draws the four uniform tensors with torch.rand in the reference's call order instead
        (4 extra launches per step).  comm: optional dist_utils.RcclComm -- the gradient average then goes through the
        library's own RCCL entry point (nerfpp_allreduce_mean) instead of torch.distributed.all_reduce."""
        self.device = torch.device(device)
        self.comm = comm
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
        # [LEVEL_PARAMS gradient | bad-camera flag | pad]: the flag rides the gradient all-reduce (a SUM over ranks) and is
        # the device-side predicate of the Adam step, so that NO rank applies an update computed from rays that left the
        # unit sphere (the reference raises before the step, ddp_train_nerf.py:62-63; here the count is read later)
        self.grads = [torch.zeros(L.LEVEL_PARAMS + 4, device=self.device) for _ in self.engines]
        self.step_count = 0
        self.seed, self.torch_rng = int(seed), bool(torch_rng)
        # loss-head gradient inside the compositing backward (nerfpp_backward_args.fused_loss; not with auto-exposure).
        # Bit-identical to the two-call path; measured in one process (tools/ab_step.py, profiles/r03_ab_*): 2.717 ms per
        # step fused vs 2.704 separate -- the loss launch it takes off the critical path (~9 us) is paid back by the recount
        # in every compositing workgroup and the extra stream events, so the default stays the separate launch.
        self.fuse_loss = bool(fuse_loss)
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
        # The parameter update of a level (split-K slab sum -> [RCCL all-reduce] -> Adam -> re-pack of the bf16 weight
        # streams: four to six short launches that leave most CUs idle) runs on a side stream under the NEXT level's
        # sampling + forward; the main stream waits for it right before the level's streams are used again, i.e. at
        # the same level of the next step.  overlap_allreduce=False keeps everything on the caller's stream.
        self.update_stream = torch.cuda.Stream(device=self.device) if overlap_allreduce else None
        self.level_streams = [torch.cuda.Stream(device=self.device) for _ in range(len(self.cascade_samples))] if overlap_allreduce else None
        self.concurrent_backward = True     # level 0's backward on its own stream under level 1's forward (False: inline)
        self._pending = {}                # level -> event recorded after its update on the side stream
        # diagnostic (bench.py, N > 1): when a list, every _update_end appends a (before, after) timing-event pair around
        # the main stream's wait for the side-stream update -- the part of [slab sum, all-reduce, Adam, re-pack] that the
        # next level's sampling + forward did NOT hide
        self.wait_taps = None
        # diagnostic (bench.py, N > 1): when a list, every gradient all-reduce appends (level, begin, end) timing events recorded
        # around it on the stream it is issued on -- the duration of the collective as the GPU saw it (queueing behind the slab
        # sum included in `begin`'s position, not in the interval)
        self.comm_taps = None

    # -- parameter update ----------------------------------------------------------------------------
    def _update(self, m, step):
        """reduce the weight-gradient slabs, average over ranks (DDP semantics: grads were pre-scaled by
        1/world_size in the reduction, so a SUM all-reduce yields the mean, ddp_train_nerf.py:323), Adam, re-pack."""
        eng = self.engines[m]
        eng.reduce_grads()
        # the slab sum wrote float(bad_cameras) -- cumulative since the last check_cameras() -- behind the gradients
        # (nerfpp_backward_args.bad_count; until round 6 a torch copy_ launch per level and step)
        flag = self.grads[m][L.LEVEL_PARAMS:L.LEVEL_PARAMS + 1]
        if self.world_size > 1 or self.comm is not None:
            import torch.distributed as dist
            if self.autoexpo is not None:
                # [n_img, 3]: grads | used flag; a few hundred bytes.  EVERY rank issues this collective every
                # step: a rank whose image has no auto-exposure entry (ddp_model.py:186 falls back to the plain
                # rgb loss) contributes zeros, otherwise the ranks' collective sequences would diverge
                if self._ae_grad[m] is None:
                    self._ae_grad[m] = torch.zeros(len(self.autoexpo[m].names), 3, device=self.device)
                if self.comm is not None:
                    self.comm.allreduce_mean(self._ae_grad[m].view(-1), prescaled=True)
                else:
                    dist.all_reduce(self._ae_grad[m])
            tap = None
            if self.comm_taps is not None:
                tap = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                tap[0].record()
            if self.comm is not None:
                self.comm.allreduce_mean(self.grads[m], prescaled=True)      # grads carry 1 / world_size already (grad_scale)
            else:
                dist.all_reduce(self.grads[m])
            if tap is not None:
                tap[1].record()
                self.comm_taps.append((m, tap[0], tap[1]))
        ops.adam_step(eng.params, self.grads[m][:L.LEVEL_PARAMS], self.exp_avg[m], self.exp_avg_sq[m], step, lr=self.lrate,
                      skip=flag)
        eng.repack()
        if self._ae_grad[m] is not None:
            self.autoexpo[m].apply(self._ae_grad[m][:, :2], self._ae_grad[m][:, 2] > 0)
            self._ae_grad[m] = None

    def _update_begin(self, m):
        if self.update_stream is None:
            self._update(m, self.step_count)
            return
        ev = torch.cuda.Event()
        ev.record()
        self.update_stream.wait_event(ev)
        if self._ae_grad[m] is not None:
            self._ae_grad[m].record_stream(self.update_stream)
        with torch.cuda.stream(self.update_stream):
            self._update(m, self.step_count)
            done = torch.cuda.Event()
            done.record()
        self._pending[m] = done

    def _update_end(self, m):
        done = self._pending.pop(m, None)
        if done is not None:
            if self.wait_taps is not None:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                torch.cuda.current_stream().wait_event(done)
                b.record()
                self.wait_taps.append((a, b))
            else:
                torch.cuda.current_stream().wait_event(done)

    def flush(self):
        """Order the caller's stream after the parameter updates still running on the side stream.  Call before
        reading parameters or using the engines directly: checkpoints, rendering, the end of a timed region."""
        for m in list(self._pending):
            self._update_end(m)

    def check_cameras(self):
        """Raise the reference's exception (ddp_train_nerf.py:62-63) if any ray of any step since the last
        call left the unit sphere -- on ANY rank: the count rides the gradient all-reduce, so every rank raises
        together instead of one raising and the others blocking in the next collective.  From the first bad step
        until this call the Adam updates are dropped on the device (adam_step(skip=...)), i.e. the parameters are
        what they were when the reference would have raised.  Synchronises (a few bytes D2H): call it where the loop
        syncs anyway -- the log line, checkpoints, evaluation, the end of training."""
        bad = int(self.bad_cameras.item())
        if self.world_size > 1 and self.step_count > 0:
            self.flush()
            bad += int(sum(float(g[L.LEVEL_PARAMS].item()) for g in self.grads))
        if bad != 0:
            self.bad_cameras.zero_()
            for g in self.grads:
                g[L.LEVEL_PARAMS:].zero_()
            raise Exception(ops.CAMERA_ERROR)

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
        ae_idx = None
        if self.autoexpo is not None:
            name = batch.get('img_name')
            if name is None and 'frame' in batch:
                name = self.img_names[int(batch['frame'])]
            ae_idx = self.autoexpo[0].lookup(name)
        for m, eng in enumerate(self.engines):
            self._update_end(m)                   # this level's update from the previous step
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
            if ae is None and self.fuse_loss:
                # The loss head's gradient is formed inside the compositing backward (nerfpp_backward_args.fused_loss);
                # the loss launch itself only yields the logged scalars, so it runs on the side stream next to the
                # backward kernels instead of between forward and backward on the critical path.
                loss_done = None
                if self.update_stream is not None:
                    fwd_done = torch.cuda.Event()
                    fwd_done.record()
                    self.update_stream.wait_event(fwd_done)
                    with torch.cuda.stream(self.update_stream):
                        sc = ops.loss_and_grads(ret, batch['rgb'], depth_sup, self.loss_type, self.lambda_depth,
                                                self.kl_sigma, fg_z, far)[0]
                        loss_done = torch.cuda.Event()
                        loss_done.record()
                eng.backward(None, None, None, grad_scale=1.0 / self.world_size, out=self.grads[m][:L.LEVEL_PARAMS + 1], bad_count=self.bad_cameras,
                             events=ev['bwd'] if ev else None, defer_reduce=True,
                             fused_loss=dict(loss_type=self.loss_type, lambda_depth=self.lambda_depth, kl_sigma=self.kl_sigma,
                                             ret=ret, rgb_gt=batch['rgb'], depth_sup=depth_sup))
                if loss_done is not None:
                    torch.cuda.current_stream().wait_event(loss_done)     # long finished: orders readers of `sc` / frees of `ret`
                else:
                    sc = ops.loss_and_grads(ret, batch['rgb'], depth_sup, self.loss_type, self.lambda_depth,
                                            self.kl_sigma, fg_z, far)[0]
                scalars.append(sc)
                self._update_begin(m)
                continue
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
            if self.concurrent_backward and m + 1 < len(self.engines) and ae is None and events is None and self.update_stream is not None:
                # Level 1 needs level 0's FORWARD only (its weights, detached: ddp_train_nerf.py:452-457): level 0's backward, weight
                # gradients and update run on their own stream under level 1's sampling and forward (HBM-bound weight gradients
                # next to the forward kernels).  Measured: -1.1 % per step; the last level's backward under the NEXT step's
                # level 0 as well: no further gain (+-0.5 %); level 0's loss head + compositing backward moved to that stream too (25 us off
                # the path to level 1's forward): 2.331 vs 2.328 ms in one process (round 4) -- the step is bound by the sum of
                # the work, not by this chain.  _update_end(m) / flush() order later readers.  A step that carries
                # event taps runs inline, so that a tap times its kernel group alone on the GPU.
                stream = self.level_streams[m]
                fwd_done = torch.cuda.Event()
                fwd_done.record()
                stream.wait_event(fwd_done)
                for t in [g_rgb, g_depth, g_w, ray_o, ray_d, far, fg_z, bg_z, batch['rgb'], depth_sup] + list(ret.values()):
                    if torch.is_tensor(t):
                        t.record_stream(stream)               # (allocated on this stream, read on the level's)
                with torch.cuda.stream(stream):
                    eng.backward(g_rgb, g_depth, g_w, grad_scale=1.0 / self.world_size, out=self.grads[m][:L.LEVEL_PARAMS + 1], bad_count=self.bad_cameras,
                                 defer_reduce=True)
                    self._update_begin(m)
                scalars.append(sc)
                continue
            eng.backward(g_rgb, g_depth, g_w, grad_scale=1.0 / self.world_size, out=self.grads[m][:L.LEVEL_PARAMS + 1], bad_count=self.bad_cameras,
                         events=ev['bwd'] if ev else None, defer_reduce=True)
            scalars.append(sc)
            self._update_begin(m)
        return scalars


def batch_to_device(b, device):
    out = {}
    for k, v in b.items():
        if isinstance(v, np.ndarray):
            out[k] = torch.from_numpy(np.ascontiguousarray(v)).to(device)
    return out
