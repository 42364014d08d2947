#!/usr/bin/env python
"""Drop-in entry point for nerf-methods/nerfplusplus/ddp_train_nerf.py on MI355X.

    python -m outdoor_nerf_depth_amd.ddp_train_nerf --config configs/kitti.txt \
        --use_depth --depth_sup_type gt --depth_loss_type mse --lambda_depth 0.1 --trainskip 4 ...

Same flags and defaults as the reference's `config_parser` (:657-727), same outputs under
{basedir}/{expname}/ (args.txt, config.txt, model_{step:06d}.pth with the reference's state-dict
keys, render_test_{step:06d}/ + psnr_/rmse_/absrel_*.txt), same log scalars
(`level_m/{rgb_loss,pnsr,loss_depth}`, `iter_time`, `resolution`).  The per-step work runs on the
HIP library (trainer.py); one process per GPU, RCCL all-reduce of the flat gradients.

Extra flags: --sample_every (the README's name for --trainskip), --precision {split,fp16_fwd,split_fwd,bf16},
--synthetic (KITTI-shaped procedural scene instead of --datadir), --N_rand_override.
"""
import argparse
import logging
import os
import sys
import time
from collections import OrderedDict

import numpy as np

logger = logging.getLogger(__package__ or 'outdoor_nerf_depth_amd')
TINY_NUMBER = 1e-6
mse2psnr = lambda x: -10. * np.log(x + TINY_NUMBER) / np.log(10.)           # utils.py:31
to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)


def setup_logger():
    lg = logging.getLogger(__package__ or 'outdoor_nerf_depth_amd')
    lg.setLevel(logging.INFO)
    if not lg.handlers:
        ch = logging.StreamHandler()
        ch.setFormatter(logging.Formatter('%(asctime)s [%(levelname)s] %(name)s: %(message)s'))
        lg.addHandler(ch)


class ConfigFileParser(argparse.ArgumentParser):
    """argparse + the config file syntax of configargparse's default parser (--config; the command line
    wins): one `key = value`, `key: value` or `key value` per line, `#` / `;` comments, a bare `key` or
    `key = true` for store_true flags.  Unknown keys are an error, like upstream."""

    @staticmethod
    def _split(line):
        for sep in ('=', ':'):
            if sep in line:
                k, v = line.split(sep, 1)
                if ' ' not in k.strip():
                    return k.strip(), v.strip()
        parts = line.split(None, 1)
        return parts[0], (parts[1].strip() if len(parts) > 1 else None)

    def parse_args(self, args=None, namespace=None):
        args = list(sys.argv[1:] if args is None else args)
        cfg = None
        for i, a in enumerate(args):
            if a == '--config' and i + 1 < len(args):
                cfg = args[i + 1]
            elif a.startswith('--config='):
                cfg = a.split('=', 1)[1]
        file_args = []
        if cfg and cfg != 'None':
            known = {}
            for a in self._actions:
                for opt in a.option_strings:
                    known[opt.lstrip('-')] = a
            for ln, raw in enumerate(open(cfg), 1):
                line = raw.strip()
                if not line or line[0] in '#;' or line.startswith('---'):
                    continue
                line = line.split(' #', 1)[0].strip()
                k, v = self._split(line)
                k = k.lstrip('-')
                if k == 'config':
                    continue
                if k not in known:
                    self.error('%s:%d: unrecognized config key %r' % (cfg, ln, k))
                if v is not None and len(v) >= 2 and v[0] == v[-1] and v[0] in '"\'':
                    v = v[1:-1]
                act = known[k]
                if isinstance(act, argparse._StoreTrueAction):
                    if v is None or v.lower() in ('true', '1', 'yes', 'on'):
                        file_args.append('--' + k)
                    elif v.lower() not in ('false', '0', 'no', 'off'):
                        self.error('%s:%d: flag %r takes true/false, got %r' % (cfg, ln, k, v))
                elif v is None:
                    self.error('%s:%d: key %r needs a value' % (cfg, ln, k))
                elif v != 'None':
                    file_args += ['--' + k, v]
        return super().parse_args(file_args + args, namespace)


def config_parser():
    """Flag for flag the reference's parser (ddp_train_nerf.py:657-727)."""
    p = ConfigFileParser()
    p.add_argument('--config', type=str, default=None, help='config file path')
    p.add_argument('--expname', type=str, help='experiment name')
    p.add_argument('--basedir', type=str, default='./logs/', help='where to store ckpts and logs')
    p.add_argument('--datadir', type=str, default=None, help='input data directory')
    p.add_argument('--scene', type=str, default=None, help='scene name')
    p.add_argument('--testskip', type=int, default=8)
    p.add_argument('--trainskip', '--sample_every', dest='trainskip', type=int, default=1,
                   help='will load 1/N images from train sets for sparse inputs')
    p.add_argument('--netdepth', type=int, default=8)
    p.add_argument('--netwidth', type=int, default=256)
    p.add_argument('--use_viewdirs', action='store_true')
    p.add_argument('--no_reload', action='store_true')
    p.add_argument('--ckpt_path', type=str, default=None)
    p.add_argument('--N_rand', type=int, default=32 * 32 * 2)
    p.add_argument('--chunk_size', type=int, default=1024 * 8)
    p.add_argument('--N_iters', type=int, default=250001)
    p.add_argument('--render_splits', type=str, default='test')
    p.add_argument('--cascade_level', type=int, default=2)
    p.add_argument('--cascade_samples', type=str, default='64,64')
    p.add_argument('--world_size', type=int, default=-1)
    p.add_argument('--optim_autoexpo', action='store_true')
    p.add_argument('--lambda_autoexpo', type=float, default=1.)
    p.add_argument('--lrate', type=float, default=5e-4)
    p.add_argument('--lrate_decay_factor', type=float, default=0.1)      # parsed, never read (as upstream)
    p.add_argument('--lrate_decay_steps', type=int, default=5000)        # parsed, never read
    p.add_argument('--det', action='store_true')                         # parsed, never read
    p.add_argument('--max_freq_log2', type=int, default=10)
    p.add_argument('--max_freq_log2_viewdirs', type=int, default=4)
    p.add_argument('--load_min_depth', action='store_true')
    p.add_argument('--i_print', type=int, default=100)
    p.add_argument('--i_img', type=int, default=500)                     # parsed, never read
    p.add_argument('--i_weights', type=int, default=10000)
    p.add_argument('--use_depth', action='store_true')
    p.add_argument('--lambda_depth', type=float, default=1.0)
    p.add_argument('--depth_loss_type', choices=['mse', 'kl', 'los', 'l1', 'nll'], default='mse')
    p.add_argument('--depth_sup_type', type=str, default='gt')
    p.add_argument('--depth_sigma', type=float, default=0.01)
    p.add_argument('--port', type=int, default=12345)
    # --- additions of this implementation
    p.add_argument('--precision', choices=['bf16', 'split', 'split_fwd', 'fp16_fwd'], default='split',
                   help='MLP arithmetic.  split (default): split-bf16 (3 MFMA passes) in forward, backward and weight gradients -- '
                        'rendered RGB / depth / loss within 1e-4 of the float32 reference, gradients at float32 grade.  split_fwd: '
                        'that forward with the single-pass bf16 backward (1.6x the throughput).  fp16_fwd: forward with fp16 '
                        'operands, weights hi + lo (2 passes: 1e-4 at initialisation, 3-4e-4 on trained weights), bf16 backward '
                        '(2.3x).  bf16: single-pass bf16 MFMA everywhere (fastest; outputs at bf16 grade, 1e-2).  With a bf16 backward a '
                        "training run leaves the reference's trajectory like the reference's own float64 run leaves its float32 "
                        'run: rgb-only within 0.01 dB at step 200, with a depth term +-0.2 ... 0.9 dB at step 200 in either '
                        'direction and no resolvable gap (|median| < 0.1 dB over seeds) at 1000 steps: DESIGN.md section 5')
    p.add_argument('--grad_comm', choices=['torch', 'rccl_abi'], default='torch',
                   help="gradient all-reduce: torch.distributed (backend nccl = RCCL), or the library's own RCCL entry point "
                        '(nerfpp_allreduce_mean; the communicator id travels over the torch process group)')
    p.add_argument('--rccl_channels', type=int, default=-1,
                   help='cap on the RCCL channels of the gradient all-reduce (NCCL_MAX_NCHANNELS; every channel holds a CU that an '
                        'MLP tile cannot use).  -1 (default): 4 when all ranks run on one node, RCCL\'s own choice across nodes; '
                        "0: never touch RCCL's environment; N > 0: N, also across nodes.  NCCL_* variables set by the caller always win; "
                        'the values in effect are logged on every rank')
    p.add_argument('--synthetic', action='store_true', help='KITTI-shaped procedural scene, no datadir')
    p.add_argument('--synthetic_hw', type=str, default=None, help="'H,W' of the synthetic frames (default 375,1242)")
    p.add_argument('--synthetic_frames', type=int, default=295)
    p.add_argument('--N_rand_override', type=int, default=None,
                   help='the reference overrides N_rand to 1024 on >14 GB GPUs; this overrides that')
    p.add_argument('--i_test', type=int, default=50000, help='test-set render period (hard-coded 50000 upstream)')
    p.add_argument('--device_sampling', action='store_true',
                   help='(default) keep the training frames in HBM and draw ray batches on the device: no host '
                        'np.random.choice over H*W (4 ms per step at 375x1242), no per-step H2D copies')
    p.add_argument('--host_sampling', action='store_true',
                   help="the reference's host-side RaySamplerSingleImage.random_sample per step (numpy RNG stream "
                        'of the reference; bounds the step at ~4 ms)')
    return p


def validate_args(args):
    if args.netdepth != 8 or args.netwidth != 256:
        raise SystemExit('only netdepth=8 / netwidth=256 (the reference configs) are implemented in HIP')
    if args.max_freq_log2 != 10 or args.max_freq_log2_viewdirs != 4:
        raise SystemExit('only max_freq_log2=10 / max_freq_log2_viewdirs=4 are implemented in HIP')
    if args.use_depth and args.depth_loss_type in ('los', 'nll'):
        raise SystemExit("depth_loss_type '%s' is dead code in the reference (depth_loss.py:46-76)" %
                         args.depth_loss_type)
    if args.cascade_level != 2:
        raise SystemExit('cascade_level must be 2')


# ------------------------------------------------------------------------------------------------
def render_single_image(rank, world_size, trainer, ray_sampler, chunk_size, keep_dists=True, mlp_events=None):
    """ddp_train_nerf.py:133-249: deterministic sampling, no perturbation, chunked, sharded over ranks
    (ragged last shard instead of raising when H*W % world_size != 0).  Returns, per level, every key of
    `ret` except the two weight tensors, in the reference's order (:210-218) -- including `fg_dists`
    [H, W, S] (0.36 GB per 375x1242 frame at S = 192; keep_dists=False drops it for timing runs).
    mlp_events: optional list; (level, rows, begin, end) HIP-event taps around the MLP kernels (fg + bg) of every
    chunk are appended to it (bench.py's inference leg reads them after a synchronise)."""
    import torch
    from . import ops
    from .dist_utils import shard_sizes, gather_ragged
    trainer.flush()
    b = ray_sampler.get_all()
    n = b['ray_d'].shape[0]
    sizes = shard_sizes(n, world_size)
    lo = sum(sizes[:rank])
    dev = trainer.device
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a[lo:lo + sizes[rank]])).to(dev)
    ray_o, ray_d, min_depth = T(b['ray_o']), T(b['ray_d']), T(b['min_depth'])
    S0, S1 = trainer.cascade_samples
    keys = ('rgb', 'fg_dists', 'fg_rgb', 'fg_depth', 'bg_rgb', 'bg_depth', 'bg_lambda', 'depth')
    if not keep_dists:
        keys = tuple(k for k in keys if k != 'fg_dists')
    out = [OrderedDict((k, []) for k in keys) for _ in range(2)]
    for s in range(0, sizes[rank], chunk_size):
        o, d, md = ray_o[s:s + chunk_size], ray_d[s:s + chunk_size], min_depth[s:s + chunk_size]
        far, fg_z, bg_z = ops.sample_coarse(o, d, md, S0, perturb=False, check=(s == 0))
        ret = None
        for m, eng in enumerate(trainer.engines):
            if m > 0:
                fg_z, bg_z = ops.sample_fine_pair(fg_z, ret['fg_weights'], bg_z, ret['bg_weights'], S1, det=True)
            ev = None
            if mlp_events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record(); ev[1].record()            # materialise the hipEvent_t handles
                mlp_events.append((m, int(o.shape[0]) * int(fg_z.shape[1]), ev[0], ev[1]))
            ret = eng.forward(o, d, far, fg_z, bg_z, training=False, events=ev)
            for k in keys:
                out[m][k].append(ret[k])
    merged = []
    for m in range(2):
        lvl = OrderedDict()
        for k in keys:
            t = gather_ragged(torch.cat(out[m][k], 0), sizes, rank, world_size)
            if rank == 0:
                lvl[k] = t.cpu().reshape((ray_sampler.H, ray_sampler.W, -1)).squeeze()
        merged.append(lvl)
    return merged if rank == 0 else None


def save_checkpoint(path, trainer, global_step):
    """{net_m: state_dict (DDP-prefixed keys), optim_m: Adam state_dict}   ddp_train_nerf.py:642-652"""
    import torch
    from .model import state_dict_from_flat, adam_state_dict
    trainer.flush()
    # never write (and later auto-reload) a checkpoint of a poisoned run.  Under data parallelism the bad-ray count is summed
    # over ranks, so a check on rank 0 alone would raise here while the other ranks block in the next step's gradient
    # all-reduce: there the training loop runs check_cameras() on EVERY rank right before this call
    if trainer.world_size == 1:
        trainer.check_cameras()
    to_save = OrderedDict()
    for m, eng in enumerate(trainer.engines):
        to_save['net_%d' % m] = OrderedDict((k, v.clone().cpu()) for k, v in
                                            state_dict_from_flat(eng.params).items())
        to_save['optim_%d' % m] = adam_state_dict(trainer.exp_avg[m].cpu(), trainer.exp_avg_sq[m].cpu(),
                                                  trainer.step_count, trainer.lrate)
        if trainer.autoexpo is not None:          # autoexpo_params follow nerf_net in parameters() order
            ae = trainer.autoexpo[m]
            for k, v in ae.state_dict_entries():
                to_save['net_%d' % m][k] = v.cpu()
            opt = to_save['optim_%d' % m]
            base = len(opt['param_groups'][0]['params'])
            for i, st in enumerate(ae.adam_entries()):
                opt['param_groups'][0]['params'].append(base + i)
                if st is not None:
                    opt['state'][base + i] = st
    torch.save(to_save, path)


def load_checkpoint(path, trainer):
    import torch
    from .model import load_state_dict_into_flat, load_adam_state_dict
    ck = torch.load(path, map_location='cpu', weights_only=False)
    for m, eng in enumerate(trainer.engines):
        load_state_dict_into_flat(eng.params, ck['net_%d' % m])
        if 'optim_%d' % m in ck:
            trainer.step_count = load_adam_state_dict(trainer.exp_avg[m], trainer.exp_avg_sq[m], ck['optim_%d' % m])
        eng.repack()
        if trainer.autoexpo is not None:
            ae = trainer.autoexpo[m]
            ae.load_state_dict_entries(ck['net_%d' % m])
            if 'optim_%d' % m in ck:
                from .model import level_param_specs
                base = len(level_param_specs())
                ae.load_adam_entries([ck['optim_%d' % m]['state'].get(base + i) for i in range(len(ae.names))])


def find_latest_checkpoint(args):
    """ddp_train_nerf.py:329-352: explicit --ckpt_path, else the newest model_*.pth by trailing integer."""
    def path2iter(path):
        tmp = os.path.basename(path)[:-4]
        return int(tmp[tmp.rfind('_') + 1:])
    if args.ckpt_path is not None and os.path.isfile(args.ckpt_path):
        ckpts = [args.ckpt_path]
    else:
        d = os.path.join(args.basedir, args.expname)
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith('.pth')] if os.path.isdir(d) else []
    ckpts = sorted(ckpts, key=path2iter)
    if ckpts and not args.no_reload:
        return ckpts[-1], path2iter(ckpts[-1])
    return None, -1


def depth_metrics(pred_depth, sampler, abs_err_map=None):
    """ddp_train_nerf.py:566-600: cap 80 m, valid 1e-3 < gt < 80, metres = value / depth_scale.
    abs_err_map: optional float array like pred_depth that receives |gt - pred| on the valid pixels (0 elsewhere), :591-592."""
    scale = sampler.get_depth_scale()
    gt = sampler.get_gt_depth_img() / scale
    pred = pred_depth / scale
    valid = (gt < 80) & (gt > 1e-3)
    vg, vp = gt[valid].clip(1e-3, 80), pred[valid].clip(1e-3, 80)
    if abs_err_map is not None:
        abs_err_map[...] = 0
        abs_err_map[valid] = np.abs(vg - vp)
    return float(np.sqrt(np.mean((vg - vp) ** 2))), float(np.mean(np.abs(vg - vp) / vg))


def minmax8(x):
    """min-max normalised uint8 map (ddp_train_nerf.py:562-564, :593-596).  A constant map is 0 / 0 upstream (NaN cast to
    uint8); here it is written as zeros."""
    x = np.asarray(x, np.float32)
    lo, hi = float(x.min()), float(x.max())
    if not hi > lo:
        return np.zeros(x.shape, np.uint8)
    return (np.clip((x - lo) / (hi - lo), 0., 1.) * 255.).astype(np.uint8)


def write_eval_images(out_dir, idx, ret, sampler):
    """The per-image artefacts of the in-loop evaluation (ddp_train_nerf.py:549-600): {idx}.png, fg_ / bg_ composites,
    error_rgb_ (mean absolute colour error, min-max normalised), depth_ (uint16 = metres x 256) and absrel_ (absolute depth
    error on the valid ground-truth pixels, min-max normalised).  Returns (psnr | None, rmse | None, absrel | None)."""
    from PIL import Image
    fname = '{:06d}.png'.format(idx)
    im = ret[-1]['rgb'].numpy()
    psnr = rmse = absrel = None
    if sampler.get_img() is not None:
        gt_im = sampler.get_img()
        psnr = float(mse2psnr(np.mean((gt_im - im) * (gt_im - im))))
        Image.fromarray(minmax8(np.abs(im - gt_im).mean(-1))).save(os.path.join(out_dir, 'error_rgb_' + fname))
    if sampler.get_gt_depth_img() is not None:
        pred = ret[-1]['depth'].numpy()
        err = np.zeros_like(pred, dtype=np.float32)
        rmse, absrel = depth_metrics(pred, sampler, err)
        d16 = ((pred / sampler.get_depth_scale()).clip(1e-3, 80) * 256.0)
        Image.fromarray(d16.astype(np.uint16)).save(os.path.join(out_dir, 'depth_' + fname))
        Image.fromarray(minmax8(err)).save(os.path.join(out_dir, 'absrel_' + fname))
    Image.fromarray(to8b(im)).save(os.path.join(out_dir, fname))
    Image.fromarray(to8b(ret[-1]['fg_rgb'].numpy())).save(os.path.join(out_dir, 'fg_' + fname))
    Image.fromarray(to8b(ret[-1]['bg_rgb'].numpy())).save(os.path.join(out_dir, 'bg_' + fname))
    return psnr, rmse, absrel


def ddp_train_nerf(rank, args):
    import torch
    from .trainer import NerfppTrainer, batch_to_device
    from .data_loader_split import load_data_split, synthetic_ray_samplers
    from . import _lib as L
    setup_logger()
    world = args.world_size
    torch.cuda.set_device(rank)
    device = torch.device('cuda', rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ['MASTER_PORT'] = str(args.port)
        from .dist_utils import apply_rccl_env_defaults
        # few channels: a CU RCCL holds is a CU a tile cannot use (dist_utils.py); process-global, so only on one node by default
        eff = apply_rccl_env_defaults(world, None if args.rccl_channels < 0 else args.rccl_channels)
        logger.info('rank %d RCCL environment: %s' % (rank, ' '.join('%s=%s' % kv for kv in sorted(eff.items())) or "RCCL's defaults"))
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)   # RCCL (gloo upstream, :298)

    # batch sizes by GPU memory                                      ddp_train_nerf.py:364-373
    if torch.cuda.get_device_properties(rank).total_memory / 1e9 > 14:
        args.N_rand, args.chunk_size = 1024, 8192
    else:
        args.N_rand, args.chunk_size = 512, 4096
    if args.N_rand_override:
        args.N_rand = args.N_rand_override
        args.chunk_size = max(args.chunk_size, 8 * args.N_rand_override)

    exp_dir = os.path.join(args.basedir, args.expname)
    if rank == 0:                                                   # :376-386
        os.makedirs(exp_dir, exist_ok=True)
        with open(os.path.join(exp_dir, 'args.txt'), 'w') as f:
            for arg in sorted(vars(args)):
                f.write('{} = {}\n'.format(arg, getattr(args, arg)))
        if args.config is not None:
            with open(os.path.join(exp_dir, 'config.txt'), 'w') as f:
                f.write(open(args.config, 'r').read())
    if world > 1:
        dist.barrier()

    if args.synthetic:
        hw = [int(x) for x in args.synthetic_hw.split(',')] if args.synthetic_hw else [None, None]
        ray_samplers = synthetic_ray_samplers('train', args.trainskip, args.depth_sup_type,
                                              args.synthetic_frames, hw[0], hw[1])
        val_ray_samplers = synthetic_ray_samplers('test', args.testskip, args.depth_sup_type,
                                                  args.synthetic_frames, hw[0], hw[1])
    else:
        ray_samplers = load_data_split(args.datadir, args.scene, split='train', skip=args.trainskip,
                                       try_load_min_depth=args.load_min_depth, depth_sup_type=args.depth_sup_type)
        val_ray_samplers = load_data_split(args.datadir, args.scene, split='test', skip=args.testskip,
                                           try_load_min_depth=args.load_min_depth, depth_sup_type=args.depth_sup_type)
    depth_scale = ray_samplers[0].get_depth_scale() or 1.0
    img_names = None
    if args.optim_autoexpo:                       # :394-399 (written before the nets need it, unlike upstream)
        img_names = [rs.img_path or 'synthetic/train/rgb/%06d.png' % i for i, rs in enumerate(ray_samplers)]
        if rank == 0:
            import json
            with open(os.path.join(exp_dir, 'train_images.json'), 'w') as f:
                json.dump(img_names, f, indent=2)
    device_samplers = None
    if not args.host_sampling:
        from .device_sampler import DeviceRaySamplers
        if len(set((rs.H, rs.W) for rs in ray_samplers)) == 1:
            device_samplers = DeviceRaySamplers(ray_samplers, device, seed=(rank + 1) * 777)
        else:
            logger.info('frames of different sizes: falling back to host-side ray sampling')

    cascade = tuple(int(x.strip()) for x in args.cascade_samples.split(','))
    comm = None
    if args.grad_comm == 'rccl_abi':
        from .dist_utils import RcclComm
        comm = RcclComm(rank, world)
    trainer = NerfppTrainer(device, precision={'bf16': L.PREC_BF16, 'split': L.PREC_SPLIT_BF16, 'split_fwd': L.PREC_SPLIT_FWD,
                                               'fp16_fwd': L.PREC_FP16_FWD}[args.precision],
                            cascade_samples=cascade, lrate=args.lrate, use_depth=args.use_depth,
                            depth_loss_type=args.depth_loss_type, lambda_depth=args.lambda_depth,
                            depth_sigma=args.depth_sigma, depth_scale=depth_scale, world_size=world,
                            optim_autoexpo=args.optim_autoexpo, img_names=img_names,
                            lambda_autoexpo=args.lambda_autoexpo, seed=(rank + 1) * 777, comm=comm)   # :406-408
    if rank == 0:
        # (the default is the slowest mode -- say so at start-up, with the alternatives and what they cost / keep)
        logger.info('precision = %s%s.  Relative throughput on one MI355X (bench.py, round 6): bf16 1.0, fp16_fwd 0.81, split_fwd 0.67, '
                    'split 0.40; split / split_fwd keep rendered RGB, depth and loss within 1e-4 of the float32 reference (split also the '
                    'gradients), bf16 is at bf16 grade (1e-2) -- see --help' % (args.precision, ' (the default)' if args.precision == 'split' else ''))
    ckpt, start = find_latest_checkpoint(args)
    if ckpt is not None:
        logger.info('Reloading from: {}'.format(ckpt))
        load_checkpoint(ckpt, trainer)

    np.random.seed((rank + 1) * 777)                                # :406
    torch.manual_seed((rank + 1) * 777)                             # :408
    writer = None
    if rank == 0:
        try:
            from tensorboardX import SummaryWriter
            writer = SummaryWriter(os.path.join(args.basedir, 'summaries', args.expname))
        except ImportError:
            logger.info('tensorboardX not installed: scalars go to the console only')

    for global_step in range(start + 1, start + 1 + args.N_iters):
        if global_step == start + 1:
            t_log, n_log = time.time(), 0
        n_log += 1
        if device_samplers is not None:
            ray_batch = device_samplers.random_sample(args.N_rand)   # two small kernels on the training stream: +0.8 % of a step
            # (drawing the next batch on a side stream instead measured +1.0 %: profiles/r04_cli_loop.md)
        else:
            i = np.random.randint(low=0, high=len(ray_samplers))
            ray_batch = batch_to_device(ray_samplers[i].random_sample(args.N_rand, center_crop=False), device)
            if img_names is not None:
                ray_batch['img_name'] = img_names[i]
        scalars = trainer.train_step(ray_batch)
        log_now = rank == 0 and (global_step % args.i_print == 0 or global_step < 10)
        if global_step % args.i_print == 0 or global_step < 10:
            trainer.check_cameras()               # every rank: the reference raises per step (:62-63)
        if log_now:
            scalars_to_log = OrderedDict([('resolution', ray_samplers[0].resolution_level)])
            for m, sc in enumerate(scalars):
                sc = sc.cpu().numpy()
                if not np.isfinite(sc[:2]).all():
                    raise FloatingPointError('non-finite loss at step %d, level %d: loss=%r rgb_loss=%r (check the '
                                             'scene normalisation / input data)' % (global_step, m, sc[0], sc[1]))
                if args.use_depth:
                    scalars_to_log['level_{}/loss_depth'.format(m)] = float(sc[2])
                if trainer.last_autoexpo[m] is not None:
                    scalars_to_log['level_{}/autoexpo_scale'.format(m)] = float(trainer.last_autoexpo[m][0])
                    scalars_to_log['level_{}/autoexpo_shift'.format(m)] = float(trainer.last_autoexpo[m][1])
                scalars_to_log['level_{}/rgb_loss'.format(m)] = float(sc[1])
                scalars_to_log['level_{}/pnsr'.format(m)] = float(mse2psnr(float(sc[1])))
            # the steps are queued asynchronously; reading the scalars above synchronised, so the wall time
            # since the previous log line over the steps it covers is the true time per iteration
            scalars_to_log['iter_time'] = (time.time() - t_log) / max(n_log, 1)
            t_log, n_log = time.time(), 0
            logstr = '{} step: {} '.format(args.expname, global_step)
            for k, v in scalars_to_log.items():
                logstr += ' {}: {:.6f}'.format(k, v)
                if writer is not None:
                    writer.add_scalar(k, v, global_step)
            logger.info(logstr)

        if (global_step % args.i_test == 0) and global_step != 0:   # :539-640
            out_dir = os.path.join(exp_dir, 'render_{}_{:06d}'.format('test', global_step))
            if rank == 0:
                os.makedirs(out_dir, exist_ok=True)
            psnrs, rmses, abs_rels = [], [], []
            trainer.check_cameras()
            for idx, sampler in enumerate(val_ray_samplers):
                ret = render_single_image(rank, world, trainer, sampler, args.chunk_size, keep_dists=False)   # fg_dists is never read below
                if rank != 0:
                    continue
                psnr, rmse, absrel = write_eval_images(out_dir, idx, ret, sampler)
                if psnr is not None:
                    psnrs.append(psnr)
                if rmse is not None:
                    rmses.append(rmse)
                    abs_rels.append(absrel)
            if rank == 0:
                for name, vals in (('psnr', psnrs), ('rmse', rmses), ('absrel', abs_rels)):
                    if vals:
                        vals = vals + [float(np.mean(vals))]
                        with open(os.path.join(out_dir, '%s_%06d.txt' % (name, global_step)), 'w') as f:
                            f.write('\n'.join(str(p) for p in vals))
                        if writer is not None:
                            writer.add_scalar('test_' + name, vals[-1], global_step)
                        logger.info('test_%s: %s' % (name, vals[-1]))

        if global_step % args.i_weights == 0 and global_step > 0:   # :642-652
            trainer.check_cameras()               # every rank, so that all raise together: never write (and later auto-reload) a checkpoint of a poisoned run
        if rank == 0 and (global_step % args.i_weights == 0 and global_step > 0):
            save_checkpoint(os.path.join(exp_dir, 'model_{:06d}.pth'.format(global_step)), trainer, global_step)

    trainer.flush()                               # the last step's level-1 update is applied lazily under DP
    trainer.check_cameras()
    on_finish = getattr(args, 'on_finish', None)  # (bench.py: times the kernel-only step on the SAME trained state)
    if on_finish is not None:
        on_finish(trainer, device_samplers if device_samplers is not None else ray_samplers)
    if comm is not None:
        torch.cuda.synchronize()
        comm.destroy()                            # the library's own communicator, before the process group that carried its id
    if world > 1:
        dist.destroy_process_group()


def train(argv=None):
    import torch
    parser = config_parser()
    args = parser.parse_args(argv)
    validate_args(args)
    if os.path.exists(os.path.join(args.basedir, args.expname)):     # :733-735
        print('already trained, exiting...')
        sys.exit(0)
    if args.world_size == -1:
        args.world_size = torch.cuda.device_count()
    if args.world_size <= 1:
        args.world_size = 1
        ddp_train_nerf(0, args)
    else:
        torch.multiprocessing.spawn(ddp_train_nerf, args=(args,), nprocs=args.world_size, join=True)


if __name__ == '__main__':
    setup_logger()
    train()
