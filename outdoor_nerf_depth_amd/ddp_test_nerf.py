#!/usr/bin/env python
"""Offline test-set render + metrics: the counterpart of nerf-methods/nerfplusplus/ddp_test_nerf.py
(:23-160) on the HIP path.  Same parser as training (`--config`, `--render_splits`, `--ckpt_path`);
for every split renders each image with the newest (or given) checkpoint -- deterministic
sampling, no perturbation -- and writes under {basedir}/{expname}/render_{split}_{step:06d}/:
  {idx:06d}.png, fg_*.png, bg_*.png, depth_*.png (uint16 = metres*256), error_rgb_*.png / absrel_*.png (the min-max
  normalised error maps of the training loop's evaluation, ddp_train_nerf.py:561-596) and
  psnr_/rmse_/absrel_{step:06d}.txt (per image, then the mean).
PSNR = mse2psnr(mean((gt-im)^2)) on float images; depth metrics use the 80 m cap and
1e-3 < gt < 80 validity of the reference (:87-116).
"""
import os
import sys

import numpy as np

from .ddp_train_nerf import (config_parser, validate_args, setup_logger, render_single_image, load_checkpoint,
                             find_latest_checkpoint, write_eval_images, logger)


def ddp_test_nerf(rank, args):
    import torch
    from .trainer import NerfppTrainer
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
        apply_rccl_env_defaults(world, None if getattr(args, 'rccl_channels', -1) < 0 else args.rccl_channels)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    cascade = tuple(int(x.strip()) for x in args.cascade_samples.split(','))
    # forward only: bf16, the two-pass fp16x2w forward (1e-4 outputs), or split-bf16 (also for split_fwd: the same forward)
    trainer = NerfppTrainer(device, precision={'bf16': L.PREC_BF16, 'fp16_fwd': L.PREC_FP16_FWD}.get(args.precision, L.PREC_SPLIT_BF16),
                            cascade_samples=cascade, use_depth=False, world_size=1)
    ckpt, start = find_latest_checkpoint(args)
    if ckpt is None:
        raise SystemExit('no checkpoint found under %s' % os.path.join(args.basedir, args.expname))
    logger.info('Reloading from: {}'.format(ckpt))
    load_checkpoint(ckpt, trainer)
    for split in [x.strip() for x in args.render_splits.strip().split(',')]:
        out_dir = os.path.join(args.basedir, args.expname, 'render_{}_{:06d}'.format(split, start))
        if rank == 0:
            os.makedirs(out_dir, exist_ok=True)
        if args.synthetic:
            hw = [int(x) for x in args.synthetic_hw.split(',')] if args.synthetic_hw else [None, None]
            samplers = synthetic_ray_samplers(split, args.testskip, args.depth_sup_type, args.synthetic_frames,
                                              hw[0], hw[1])
        else:
            samplers = load_data_split(args.datadir, args.scene, split, skip=args.testskip,
                                       try_load_min_depth=args.load_min_depth, depth_sup_type=args.depth_sup_type)
        psnrs, rmses, abs_rels = [], [], []
        for idx, sampler in enumerate(samplers):
            ret = render_single_image(rank, world, trainer, sampler, args.chunk_size, keep_dists=False)   # fg_dists is never read below
            if rank != 0:
                continue
            psnr, rmse, absrel = write_eval_images(out_dir, idx, ret, sampler)      # incl. error_rgb_ / absrel_ (ddp_train_nerf.py:561-596)
            if psnr is not None:
                psnrs.append(psnr)
            if rmse is not None:
                rmses.append(rmse)
                abs_rels.append(absrel)
        if rank == 0:
            for name, vals in (('psnr', psnrs), ('rmse', rmses), ('absrel', abs_rels)):
                if vals:
                    vals = vals + [float(np.mean(vals))]
                    with open(os.path.join(out_dir, '%s_%06d.txt' % (name, start)), 'w') as f:
                        f.write('\n'.join(str(p) for p in vals))
                    logger.info('%s %s: %s' % (split, name, vals[-1]))
    if world > 1:
        dist.destroy_process_group()


def test(argv=None):
    import torch
    args = config_parser().parse_args(argv)
    validate_args(args)
    if args.world_size == -1:
        args.world_size = torch.cuda.device_count()
    if args.world_size <= 1:
        args.world_size = 1
        ddp_test_nerf(0, args)
    else:
        torch.multiprocessing.spawn(ddp_test_nerf, args=(args,), nprocs=args.world_size, join=True)


if __name__ == '__main__':
    setup_logger()
    test()
