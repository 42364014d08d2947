#!/usr/bin/env python
"""Headline benchmark: NeRF++ training ray-steps per second on synthetic KITTI-shaped batches.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one optimisation step of nerf-methods/nerfplusplus/ddp_train_nerf.py:417-498 for
N_rand rays per GPU: stratified + inverse-CDF sampling, both cascade levels (64 and 64+128 samples
per ray, fg + bg networks), loss (depth_sup_type=gt, depth_loss_type=mse, lambda_depth=0.1 =
BASELINE config 2), backward, gradient all-reduce (N > 1) and Adam.  Ray batches are resident in HBM
before the timed region.  Prints ONE JSON line (rank 0).

value            : single-pass bf16 MFMA (the arithmetic north_star names), rays/s over all GPUs
parity_mode      : the same step in split-bf16 precision (3 MFMA passes), the mode the 1e-4 parity
                   tests run in -- both numbers come from the same invocation
roofline         : dominant kernel of the bf16 run, timed live with HIP events on its stream
cpu_baseline     : the numpy oracle (a port of the reference's PyTorch path) on the host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0            # HBM3E spec (6.3 TB/s measured copy ceiling)
# algorithmic bytes per sample row read by the weight-gradient GEMMs (every operand once per job):
# sum over jobs of (dZ cols + input cols) * 2 B, fg + bg
DW_BYTES_PER_ROW = (4960 + 5024) * 2
# level-1 dW launch pair at N_rand=1024, bf16: FETCH_SIZE (x2 gfx950 wide-stream correction) + WRITE_SIZE
DW_TRAFFIC_PMC_BYTES = (2 * 1.5 * (917.6e6 + 356.3e6)) + 1.5 * (64.7e6 + 23.1e6)
DW_TRAFFIC_SOURCE = 'profiles/r01_j_kernel_stats_timeline_hbm.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)'


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--n_rand', type=int, default=1024, help='rays per GPU per step (reference forces 1024)')
    p.add_argument('--precision', choices=['both', 'bf16', 'split'], default='both')
    p.add_argument('--no_cpu_baseline', action='store_true')
    p.add_argument('--large_batch', type=int, default=8192, help='also report this N_rand (0 = skip)')
    # BASELINE.json configs[1] by default (gt / mse / 0.1); configs[2] = mono_crop / kl, configs[3] = stereo_crop / l1
    p.add_argument('--depth_sup_type', default='gt')
    p.add_argument('--depth_loss_type', default='mse', choices=['mse', 'l1', 'kl'])
    p.add_argument('--lambda_depth', type=float, default=0.1)
    p.add_argument('--cpu_rays', type=int, default=128)
    return p.parse_args()


def run_mode(args, precision, rank, world, device, batches):
    import torch
    from outdoor_nerf_depth_amd.trainer import NerfppTrainer, ALGO_MACS
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti

    scale = float(SyntheticKitti().depth_scale)
    tr = NerfppTrainer(device, precision=precision, use_depth=True, depth_loss_type=args.depth_loss_type,
                       lambda_depth=args.lambda_depth,
                       depth_scale=scale, world_size=world)
    K, W = args.steps, args.warmup
    # live kernel taps: HIP events recorded by the library around the level-1 kernels, on the launch stream,
    # inside the timed region.  An event record costs ~6 us of queue time, so only every TAP-th timed step
    # carries them (>= 3 tapped steps) and level 0 is not tapped.
    mk = lambda: torch.cuda.Event(enable_timing=True)
    TAP = max(1, min(4, K // 3))
    tapped = [i for i in range(K) if i % TAP == 0]
    events = {}
    for i in tapped:
        ev = {'fwd': (mk(), mk()), 'bwd': (mk(), mk(), mk(), mk())}
        for e in ev['fwd'] + ev['bwd']:
            e.record()                           # materialise the hipEvent_t handles
        events[i] = [None, ev]
    for i in range(W):
        tr.train_step(batches[i])
    tr.flush()
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for i in range(K):
        last = tr.train_step(batches[W + i], events=events.get(i))
    tr.flush()                                   # the last step's level-1 all-reduce + Adam belong to the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = [float(s[0]) for s in last]
    assert all(np.isfinite(loss)), 'non-finite loss %r' % (loss,)

    # live per-kernel timing (ms) of the level-1 foreground kernels + the weight-gradient GEMM
    n, S1 = args.n_rand, 192
    rows = n * S1
    med = lambda xs: float(np.median(xs))
    fwd_ms = med([events[i][1]['fwd'][0].elapsed_time(events[i][1]['fwd'][1]) for i in tapped])
    bwd_ms = med([events[i][1]['bwd'][0].elapsed_time(events[i][1]['bwd'][1]) for i in tapped])
    dw_ms = med([events[i][1]['bwd'][2].elapsed_time(events[i][1]['bwd'][3]) for i in tapped])
    P = 2 if precision == 2 else 1            # precision of the backward kernels / saved planes the dW GEMM reads
    kernels = {
        'mlp_fwd_fg_L1': dict(ms=fwd_ms, flop=2.0 * ALGO_MACS['fwd'][0] * rows, bound='mfma'),
        'mlp_bwd_fg_L1': dict(ms=bwd_ms, flop=2.0 * ALGO_MACS['dx'][0] * rows, bound='mfma'),
        'dw_both_L1': dict(ms=dw_ms, flop=2.0 * (ALGO_MACS['fwd'][0] + ALGO_MACS['fwd'][1]) * rows, bound='hbm',
                           bytes=float(DW_BYTES_PER_ROW) * P * rows),
    }
    for k in kernels.values():
        k['tflops'] = k['flop'] / (k['ms'] * 1e-3) / 1e12
        if 'bytes' in k:
            k['gbs'] = k['bytes'] / (k['ms'] * 1e-3) / 1e9
    # per-step share: fwd and bwd kernels run for fg and bg at both levels, dw once per level
    share = {'mlp_fwd_fg_L1': fwd_ms * 2 * (1 + 64.0 / 192), 'mlp_bwd_fg_L1': bwd_ms * 2 * (1 + 64.0 / 192),
             'dw_both_L1': dw_ms * (1 + 64.0 / 192)}
    dominant = max(share, key=share.get)
    # HBM bytes of the dominant launch from the PMC passes committed under profiles/ (cannot be collected
    # inside this process): only quoted for the configuration they were measured on
    traffic = DW_TRAFFIC_PMC_BYTES if (DW_TRAFFIC_PMC_BYTES and n == 1024 and precision == 1) else None
    return dict(elapsed=elapsed, ms_per_step=1e3 * elapsed / K, value=n * K * world / elapsed, loss=loss,
                value_per_gpu=n * K / elapsed, kernels=kernels, dominant=dominant, share_ms=share,
                dw_traffic_pmc=traffic)


def roofline(r):
    """Dominant kernel (largest share of the step) against the roof that bounds it: the fused MLP
    kernels against dense bf16 MFMA, the weight-gradient GEMM (split-K over the samples, every
    operand streamed once) against HBM.  Durations are HIP-event times recorded by the library on the
    launch stream inside the timed region."""
    dom = r['kernels'][r['dominant']]
    # whole-step figure of SURVEY 8(d): 1.797 GFLOP of dense-layer work per ray-step against the bf16 peak
    whole = {'tflops': r['value_per_gpu'] * 1.797e9 / 1e12, 'frac_of_bf16_mfma_peak': r['value_per_gpu'] * 1.797e9 / 1e12 / PEAK_BF16_TFLOPS}
    allk = {k: {kk: round(vv, 4) for kk, vv in v.items() if kk in ('ms', 'tflops', 'gbs')}
            for k, v in r['kernels'].items()}
    if dom['bound'] == 'hbm':
        return {'bound': 'hbm', 'kernel': r['dominant'], 'achieved': dom['gbs'], 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                'frac': dom['gbs'] / PEAK_HBM_GBS, 'traffic': r.get('dw_traffic_pmc'), 'launch_ms': dom['ms'],
                'traffic_source': DW_TRAFFIC_SOURCE if r.get('dw_traffic_pmc') else None,
                'mfma_tflops_of_same_kernel': dom['tflops'], 'whole_step': whole, 'all_kernels': allk,
                'share_ms_per_step': {k: round(v, 4) for k, v in r['share_ms'].items()}}
    return {'bound': 'mfma', 'kernel': r['dominant'], 'achieved': dom['tflops'], 'peak': PEAK_BF16_TFLOPS,
            'unit': 'TFLOP/s', 'frac': dom['tflops'] / PEAK_BF16_TFLOPS, 'traffic': None, 'launch_ms': dom['ms'],
            'whole_step': whole, 'all_kernels': allk, 'share_ms_per_step': {k: round(v, 4) for k, v in r['share_ms'].items()}}


def cpu_baseline(args):
    """The numpy oracle (a port of the reference's PyTorch-CPU path, validated against it in
    tests/test_oracle_golden.py) on a bounded sample: cpu_rays rays, both levels, fwd+bwd+Adam."""
    from oracle import nerfpp_oracle as O
    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    n = args.cpu_rays
    levels = O.init_params_like_reference(2)
    opt = O.new_opt_state(levels)
    scene = SyntheticKitti(depth_sup_type='gt')
    rng = np.random.RandomState(777)
    times = []
    for step in range(1, 4):
        b = scene.random_batch(n, rng)
        uni = dict(t_fg=rng.rand(n, 64).astype(np.float32), t_bg=rng.rand(n, 64).astype(np.float32),
                   u_fg=rng.rand(n, 128).astype(np.float32), u_bg=rng.rand(n, 128).astype(np.float32))
        t0 = time.perf_counter()
        O.train_step(levels, opt, step, b, uni, use_depth=True, depth_loss_type='mse', lambda_depth=0.1)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:]))
    return dict(value=n / t, unit='rays/s', cores=os.cpu_count(), kind='port',
                sample='%d rays/step, 1 warm-up + 2 timed steps, both levels fwd+bwd+Adam, float32 numpy '
                       '(OpenBLAS threads = all host cores)' % n)


def main():
    args = parse()
    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # test hooks: NERFPP_SHARE_GPU=1 puts every rank on cuda:0 and NERFPP_DIST_BACKEND=gloo replaces RCCL,
    # so the N > 1 code path can be smoke-tested on a 1-GPU box (numbers from such a run mean nothing)
    if os.environ.get('NERFPP_SHARE_GPU'):
        local = 0
    backend = os.environ.get('NERFPP_DIST_BACKEND', 'nccl')
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert torch.cuda.is_available(), 'bench.py needs a GPU: the NeRF++ hot path has no CPU fallback'
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)

    from outdoor_nerf_depth_amd.synthetic import SyntheticKitti
    from outdoor_nerf_depth_amd.trainer import batch_to_device
    from outdoor_nerf_depth_amd import _lib as L
    scene = SyntheticKitti(depth_sup_type=args.depth_sup_type)
    rng = np.random.RandomState((rank + 1) * 777)                 # ddp_train_nerf.py:406
    torch.manual_seed((rank + 1) * 777)                           # :408
    batches = [batch_to_device(scene.random_batch(args.n_rand, rng), device)
               for _ in range(args.steps + args.warmup)]

    res = {}
    if world > 1 and args.precision == 'both':
        args.precision = 'bf16'                  # the scaling runs measure the headline precision only
    if args.precision in ('both', 'bf16'):
        res['bf16'] = run_mode(args, L.PREC_BF16, rank, world, device, batches)
    if args.precision in ('both', 'split'):
        res['split'] = run_mode(args, L.PREC_SPLIT_BF16, rank, world, device, batches)
    if rank != 0:
        return
    main_key = 'bf16' if 'bf16' in res else 'split'
    r = res[main_key]
    dom = r['kernels'][r['dominant']]
    out = {
        'metric': 'train rays/sec, KITTI 375x1242, 64+128 samples/ray',
        'value': r['value'], 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': r['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16 MFMA operands, f32 accumulate / f32 master weights' if main_key == 'bf16'
                 else 'split-bf16 (hi+lo, 3 MFMA passes), f32 accumulate',
        'data': 'synthetic',
        'config': {'workload': 'NeRF++ KITTI seq00-shaped (295 fr, 375x1242), depth_sup_type=%s, '
                               'depth_loss_type=%s, lambda_depth=%g, N_rand=%d rays/GPU/step, cascade 64+128, '
                               'both levels fwd+bwd+Adam' % (args.depth_sup_type, args.depth_loss_type,
                                                             args.lambda_depth, args.n_rand),
                   'n_rand_per_gpu': args.n_rand, 'parallelism': 'dp%d (ray batches, RCCL grad all-reduce)' % world},
        'roofline': roofline(r),
        'final_loss': r['loss'],
    }
    if world == 1 and args.precision == 'both':
        # split-bf16 forward (rendered RGB / depth / loss within 1e-4 of float32) + bf16 backward over its hi planes
        h = run_mode(args, L.PREC_SPLIT_FWD, rank, world, device, batches)
        out['parity_forward_mode'] = {'dtype': 'split-bf16 forward (outputs and loss at 1e-4), bf16 backward / weight gradients',
                                      'value': h['value'], 'ms_per_step': h['ms_per_step']}
    if 'split' in res and main_key != 'split':
        s = res['split']
        out['parity_mode'] = {'dtype': 'split-bf16 (hi+lo, 3 MFMA passes): the precision the 1e-4 parity tests use',
                              'value': s['value'], 'ms_per_step': s['ms_per_step'],
                              'roofline': roofline(s)}
    if world == 1 and args.large_batch > 0 and main_key == 'bf16':
        # "report at N_rand=1024 (reference value) and at the largest N_rand that fits, labelled"
        import copy
        a2 = copy.copy(args)
        a2.n_rand, a2.steps, a2.warmup = args.large_batch, 5, 2
        b2 = [batch_to_device(scene.random_batch(a2.n_rand, rng), device) for _ in range(a2.steps + a2.warmup)]
        r2 = run_mode(a2, L.PREC_BF16, rank, world, device, b2)
        out['large_batch'] = {'n_rand_per_gpu': a2.n_rand, 'value': r2['value'], 'unit': 'rays/s',
                              'ms_per_step': r2['ms_per_step'], 'steps': a2.steps,
                              'note': 'same workload at a larger ray batch than the 1024 of the reference (labelled, not the headline)'}
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(args)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
