#!/usr/bin/env python
"""Assemble the pieces tools/probes/profile_round.sh wrote into the markdown files that get committed under profiles/:
    <tag>_kernel_stats_timeline_hbm.md   NeRF++ step: rocprofv3 kernel stats, one step dispatch by dispatch, HBM PMC passes
                                         (bench.py parses the `## kernel (dispatches: n)` / `FETCH_SIZE avg` blocks of this file)
    <tag>_render_kernel_stats.md         rendering leg (render_single_image), bf16 and split-bf16
    <tag>_mip360_kernel_stats_timeline.md
usage: python tools/probes/assemble_profile.py gpurun_out/<tag> <tag>"""
import os
import sys


def rd(d, name):
    p = os.path.join(d, name)
    return open(p).read().rstrip() + '\n' if os.path.exists(p) else '(missing: %s)\n' % name


def main(d, tag):
    with open(os.path.join(d, tag + '_kernel_stats_timeline_hbm.md'), 'w') as f:
        f.write('# %s -- NeRF++ training step (bench.py, N_rand = 1024, 64 + 128 samples), one MI355X\n\n' % tag)
        f.write('Collected by `tools/probes/profile_round.sh %s` (rocprofv3 --kernel-trace --stats; bf16, split-bf16 and split_fwd in '
                'one process).\n\n' % tag)
        # bench.py re-computes this hash of csrc/nerfpp_{mlp,dw}.hip + nerfpp_common.h and withholds `roofline.traffic` when the
        # kernels have changed since these PMC passes
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
        import bench
        f.write('sources_sha256: %s\n\n' % bench.kernel_sources_sha256())
        f.write("Level 0's backward and weight-gradient dispatches run on their own stream under level 1's sampling and forward (`NerfppTrainer.concurrent_backward`): their durations and those of level 1's `mlp_fwd_pair_kernel` below are wall time while sharing the GPU (negative gaps in the timeline).  Level 1's `mlp_bwd_pair_kernel` / `dw_kernel` dispatches are never overlapped; a `*_pair_kernel` dispatch covers the fg and the bg net of a level; the average rows mix level-0 (1024 x 64 samples) and level-1 (1024 x 192) launches -- the timeline separates them." + '\n\n')
        f.write(rd(d, 'kernel_stats.md'))
        f.write('\n## One bf16 training step, dispatch by dispatch (tools/rocpd_timeline.py)\n\n')
        f.write(rd(d, 'timeline_bf16.md'))
        f.write('\n## HBM traffic per dispatch (separate --pmc passes FETCH_SIZE, WRITE_SIZE; KB; FETCH_SIZE on gfx950 counts 64 B '
                'per 128-B request for wide streams: double it)\n\n```\n')
        f.write(rd(d, 'hbm_traffic.txt'))
        f.write('```\n')
    with open(os.path.join(d, tag + '_render_kernel_stats.md'), 'w') as f:
        f.write('# %s -- rendering leg: render_single_image on one 375 x 1242 frame (tools/render_bench.py, 1 warm-up + 2 timed '
                'frames), rocprofv3 --kernel-trace --stats\n' % tag)
        for prec in ('bf16', 'split'):
            f.write('\n## %s\n\n```\n%s```\n\n' % (prec, rd(d, 'render_%s.json' % prec)))
            f.write(rd(d, 'render_%s_kernel_stats.md' % prec))
    with open(os.path.join(d, tag + '_mip360_kernel_stats_timeline.md'), 'w') as f:
        f.write('# %s -- MipNeRF-360 step (BASELINE config 5), 4096 rays, configs/360.gin shape\n\n```\n%s```\n\n' % (tag, rd(d, 'mip360_bench.json')))
        f.write('## rocprofv3 --kernel-trace --stats (6 timed + 2 warm-up steps)\n\n')
        f.write(rd(d, 'mip360_kernel_stats.md'))
        f.write('\n## One training step, dispatch by dispatch\n\n')
        f.write(rd(d, 'mip360_timeline.md'))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
