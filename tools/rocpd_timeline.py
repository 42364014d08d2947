#!/usr/bin/env python
"""Per-step timeline from a rocprofv3 rocpd (.db) kernel trace: every dispatch of the LAST complete
training step (a step starts at sample_coarse_kernel of level 0) with its start offset, duration and
the idle gap since the previous kernel ended; then busy / idle totals.

    python tools/rocpd_timeline.py gpurun_out/prof/*/*.db [anchor_kernel_substring [anchors_per_step]]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    return name.replace('nerfpp::', '').replace('mip360::', '').replace('void ', '')[:60]


def main(path, anchor='sample_coarse_kernel', per_step=1):
    per_step = int(per_step)
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else [x for x in cols if 'name' in x][0]
    rows = sorted(c.execute('select %s, start, end from kernels' % name_col).fetchall(), key=lambda r: r[1])
    starts = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(starts) < 3 * per_step:
        raise SystemExit('need at least 3 steps of anchors')
    starts = starts[len(starts) % per_step::per_step]          # first anchor of every step
    lo, hi = starts[-3], starts[-2]
    step = rows[lo:hi]
    t0 = step[0][1]
    prev_end = t0
    busy = 0.0
    print('| # | kernel | start us | dur us | gap before us |')
    print('|---|---|---|---|---|')
    for i, (n, s, e) in enumerate(step):
        print('| %d | %s | %.1f | %.1f | %.1f |' % (i, short(n), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
        busy += (e - s) / 1e3
        prev_end = max(prev_end, e)
    span = (rows[hi][1] - t0) / 1e3
    print('\nstep span %.1f us, kernels busy %.1f us, idle %.1f us (%d dispatches)' % (span, busy, span - busy, len(step)))


if __name__ == '__main__':
    main(*sys.argv[1:4])
