#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace as a per-kernel stats table (markdown/CSV).

    python tools/rocpd_stats.py gpurun_out/prof1/*/*.db > profiles/r01_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    name = name.replace('nerfpp::', '').replace('void ', '')
    return name[:70]


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else [x for x in cols if 'name' in x][0]
    rows = c.execute('select %s, start, end from kernels' % name_col).fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print('| kernel | calls | total us | avg us | min us | max us | % |')
    print('|---|---|---|---|---|---|---|')
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('| %s | %d | %.1f | %.2f | %.2f | %.2f | %.2f |' % (short(name), a[0], a[1], a[1] / a[0], a[2], a[3],
                                                                 100 * a[1] / tot))


if __name__ == '__main__':
    main(sys.argv[1])
