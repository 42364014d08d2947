#!/usr/bin/env python
"""Per-kernel PMC counter averages from a rocprofv3 rocpd database.

    python tools/rocpd_pmc.py gpurun_out/pmc1/*/*.db [kernel-substring]
"""
import collections
import re
import sqlite3
import sys


def main(path, filt=''):
    c = sqlite3.connect(path)
    rows = c.execute('select kernel_name, counter_name, value, dispatch_id from counters_collection').fetchall()
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, n, v, d in rows:
        k = re.sub(r'\(.*$', '', k).replace('nerfpp::', '').replace('void ', '')
        if filt in k:
            agg[k][n].append(v)
    for k, cs in agg.items():
        print('## %s  (dispatches: %d)' % (k[:80], max(len(v) for v in cs.values())))
        for n, v in sorted(cs.items()):
            print('  %-28s avg %.4g' % (n, sum(v) / len(v)))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
