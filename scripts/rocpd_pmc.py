#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database.  usage: rocpd_pmc.py db [substr]"""
import sqlite3
import sys
from collections import defaultdict


def main(db, sub=''):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    namecol = 'kernel_name' if 'kernel_name' in cols else 'name'
    rows = c.execute("select %s, counter_name, value, dispatch_id from counters_collection" % namecol).fetchall()
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for k, cn, v, d in rows:
        if sub in k and any(t in k for t in ('gemm', 'wres', 'imgres', 'dgrad', 'heads', 'mlp_step', 'reduce_slabs', 'adam', 'lstm')):
            acc[k][cn] += v
            disp[k].add(d)
    for k in acc:
        n = len(disp[k])
        print(k[:150], 'dispatches', n)
        for cn in sorted(acc[k]):
            print('    %-32s %16.0f' % (cn, acc[k][cn] / n))


if __name__ == '__main__':
    main(*sys.argv[1:3])
