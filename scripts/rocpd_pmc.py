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
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    for k, cn, v, d in rows:
        if sub in k and any(t in k for t in ('gemm', 'wres', 'imgres', 'dgrad', 'heads', 'mlp_step', 'reduce_slabs', 'adam', 'lstm',
                                             'c1fwd', 'c1wgrad', 'wgrad_x8', 'wgrad_tr', 'conv_x6c', 'conv_skinny', 'q_lat', 'q_adam')):
            acc[k][cn] += v
            disp[k].add(d)
            per[k][cn][d] += v
    for k in acc:
        n = len(disp[k])
        print(k[:150], 'dispatches', n)
        for cn in sorted(acc[k]):
            print('    %-32s %16.0f' % (cn, acc[k][cn] / n))
            # the minibatch-shaped launches of a kernel that also serves the act side: the 8 largest dispatches, in launch
            # order (a kernel name shared by two layers alternates between them)
            if n > 8:
                top = sorted(sorted(per[k][cn].items(), key=lambda kv: -kv[1])[:8])
                print('    %-32s %s' % (cn + '.top8', ' '.join('%.0f' % v for _, v in top)))


if __name__ == '__main__':
    main(*sys.argv[1:3])
