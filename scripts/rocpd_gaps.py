#!/usr/bin/env python
"""Execution time and idle gaps of a launch-bound kernel sequence in a rocprofv3 rocpd database (kernel trace): per kernel
name the average duration, and the average gap between the END of the previous kernel (any name) and its START.
python scripts/rocpd_gaps.py <db> [name-substring to restrict the window, e.g. mlp_step]"""
import sqlite3
import sys


def main(db, key=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    if key:
        idx = [i for i, r in enumerate(rows) if key in r[0]]
        rows = rows[idx[len(idx) // 2]:idx[-1] + 3]            # second half of the window (after warm-up), incl. the step's tail kernels
    agg = {}
    for prev, cur in zip(rows, rows[1:]):
        name = cur[0].split('(')[0][-60:]
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += cur[2] - cur[1]
        a[2] += max(0, cur[1] - prev[2])
    print('%-62s %8s %12s %12s' % ('kernel', 'calls', 'avg_exec_us', 'avg_gap_us'))
    for k, (n, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-62s %8d %12.2f %12.2f' % (k, n, d / n / 1e3, g / n / 1e3))
    span = rows[-1][2] - rows[0][1]
    busy = sum(r[2] - r[1] for r in rows)
    print('window %.3f ms, kernels busy %.3f ms (%.1f %%)' % (span / 1e6, busy / 1e6, 100.0 * busy / span))


if __name__ == '__main__':
    main(*sys.argv[1:3])
