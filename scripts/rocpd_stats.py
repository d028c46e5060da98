#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (ROCm 7 default output of `rocprofv3 --kernel-trace --stats`)
into the per-kernel table `--stats` prints: calls, total / average / min / max duration, share."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ['%-110s %8s %14s %12s %12s %12s %7s' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', 'pct')]
    for r in rows:
        name = r[0] if len(r[0]) <= 110 else r[0][:107] + '...'
        lines.append('%-110s %8d %14d %12.0f %12d %12d %6.2f%%' % (name, r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
    text = '\n'.join(lines) + '\n'
    if out:
        open(out, 'w').write(text)
    else:
        sys.stdout.write(text)


if __name__ == '__main__':
    main(*sys.argv[1:3])
