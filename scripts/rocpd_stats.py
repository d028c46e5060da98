#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (ROCm 7 default output of `rocprofv3 --kernel-trace --stats`)
into the per-kernel table `--stats` prints: calls, total / average / min / max duration, share.
    rocpd_stats.py <db> [out.txt] [--window <kernel substring>]
--window K (round 6): only the dispatches from the LAST launch of a kernel whose name contains K to the end of the trace.  `bench.py
--steps 1` opens its timed region with the GAE kernel, so `--window gae_` restricts the table to the timed update -- the per-kernel
averages then are the minibatch-shaped launches only (the whole-trace table mixes them with the act-side launches of the same
kernels during the rollout)."""
import sqlite3
import sys


def main(argv):
    args = [a for a in argv if not a.startswith('--')]
    window = argv[argv.index('--window') + 1] if '--window' in argv else None
    if window in args:
        args.remove(window)
    db, out = args[0], (args[1] if len(args) > 1 else None)
    c = sqlite3.connect(db)
    where, head = '', ''
    if window:
        row = c.execute("select max(start) from kernels where name like ?", ('%' + window + '%',)).fetchone()
        if row and row[0] is not None:
            where = ' where start >= %d' % row[0]
            n, span = c.execute("select count(*), max(end) - min(start) from kernels" + where).fetchone()
            head = ('# window: from the last launch of *%s* to the end of the trace: %d dispatches, %.3f ms wall, kernels busy %.3f ms\n'
                    % (window, n, span / 1e6, c.execute("select sum(end-start) from kernels" + where).fetchone()[0] / 1e6))
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels" + where + " group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ['%-110s %8s %14s %12s %12s %12s %7s' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', 'pct')]
    for r in rows:
        name = r[0] if len(r[0]) <= 110 else r[0][:107] + '...'
        lines.append('%-110s %8d %14d %12.0f %12d %12d %6.2f%%' % (name, r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
    text = head + '\n'.join(lines) + '\n'
    if out:
        open(out, 'w').write(text)
    else:
        sys.stdout.write(text)


if __name__ == '__main__':
    main(sys.argv[1:])
