#!/usr/bin/env python
"""One period of a repeating kernel sequence out of a rocprofv3 rocpd database (kernel trace), as a timeline: start offset, duration,
queue / stream and name of every kernel between two consecutive occurrences of an anchor kernel (taken late in the run, after warm-up).
    python scripts/rocpd_timeline.py <db> <anchor-name-substring> [periods-from-the-end, default 3]"""
import sqlite3
import sys


def short(name):
    n = name
    for pre in ('void ', 'mrl::', '(anonymous namespace)::', 'at::native::'):
        n = n.replace(pre, '')
    depth, out = 0, []
    for ch in n:                       # cut at the first '(' outside template brackets: the argument list
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            break
        out.append(ch)
    return ''.join(out)[:110]


def main(db, anchor, back=3):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    extra = [k for k in ('queue_id', 'stream_id') if k in cols]
    rows = c.execute("select name, start, end%s from kernels order by start" % ''.join(', ' + k for k in extra)).fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    lo, hi = idx[-int(back) - 1], idx[-int(back)]
    t0 = rows[lo][1]
    print('%9s %8s %8s  %-8s %s' % ('start_us', 'dur_us', 'gap_us', '/'.join(extra) or '-', 'kernel'))
    prev_end = t0
    for r in rows[lo:hi]:
        print('%9.1f %8.1f %8.1f  %-8s %s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - prev_end) / 1e3,
                                             '/'.join(str(x) for x in r[3:]), short(r[0])))
        prev_end = max(prev_end, r[2])
    print('period %.1f us, %d kernels, busy %.1f us' % ((rows[hi][1] - t0) / 1e3, hi - lo, sum(r[2] - r[1] for r in rows[lo:hi]) / 1e3))


if __name__ == '__main__':
    main(*sys.argv[1:4])
