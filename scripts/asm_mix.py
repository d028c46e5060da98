#!/usr/bin/env python3
"""Instruction mix of the MFMA main loop of selected kernels in a gfx950 assembly listing
(hipcc --cuda-device-only -S).  usage: asm_mix.py file.s substr [substr2 ...]"""
import re
import sys
from collections import Counter


def main(path, subs):
    txt = open(path).read()
    funcs = re.split(r'\n(?=_Z\S+:)', txt)
    for f in funcs:
        head = f.split('\n')[0]
        if not all(s in head for s in subs):
            continue
        lines = f.split('\n')
        idx = [i for i, l in enumerate(lines) if 'v_mfma' in l]
        if not idx:
            continue
        lo, hi = idx[0], idx[-1]
        while lo > 0 and not re.match(r'\.LBB\d+_\d+:', lines[lo]):
            lo -= 1
        while hi < len(lines) - 1 and 's_cbranch' not in lines[hi]:
            hi += 1
        c = Counter()
        for l in lines[lo:hi + 1]:
            l = l.strip()
            if not l or l.startswith(';') or l.startswith('.'):
                continue
            op = l.split()[0]
            if op.startswith('v_mfma'):
                c['mfma'] += 1
            elif op.startswith('v_'):
                c['valu'] += 1
            elif op.startswith('s_waitcnt'):
                c['waitcnt'] += 1
            elif op.startswith('s_'):
                c['salu'] += 1
            elif op.startswith('ds_'):
                c['ds'] += 1
            elif op.startswith(('global_', 'buffer_', 'flat_')):
                c['vmem'] += 1
            else:
                c[op] += 1
        print(head[:100], dict(c))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:])
