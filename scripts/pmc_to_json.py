#!/usr/bin/env python
"""profiles/<tag>_pmc_hbm.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_epoch.sh (one epoch at the bench
shape, B = 131072 per launch).  Corrections per MI355X_MICROARCH.md "HBM": the counters are in KB; on gfx950 FETCH_SIZE
tallies the 128-byte requests of wide coalesced reads at 64 bytes -> doubled; WRITE_SIZE taken as is (checked here:
c2.dgrad writes exactly its 6.71 GB).  Only kernels whose every dispatch in the run has the minibatch shape are listed
(the backward kernels: 4 dispatches); forward kernels also serve the act side with other batch sizes: for them the
largest dispatches of the run (the minibatch launches) are averaged.
usage: pmc_to_json.py fetch.txt write.txt out.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_sha16      # noqa: E402  (identity of the kernel sources the counters were collected from)

LABELS = {
    'c2.dgrad': ('dgrad_x6p_kernel<20, 20, 32', 'dgrad_x6_kernel<20, 20, 32'),      # round 6: dgrad_x6p_kernel (dgradx6.hip.h)
    'c3.dgrad': ('dgrad_x6p_kernel<9, 9, 64', 'dgrad_x6_kernel<9, 9, 64'),
    'c1.wgrad': 'c1wgrad_half_kernel',
    'c2.wgrad': 'wgrad_tr_kernel<20, 20, 32',
    'c3.wgrad': 'wgrad_tr_kernel<9, 9, 64',
    'fc1.wgrad': 'wgrad_tr_dense_kernel',
    'fc1.dgrad': 'gemm_x6_kernel<mrl::X6DenseA, mrl::TrMaskRelu',      # transposed-accumulator epilogue (planes.hip.h)
}


# forward kernels also serve the act side (other batch sizes): their minibatch launches are the largest dispatches
# (".top8" lines of scripts/rocpd_pmc.py, launch order); (kernel substring, which of the alternating layers, of how many)
TOP = {
    'c1.fwd': ('c1fwd3_kernel', 0, 1),
    'c2.fwd': ('conv_x6c_kernel<20, 20, 32', 0, 1),          # round 5: class-resident conv forward (convx6c.hip.h)
    'c3.fwd': ('conv_x6c_kernel<9, 9, 64', 0, 1),
    'fc1.fwd': ('gemm_x6_kernel<mrl::X6DenseA, mrl::TrBiasRelu', 0, 1),
}


def parse_top(path, counter):
    out, name = {}, None
    for line in open(path):
        if not line.startswith(' '):
            name = line
        elif (counter + '.top8') in line and name:
            out[name] = [float(x) for x in line.split()[1:]]
    return out


def parse(path, counter):
    out, name = {}, None
    for line in open(path):
        if not line.startswith(' '):
            name = line
        elif counter in line and name:
            out[name] = float(line.split()[-1])
    return out


def main(fetch_txt, write_txt, out_json):
    f, w = parse(fetch_txt, 'FETCH_SIZE'), parse(write_txt, 'WRITE_SIZE')
    res = {}
    for label, subs in LABELS.items():
        fk = wk = []
        for sub in (subs if isinstance(subs, tuple) else (subs,)):       # first alternative that ran
            fk = [v for k, v in f.items() if sub in k]
            wk = [v for k, v in w.items() if sub in k]
            if fk and wk:
                break
        if fk and wk:
            res[label] = {'fetch_bytes': fk[0] * 1024 * 2, 'write_bytes': wk[0] * 1024,
                          'hbm_bytes': fk[0] * 1024 * 2 + wk[0] * 1024,
                          'raw': {'FETCH_SIZE_KB': fk[0], 'WRITE_SIZE_KB': wk[0]}}
    ft, wt = parse_top(fetch_txt, 'FETCH_SIZE'), parse_top(write_txt, 'WRITE_SIZE')
    for label, (sub, which, of) in TOP.items():
        fk = [v for k, v in ft.items() if sub in k]
        wk = [v for k, v in wt.items() if sub in k]
        if not (fk and wk):
            # eight dispatches or fewer: every launch of this kernel had the minibatch shape (round 6: the act side's fc1 runs the
            # split-K instantiation, a different kernel) -- the plain per-dispatch average
            fa = [v for k, v in f.items() if sub in k]
            wa = [v for k, v in w.items() if sub in k]
            if fa and wa:
                res[label] = {'fetch_bytes': fa[0] * 1024 * 2, 'write_bytes': wa[0] * 1024, 'hbm_bytes': fa[0] * 1024 * 2 + wa[0] * 1024,
                              'raw': {'FETCH_SIZE_KB': fa[0], 'WRITE_SIZE_KB': wa[0]}}
            continue
        if fk and wk:
            # the largest dispatches of the run are the minibatch launches; two layers sharing a kernel name alternate
            fv = [v for v in fk[0] if v >= 0.25 * max(fk[0])][which::of]
            wv = [v for v in wk[0] if v >= 0.25 * max(wk[0])][which::of]
            if fv and wv:
                f1, w1 = sum(fv) / len(fv), sum(wv) / len(wv)
                res[label] = {'fetch_bytes': f1 * 1024 * 2, 'write_bytes': w1 * 1024, 'hbm_bytes': f1 * 1024 * 2 + w1 * 1024,
                              'raw': {'FETCH_SIZE_KB': f1, 'WRITE_SIZE_KB': w1, 'dispatches': len(fv)}}
    json.dump({'note': __doc__.split('usage')[0].strip(), 'source_sha16': source_sha16(), 'per_launch': res}, open(out_json, 'w'), indent=1)
    for k, v in res.items():
        print('%-10s fetch %.2f GB  write %.2f GB' % (k, v['fetch_bytes'] / 1e9, v['write_bytes'] / 1e9))


if __name__ == '__main__':
    main(*sys.argv[1:4])
