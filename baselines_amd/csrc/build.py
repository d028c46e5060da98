"""Build libmrl.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m baselines_amd.csrc.build [--force]

The shared library is git-ignored but travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['rollout.hip', 'model.hip', 'envs.hip', 'replay.hip', 'wrappers.hip', 'comm.hip']
HEADERS = ['common.hip.h', 'gemm.hip.h', 'wres.hip.h', 'imgres.hip.h', 'ldsdgrad.hip.h', 'mlpstep.hip.h', 'mlpact.hip.h', 'powcr.hip.h', 'gemmx6.hip.h', 'dgradx6.hip.h', 'gemmx6s.hip.h', 'gemmx6r.hip.h', 'convx6c.hip.h', 'c1fwd.hip.h', 'wgradx8.hip.h', 'wgradtr.hip.h', 'c1wgrad.hip.h', 'comm.hip.h', 'lstm.hip.h', 'qnet.hip.h', 'qheads.hip.h', 'convskinny.hip.h', 'planes.hip.h',
           os.path.join('..', '..', 'include', 'mrl.h')]
LIB = os.path.join(HERE, 'libmrl.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall',
         '-Wno-unused-function', '-Wno-unused-variable']
# MRL_BUILD_DEFINES="-DMRL_X6_EXPERIMENTS": extra kernel instantiations for the timing experiments (scripts/ab_options.py)
FLAGS += os.environ.get('MRL_BUILD_DEFINES', '').split()


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def build(force=False, verbose=True):
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    hdr_t = max(_mtime(os.path.join(HERE, h)) for h in HEADERS)
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(HERE, s)
        obj = os.path.join(HERE, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or _mtime(obj) < max(_mtime(src), hdr_t):
            jobs.append([HIPCC] + FLAGS + ['-x', 'hip', '-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _mtime(LIB) < max([_mtime(o) for o in objs] + [0.0]):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
