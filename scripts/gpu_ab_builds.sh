#!/bin/bash
# same-box A/B of two BUILDS (baselines_amd/csrc/libmrl_base.so = the previous build, libmrl.so = the current one): per-kernel times
# of one epoch at the bench shape, A B A B, grad + stats SHA-1 per run (equal digests <=> bit-identical results)
# usage (through gpurun): bash scripts/gpu_ab_builds.sh <tag> [ab_options arguments]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-ab}; shift
cd $R; mkdir -p $O
for v in base new base new; do
    lib=$R/baselines_amd/csrc/libmrl.so; [ $v = base ] && lib=$R/baselines_amd/csrc/libmrl_base.so
    echo "== $v" >> $O/${TAG}_ab_builds.txt
    MRL_LIB_PATH=$lib python scripts/ab_options.py 4096 "$@" 2>&1 | grep -v amdgpu.ids | head -${AB_LINES:-2} >> $O/${TAG}_ab_builds.txt
done
cut -c1-260 $O/${TAG}_ab_builds.txt
