#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the kernels matching PMC_PAT over one epoch, under the environment given on the command line
#   usage: PMC_PAT=wgrad_tr_dense bash scripts/pmc_one.sh <tag> [N]     (e.g. MRL_WGRAD_XCD=0 PMC_PAT=... bash scripts/pmc_one.sh xcd0)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=$1; N=${2:-4096}
mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc1_$TAG
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc1_$TAG -o e -- python $R/scripts/one_epoch.py $N > /dev/null 2> $O/pmc1_$TAG.err
    python $R/scripts/rocpd_pmc.py $(ls $O/pmc1_$TAG/*.db | head -1) "$PMC_PAT" | grep -v top8 >> $O/pmc1_${TAG}.txt
    rm -rf $O/pmc1_$TAG
done
echo "== $TAG"; cut -c1-120 $O/pmc1_${TAG}.txt
