#!/bin/bash
# PMC passes over one PPO2 epoch at the bench shape (run on the GPU box via gpurun; counters in their own passes,
# kernel trace only -- MI355X_MICROARCH.md "rocprofv3 PMC slots").  Results: gpurun_out/pmc_<tag>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
N=${1:-4096}
mkdir -p $O
export TMPDIR=/tmp; cd /tmp
run() {  # tag, counters...
    tag=$1; shift
    rm -rf $O/pmc_$tag
    timeout 900 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$tag -o e -- python $R/scripts/one_epoch.py $N > /dev/null 2> $O/pmc_$tag.err
    python $R/scripts/rocpd_pmc.py $(ls $O/pmc_$tag/*.db | head -1) > $O/pmc_$tag.txt 2>> $O/pmc_$tag.err
    rm -rf $O/pmc_$tag
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS
ls -la $O/pmc_*.txt
