#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
run() { tag=$1; shift; rm -rf $O/pmc_$tag; timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$tag -o e -- python $R/scripts/one_epoch.py 4096 > /dev/null 2> $O/pmc_$tag.err; python $R/scripts/rocpd_pmc.py $(ls $O/pmc_$tag/*.db $O/pmc_$tag/*/*.db 2>/dev/null | head -1) > $O/pmc_$tag.txt 2>> $O/pmc_$tag.err; rm -rf $O/pmc_$tag; }
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS
grep -A16 "${PMC_PAT:-c1fwd3\|c1wgrad_half}" $O/pmc_sq.txt | grep -v top8 | head -40
grep -A16 "${PMC_PAT:-c1fwd3\|c1wgrad_half}" $O/pmc_sq2.txt | grep -v top8 | head -40
tail -3 $O/pmc_sq2.err
