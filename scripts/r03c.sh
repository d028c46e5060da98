#!/bin/bash
# round 3, GPU call 3: phase experiments of the transpose-read weight gradient + SQ counter pass over one epoch
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r03c}
cd $R; mkdir -p $O
t0=$(date +%s)
timeout 900 python scripts/ab_options.py 4096 wgrad_tr=2,3,4 > $O/${TAG}_ab.log 2>&1
echo "ab rc=$? ($(( $(date +%s) - t0 )) s)"; cat $O/${TAG}_ab.log | tail -6
export TMPDIR=/tmp; cd /tmp
t0=$(date +%s)
rm -rf $O/pmc_sq
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS -d $O/pmc_sq -o e -- python $R/scripts/one_epoch.py 4096 > /dev/null 2> $O/${TAG}_pmc_sq.err
python $R/scripts/rocpd_pmc.py $(ls $O/pmc_sq/*.db | head -1) > $O/${TAG}_pmc_sq.txt 2>> $O/${TAG}_pmc_sq.err
rm -rf $O/pmc_sq
echo "pmc sq rc=$? ($(( $(date +%s) - t0 )) s)"; grep -c dispatches $O/${TAG}_pmc_sq.txt
t0=$(date +%s)
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d $O/pmc_lds -o e -- python $R/scripts/one_epoch.py 4096 > /dev/null 2> $O/${TAG}_pmc_lds.err
python $R/scripts/rocpd_pmc.py $(ls $O/pmc_lds/*.db | head -1) > $O/${TAG}_pmc_lds.txt 2>> $O/${TAG}_pmc_lds.err
rm -rf $O/pmc_lds
echo "pmc lds rc=$? ($(( $(date +%s) - t0 )) s)"; tail -3 $O/${TAG}_pmc_lds.err
