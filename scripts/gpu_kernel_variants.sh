R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
for v in ${VARIANTS:-0}; do
rm -rf $O/pk; env $VAR=$v timeout 600 rocprofv3 --kernel-trace --stats -d $O/pk -o e -- python $R/scripts/one_epoch.py 4096 > /dev/null 2> $O/pk.err
echo "== $VAR=$v"; python $R/scripts/rocpd_stats.py $(ls $O/pk/*.db | head -1) 2>/dev/null | grep -i -E "${PAT:-c1wgrad}" | cut -c1-75,110-170
done
rm -rf $O/pk
