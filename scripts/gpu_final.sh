# full round-end check on one box: parity suite, smoke, bench, rocprof kernel trace of the bench command, PMC passes
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r02z}
cd $R; mkdir -p $O
bash scripts/gpu_check.sh $TAG "${@:2}"      # further arguments go to pytest (e.g. -k "not full_size")
export TMPDIR=/tmp; cd /tmp
rm -rf $O/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof --no-other-configs > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof.err
DB=$(ls $O/prof_$TAG/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/scripts/rocpd_stats.py $DB > $O/${TAG}_kernel_stats.txt 2>> $O/${TAG}_prof.err
rm -rf $O/prof_$TAG
head -14 $O/${TAG}_kernel_stats.txt
cd $R; bash scripts/pmc_epoch.sh 4096 > $O/${TAG}_pmc.log 2>&1; tail -5 $O/${TAG}_pmc.log
python scripts/pmc_to_json.py $O/pmc_fetch.txt $O/pmc_write.txt $O/${TAG}_pmc_hbm.json
