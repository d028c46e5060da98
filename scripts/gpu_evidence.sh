#!/bin/bash
# round-end evidence of the CURRENT kernel sources on one box, without the parity suite (scripts/gpu_subset.sh / gpu_check.sh run that):
# rocprofv3 kernel trace of the bench command, PMC passes over one epoch -> profiles-shaped JSON with the source hash, then the
# driver-shaped `python bench.py` (which attaches the counter traffic because the JSON now matches the running sources).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r06z}
cd $R; mkdir -p $O
(rocm-smi --showproductname 2>/dev/null | head -12; lscpu | head -20; free -g | head -2) > $O/${TAG}_box.txt 2>&1
export TMPDIR=/tmp; cd /tmp
rm -rf $O/prof_$TAG
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof --no-other-configs > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof.err
DB=$(ls $O/prof_$TAG/*.db $O/prof_$TAG/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/scripts/rocpd_stats.py $DB > $O/${TAG}_bench_atari4096_kernel_stats.txt 2>> $O/${TAG}_prof.err
# the timed update alone (from its GAE launch on): per-kernel averages = the minibatch-shaped launches
[ -n "$DB" ] && python $R/scripts/rocpd_stats.py $DB --window gae_ > $O/${TAG}_bench_atari4096_kernel_stats_timed_window.txt 2>> $O/${TAG}_prof.err
rm -rf $O/prof_$TAG
head -8 $O/${TAG}_bench_atari4096_kernel_stats.txt | cut -c1-150
cd $R; bash scripts/pmc_epoch.sh 4096 > $O/${TAG}_pmc.log 2>&1; tail -3 $O/${TAG}_pmc.log
python scripts/pmc_to_json.py $O/pmc_fetch.txt $O/pmc_write.txt $O/${TAG}_pmc_hbm.json && cp $O/${TAG}_pmc_hbm.json profiles/${TAG}_pmc_hbm.json
for t in fetch write sq; do cp $O/pmc_$t.txt $O/${TAG}_pmc_$t.txt; done
t0=$(date +%s)
timeout 600 python bench.py > $O/${TAG}_bench_atari4096.json 2> $O/${TAG}_bench.err
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"; tail -2 $O/${TAG}_bench.err
python - <<PY
import json
d = json.loads(open('$O/${TAG}_bench_atari4096.json').read().strip().splitlines()[-1])
r = d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roof', r['kernel'], r['frac'], 'traffic', r.get('traffic_over_algorithmic'), r.get('traffic_source'), r.get('all_gemm_sites_traffic_over_algorithmic'))
print(d.get('device_state'), d.get('self_check'))
for o in d.get('other_configs', []):
    print(o.get('workload'), o.get('value'), o.get('error'))
print(d.get('cpu_baseline'))
PY
