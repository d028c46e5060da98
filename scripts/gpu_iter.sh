#!/bin/bash
# Short GPU iteration: selected parity tests + a bench run + (optionally) the rocprofv3 kernel trace of the bench command.
# usage (through gpurun): bash scripts/gpu_iter.sh <tag> "<pytest selection>" [prof]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=$1; SEL=$2; PROF=$3
mkdir -p $O
cd $R
t0=$(date +%s)
timeout 1200 python -m pytest $SEL -m gpu -x -q --durations=12 > $O/${TAG}_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; grep -E "passed|failed|error" $O/${TAG}_pytest.log | tail -3; grep -E "^(FAILED|ERROR)|Error|assert" $O/${TAG}_pytest.log | head -20
t0=$(date +%s)
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"; tail -3 $O/${TAG}_bench.err
python - <<PY
import json
try:
    d = json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms/step', d['ms_per_step'], 'roof', d['roofline']['kernel'], d['roofline']['frac'])
    print(d.get('kernel_ms_per_step'))
    for k, v in d.get('kernel_rooflines', {}).items():
        print('  %-12s %8.3f ms  %8.1f %s  frac %.3f' % (k, v['avg_ms'], v['achieved'], v['unit'], v['frac']))
    for o in d.get('other_configs', []):
        print(o.get('workload'), o.get('value'), o.get('error'), o.get('kernel_ms_per_step'))
except Exception as e:
    print('bench parse failed', e)
PY
if [ -n "$PROF" ]; then
    export TMPDIR=/tmp; cd /tmp
    rm -rf $O/prof_$TAG
    timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof --no-other-configs > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof.err
    DB=$(ls $O/prof_$TAG/*.db 2>/dev/null | head -1)
    if [ -n "$DB" ]; then python $R/scripts/rocpd_stats.py $DB > $O/${TAG}_kernel_stats.txt 2>> $O/${TAG}_prof.err; else ls -R $O/prof_$TAG | head; fi
    head -30 $O/${TAG}_kernel_stats.txt
    rm -rf $O/prof_$TAG
fi
