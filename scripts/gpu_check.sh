#!/bin/bash
# One GPU-box session: parity suite, smoke, headline bench (+ other configs), optional rocprof of the bench command.
# usage (through gpurun): bash scripts/gpu_check.sh [tag] [pytest-args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r02a}; shift
mkdir -p $O
cd $R
(rocm-smi --showproductname 2>/dev/null | head -12; lscpu | head -20; free -g | head -2) > $O/${TAG}_box.txt 2>&1
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q "$@" > $O/${TAG}_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -5 $O/${TAG}_pytest.log
t0=$(date +%s)
timeout 300 python __graft_entry__.py smoke > $O/${TAG}_smoke.log 2>&1
echo "smoke rc=$? ($(( $(date +%s) - t0 )) s)"; tail -2 $O/${TAG}_smoke.log
t0=$(date +%s)
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"; tail -3 $O/${TAG}_bench.err
python - <<PY
import json
try:
    d = json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms/step', d['ms_per_step'], 'roof', d['roofline']['kernel'], d['roofline']['frac'])
    print(d.get('kernel_ms_per_step'))
    for o in d.get('other_configs', []):
        print(o.get('workload'), o.get('value'), o.get('error'))
    print(d.get('cpu_baseline'))
except Exception as e:
    print('bench parse failed', e)
PY
