#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r03i}
cd $R; mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_lstm.py -m gpu -x -q --durations=5 > $O/${TAG}_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -12 $O/${TAG}_pytest.log
t0=$(date +%s)
timeout 900 python bench.py --workload atari_lstm --num-envs 256 --no-cpu-baseline --no-other-configs > $O/${TAG}_bench_lstm.json 2> $O/${TAG}_bench_lstm.err
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"; tail -3 $O/${TAG}_bench_lstm.err
python - <<PY
import json
d = json.loads(open('$O/${TAG}_bench_lstm.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step']); print(d.get('kernel_ms_per_step'))
PY
