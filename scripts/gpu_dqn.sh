#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-gpu}
cd $R; mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_dqn.py tests/test_gpu_replay.py tests/test_gpu_microbatch.py -m gpu -x -q --durations=5 > $O/${TAG}_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -12 $O/${TAG}_pytest.log
t0=$(date +%s)
timeout 900 python scripts/bench_replay.py > $O/${TAG}_replay.json 2> $O/${TAG}_replay.err
echo "replay rc=$? ($(( $(date +%s) - t0 )) s)"; tail -3 $O/${TAG}_replay.err
python - <<PY
import json
d = json.loads(open('$O/${TAG}_replay.json').read().strip().splitlines()[-1])
print({k: v for k, v in d.items() if not k.startswith('batch_') and 'kernels' not in k})
for b in ('batch_32', 'batch_4096'):
    print(b, {k: (v['us'] if isinstance(v, dict) else v) for k, v in d[b].items()})
print({k: (v['us'], v['calls_per_step']) for k, v in d['dqn_learner_step_batch32_kernels_us'].items()})
PY
