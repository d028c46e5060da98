#!/bin/bash
# repeated MuJoCo-shaped (config 2) bench lines on one box: 40 timed updates each
cd ${GRAFT_REPO_ROOT:-/root/repo}
for s in 1 2 3; do
  timeout 600 python bench.py --workload mujoco --no-cpu-baseline --no-other-configs --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $s', round(d['value']), round(d['ms_per_step'],2), d['kernel_ms_per_step'])"
done
