#!/bin/bash
# A/B of the fused MLP step kernel's launch shapes on one box (MuJoCo-shaped config 2, 40 timed updates each)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for s in 0 1 0 1; do
  MRL_MLP_SLICE=$s timeout 600 python bench.py --workload mujoco --no-cpu-baseline --no-other-configs --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slice $s', round(d['value']), round(d['ms_per_step'],2), d['kernel_ms_per_step'])"
done
