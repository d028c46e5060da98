cd ${GRAFT_REPO_ROOT:-/root/repo}
for w in 4 8 4 8 4 8; do
  MRL_MLP_WAVES=$w timeout 600 python bench.py --workload mujoco --no-cpu-baseline --no-other-configs --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves $w', round(d['value']), round(d['ms_per_step'],2), d['kernel_ms_per_step'])"
done
