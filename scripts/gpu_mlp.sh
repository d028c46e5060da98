#!/bin/bash
# MLP (config 2) iteration: parity tests that exercise the fused step kernel, then the MuJoCo-shaped bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-gpu}
cd $R; mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ppo2.py tests/test_gpu_probtypes.py tests/test_gpu_microbatch.py -m gpu -x -q -k "mlp or cartpole or mujoco or engine or epoch or probtype" > $O/${TAG}_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -6 $O/${TAG}_pytest.log
timeout 600 python bench.py --workload mujoco --no-cpu-baseline --no-other-configs --steps 5 > $O/${TAG}_bench_mujoco.json 2> $O/${TAG}_bench_mujoco.err
echo "bench rc=$?"; tail -2 $O/${TAG}_bench_mujoco.err
python - <<PY
import json
d = json.loads(open('$O/${TAG}_bench_mujoco.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step']); print(d.get('kernel_ms_per_step'))
PY
