#!/bin/bash
# act_planes bring-up: engine-agreement + large-batch oracle parity, then A/B timing of the plane modes on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r02m}
mkdir -p $O; cd $R
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_kernels.py::test_engine_options_agree -m gpu -x -q --durations=8 > $O/${TAG}_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -25 $O/${TAG}_pytest.log
t0=$(date +%s)
timeout 600 python scripts/ab_options.py 4096 act_planes=${MODES:-4,8,16,12,28} > $O/${TAG}_ab.log 2>&1
echo "ab rc=$? ($(( $(date +%s) - t0 )) s)"; cat $O/${TAG}_ab.log | tail -12
