#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-gpu}
cd $R; mkdir -p $O
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "engine_options" > $O/${TAG}_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -4 $O/${TAG}_pytest.log
timeout 900 python scripts/ab_options.py 4096 $2 > $O/${TAG}_ab.log 2>&1
echo "ab rc=$?"; cat $O/${TAG}_ab.log | tail -12
