#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-gpu}
cd $R; mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest $2 -m gpu -x -q --durations=5 > $O/${TAG}_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -15 $O/${TAG}_pytest.log
