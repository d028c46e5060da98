#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r03f}
cd $R; mkdir -p $O
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q -k "not full_size_minibatch_backward" --durations=6 > $O/${TAG}_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -12 $O/${TAG}_pytest.log
