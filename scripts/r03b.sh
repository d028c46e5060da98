#!/bin/bash
# round 3, GPU call 2: image-resident transpose-read conv weight gradients (correctness + A/B timing), panel-group tile order
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r03b}
cd $R; mkdir -p $O
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "engine_options" > $O/${TAG}_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -12 $O/${TAG}_pytest.log
t0=$(date +%s)
timeout 900 python scripts/ab_options.py 4096 wgrad_tr=0,1,2 x6_pg=2,4,8 > $O/${TAG}_ab.log 2>&1
echo "ab rc=$? ($(( $(date +%s) - t0 )) s)"; cat $O/${TAG}_ab.log | tail -12
