cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dqn.py tests/test_gpu_kernels.py tests/test_gpu_replay.py -m gpu -x -q --durations=8 > gpurun_out/r02d_pytest.log 2>&1
echo "pytest rc=$?"; tail -40 gpurun_out/r02d_pytest.log
