cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_kernels.py tests/test_gpu_microbatch.py "tests/test_gpu_ppo2.py" -m gpu -x -q --durations=8 > gpurun_out/r02c_pytest.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/r02c_pytest.log
timeout 300 python scripts/dgx6_experiments.py 4096 > gpurun_out/r02c_dgx6.txt 2>&1; cat gpurun_out/r02c_dgx6.txt | tail -9
timeout 300 python bench.py --workload mujoco --steps 5 --no-cpu-baseline > gpurun_out/r02c_mujoco.json 2> gpurun_out/r02c_mujoco.err; python -c "
import json; d=json.loads(open('gpurun_out/r02c_mujoco.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
