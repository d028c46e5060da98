cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -2
python scripts/prof_labels.py 4096 2>&1 | tail -1
