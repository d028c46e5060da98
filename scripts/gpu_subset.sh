R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; mkdir -p $O
t0=$(date +%s)
timeout 700 python -m pytest tests -m gpu -q --durations=4 \
  --deselect tests/test_gpu_large_batch.py::test_full_size_minibatch_backward_vs_oracle_every_entry \
  --deselect tests/test_gpu_benched_shapes.py::test_config3_whole_update_vs_oracle_step_by_step > $O/r05zz_gpu_pytest.txt 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -8 $O/r05zz_gpu_pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python scripts/ab_options.py 4096 heads_wave=1,2 2>&1 | grep -v amdgpu | cut -c1-60,200-262 > $O/r05zz_ab_heads.txt; cat $O/r05zz_ab_heads.txt
