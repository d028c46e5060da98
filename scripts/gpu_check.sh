#!/bin/bash
# Run on the GPU box (via gpurun): parity tests, bench line, rocprof kernel stats.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu.txt
nproc >> $O/gpu.txt; lscpu | grep "Model name" >> $O/gpu.txt; free -g | head -2 >> $O/gpu.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cat $O/bench.json
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/prof_bench.json 2> $O/prof.err; echo "rocprof rc=$?"
ls -R $O/prof | head
