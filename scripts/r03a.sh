#!/bin/bash
# round 3, first GPU call: new parity tests (full-size backward, 2-rank bench), GAE rewrite, tr-read probe, bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r03a}
cd $R; mkdir -p $O
t0=$(date +%s)
scripts/tr_probe.bin > $O/${TAG}_tr_probe.txt 2>&1; echo "probe rc=$? ($(( $(date +%s) - t0 )) s)"
grep -E "model A|cycles per" $O/${TAG}_tr_probe.txt | head -60
t0=$(date +%s)
timeout 1500 python -m pytest tests/test_gpu_large_batch.py tests/test_gpu_bench_multirank.py tests/test_gpu_kernels.py tests/test_gpu_ppo2.py -m gpu -x -q --durations=8 > $O/${TAG}_pytest.log 2>&1
echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -15 $O/${TAG}_pytest.log
t0=$(date +%s)
timeout 900 python bench.py --no-other-configs > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"; tail -3 $O/${TAG}_bench.err
python - <<PY
import json
try:
    d = json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms/step', d['ms_per_step'], 'roof', d['roofline']['kernel'], d['roofline']['bound'], d['roofline']['frac'])
    print('self_check', d.get('self_check'))
    print(d.get('kernel_ms_per_step'))
    for k, v in d.get('kernel_rooflines', {}).items():
        print('  %-12s %8.3f ms  %-5s %8.1f %s  frac %.3f  (mfma %.3f hbm %.3f)' % (k, v['avg_ms'], v['bound'], v['achieved'], v['unit'], v['frac'], v.get('mfma_frac', 0), v.get('hbm_frac', 0)))
except Exception as e:
    print('bench parse failed', e)
PY
