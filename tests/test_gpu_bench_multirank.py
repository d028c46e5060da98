"""GPU: `bench.py --gpus 2` end to end on the one GPU of the test box.

The driver measures the multi-GPU scaling curve by launching bench.py through torch.distributed.run with one rank per GPU
over RCCL.  RCCL refuses two ranks on one device, so here the same N > 1 code path -- self-relaunch through
torch.distributed.run, env sharding (common/dist.shard_envs), sync_from_root, one all-reduce(sum) of the flat gradient per
minibatch step, / total weight -> clip -> Adam replicated, max-over-ranks timing, rank 0 printing ONE JSON line -- runs
with two processes sharing the device and `MRL_BENCH_BACKEND=gloo` carrying the collectives
(common/mpi_adam_optimizer.py:18-51, common/mpi_util.py:15-26).  Not a measurement; a does-it-run-and-agree check."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('num_envs', [64, 1024])
def test_bench_two_ranks_one_json_line_and_synced_parameters(num_envs):
    """num_envs=1024: 512 envs per rank -> 16384-sample minibatches, i.e. every rank runs the engines of the benched shapes
    (tiled split engines need B >= 1024)"""
    env = dict(os.environ, MRL_BENCH_BACKEND='gloo', MRL_BENCH_SMI='0', HSA_ENABLE_IPC_MODE_LEGACY='0', MRL_BENCH_DP_STRICT='1')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--num-envs', str(num_envs),
           '--no-cpu-baseline', '--no-other-configs']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['world_size_observed'] == 2
    assert d['config']['envs_per_gpu'] == num_envs // 2 and d['scaling'] == 'strong'
    assert d['metric'].startswith('env-steps/sec') and d['value'] > 0
    assert all(abs(x) < 1e6 and x == x for x in d['loss'])                   # finite
    assert d['params_synced_across_ranks'] is True
    assert 'gloo' in d['config']['collective']
    # the data-parallel self-validation of the first multi-GPU run (bench.dp_verify), rehearsed over gloo: the all-reduced
    # local gradients against rank 0 recomputing every rank's gathered minibatch on its own device
    v = d['dp_verify']
    assert v['ok'] is True and v['ranks'] == 2 and v['minibatch'] == d['config']['nbatch_train_per_gpu']
    assert v['max_abs_diff_over_scale'] <= 1e-6 and v['recompute_vs_plain_max_abs_diff_over_scale'] <= 1e-6
    assert v['rank0_gathered_rows_reproduce_its_indexed_gradient_bitwise'] is True
    assert v['overlapped_equals_plain'] is None and 'not exercised' in v['note']      # RCCL refuses two ranks on one device


def test_bench_two_ranks_over_the_in_library_rccl_path_when_two_gpus_are_visible():
    """The native path itself -- mrl_comm over RCCL, the two-slice all-reduce issued from inside the backward pass -- with 2
    ranks on 2 GPUs: skipped on the one-GPU test box, runs wherever the suite meets a multi-GPU node.  bench.py's dp_verify
    compares the overlapped gradient with the plain all-reduce and with a single-device recomputation (1e-6 of its scale)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 HIP devices (RCCL refuses two ranks on one device)')
    env = dict(os.environ, MRL_BENCH_SMI='0', HSA_ENABLE_IPC_MODE_LEGACY='0', MRL_BENCH_DP_STRICT='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MRL_BENCH_BACKEND', 'MRL_NATIVE_COMM'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1', '--num-envs', '1024',
           '--no-cpu-baseline', '--no-other-configs']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['params_synced_across_ranks'] is True
    assert d['config']['native_dp'] is True and 'in-library RCCL' in d['config']['collective']
    v = d['dp_verify']
    assert v['ok'] is True and v['overlapped_equals_plain'] is True and v['max_abs_diff_over_scale'] <= 1e-6
    assert v['overlapped_bit_identical_to_plain'] is True          # two addends: the sum does not depend on the ring's order
