"""CPU, world_size 2 / 4 / 8 over gloo: the data-parallel path of SURVEY.md 8(e) -- contiguous env shards,
rank-local advantage normalisation, ONE all-reduce(sum) of the flat gradient per minibatch step,
/ sum of rank weights BEFORE the global-norm clip, replicated Adam, broadcast of the initial state
(common/mpi_adam_optimizer.py:18-51, common/mpi_util.py:15-26, ppo2/model.py:97-131).

The communicator under test is the product's `baselines_amd.common.dist.Comm` (RCCL on the GPU box,
gloo here); the arithmetic around it is the oracle's torch-CPU restatement of the TF graph."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NENV = 16                    # whole-job environments: 2 per rank at world 8
KW = dict(network='mlp', ob_shape=(11,), ob_dtype=np.float32, pd_kind='gaussian', nact=3, value_network='copy',
          ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _data(N, T, seed):
    rng = np.random.RandomState(seed)
    return dict(obs=rng.randn(N * T, 11).astype(np.float32), returns=rng.randn(N * T).astype(np.float32),
                actions=rng.randn(N * T, 3).astype(np.float32), values=rng.randn(N * T).astype(np.float32),
                neglogpacs=(3.0 + 0.1 * rng.randn(N * T)).astype(np.float32))


def _worker(rank, port, weights, out, WORLD):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    torch.set_num_threads(1)
    from baselines_amd.common.dist import Comm, default_comm, shard_envs
    from oracle.ppo2_torch import OracleModel
    comm = default_comm()
    assert isinstance(comm, Comm) and comm.Get_size() == WORLD and comm.Get_rank() == rank
    total_w = comm.total_weight(weights[rank])
    assert total_w == pytest.approx(sum(weights))
    # every rank draws the same init stream (Appendix D: set_global_seeds never applies the rank offset);
    # perturb every rank but the root on purpose and let sync_from_root repair it
    np.random.seed(0)
    om = OracleModel(**KW)
    if rank != 0:
        with torch.no_grad():
            for p in om.p.values():
                p.add_(float(rank))
    for p in om.p.values():
        comm.bcast_(p.data, 0)
    comm.check_synced(torch.from_numpy(om.flat_params()[:64].copy()).sum().reshape(1))
    om.allreduce = lambda flat: comm.allreduce_sum_(flat.clone())
    om.rank_weight, om.total_weight = float(weights[rank]), float(total_w)
    N, T = NENV, 4
    d = _data(N, T, 123)                                  # the WHOLE job's batch, env-major
    lo, hi = shard_envs(N, rank, WORLD)
    rows = np.arange(lo * T, hi * T)                      # this rank's envs (contiguous shard)
    stats = []
    for step in range(3):
        np.random.seed(step)
        idx = rows[np.random.permutation(len(rows))[:len(rows) // 2 * 2]]
        stats.append(om.train(3e-4, 0.2, d['obs'][idx], d['returns'][idx], None, d['actions'][idx], d['values'][idx],
                              d['neglogpacs'][idx]))
    comm.check_synced(torch.from_numpy(om.flat_params().copy()).sum().reshape(1))
    out[rank] = (om.flat_params(), np.asarray(stats))
    dist.barrier()
    dist.destroy_process_group()


def _emulate(weights, WORLD):
    """single-process statement of the same algebra: g = sum_r w_r g_r / sum_r w_r -> clip -> Adam on every replica"""
    from baselines_amd.common.dist import shard_envs
    from oracle.ppo2_torch import OracleModel
    np.random.seed(0)
    reps = [OracleModel(**KW) for _ in range(WORLD)]
    for r in reps[1:]:
        for k in r.p:
            r.p[k].data.copy_(reps[0].p[k].data)
    N, T = NENV, 4
    d = _data(N, T, 123)
    for step in range(3):
        flats = []
        for rank, om in enumerate(reps):
            lo, hi = shard_envs(N, rank, WORLD)
            rows = np.arange(lo * T, hi * T)
            np.random.seed(step)
            idx = rows[np.random.permutation(len(rows))[:len(rows) // 2 * 2]]
            _, flat = om.compute_grads(0.2, d['obs'][idx], d['returns'][idx], d['actions'][idx], d['values'][idx],
                                       d['neglogpacs'][idx])
            flats.append(flat * float(weights[rank]))
        tot = flats[0]
        for fl in flats[1:]:
            tot = tot + fl
        avg = tot / float(sum(weights))
        for om in reps:
            om.apply_flat_grad(3e-4, avg.clone())
    return reps[0].flat_params()


@pytest.mark.parametrize('world,unequal', [(2, False), (2, True), (4, False), (4, True), (8, False), (8, True)])
def test_gradient_allreduce_matches_reference_algebra(world, unequal):
    """shard cover, rank weights (equal and unequal), average BEFORE the clip, sync_from_root and check_synced at every
    world size the 8-GPU node can run (mpi_adam_optimizer.py:21,39-43,53-68; mpi_util.py:15-26)."""
    weights = tuple((1, 3, 2, 5, 1, 4, 2, 7)[:world]) if unequal else (1,) * world
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(port, weights, out, world), nprocs=world, join=True)
    p0, s0 = out[0]
    for r in range(1, world):
        np.testing.assert_array_equal(p0, out[r][0])      # replicas stay bit-identical
        assert not np.array_equal(s0, out[r][1])          # loss statistics are rank-local (not reduced)
    ref = _emulate(weights, world)
    # gloo's ring adds the ranks' contributions in its own order: one fp32 rounding per addend on a gradient of O(1)
    np.testing.assert_allclose(p0, ref, rtol=0, atol=1e-7 * max(1, world // 2))


def test_shard_envs_contiguous_cover():
    from baselines_amd.common.dist import shard_envs
    for n, w in [(4096, 8), (4096, 4), (4096, 2), (256, 2), (7, 3), (5, 8), (16, 8)]:
        spans = [shard_envs(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def _bringup_worker(rank, port, failing_rank, out, WORLD):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    from baselines_amd import _lib
    from baselines_amd.common.dist import Comm
    entered = []

    class FakeLib(object):                # stands in for libmrl.so: an id can be drawn, creation would block forever alone
        def mrl_comm_unique_id(self, buf):
            return 0

        def mrl_comm_create(self, *a):
            entered.append(rank)
            return 0

        def mrl_comm_destroy(self, h):
            return 0

    def load():
        if rank == failing_rank:
            raise OSError('libmrl.so: cannot open shared object file')
        return FakeLib()
    _lib.load = load
    comm = Comm()
    ok = comm.enable_native()
    out[rank] = (ok, list(entered), comm.native is None, comm.native_error)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,failing_rank', [(2, 0), (2, 1), (4, 2), (8, 0), (8, 5)])
def test_enable_native_agrees_before_the_collective_communicator_creation(world, failing_rank):
    """ncclCommInitRank waits for every rank, so a rank whose library failed to load must keep ALL ranks out of
    mrl_comm_create (not just itself) -- everybody falls back to the torch.distributed collectives, nobody blocks."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bringup_worker, args=(port, failing_rank, out, world), nprocs=world, join=True)
    for rank in range(world):
        ok, entered, no_handle, err = out[rank]
        assert ok is False and entered == [] and no_handle and err
