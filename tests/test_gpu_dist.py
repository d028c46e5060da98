"""GPU: the data-parallel branch of the PRODUCT -- `Model.train_indexed` / `_apply_gradients`
under `self.multi`, `MicrobatchedModel._micro_steps`, and the in-library RCCL communicator.

  * two processes share the one GPU of the test box and talk over gloo (RCCL refuses two ranks on one device): each owns
    a contiguous env shard of the same whole-job batch, rank weights (1,1) and (1,3), different initial parameters that
    `sync_from_root` has to repair; the resulting parameters are bit-identical across the ranks and equal the
    single-process statement of the reference's algebra (tests/test_dist_gloo.py::_emulate: g = sum_r w_r g_r / sum_r w_r
    -> clip_by_global_norm -> Adam, common/mpi_adam_optimizer.py:18-51) evaluated by the oracle;
  * the RCCL path inside libmrl.so (mrl_comm_*, mrl_allreduce_grads, mrl_broadcast_state, mrl_model_attach_comm) runs
    with a one-rank communicator: the collective calls, the communication-stream / event choreography inside the
    backward pass and the rank-weight scaling are the code that runs on 8 GPUs, only the ring is trivial.
"""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.test_dist_gloo import KW, NENV, _data, _emulate, _free_port      # noqa: E402

WORLD = 2

pytestmark = pytest.mark.gpu


class _Spaces(object):
    """the two attributes build_policy reads from an env (common/policies.py:127)"""

    def __init__(self):
        from baselines_amd.common.spaces import Box
        self.observation_space = Box(-10.0, 10.0, (11,), np.float32)
        self.action_space = Box(-1.0, 1.0, (3,), np.float32)
        self.num_envs = 1


def _shard_rollout(d, lo, hi, T, device):
    """time-major device Rollout holding envs [lo, hi) of the env-major whole-job batch `d`"""
    from baselines_amd.ppo2.runner import Rollout
    n = hi - lo
    ro = Rollout(T, n, (11,), torch.float32, 'gaussian', 3, device)

    def tm(x):      # env-major flat rows of the shard -> [T, n, ...]
        x = x[lo * T:hi * T]
        return torch.from_numpy(np.ascontiguousarray(x.reshape((n, T) + x.shape[1:]).swapaxes(0, 1))).to(device)
    ro.obs.copy_(tm(d['obs']))
    ro.actions.copy_(tm(d['actions']))
    ro.values.copy_(tm(d['values']))
    ro.neglogpacs.copy_(tm(d['neglogpacs']))
    ro.returns = tm(d['returns']).contiguous()
    return ro


def _worker(rank, port, weights, micro, out):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    from baselines_amd.common.dist import default_comm, shard_envs
    from baselines_amd.common.policies import build_policy
    from baselines_amd.ppo2 import Model
    from baselines_amd.ppo2.microbatched_model import MicrobatchedModel
    comm = default_comm()
    assert comm.Get_size() == WORLD and comm.native is None          # gloo: collectives through torch.distributed
    N, T = NENV, 4
    lo, hi = shard_envs(N, rank, WORLD)
    nloc = (hi - lo) * T
    np.random.seed(rank)                       # ranks draw DIFFERENT initial weights; rank 0's must win (sync_from_root)
    policy = build_policy(_Spaces(), 'mlp', value_network='copy')
    kw = dict(policy=policy, ob_space=None, ac_space=None, nbatch_act=hi - lo, nbatch_train=nloc, nsteps=T,
              ent_coef=KW['ent_coef'], vf_coef=KW['vf_coef'], max_grad_norm=KW['max_grad_norm'],
              mpi_rank_weight=weights[rank], comm=comm)
    model = MicrobatchedModel(microbatch_size=nloc // 2, **kw) if micro else Model(**kw)
    assert model.multi and not model.native_dp and model.total_weight == pytest.approx(sum(weights))
    ro = _shard_rollout(_data(N, T, 123), lo, hi, T, model.device)
    stats = []
    for step in range(3):
        np.random.seed(step)
        idx = np.random.permutation(nloc)[:nloc // 2 * 2]          # local env-major indices == _emulate's rows - lo*T
        stats.append(model.train_indexed(3e-4, 0.2, ro, torch.from_numpy(idx).to(model.device)).cpu().numpy())
    comm.check_synced(model.params.sum().reshape(1))
    out[rank] = (model.get_flat_params(), np.asarray(stats))
    dist.barrier()
    dist.destroy_process_group()


def _run(weights, micro):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(port, weights, micro, out), nprocs=WORLD, join=True)
    return out[0], out[1]


@pytest.mark.parametrize('weights', [(1, 1), (1, 3)])
def test_two_product_models_on_one_gpu_match_reference_algebra(weights):
    (p0, s0), (p1, s1) = _run(weights, micro=False)
    np.testing.assert_array_equal(p0, p1)                 # replicas stay bit-identical (check_synced also ran inside)
    assert not np.array_equal(s0, s1)                     # loss statistics are rank-local, not reduced
    np.random.seed(0)
    ref = _emulate(weights, WORLD)                               # rank 0's initial weights == np.random.seed(0) stream
    np.testing.assert_allclose(p0, ref, rtol=0, atol=5e-6)


def test_two_microbatched_models_on_one_gpu_stay_synced():
    """MicrobatchedModel._micro_steps under `self.multi`: per-slice all-reduce -> / total weight -> per-slice clip -> sum"""
    (p0, s0), (p1, s1) = _run((1, 3), micro=True)
    np.testing.assert_array_equal(p0, p1)
    assert np.all(np.isfinite(p0)) and not np.array_equal(s0, s1)


# ------------------------------------------------------------------------------------------------------------------
def _native_comm():
    from baselines_amd import _lib
    lib = _lib.load()
    buf = ctypes.create_string_buffer(128)
    _lib.check(lib.mrl_comm_unique_id(buf), 'mrl_comm_unique_id')
    h = ctypes.c_void_p()
    _lib.check(lib.mrl_comm_create(bytes(buf.raw), 1, 0, ctypes.byref(h)), 'mrl_comm_create')
    assert lib.mrl_comm_size(h) == 1 and lib.mrl_comm_rank(h) == 0
    return lib, h


def test_rccl_entry_points_one_rank():
    from baselines_amd import _lib
    lib, h = _native_comm()
    try:
        x = torch.randn(1687719, device='cuda')
        ref = x.clone()
        _lib.check(lib.mrl_allreduce_grads(h, _lib.ptr(x), x.numel(), _lib.stream_ptr()), 'mrl_allreduce_grads')
        _lib.check(lib.mrl_broadcast_state(h, _lib.ptr(x), x.numel() * 4, 0, _lib.stream_ptr()), 'mrl_broadcast_state')
        torch.cuda.synchronize()
        assert torch.equal(x, ref)
        assert lib.mrl_allreduce_grads(h, None, 5, _lib.stream_ptr()) == -1            # MRL_EINVAL
        assert lib.mrl_broadcast_state(h, _lib.ptr(x), 4, 3, _lib.stream_ptr()) == -1   # root out of range
    finally:
        lib.mrl_comm_destroy(h)


@pytest.mark.parametrize('network', ['cnn', 'mlp'])
def test_attached_communicator_inside_the_backward_pass(network):
    """mrl_model_grad / mrl_model_train_step with a communicator attached: the tail of the flat gradient is all-reduced
    on the communication stream while the conv layers are still being back-propagated (cnn), the whole gradient after the
    fused step (mlp); with one rank the result must equal the un-attached gradient, times the rank weight."""
    from baselines_amd import _lib, ops
    lib, h = _native_comm()
    rng = np.random.RandomState(3)
    if network == 'cnn':
        B = 96
        dm = ops.DeviceModel(network='cnn', ob_shape=(84, 84, 4), ob_dtype=np.uint8, pd_kind='categorical', nact=6, chunk=64)
        obs = torch.from_numpy(rng.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)).cuda()
        act = torch.from_numpy(rng.randint(0, 6, B).astype(np.int32)).cuda()
    else:
        B = 256
        dm = ops.DeviceModel(network='mlp', ob_shape=(376,), ob_dtype=np.float32, pd_kind='gaussian', nact=17, value_copy=True,
                             chunk=B)
        obs = torch.from_numpy(rng.randn(B, 376).astype(np.float32)).cuda()
        act = torch.from_numpy(rng.randn(B, 17).astype(np.float32)).cuda()
    params = torch.from_numpy((rng.randn(dm.P) * 0.05).astype(np.float32)).cuda()
    ret, val = (torch.from_numpy(rng.randn(B).astype(np.float32)).cuda() for _ in range(2))
    nlp = torch.from_numpy((np.abs(rng.randn(B)) + 1).astype(np.float32)).cuda()

    def grad():
        g = torch.empty(dm.P, dtype=torch.float32, device='cuda')
        st = torch.empty(5, dtype=torch.float32, device='cuda')
        dm.grad(params, obs, act, ret, val, nlp, None, B, 1, 1, 0.2, 0.01, 0.5, g, st)
        torch.cuda.synchronize()
        return g, st

    def step(total_weight):
        p, m, v = params.clone(), torch.zeros_like(params), torch.zeros_like(params)
        g = torch.empty_like(params)
        st = torch.empty(5, dtype=torch.float32, device='cuda')
        gn = torch.empty(1, dtype=torch.float32, device='cuda')
        _lib.check(lib.mrl_model_train_step(dm.handle, _lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), _lib.ptr(obs),
                                            _lib.ptr(act), _lib.ptr(ret), _lib.ptr(val), _lib.ptr(nlp), None, B, 1, 1, 0.2,
                                            0.01, 0.5, 1e-3, None, 0.9, 0.999, 1e-5, 0.5, float(total_weight), _lib.ptr(st),
                                            _lib.ptr(gn), _lib.ptr(dm.workspace), dm.workspace.numel(), dm.chunk,
                                            _lib.stream_ptr()), 'mrl_model_train_step')
        torch.cuda.synchronize()
        return p, float(gn.cpu())

    try:
        g0, s0 = grad()
        p0, gn0 = step(1.0)
        dm.attach_comm(h, 1.0)
        g1, s1 = grad()
        assert torch.equal(g1, g0) and torch.equal(s1, s0)
        p1, gn1 = step(1.0)
        assert abs(gn1 - gn0) <= 1e-6 * gn0                  # norm from a separate pass instead of the fused partials
        np.testing.assert_allclose(p1.cpu().numpy(), p0.cpu().numpy(), rtol=0, atol=1e-7)
        dm.attach_comm(h, 3.0)                               # mpi_adam_optimizer.py:21 flat_grad * mpi_rank_weight
        g3, _ = grad()
        assert torch.equal(g3, g0 * 3.0)
        p3, gn3 = step(3.0)                                  # ... / total weight BEFORE the clip (:40)
        assert abs(gn3 - gn0) <= 1e-6 * gn0
        np.testing.assert_allclose(p3.cpu().numpy(), p0.cpu().numpy(), rtol=0, atol=1e-7)
    finally:
        dm.attach_comm(None, 1.0)
        lib.mrl_comm_destroy(h)
