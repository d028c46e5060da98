"""GPU parity at the shapes `bench.py` TIMES for BASELINE.json's config 2 and the recurrent row.

  * config 2 -- MuJoCo-shaped `mlp`, value_network='copy', num_envs=1024, nsteps=128, 32 minibatches of 4096: whole epochs
    through `Model.train_epoch` (first epoch = individually launched steps, second epoch = the replayed hipGraph with
    `mrl_advstat_minibatches`, the 8-wave fused step on 256 workgroups, the slab reduction and Adam) against
    `OracleModel.train` step by step (ppo2/model.py:133-158, ppo2/ppo2.py:154-166), then one 4096-sample `mrl_model_grad`
    against the fp64 oracle on every entry;
  * recurrent row -- `cnn_lstm` nlstm=128 with 64 envs x 128 steps per minibatch (the bench's N=256 / 4 minibatches) and
    `lstm` nlstm=128 with 64 envs x 128 steps: act + BPTT gradient through all 128 steps of the MFMA 4x4x1 scans against
    the fp32 / fp64 oracle (a2c/utils.py:81-102, ppo2/ppo2.py:167-180).

Measured errors are appended to $MRL_PARITY_REPORT (JSON lines) when that variable is set; profiles/ keeps a copy.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ppo2_numpy as O
from oracle.ppo2_torch import OracleModel

pytestmark = pytest.mark.gpu


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def report(**kw):
    path = os.environ.get('MRL_PARITY_REPORT')
    if path:
        with open(path, 'a') as fh:
            fh.write(json.dumps(kw) + '\n')


def per_tensor_errors(tensors, g_d, g_32, g_64):
    """per gradient tensor: max |device - fp64| and max |fp32 oracle - fp64| over all entries, both over the tensor's scale"""
    out = {}
    for t in tensors:
        sl = slice(t['offset'], t['offset'] + t['size'])
        sc = max(float(np.abs(g_64[sl]).max()), 1e-30)
        out[t['name']] = dict(scale=sc, err_device=float(np.abs(g_d[sl] - g_64[sl]).max()) / sc,
                              err_fp32_oracle=float(np.abs(g_32[sl] - g_64[sl]).max()) / sc)
    return out


def test_config2_epochs_at_the_benched_shape_vs_oracle_step_by_step():
    from baselines_amd import _lib
    from baselines_amd.common import set_global_seeds
    from baselines_amd.common.policies import build_policy
    from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv
    from baselines_amd.ppo2 import Model, Runner
    N, T, M = 1024, 128, 32
    B = N * T // M
    lr, clip, ent = 3e-4, 0.2, 0.0                     # ppo2/defaults.py:3-13 (mujoco), as bench.py runs the row
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    env = SyntheticVecEnv('mujoco', N, seed=1000)
    set_global_seeds(0)
    policy = build_policy(env, 'mlp', value_network='copy')
    model = Model(policy=policy, ob_space=env.observation_space, ac_space=env.action_space, nbatch_act=N, nbatch_train=B,
                  nsteps=T, ent_coef=ent, vf_coef=0.5, max_grad_norm=0.5)
    assert _lib.get_option('mlp_fused') == 1 and _lib.get_option('mlp_waves') == 8 and _lib.get_option('mlp_slice') == 1
    np.random.seed(0)                                  # the oracle draws the same ortho-init stream (a2c/utils.py:20-35)
    kw = dict(network='mlp', ob_shape=(376,), ob_dtype=np.float32, pd_kind='gaussian', nact=17, value_network='copy',
              ent_coef=ent, vf_coef=0.5, max_grad_norm=0.5)
    om = OracleModel(**kw)
    np.testing.assert_array_equal(model.get_flat_params(), om.flat_params())
    om64 = OracleModel(dtype=torch.float64, params=om.params_numpy(), **kw)       # the yardstick for free-running trajectories
    runner = Runner(env=env, model=model, nsteps=T, gamma=0.99, lam=0.95, return_host=False)
    runner.run()
    runner.run()                                       # second rollout: episodes have ended, dones are mixed in
    ro = runner.rollout
    f = {k: O.sf01(getattr(ro, k).cpu().numpy()) for k in ('obs', 'actions', 'returns', 'values', 'neglogpacs')}
    worst_stats, worst_par, worst_o32 = 0.0, [], []
    inds = np.arange(N * T)
    for epoch in range(3):
        np.random.shuffle(inds)                        # ppo2.py:156-158
        st = model.train_epoch(lr, clip, ro, model.indices_to_device(inds)).cpu().numpy()
        assert st.shape == (M, 5)
        for k in range(M):
            idx = inds[k * B:(k + 1) * B]
            so = om.train(lr, clip, f['obs'][idx], f['returns'][idx], None, f['actions'][idx], f['values'][idx],
                          f['neglogpacs'][idx])
            s64 = om64.train(lr, clip, f['obs'][idx], f['returns'][idx], None, f['actions'][idx], f['values'][idx],
                             f['neglogpacs'][idx])
            np.testing.assert_allclose(st[k], np.array(so), rtol=1e-5, atol=1e-5, err_msg='epoch %d step %d' % (epoch, k))
            np.testing.assert_allclose(st[k], np.array(s64), rtol=1e-5, atol=1e-5, err_msg='epoch %d step %d (fp64)' % (epoch, k))
            worst_stats = max(worst_stats, float(np.abs(st[k] - np.array(s64)).max()))
        # epoch 0 ran as individually launched steps (nothing to replay yet), epochs 1, 2 as the captured graph and its replay
        assert (model._epoch_graph is None) == (epoch == 0)
        # Adam steps are sign-like (|step| ~ lr whatever |g|), so two fp32 implementations drift apart along a free-running
        # trajectory on the entries whose gradient is a cancellation residue; the yardstick is the fp64 trajectory: the device
        # has to stay within 5e-6 of it for the first two epochs (one launched step by step, one replayed) and as close
        # to it as the fp32 CPU restatement does thereafter
        p64 = om64.flat_params()
        worst_par.append(float(np.abs(model.get_flat_params() - p64).max()))
        worst_o32.append(float(np.abs(om.flat_params() - p64).max()))
    report(test='config2_epochs_benched_shape.trajectory', max_param_abs_diff_vs_fp64_after_each_epoch=worst_par,
           fp32_oracle_max_param_abs_diff_vs_fp64_after_each_epoch=worst_o32, worst_stat_abs_diff_vs_fp64=worst_stats)
    for epoch in range(3):
        assert worst_par[epoch] <= max(5e-6 if epoch < 2 else 0.0, 2.0 * worst_o32[epoch]), (epoch, worst_par, worst_o32)
    assert model._train_calls == 3 * M and isinstance(model._epoch_graph, dict)
    v64 = np.concatenate([om64.v[k].numpy().reshape(-1) for k in om.names])
    v32 = np.concatenate([om.v[k].numpy().reshape(-1) for k in om.names]).astype(np.float64)
    e_d, e_o = np.abs(model.adam_v.cpu().numpy() - v64).max(), np.abs(v32 - v64).max()
    assert e_d <= max(1e-6 * v64.max(), 4.0 * e_o), (e_d, e_o, v64.max())          # second-moment slots: same yardstick (a trajectory, not a step)

    # ---- one 4096-sample gradient, every entry against the fp64 oracle (the policy has moved: ratios != 1, clipping active)
    flat = model.get_flat_params()
    params = {t['name']: flat[t['offset']:t['offset'] + t['size']].reshape(t['shape']) for t in model.dm.tensors}
    om32 = OracleModel(params=params, **kw)
    om64 = OracleModel(dtype=torch.float64, params=params, **kw)
    idx = inds[5 * B:6 * B]
    args = (clip, f['obs'][idx], f['returns'][idx], f['actions'][idx], f['values'][idx], f['neglogpacs'][idx])
    s32, g32 = om32.compute_grads(*args)
    s64, g64 = om64.compute_grads(*args)
    grads = torch.empty(model.dm.P, dtype=torch.float32, device='cuda')
    stats = torch.empty(5, dtype=torch.float32, device='cuda')
    model.dm.grad(model.params, ro.obs, ro.actions, ro.returns, ro.values, ro.neglogpacs, dev(idx), B, T, N, clip, ent, 0.5,
                  grads, stats)
    np.testing.assert_allclose(stats.cpu().numpy(), np.array(s64), rtol=1e-5, atol=1e-5)
    assert s64[4] > 0.0, 'some samples are clipped after three epochs'
    g_d, g64n, g32n = grads.cpu().numpy().astype(np.float64), g64.numpy(), g32.numpy().astype(np.float64)
    errs = per_tensor_errors(model.dm.tensors, g_d, g32n, g64n)
    scale = np.abs(g64n).max()
    for t in model.dm.tensors:
        sl = slice(t['offset'], t['offset'] + t['size'])
        tol = 5e-5 * max(np.abs(g64n[sl]).max(), 1e-3 * scale)
        assert np.abs(g_d[sl] - g64n[sl]).max() <= tol, (t['name'], errs[t['name']])
    report(test='config2_epochs_benched_shape', N=N, T=T, minibatch=B, steps_checked=3 * M,
           worst_stat_abs_diff_vs_fp64_oracle=worst_stats, max_param_abs_diff_vs_fp64_after_each_epoch=worst_par,
           fp32_oracle_max_param_abs_diff_vs_fp64_after_each_epoch=worst_o32,
           grad_errors_over_tensor_scale=errs)


def _min_abs_preact(om64, obs):
    from tests.test_gpu_large_batch import _min_abs_preact as f
    return np.concatenate([f(om64, obs[s:s + 4096]) for s in range(0, obs.shape[0], 4096)])


RECURRENT = {
    # the recurrent row of bench.py: cnn_lstm, N=256 envs / 4 minibatches = 64 trajectories of 128 steps per minibatch
    'cnn_lstm_64x128': dict(network='cnn_lstm', ob_shape=(84, 84, 4), ob_dtype=np.uint8, nlstm=128, nseq=64, T=128),
    'lstm_nh128_64x128': dict(network='lstm', ob_shape=(376,), ob_dtype=np.float32, nlstm=128, nseq=64, T=128),
}


@pytest.mark.parametrize('name', list(RECURRENT))
def test_recurrent_bptt_over_128_steps_at_the_benched_shape_vs_oracle(name):
    from baselines_amd import ops
    c = dict(RECURRENT[name])
    nseq, T, nh = c.pop('nseq'), c.pop('T'), c['nlstm']
    B = nseq * T
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    rng = np.random.RandomState(31)
    np.random.seed(31)
    kw = dict(pd_kind='categorical', nact=6, value_network=None, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, **c)
    om = OracleModel(**kw)
    with torch.no_grad():
        for k in om.names:
            om.p[k] += torch.tensor(0.02 * rng.randn(*om.p[k].shape), dtype=torch.float32)
    om64 = OracleModel(dtype=torch.float64, params=om.params_numpy(), **kw)
    if c['ob_dtype'] == np.uint8:
        # ReLU kinks (tests/test_gpu_large_batch.py): images with a conv / fc1 pre-activation within 1e-5 of zero are replaced
        pool = rng.randint(0, 256, (B + B // 3, 84, 84, 4), dtype=np.uint8)
        keep = np.nonzero(_min_abs_preact(om64, pool) > 1e-5)[0]
        assert keep.size >= B, (keep.size, B)
        obs = pool[keep[:B]]
        del pool
    else:
        obs = rng.randn(B, *c['ob_shape']).astype(np.float32)
    masks = rng.rand(B) < 0.02                                   # a few episode boundaries inside the 128-step trajectories
    masks[::T][::2] = True
    S0 = (0.5 * rng.randn(nseq, 2 * nh)).astype(np.float32)
    dm = ops.DeviceModel(network=c['network'], ob_shape=c['ob_shape'], ob_dtype=c['ob_dtype'], pd_kind='categorical', nact=6,
                         nlstm=nh, chunk=B)
    assert [t['name'] for t in dm.tensors] == om.names
    params = dev(om.flat_params().astype(np.float32))

    # ---- act side: the 128 steps of the 64 trajectories one at a time, state carried on the device (runner.py:23-50),
    #      against ONE oracle scan over the trajectories -- values and the final state
    with torch.no_grad():
        pi_o, v_o = om64.forward(obs, S0, masks, nenv=nseq)
    st = dev(S0).clone()
    tm = lambda x: np.ascontiguousarray(x.reshape((nseq, T) + x.shape[1:]).swapaxes(0, 1))      # env-major -> time-major
    d_obs_tm, d_m_tm = dev(tm(obs)), dev(tm(masks.view(np.uint8)))
    v_d = torch.empty((T, nseq), dtype=torch.float32, device='cuda')
    pd_d = torch.empty((T, nseq, 6), dtype=torch.float32, device='cuda')
    for t in range(T):
        dm.act_rnn_into(params, d_obs_tm[t], None, st, d_m_tm[t], st, None, v_d[t], None, pdparam=pd_d[t])
    np.testing.assert_allclose(v_d.cpu().numpy().T.reshape(-1), v_o.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(pd_d.cpu().numpy().swapaxes(0, 1).reshape(B, 6), pi_o.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(st.cpu().numpy(), om64.last_state.numpy(), rtol=1e-4, atol=5e-6)

    # ---- learner: loss + gradient through all 128 steps
    actions = rng.randint(0, 6, B)
    values = (v_o.numpy() + 0.05 * rng.randn(B)).astype(np.float32)
    returns = (values + 0.5 * rng.randn(B)).astype(np.float32)
    with torch.no_grad():
        nlp_now = om64._neglogp(pi_o, torch.as_tensor(actions)).numpy()
    nlps = (nlp_now + 0.05 * rng.randn(B)).astype(np.float32)      # ratios around 1, some clipped at 0.1
    so, go = om.compute_grads(0.1, obs, returns, actions, values, nlps, states=S0, masks=masks)
    s64, g64 = om64.compute_grads(0.1, obs, returns, actions, values, nlps, states=S0, masks=masks)
    assert s64[4] > 0.01
    grads = torch.empty(dm.P, dtype=torch.float32, device='cuda')
    stats = torch.empty(5, dtype=torch.float32, device='cuda')
    d_obs, d_act, d_ret, d_val, d_nlp = dev(obs), dev(actions.astype(np.int32)), dev(returns), dev(values), dev(nlps)
    d_m, d_s = dev(masks.view(np.uint8)), dev(S0)
    dm.grad_rnn(params, d_obs, d_act, d_ret, d_val, d_nlp, d_m, d_s, nseq, None, B, 1, 1, 0.1, 0.01, 0.5, grads, stats)
    np.testing.assert_allclose(stats.cpu().numpy(), np.array(s64), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(stats.cpu().numpy(), np.array(so), rtol=1e-5, atol=1e-5)
    g_d, ref, g32 = grads.cpu().numpy().astype(np.float64), g64.numpy(), go.numpy().astype(np.float64)
    errs = per_tensor_errors(dm.tensors, g_d, g32, ref)
    report(test='recurrent_bptt_benched_shape', case=name, nseq=nseq, T=T,
           stats_abs_diff_vs_fp64=[float(x) for x in np.abs(stats.cpu().numpy() - np.array(s64))],
           grad_errors_over_tensor_scale=errs)
    scale = np.abs(ref).max()
    for t in dm.tensors:
        sl = slice(t['offset'], t['offset'] + t['size'])
        tol = 5e-5 * max(np.abs(ref[sl]).max(), 1e-3 * scale)       # the bar of test_recurrent_act_and_bptt_gradient_vs_oracle
        assert np.abs(g_d[sl] - ref[sl]).max() <= tol, (t['name'], errs[t['name']], tol)

    # ---- the same minibatch addressed in place: env-major indices into a time-major rollout of 4 x 64 environments
    N = 4 * nseq
    envs = np.sort(rng.permutation(N)[:nseq])
    rng.shuffle(envs)
    def to_rollout(x, fill):
        full = np.full((T, N) + x.shape[1:], fill, dtype=x.dtype)
        full[:, envs] = x.reshape((nseq, T) + x.shape[1:]).swapaxes(0, 1)
        return full
    idx = (envs[:, None] * T + np.arange(T)[None, :]).ravel().astype(np.int64)
    grads2, stats2 = torch.empty_like(grads), torch.empty_like(stats)
    r_obs, r_act = dev(to_rollout(obs, 0)), dev(to_rollout(actions.astype(np.int32), 0))
    r_ret, r_val, r_nlp = dev(to_rollout(returns, 0)), dev(to_rollout(values, 0)), dev(to_rollout(nlps, 1))
    r_m = dev(to_rollout(masks.view(np.uint8), 0))
    dm.grad_rnn(params, r_obs, r_act, r_ret, r_val, r_nlp, r_m, d_s, nseq, dev(idx), B, T, N, 0.1, 0.01, 0.5, grads2, stats2)
    np.testing.assert_array_equal(stats2.cpu().numpy(), stats.cpu().numpy())
    np.testing.assert_array_equal(grads2.cpu().numpy(), grads.cpu().numpy())


# ---- config 4's per-rank shape and config 3's whole update ---------------------------
CNN_KW = dict(network='cnn', ob_shape=(84, 84, 4), ob_dtype=np.uint8, pd_kind='categorical', nact=6, value_network=None,
              ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5)


def test_rank_shard_minibatch_backward_vs_oracle_every_entry():
    """Config 4's PER-RANK shape (num_envs = 4096 over 8 GPUs = 512 envs per rank, 4 minibatches of 16384, chunk 16384 -- the
    shape `bench.py --gpus 8` and the N = 512 row of `other_configs` time): `mrl_model_grad` gathering 16384 samples through
    env-major indices out of a time-major rollout, every entry of every gradient tensor and the 5 statistics against the
    fp64 oracle (ppo2/model.py:57-114, 136-139; common/models.py:15-26) at the bar of the full-size test (5e-6 of each
    tensor's scale).  The engines size their persistent-workgroup grids and slab counts from B, so neither the B = 8192 nor
    the B = 131072 test reaches this configuration."""
    from baselines_amd import ops
    from tests.test_gpu_large_batch import MARGIN, _sliced
    B, S, clip = 16384, 8192, 0.1
    T, N = 48, 512                                     # a 512-env shard; 24576 stored samples to screen 16384 out of
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    rng = np.random.RandomState(31)
    np.random.seed(31)
    om = OracleModel(**CNN_KW)
    with torch.no_grad():
        for k in om.names:
            om.p[k] += torch.tensor(0.02 * rng.randn(*om.p[k].shape), dtype=torch.float32)
    obs_tm = rng.randint(0, 256, (T, N, 84, 84, 4), dtype=np.uint8)           # storage order of the rollout buffer: [T][N]
    flat = O.sf01(obs_tm)                                                      # the reference's env-major view, i = e*T + t
    n = T * N
    a_f, v_f, nlp_f = np.empty(n, np.int64), np.empty(n, np.float32), np.empty(n, np.float32)
    for sl in _sliced(n, S):
        a_f[sl], v_f[sl], _, nlp_f[sl] = om.step(flat[sl], rng.rand(sl.stop - sl.start, 6).astype(np.float32))
    with torch.no_grad():
        for k in om.names:
            om.p[k] += torch.tensor(0.003 * rng.randn(*om.p[k].shape), dtype=torch.float32)
    om64 = OracleModel(dtype=torch.float64, params=om.params_numpy(), **CNN_KW)
    ret_f = (v_f + 0.5 * rng.randn(n)).astype(np.float32)
    keep = np.nonzero(_min_abs_preact(om64, flat) > MARGIN)[0]
    assert keep.size >= B, 'screening left %d of %d samples' % (keep.size, B)
    idx = rng.permutation(keep)[:B]                                            # a minibatch: B env-major indices, shuffled

    def tm(x):                                                                 # env-major flat -> time-major [T][N]
        return np.ascontiguousarray(x.reshape(N, T).swapaxes(0, 1))

    dm = ops.DeviceModel(chunk=B, **{k: CNN_KW[k] for k in ('network', 'ob_shape', 'ob_dtype', 'pd_kind', 'nact')})
    assert dm.chunk == B
    params = dev(om.flat_params().astype(np.float32))
    d = dict(obs=dev(obs_tm), act=dev(tm(a_f).astype(np.int32)), ret=dev(tm(ret_f)), val=dev(tm(v_f)), nlp=dev(tm(nlp_f)))
    grads = torch.empty(dm.P, dtype=torch.float32, device='cuda')
    stats = torch.empty(5, dtype=torch.float32, device='cuda')
    dm.grad(params, d['obs'], d['act'], d['ret'], d['val'], d['nlp'], dev(idx), B, T, N, clip, 0.01, 0.5, grads, stats)
    g_d, s_d = grads.cpu().numpy().astype(np.float64), stats.cpu().numpy()
    assert np.isfinite(g_d).all() and np.isfinite(s_d).all()

    obs, rets, acts, vals, nlps = flat[idx], ret_f[idx], a_f[idx], v_f[idx], nlp_f[idx]
    advs64, advs32 = om64._normalised_advs(rets, vals), om._normalised_advs(rets, vals)
    g64, g32, s64 = np.zeros(dm.P), np.zeros(dm.P), np.zeros(5)
    for sl in _sliced(B, S):                           # two slices with the advantage statistics of the WHOLE minibatch
        st, fl = om64.compute_grads(clip, obs[sl], rets[sl], acts[sl], vals[sl], nlps[sl], advs=advs64[sl])
        g64 += fl.numpy() / (B // S)
        s64 += np.array(st) / (B // S)
        _, fl32 = om.compute_grads(clip, obs[sl], rets[sl], acts[sl], vals[sl], nlps[sl], advs=advs32[sl])
        g32 += fl32.numpy().astype(np.float64) / (B // S)
    np.testing.assert_allclose(s_d, s64, rtol=1e-5, atol=1e-5)
    assert s64[4] > 0.0                                # some samples are clipped
    errs = per_tensor_errors(dm.tensors, g_d, g32, g64)
    report(test='rank_shard_minibatch_backward', B=B, chunk=B, T=T, N=N,
           stats_abs_diff_vs_fp64=[float(x) for x in np.abs(s_d - s64)], grad_errors_over_tensor_scale=errs)
    scale = np.abs(g64).max()
    for t in dm.tensors:
        sl = slice(t['offset'], t['offset'] + t['size'])
        tol = 5e-6 * max(np.abs(g64[sl]).max(), 1e-3 * scale)
        assert np.abs(g_d[sl] - g64[sl]).max() <= tol, (t['name'], errs[t['name']])


def test_config3_whole_update_vs_oracle_step_by_step():
    """BASELINE.json config 3 exactly as `learn()` runs it (ppo2/ppo2.py:154-166, ppo2/defaults.py:15-22): nature_cnn,
    num_envs = 256, nsteps = 128, 4 epochs x 4 minibatches of 8192 out of the rollout the device Runner stored -- 16
    `Model.train_indexed` steps with permutations from the global NumPy stream, each followed by `OracleModel.train`
    (fp32 and fp64; ppo2/model.py:133-158) on the same gathered samples.  north_star's bar: the 5 loss statistics within
    1e-5 and the parameters within 5e-6 of the fp64 trajectory after the optimizer steps -- held strictly through the first
    epoch; from the second epoch on both are allowed twice the fp32 CPU restatement's OWN deviation from fp64 on top
    (Adam's sign-like steps amplify round-off on cancellation-residue entries, so any two fp32 implementations leave the
    fp64 trajectory, and each other, by ~1e-5 within 16 steps; same yardstick as the config-2 test).  No ReLU-margin screening here: these are the samples the runner produced."""
    from baselines_amd.common import set_global_seeds
    from baselines_amd.common.policies import build_policy
    from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv
    from baselines_amd.ppo2 import Model, Runner
    N, T, M, E = 256, 128, 4, 4
    B = N * T // M
    lr, clip = 2.5e-4, 0.1
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    env = SyntheticVecEnv('atari', N, seed=1003)
    set_global_seeds(0)
    policy = build_policy(env, 'cnn')
    model = Model(policy=policy, ob_space=env.observation_space, ac_space=env.action_space, nbatch_act=N, nbatch_train=B,
                  nsteps=T, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5)
    np.random.seed(0)
    om = OracleModel(**CNN_KW)
    np.testing.assert_array_equal(model.get_flat_params(), om.flat_params())
    om64 = OracleModel(dtype=torch.float64, params=om.params_numpy(), **CNN_KW)
    runner = Runner(env=env, model=model, nsteps=T, gamma=0.99, lam=0.95, return_host=False)
    runner.run()
    runner.run()                                       # second rollout: episodes have ended, dones are mixed in
    ro = runner.rollout
    f = {k: O.sf01(getattr(ro, k).cpu().numpy()) for k in ('obs', 'actions', 'returns', 'values', 'neglogpacs')}
    worst_stats, drift_d, drift_o, clipfrac, stat_err_d, stat_err_o = 0.0, [], [], [], [], []
    inds = np.arange(N * T)
    step = 0
    for epoch in range(E):
        np.random.shuffle(inds)                        # ppo2.py:157-158
        inds_dev = model.indices_to_device(inds)
        for lo in range(0, N * T, B):
            idx = inds[lo:lo + B]
            st = model.train_indexed(lr, clip, ro, inds_dev[lo:lo + B]).cpu().numpy()
            args = (lr, clip, f['obs'][idx], f['returns'][idx], None, f['actions'][idx], f['values'][idx], f['neglogpacs'][idx])
            so, s64 = np.array(om.train(*args)), np.array(om64.train(*args))
            # the yardstick is the fp64 trajectory.  First epoch (the parameters have moved by four Adam steps at most): 1e-5 against
            # both oracles.  Later steps evaluate the loss at parameters that two fp32 implementations have each carried 5e-6..1e-5
            # away from fp64 by then -- the statistics of a free-running fp32 trajectory differ from fp64's by that drift, the fp32
            # CPU restatement's included -- so the device is held to 1e-5 plus twice the fp32 restatement's own deviation.
            dev_err, o32_err = np.abs(st - s64), np.abs(so - s64)
            tol = 1e-5 + 1e-5 * np.abs(s64) + (2.0 * o32_err if epoch > 0 else 0.0)
            assert (dev_err <= tol).all(), ('step %d' % step, st.tolist(), s64.tolist(), so.tolist())
            if epoch == 0:
                np.testing.assert_allclose(st, so, rtol=1e-5, atol=1e-5, err_msg='step %d' % step)
            stat_err_d.append(float((dev_err / (1.0 + np.abs(s64))).max()))
            stat_err_o.append(float((o32_err / (1.0 + np.abs(s64))).max()))
            worst_stats = max(worst_stats, float(dev_err.max()))
            p64 = om64.flat_params()
            drift_d.append(float(np.abs(model.get_flat_params() - p64).max()))
            drift_o.append(float(np.abs(om.flat_params() - p64).max()))
            clipfrac.append(float(s64[4]))
            step += 1
    report(test='config3_whole_update', N=N, T=T, minibatch=B, steps_checked=step, worst_stat_abs_diff_vs_fp64=worst_stats,
           max_param_abs_diff_vs_fp64_after_each_step=drift_d, fp32_oracle_max_param_abs_diff_vs_fp64_after_each_step=drift_o,
           clipfrac_per_step=clipfrac, stat_err_over_1_plus_abs_per_step=stat_err_d,
           fp32_oracle_stat_err_over_1_plus_abs_per_step=stat_err_o)
    assert step == E * M == 16 and model._train_calls == 16
    assert max(clipfrac) > 0.0                         # the policy moved: ratios != 1, the clip is active in later epochs
    for k in range(step):
        assert drift_d[k] <= max(5e-6, 2.0 * drift_o[k]), (k, drift_d, drift_o)


def test_config3_update_teacher_forced():
    """Per-step EVALUATION error of the device update, separated from trajectory drift.
    Before each of the 16 minibatch steps of a config-3 update the device parameters AND Adam slots are overwritten with the
    fp64 teacher's, rounded to fp32 (the teacher continues from the same rounded state: tests/_teacher_forced.py), so every
    step starts from bit-identical state and what is compared is one step's own arithmetic (ppo2/model.py:57-91 loss,
    :97-114 clip + Adam, :133-158 train).  Bars, with NO allowance for the fp32 restatement's own error: the 5 statistics
    within 5e-7 * (1 + |s|) of fp64, the post-step parameters within 1e-6.  Measured (profiles/r06a_teacher_forced_modes.txt):
    statistics <= 4.4e-8 * (1 + |s|) -- one ulp of the fp32 word they are returned in; the fp32 CPU restatement: 5.8e-8 --,
    parameters <= 4.8e-7 at the first step (m = v = 0: Adam's step is g / (|g| + eps), sensitivity 1 / eps = 1e5 to an entry's
    gradient) and <= 2.3e-7 afterwards (the restatement: 2.0e-7 / 1.6e-7).  The free-running test above therefore measures
    trajectory drift, not evaluation error."""
    from tests import _teacher_forced as TF
    model, ro = TF.make_model_and_rollout()
    traj = TF.teacher_trajectory(model, ro)
    res = TF.device_teacher_forced(model, ro, traj)
    er = TF.errors(traj, res)
    assert len(er) == 16
    report(test='config3_update_teacher_forced', steps=len(er),
           stat_err_over_1_plus_abs_per_step=[e['stat_err'].tolist() for e in er],
           fp32_oracle_stat_err_over_1_plus_abs_per_step=[e['stat_err_fp32'].tolist() for e in er],
           max_param_abs_diff_vs_fp64_after_each_step=[e['param_err'] for e in er],
           fp32_oracle_max_param_abs_diff_vs_fp64_after_each_step=[e['param_err_fp32'] for e in er],
           clipfrac_per_step=[float(r['s64'][4]) for r in traj])
    assert max(float(r['s64'][4]) for r in traj) > 0.0            # the policy moved: the clip is active in later epochs
    for k, e in enumerate(er):
        assert (e['stat_err'] <= 5e-7).all(), ('step %d' % k, e['stat_err'].tolist(), res[k]['stats'].tolist(), traj[k]['s64'].tolist())
        assert e['param_err'] <= 1e-6, ('step %d' % k, e['param_err'], e['param_err_fp32'])
