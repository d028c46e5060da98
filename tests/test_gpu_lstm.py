"""GPU: recurrent policies -- `lstm` / `cnn_lstm` networks (common/models.py:132-206), the LSTM cell
with episode masks (a2c/utils.py:81-102), env-wise minibatching (ppo2/ppo2.py:167-180), states / masks through
Runner.run and Model.train (ppo2/runner.py:23-50, ppo2/model.py:153-155) -- against the oracle
(oracle/ppo2_torch.py `_lstm`, torch autograd through time, fp32 and fp64; the cell itself is cross-checked against the
closed-form NumPy backward of oracle/lstm_numpy.py on CPU).
"""
import numpy as np
import pytest
import torch

from oracle import ppo2_numpy as O
from oracle.ppo2_torch import OracleModel

pytestmark = pytest.mark.gpu


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


CASES = {
    'lstm_nh32': dict(network='lstm', ob_shape=(12,), ob_dtype=np.float32, nlstm=32, nseq=6, T=5),
    'lstm_nh128': dict(network='lstm', ob_shape=(376,), ob_dtype=np.float32, nlstm=128, nseq=9, T=16),
    'lstm_nh96': dict(network='lstm', ob_shape=(20,), ob_dtype=np.float32, nlstm=96, nseq=5, T=7),
    # widths without a register-resident instantiation: recurrent weights streamed from L2 (a2c/utils.py:81-102, any nh)
    'lstm_nh48_streamed': dict(network='lstm', ob_shape=(12,), ob_dtype=np.float32, nlstm=48, nseq=6, T=5),
    'lstm_nh256_streamed': dict(network='lstm', ob_shape=(24,), ob_dtype=np.float32, nlstm=256, nseq=5, T=6),
    'cnn_lstm': dict(network='cnn_lstm', ob_shape=(84, 84, 4), ob_dtype=np.uint8, nlstm=128, nseq=3, T=4),
    # layer-normalised cell (a2c/utils.py:104-140 lnlstm; lstm(layer_norm=True) / cnn_lnlstm)
    'lnlstm_nh32': dict(network='lstm', ob_shape=(12,), ob_dtype=np.float32, nlstm=32, nseq=6, T=5, layer_norm=True),
    'lnlstm_nh128': dict(network='lstm', ob_shape=(40,), ob_dtype=np.float32, nlstm=128, nseq=9, T=12, layer_norm=True),
    'lnlstm_nh48': dict(network='lstm', ob_shape=(20,), ob_dtype=np.float32, nlstm=48, nseq=5, T=7, layer_norm=True),
    'cnn_lnlstm': dict(network='cnn_lstm', ob_shape=(84, 84, 4), ob_dtype=np.uint8, nlstm=64, nseq=3, T=4, layer_norm=True),
}


@pytest.mark.parametrize('name', list(CASES))
def test_recurrent_act_and_bptt_gradient_vs_oracle(name):
    from baselines_amd import ops
    c = dict(CASES[name])
    nseq, T, nh = c.pop('nseq'), c.pop('T'), c['nlstm']
    B = nseq * T
    rng = np.random.RandomState(3)
    np.random.seed(3)
    kw = dict(pd_kind='categorical', nact=4, value_network=None, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, **c)
    om = OracleModel(**kw)
    with torch.no_grad():
        for k in om.names:
            om.p[k] += torch.tensor(0.05 * rng.randn(*om.p[k].shape), dtype=torch.float32)
    om64 = OracleModel(dtype=torch.float64, params=om.params_numpy(), **kw)
    dm = ops.DeviceModel(network=c['network'], ob_shape=c['ob_shape'], ob_dtype=c['ob_dtype'], pd_kind='categorical', nact=4,
                         nlstm=nh, layer_norm=c.get('layer_norm', False), chunk=B)
    assert [t['name'] for t in dm.tensors] == om.names and dm.state_size == 2 * nh
    params = dev(om.flat_params().astype(np.float32))
    if c['ob_dtype'] == np.uint8:
        obs = rng.randint(0, 256, (B,) + c['ob_shape']).astype(np.uint8)
    else:
        obs = rng.randn(B, *c['ob_shape']).astype(np.float32)
    masks = rng.rand(B) < 0.25                                   # episode boundaries inside the trajectories
    masks[0] = True
    S0 = (0.5 * rng.randn(nseq, 2 * nh)).astype(np.float32)

    # ---- act side: one step of nseq sequences (first sample of every trajectory), state in/out
    first = np.arange(nseq) * T
    noise = rng.rand(nseq, 4).astype(np.float32)
    a_o, v_o, s_o, nlp_o = om.step(obs[first], noise, S=S0, M=masks[first])
    st = dev(S0).clone()
    a_d = torch.empty(nseq, dtype=torch.int32, device='cuda')
    v_d = torch.empty(nseq, dtype=torch.float32, device='cuda')
    nlp_d = torch.empty(nseq, dtype=torch.float32, device='cuda')
    dm.act_rnn_into(params, dev(obs[first]), dev(noise), st, dev(masks[first].view(np.uint8)), st, a_d, v_d, nlp_d)
    np.testing.assert_array_equal(a_d.cpu().numpy().astype(np.int64), a_o)
    np.testing.assert_allclose(v_d.cpu().numpy(), v_o, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(nlp_d.cpu().numpy(), nlp_o, rtol=1e-4, atol=2e-5)
    # (c | h) after the step; two fp32 implementations of gate pre-activations of size O(10) differ by a few 1e-6 in a cell state
    np.testing.assert_allclose(st.cpu().numpy(), s_o, rtol=1e-4, atol=1e-5)

    # ---- learner: loss + gradient through all T steps of the nseq trajectories
    actions = rng.randint(0, 4, B)
    returns, values = rng.randn(B).astype(np.float32), rng.randn(B).astype(np.float32)
    nlps = (np.log(4.0) + 0.1 * rng.randn(B)).astype(np.float32)
    so, go = om.compute_grads(0.2, obs, returns, actions, values, nlps, states=S0, masks=masks)
    s64, g64 = om64.compute_grads(0.2, obs, returns, actions, values, nlps, states=S0, masks=masks)
    grads = torch.empty(dm.P, dtype=torch.float32, device='cuda')
    stats = torch.empty(5, dtype=torch.float32, device='cuda')
    d_obs, d_act, d_ret, d_val, d_nlp = dev(obs), dev(actions.astype(np.int32)), dev(returns), dev(values), dev(nlps)
    d_m, d_s = dev(masks.view(np.uint8)), dev(S0)
    dm.grad_rnn(params, d_obs, d_act, d_ret, d_val, d_nlp, d_m, d_s, nseq, None, B, 1, 1, 0.2, 0.01, 0.5, grads, stats)
    np.testing.assert_allclose(stats.cpu().numpy(), np.array(so), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(stats.cpu().numpy(), np.array(s64), rtol=1e-5, atol=1e-5)
    g_d, ref = grads.cpu().numpy().astype(np.float64), g64.numpy()
    scale = np.abs(ref).max()
    for t in dm.tensors:
        sl = slice(t['offset'], t['offset'] + t['size'])
        tol = 5e-5 * max(np.abs(ref[sl]).max(), 1e-3 * scale)
        err = np.abs(g_d[sl] - ref[sl]).max()
        assert err <= tol, (t['name'], err, np.abs(go.numpy()[sl] - ref[sl]).max(), tol)
    # every recurrent tensor does receive gradient (the scan, the three GEMMs and the bias column sums)
    for nm in ('lstm/wx', 'lstm/wh', 'lstm/b'):
        t = [t for t in dm.tensors if t['name'].endswith(nm)][0]
        assert np.abs(g_d[t['offset']:t['offset'] + t['size']]).max() > 1e-8

    # ---- the same minibatch addressed in place through env-major indices into a time-major rollout of N > nseq envs
    N = nseq + 2
    envs = rng.permutation(N)[:nseq]
    def to_rollout(x, fill):
        full = np.full((T, N) + x.shape[1:], fill, dtype=x.dtype)
        full[:, envs] = x.reshape((nseq, T) + x.shape[1:]).swapaxes(0, 1)
        return full
    idx = (envs[:, None] * T + np.arange(T)[None, :]).ravel().astype(np.int64)
    grads2 = torch.empty_like(grads)
    stats2 = torch.empty_like(stats)
    r_obs, r_act = dev(to_rollout(obs, 0)), dev(to_rollout(actions.astype(np.int32), 0))
    r_ret, r_val, r_nlp = dev(to_rollout(returns, 0)), dev(to_rollout(values, 0)), dev(to_rollout(nlps, 1))
    r_m = dev(to_rollout(masks.view(np.uint8), 0))
    dm.grad_rnn(params, r_obs, r_act, r_ret, r_val, r_nlp, r_m, d_s, nseq, dev(idx), B, T, N, 0.2, 0.01, 0.5, grads2, stats2)
    np.testing.assert_array_equal(stats2.cpu().numpy(), stats.cpu().numpy())
    np.testing.assert_array_equal(grads2.cpu().numpy(), grads.cpu().numpy())


@pytest.mark.parametrize('kind,net,N,T,nmb,nep', [('cartpole', 'lstm', 8, 32, 4, 2), ('atari', 'cnn_lstm', 4, 6, 2, 2),
                                                  ('cartpole', 'lnlstm', 8, 16, 2, 2), ('atari', 'cnn_lnlstm', 4, 6, 2, 1)])
def test_learn_recurrent_two_updates_match_oracle(kind, net, N, T, nmb, nep):
    """ppo2.learn with a recurrent policy on the device env: env-wise minibatches from the global NumPy stream,
    states = LSTM state when the rollout started, masks = done flags; every recorded minibatch step is replayed by the
    oracle's train() (model.py:133-158 with S / M fed) -- stats 1e-5, parameters 1e-5 after two updates.  The first step of
    an update also ties the act side (T one-step calls carrying the state) to the train side (one scan over T steps):
    the policy has not changed yet, so approxkl and clipfrac must be zero."""
    from baselines_amd import ppo2
    from baselines_amd.ppo2 import Model
    from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv
    rec = {}
    net_kw, onet = {}, net
    if net == 'lnlstm':                       # common/models.py:132: lstm(nlstm, layer_norm=True)
        net, onet, net_kw = 'lstm', 'lstm', dict(layer_norm=True)
    elif net == 'cnn_lnlstm':                 # the registered name (models.py:216-218)
        onet = 'cnn_lstm'
    ln = net_kw.get('layer_norm', False) or net == 'cnn_lnlstm'

    class RecModel(Model):          # the reference's model_fn plug point (ppo2.py:103-109)
        def train_indexed(self, lr, cliprange, rollout, idx_dev, stats_out=None, states=None):
            first = len(rec.get('calls', [])) % (nmb * nep) == 0
            rec.setdefault('calls', []).append(dict(
                lr=lr, clip=cliprange, idx=idx_dev.cpu().numpy().copy(), states=states.cpu().numpy().copy(),
                fields={k: getattr(rollout, k).cpu().numpy().copy() for k in
                        ('obs', 'actions', 'returns', 'values', 'neglogpacs', 'dones')} if first else None,
                params_before=self.get_flat_params()))
            s = super().train_indexed(lr, cliprange, rollout, idx_dev, stats_out, states=states)
            rec['calls'][-1]['stats'] = s.cpu().numpy().copy()
            return s

    env = SyntheticVecEnv(kind, N, seed=2)
    updates = []
    model = ppo2.learn(network=net, env=env, total_timesteps=2 * N * T, seed=0, nsteps=T, nminibatches=nmb, noptepochs=nep,
                       ent_coef=0.01, lr=lambda f: 3e-4 * f, cliprange=0.2, log_interval=1, model_fn=RecModel,
                       update_fn=updates.append, nlstm=32 if net == 'lstm' else 128, **net_kw)
    assert updates == [1, 2] and model.recurrent and model.initial_state.shape == (N, 2 * model.dm.state_size // 2)
    calls = rec['calls']
    assert len(calls) == 2 * nmb * nep
    np.random.seed(0)
    om = OracleModel(network=onet, ob_shape=env.observation_space.shape, ob_dtype=env.observation_space.dtype,
                     pd_kind=model.pd_kind, nact=model.nact, value_network=None, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5,
                     nlstm=32 if net == 'lstm' else 128, layer_norm=ln)
    np.testing.assert_array_equal(calls[0]['params_before'], om.flat_params())       # same seeded init stream
    envsper = N // nmb
    fields = None
    for i, c in enumerate(calls):
        if c['fields'] is not None:
            fields = {k: O.sf01(v) for k, v in c['fields'].items()}
            # first minibatch step of an update: same policy as the rollout -> ratio == 1 everywhere
            assert c['stats'][3] < 1e-9 and c['stats'][4] == 0.0, c['stats']
            if i == 0:
                assert not c['states'].any()                                          # initial_state = zeros (models.py:171)
            else:
                assert c['states'].any()                                              # the state is carried across rollouts
        idx = c['idx']
        # env-wise minibatch: whole trajectories, env after env (ppo2.py:172-177)
        assert idx.shape == (envsper * T,) and np.array_equal(idx.reshape(envsper, T) % T, np.tile(np.arange(T), (envsper, 1)))
        so = om.train(c['lr'], c['clip'], fields['obs'][idx], fields['returns'][idx], fields['dones'][idx].astype(bool),
                      fields['actions'][idx], fields['values'][idx], fields['neglogpacs'][idx], states=c['states'])
        np.testing.assert_allclose(c['stats'], so, rtol=1e-4, atol=1e-5)
    assert sorted(np.concatenate([c['idx'] for c in calls[:nmb]]).tolist()) == list(range(N * T))
    np.testing.assert_allclose(model.get_flat_params(), om.flat_params(), rtol=0, atol=1e-5)


def test_recurrent_model_reference_signature_host_env():
    """Model.step(obs, S=, M=) / value(obs, S=, M=) / train(..., states=) with host arrays (the reference's protocol,
    runner.py:28,50 and ppo2.py:172-179) through a host environment: the Runner's returned states are those at the START
    of the rollout, masks are bools, and train() accepts exactly the slices the reference passes."""
    from baselines_amd.common import set_global_seeds
    from baselines_amd.common.policies import build_policy
    from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnvCPU
    from baselines_amd.ppo2 import Model, Runner
    N, T = 4, 12
    env = SyntheticVecEnvCPU('cartpole', N, seed=5)
    set_global_seeds(1)
    policy = build_policy(env, 'lstm', nlstm=32)
    model = Model(policy=policy, ob_space=env.observation_space, ac_space=env.action_space, nbatch_act=N,
                  nbatch_train=N * T // 2, nsteps=T, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5)
    assert model.initial_state.shape == (N, 64) and model.initial_state.dtype == np.float64
    runner = Runner(env=env, model=model, nsteps=T, gamma=0.99, lam=0.95)
    obs, returns, masks, actions, values, neglogpacs, states, _ = runner.run()
    assert states.shape == (N, 64) and not states.any() and masks.dtype == np.bool_
    obs2, _, _, _, _, _, states2, _ = runner.run()
    assert states2.any()
    # step / value with explicit state, reference style
    a, v, s1, nlp = model.step(obs2[:N], S=states2, M=np.zeros(N, bool))
    assert s1.shape == (N, 64) and s1.dtype == np.float32
    v2 = model.value(obs2[:N], S=states2, M=np.zeros(N, bool))
    np.testing.assert_allclose(v, v2, rtol=1e-6, atol=1e-7)
    v3 = model.value(obs2[:N], S=states2, M=np.ones(N, bool))            # mask set: the state is ignored
    v4 = model.value(obs2[:N], S=np.zeros_like(states2), M=np.zeros(N, bool))
    np.testing.assert_allclose(v3, v4, rtol=1e-6, atol=1e-7)
    envinds = np.array([2, 0])
    flat = np.arange(N * T).reshape(N, T)[envinds].ravel()
    stats = model.train(3e-4, 0.2, obs[flat], returns[flat], masks[flat], actions[flat], values[flat], neglogpacs[flat],
                        states[envinds])
    assert len(stats) == 5 and np.all(np.isfinite(stats)) and stats[3] < 1e-9     # same policy as the rollout: approxkl 0


@pytest.mark.parametrize('nh,nseq,T', [(128, 9, 16), (64, 5, 33), (128, 64, 128)])
def test_one_environment_per_workgroup_scans_are_bit_identical(nh, nseq, T):
    """option lstm_e1 (csrc/lstm.hip.h lstm_fwd1_kernel / lstm_bwd1_kernel): the scans with one environment per workgroup
    (plain fmaf chains in k order, the masked state carried in registers) against the four-environment 4x4x1-MFMA scans --
    same chains, same gate functions: statistics and every gradient entry bit-identical (a2c/utils.py:81-102)."""
    from baselines_amd import _lib as L
    from baselines_amd import ops
    B = nseq * T
    rng = np.random.RandomState(nh + T)
    dm = ops.DeviceModel(network='lstm', ob_shape=(24,), ob_dtype=np.float32, pd_kind='categorical', nact=4, nlstm=nh, chunk=B)
    params = dev((rng.randn(dm.P) * 0.1).astype(np.float32))
    obs = dev(rng.randn(B, 24).astype(np.float32))
    masks = rng.rand(B) < 0.05
    masks[0] = True
    act = dev(rng.randint(0, 4, B).astype(np.int32))
    ret, val_ = dev(rng.randn(B).astype(np.float32)), dev(rng.randn(B).astype(np.float32))
    nlp = dev((np.log(4.0) + 0.1 * rng.randn(B)).astype(np.float32))
    S0 = dev((0.5 * rng.randn(nseq, 2 * nh)).astype(np.float32))
    d_m = dev(masks.view(np.uint8))
    old = L.get_option('lstm_e1')
    outs = []
    try:
        for v in (1, 0):
            L.set_option('lstm_e1', v)
            g = torch.empty(dm.P, dtype=torch.float32, device='cuda')
            st = torch.empty(5, dtype=torch.float32, device='cuda')
            dm.grad_rnn(params, obs, act, ret, val_, nlp, d_m, S0, nseq, None, B, 1, 1, 0.2, 0.01, 0.5, g, st)
            outs.append((g.cpu(), st.cpu()))
    finally:
        L.set_option('lstm_e1', old)
    assert float(outs[0][0].abs().max()) > 0 and bool(torch.isfinite(outs[0][0]).all())
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
