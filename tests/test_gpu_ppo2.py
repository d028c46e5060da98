"""GPU: Runner / learn() end to end against the oracle and against the host-env path."""
import os

import numpy as np
import pytest
import torch

from oracle import ppo2_numpy as O
from oracle.ppo2_torch import OracleModel

pytestmark = pytest.mark.gpu


def _mk(kind, N, seed, device_env):
    from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv, SyntheticVecEnvCPU
    return SyntheticVecEnv(kind, N, seed=seed) if device_env else SyntheticVecEnvCPU(kind, N, seed=seed)


@pytest.mark.parametrize('kind', ['atari', 'mujoco', 'cartpole'])
def test_device_env_equals_cpu_twin(kind):
    """DummyVecEnv-as-oracle pattern of the reference (vec_env/test_vec_env.py:14-44): the device env
    and its NumPy twin produce identical obs / rewards / dones / episode infos for the same actions."""
    N = 37
    dev, cpu = _mk(kind, N, 5, True), _mk(kind, N, 5, False)
    np.testing.assert_array_equal(dev.reset().cpu().numpy(), cpu.reset())
    rng = np.random.RandomState(0)
    for t in range(120):
        if cpu.discrete:
            a = rng.randint(0, cpu.action_space.n, N)
            ad = torch.from_numpy(a.astype(np.int32)).cuda()
        else:
            a = rng.randn(N, cpu.action_space.shape[0]).astype(np.float32)
            ad = torch.from_numpy(a).cuda()
        o1, r1, d1, i1 = dev.step(ad)
        o2, r2, d2, i2 = cpu.step(a)
        np.testing.assert_array_equal(o1.cpu().numpy(), o2)
        np.testing.assert_array_equal(r1.cpu().numpy(), r2)
        np.testing.assert_array_equal(d1.cpu().numpy().astype(bool), d2)
        fl = i1['fin_l'].cpu().numpy()
        fr = i1['fin_r'].cpu().numpy()
        for e in range(N):
            if d2[e]:
                assert fl[e] == i2[e]['episode']['l'] and fr[e] == np.float32(i2[e]['episode']['r'])
            else:
                assert fl[e] == 0 and i2[e] == {}


def _models(kind, N, T, nmb, seed=0):
    from baselines_amd.common import set_global_seeds
    from baselines_amd.common.policies import build_policy
    from baselines_amd.ppo2 import Model
    env = _mk(kind, N, 3, False)
    net = {'atari': 'cnn', 'mujoco': 'mlp', 'cartpole': 'mlp'}[kind]
    vn = 'copy' if kind == 'mujoco' else None
    set_global_seeds(seed)
    policy = build_policy(env, net, value_network=vn)
    model = Model(policy=policy, ob_space=env.observation_space, ac_space=env.action_space, nbatch_act=N,
                  nbatch_train=N * T // nmb, nsteps=T, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5)
    np.random.seed(seed)     # the oracle draws the same ortho-init stream (a2c/utils.py:20-35)
    om = OracleModel(network=net, ob_shape=env.observation_space.shape, ob_dtype=env.observation_space.dtype,
                     pd_kind=model.pd_kind, nact=model.nact, value_network=vn, ent_coef=0.01, vf_coef=0.5,
                     max_grad_norm=0.5)
    return model, om


@pytest.mark.parametrize('kind', ['cartpole', 'mujoco', 'atari'])
def test_seeded_init_matches_reference_stream(kind):
    model, om = _models(kind, 4, 4, 1)
    np.testing.assert_array_equal(model.get_flat_params(), om.flat_params())


@pytest.mark.parametrize('kind,net', [('atari', 'nature_cnn_nact6'), ('mujoco', 'mlp_copy_376_17')])
def test_model_init_matches_the_reference_ortho_init_function(kind, net, golden_dir):
    """`Model` init against tests/golden/ortho_init.npz = the REFERENCE's ortho_init (a2c/utils.py:20-35, exec'd from
    its source by oracle/make_golden.py) called in TF variable-creation order on the stream of set_global_seeds(3)"""
    g = np.load(os.path.join(golden_dir, 'ortho_init.npz'))
    model, _ = _models(kind, 4, 4, 1, seed=3)
    flat = model.get_flat_params()
    ws = [t for t in model.dm.tensors if t['name'].endswith('/w')]
    assert len(ws) == 6
    for i, t in enumerate(ws):
        w = flat[t['offset']:t['offset'] + t['size']]
        np.testing.assert_allclose(w[::37], g['%s_%d_sample' % (net, i)], rtol=0, atol=2e-7, err_msg=t['name'])
        np.testing.assert_allclose([w.astype(np.float64).sum(), np.abs(w.astype(np.float64)).sum()],
                                   g['%s_%d_sums' % (net, i)], rtol=0, atol=2e-7 * w.size ** 0.5)
    others = [t for t in model.dm.tensors if not t['name'].endswith('/w')]
    for t in others:       # biases and logstd are zeros (a2c/utils.py:45,62; distributions.py:104)
        assert not flat[t['offset']:t['offset'] + t['size']].any(), t['name']


class _NoiseModel(object):
    """wraps our Model so that the Runner's act calls use a recorded noise stream (teacher forcing)"""

    def __init__(self, model, noises):
        self.m, self.noises, self.t = model, noises, 0
        self.initial_state = None
        self.device, self.pd_kind, self.nact = model.device, model.pd_kind, model.nact

    def step_into(self, obs, a, v, nlp):
        self.m.step_into(obs, a, v, nlp, noise=torch.from_numpy(self.noises[self.t]).cuda())
        self.t += 1

    def value_dev(self, obs):
        return self.m.value_dev(obs)


@pytest.mark.parametrize('kind,N,T', [('cartpole', 8, 128), ('mujoco', 12, 16), ('atari', 6, 5)])
def test_runner_device_vs_host_env_vs_oracle(kind, N, T):
    """Same seeded env, same noise: (a) our Runner on the device env, (b) our Runner on the NumPy twin
    (host-env path), (c) the reference algorithm restated by the oracle -> identical rollouts;
    returns bit-exact given the same values."""
    from baselines_amd.ppo2 import Runner
    model, om = _models(kind, N, T, 1)
    rng = np.random.RandomState(4)
    noises = [(rng.rand(N, model.nact) if model.pd_kind == 'categorical' else rng.randn(N, model.nact)).astype(np.float32)
              for _ in range(T)]
    outs = []
    for device_env in (True, False):
        env = _mk(kind, N, 11, device_env)
        r = Runner(env=env, model=_NoiseModel(model, noises), nsteps=T, gamma=0.99, lam=0.95, return_host=True)
        outs.append(r.run())
    for a, b in zip(outs[0][:6], outs[1][:6]):
        assert a.dtype == b.dtype and a.shape == b.shape
        np.testing.assert_array_equal(a, b)
    assert outs[0][6] is None
    assert sorted((e['l'], e['r']) for e in outs[0][7]) == sorted((e['l'], e['r']) for e in outs[1][7])
    # the episode records have the keys of bench/monitor.py:58-77 on both paths
    assert all(set(e) == {'r', 'l', 't'} and e['t'] >= 0 for o in outs for e in o[7])
    obs, returns, masks, actions, values, neglogpacs = outs[0][:6]
    assert returns.dtype == np.float32 and masks.dtype == np.bool_ and values.dtype == np.float32
    assert actions.dtype == (np.int64 if model.pd_kind == 'categorical' else np.float32)
    # oracle rollout with the reference's loop structure (runner.py:20-67) on the twin
    env = _mk(kind, N, 11, False)
    ob = env.reset()
    dones = np.zeros(N, bool)
    mb = dict(obs=[], rew=[], act=[], val=[], nlp=[], done=[])
    for t in range(T):
        a, v, _, nlp = om.step(ob, noises[t])
        mb['obs'].append(ob.copy()); mb['act'].append(a); mb['val'].append(v); mb['nlp'].append(nlp)
        mb['done'].append(dones)
        ob, rew, dones, _ = env.step(actions.reshape((N, T) + actions.shape[1:]).swapaxes(0, 1)[t])  # teacher-force OUR actions
        mb['rew'].append(rew)
    np.testing.assert_array_equal(O.sf01(np.asarray(mb['obs'])), obs)
    np.testing.assert_array_equal(O.sf01(np.asarray(mb['done'])), masks)
    if model.pd_kind == 'categorical':
        assert (O.sf01(np.asarray(mb['act'])) == actions).mean() > 0.98      # argmax ties/rounding aside
    np.testing.assert_allclose(O.sf01(np.asarray(mb['val'], np.float32)), values, rtol=1e-4, atol=2e-5)
    # GAE on OUR values must be bit-exact
    vals_tm = values.reshape(N, T).swapaxes(0, 1)
    last_v = model.value(ob)
    ret_o, _ = O.gae(np.asarray(mb['rew'], np.float32), np.ascontiguousarray(vals_tm), np.asarray(mb['done']), last_v,
                     dones, 0.99, 0.95)
    np.testing.assert_array_equal(O.sf01(ret_o), returns)


@pytest.mark.parametrize('kind,N,T,nmb,nep', [('cartpole', 8, 128, 4, 4), ('mujoco', 16, 32, 4, 2), ('atari', 8, 8, 2, 2),
                                              ('atari_cnn_small', 8, 8, 2, 2)])
def test_learn_two_updates_match_oracle(kind, N, T, nmb, nep):
    """ppo2.learn on the device env vs the reference algorithm driven by the oracle model on the
    teacher-forced rollouts: same minibatch permutations (global NumPy stream), loss stats within
    1e-5 and parameters after 2 updates within 1e-5."""
    from baselines_amd import ppo2
    from baselines_amd.ppo2 import Model
    from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv
    okw = {}
    if kind == 'atari_cnn_small':                     # learn(network='cnn_small') (common/models.py:117-129)
        kind, net, onet, okw = 'atari', 'cnn_small', 'cnn', dict(convs=((8, 8, 4), (16, 4, 2)), fc_hidden=128)
    else:
        net = onet = {'atari': 'cnn', 'mujoco': 'mlp', 'cartpole': 'mlp'}[kind]
    vn = 'copy' if kind == 'mujoco' else None
    rec = {}

    class RecModel(Model):          # the reference's model_fn plug point (ppo2.py:103-109)
        def train_indexed(self, lr, cliprange, rollout, idx_dev, stats_out=None):
            rec.setdefault('calls', []).append(dict(
                lr=lr, clip=cliprange, idx=idx_dev.cpu().numpy().copy(),
                fields={k: getattr(rollout, k).cpu().numpy().copy() for k in
                        ('obs', 'actions', 'returns', 'values', 'neglogpacs')} if len(rec.get('calls', [])) % (nmb * nep) == 0 else None,
                params_before=self.get_flat_params()))
            s = super().train_indexed(lr, cliprange, rollout, idx_dev, stats_out)
            rec['calls'][-1]['stats'] = s.cpu().numpy().copy()
            return s

    env = SyntheticVecEnv(kind, N, seed=2)
    updates = []
    model = ppo2.learn(network=net, env=env, total_timesteps=2 * N * T, seed=0, nsteps=T, nminibatches=nmb,
                       noptepochs=nep, ent_coef=0.01, lr=lambda f: 3e-4 * f, cliprange=0.2, log_interval=1,
                       value_network=vn, model_fn=RecModel, update_fn=updates.append)
    assert updates == [1, 2]
    calls = rec['calls']
    assert len(calls) == 2 * nmb * nep
    # oracle replays: same init (seed 0), then for each recorded minibatch the reference's train()
    np.random.seed(0)
    om = OracleModel(network=onet, ob_shape=env.observation_space.shape, ob_dtype=env.observation_space.dtype,
                     pd_kind=model.pd_kind, nact=model.nact, value_network=vn, ent_coef=0.01, vf_coef=0.5,
                     max_grad_norm=0.5, **okw)
    np.testing.assert_array_equal(calls[0]['params_before'], om.flat_params())
    fields = None
    for i, c in enumerate(calls):
        if c['fields'] is not None:
            fields = {k: O.sf01(v) for k, v in c['fields'].items()}
        assert c['lr'] == 3e-4 * (1.0 - (i // (nmb * nep)) / 2.0)       # ppo2.py:133-135 schedule
        idx = c['idx']
        so = om.train(c['lr'], c['clip'], fields['obs'][idx], fields['returns'][idx], None, fields['actions'][idx],
                      fields['values'][idx], fields['neglogpacs'][idx])
        np.testing.assert_allclose(c["stats"], so, rtol=1e-5, atol=1e-5)
    # permutations: exactly the reference's stream (set_global_seeds(0) -> ortho draws -> shuffles)
    assert sorted(np.concatenate([c['idx'] for c in calls[:nmb]]).tolist()) == list(range(N * T))
    # (a free-running Adam trajectory: sign-like steps amplify fp32 noise on entries whose gradient is a cancellation residue.
    # profiles/r06b_free_running_spread.txt: ANY two fp32 implementations of this update -- the CPU restatement, the device in four
    # arithmetic modes -- end 7e-6 .. 2e-5 apart after 16 steps while each single step, started from identical state, agrees to
    # 5e-7 (test_config3_update_teacher_forced).  Seen here: cnn_small 1 of 171039 entries at 1.2e-5, nature_cnn 3 of 1687719 at 1.02e-5.)
    np.testing.assert_allclose(model.get_flat_params(), om.flat_params(), rtol=0, atol=2e-5)


def test_model_train_reference_signature_and_save_load(tmp_path):
    """model.train(lr, cliprange, obs, returns, masks, actions, values, neglogpacs) with host arrays
    == train_indexed on the device rollout; save/load round trip in the reference's joblib format."""
    from baselines_amd.ppo2 import Runner
    model, om = _models('cartpole', 8, 16, 1)
    model2, _ = _models('cartpole', 8, 16, 1)
    env = _mk('cartpole', 8, 1, True)
    r = Runner(env=env, model=model, nsteps=16, gamma=0.99, lam=0.95, return_host=True)
    obs, returns, masks, actions, values, neglogpacs, _, _ = r.run()
    idx = np.random.RandomState(0).permutation(128)[:64]
    s1 = model.train(1e-3, 0.2, obs[idx], returns[idx], masks[idx], actions[idx], values[idx], neglogpacs[idx])
    s2 = model2.train_indexed(1e-3, 0.2, r.rollout, torch.from_numpy(idx).cuda()).cpu().numpy()
    np.testing.assert_array_equal(np.float32(s1), s2)
    np.testing.assert_array_equal(model.get_flat_params(), model2.get_flat_params())
    assert model.loss_names == ['policy_loss', 'value_loss', 'policy_entropy', 'approxkl', 'clipfrac']
    path = str(tmp_path / 'ckpt' / '00001')
    model.save(path)
    import joblib
    d = joblib.load(path)
    assert 'ppo2_model/pi/mlp_fc0/w:0' in d and 'ppo2_model/pi/mlp_fc0/w/Adam_1:0' in d and 'beta1_power:0' in d
    assert d['ppo2_model/pi/mlp_fc0/w:0'].shape == (4, 64) and d['ppo2_model/vf/w:0'].shape == (64, 1)
    model3, _ = _models('cartpole', 8, 16, 1, seed=5)
    model3.load(path)
    np.testing.assert_array_equal(model3.get_flat_params(), model.get_flat_params())
    np.testing.assert_array_equal(model3.adam_v.cpu().numpy(), model.adam_v.cpu().numpy())
    assert model3.beta1_power == model.beta1_power
    # the legacy LIST checkpoint (tf_util.py:362-366): GLOBAL_VARIABLES order = variables, beta powers, (m, v) per variable
    names = [t['name'] for t in model.dm.tensors]
    legacy = [d[n + ':0'] for n in names] + [d['beta1_power:0'], d['beta2_power:0']]
    for n in names:
        legacy += [d[n + '/Adam:0'], d[n + '/Adam_1:0']]
    joblib.dump(legacy, path + '_legacy')
    model4, _ = _models('cartpole', 8, 16, 1, seed=6)
    model4.load(path + '_legacy')
    np.testing.assert_array_equal(model4.get_flat_params(), model.get_flat_params())
    np.testing.assert_array_equal(model4.adam_m.cpu().numpy(), model.adam_m.cpu().numpy())
    np.testing.assert_array_equal(model4.adam_v.cpu().numpy(), model.adam_v.cpu().numpy())
    assert model4.beta2_power == model.beta2_power


def test_total_timesteps_zero_builds_model_only():
    """ppo2.py:128-129: total_timesteps=0 -> zero updates (used by the reference's tests to build/load)"""
    from baselines_amd import ppo2
    env = _mk('cartpole', 4, 0, True)
    model = ppo2.learn(network='mlp', env=env, total_timesteps=0, seed=1, nsteps=8)
    a, v, s, nlp = model.step(np.zeros((4, 4), np.float32))
    assert a.shape == (4,) and a.dtype == np.int64 and v.shape == (4,) and s is None and nlp.shape == (4,)
    with pytest.raises(ValueError):                       # common/models.py:275 'Unknown network type'
        ppo2.learn(network='impala_cnn', env=env, total_timesteps=0)
    rec = ppo2.learn(network='lstm', env=env, total_timesteps=0, nlstm=32)     # recurrent policies build too
    assert rec.initial_state.shape == (4, 64)


@pytest.mark.parametrize('kind', ['cartpole', 'mujoco'])
def test_epoch_graph_replay_equals_step_loop(kind, monkeypatch):
    """Model.train_epoch: an epoch replayed as one hipGraph (indices and Adam step sizes refreshed in device buffers)
    ends bit-identical to the individually launched steps -- parameters after 3 updates and the logged loss stats."""
    from baselines_amd import ppo2
    from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv
    vn = 'copy' if kind == 'mujoco' else None

    def run(flag):
        monkeypatch.setenv('MRL_EPOCH_GRAPH', flag)
        seen = []
        model = ppo2.learn(network='mlp', env=SyntheticVecEnv(kind, 16, seed=4), total_timesteps=3 * 16 * 32, seed=1, nsteps=32,
                           nminibatches=4, noptepochs=3, lr=lambda f: 1e-3 * f, cliprange=lambda f: 0.2 * f, value_network=vn,
                           log_interval=100, update_fn=seen.append)
        assert seen == [1, 2, 3]
        return model
    eager = run('0')
    graphed = run('1')
    assert graphed._epoch_graph is not None and eager._epoch_graph is None
    np.testing.assert_array_equal(eager.get_flat_params(), graphed.get_flat_params())
    np.testing.assert_array_equal(eager.adam_v.cpu().numpy(), graphed.adam_v.cpu().numpy())
    assert eager.beta1_power == graphed.beta1_power and eager._train_calls == graphed._train_calls == 36


@pytest.mark.parametrize('kind', ['cartpole', 'atari'])
def test_rollout_graph_replay_equals_step_loop(kind, monkeypatch):
    """Runner: the rollout replayed as one hipGraph (policy noise from the registered torch generator, env state in device
    arrays) fills the HBM rollout exactly like the step loop -- compared over 3 updates of learn()."""
    from baselines_amd import ppo2
    from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv
    net = 'cnn' if kind == 'atari' else 'mlp'
    N, T = (4, 8) if kind == 'atari' else (16, 32)

    def run(flag):
        monkeypatch.setenv('MRL_ROLLOUT_GRAPH', flag)
        got = {}

        def grab(update):
            got[update] = {k: getattr(runner_ref[0].rollout, k).cpu().numpy().copy() for k in
                           ('obs', 'actions', 'values', 'neglogpacs', 'rewards', 'dones', 'returns')}
        runner_ref = []
        orig = ppo2.ppo2.Runner

        class Spy(orig):
            def __init__(self, **kw):
                super().__init__(**kw)
                runner_ref.append(self)
        monkeypatch.setattr(ppo2.ppo2, 'Runner', Spy)
        model = ppo2.learn(network=net, env=SyntheticVecEnv(kind, N, seed=4), total_timesteps=3 * N * T, seed=1, nsteps=T,
                           nminibatches=2, noptepochs=2, log_interval=100, update_fn=grab)
        monkeypatch.setattr(ppo2.ppo2, 'Runner', orig)
        return model, got, runner_ref[0]
    m0, r0, run0 = run('0')
    m1, r1, run1 = run('1')
    assert run0._graph is None and run1._graph not in (None, False)
    for u in (1, 2, 3):
        for k in r0[u]:
            np.testing.assert_array_equal(r0[u][k], r1[u][k], err_msg='update %d field %s' % (u, k))
    np.testing.assert_array_equal(m0.get_flat_params(), m1.get_flat_params())


def test_normalize_observations_is_the_references_clip():
    """build_policy(normalize_observations=True) under ppo2 (policies.py:133-135,182-185 with a never-updated
    RunningMeanStd): observations are clipped to [-5, 5] on both the act and the train side."""
    from baselines_amd import ppo2
    from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv

    class Wide(SyntheticVecEnv):                     # observations scaled to +-20 so that the clip matters
        def reset(self, out=None):
            o = super().reset(out)
            return o.mul_(20.0)

        def step_into(self, actions, obs_out=None, rew_out=None, done_out=None):
            o, r, d, i = super().step_into(actions, obs_out=obs_out, rew_out=rew_out, done_out=done_out)
            o.mul_(20.0)
            return o, r, d, i
    m = ppo2.learn(network='mlp', env=Wide('mujoco', 8, seed=2), total_timesteps=2 * 8 * 16, seed=0, nsteps=16, nminibatches=2,
                   noptepochs=1, value_network='copy', normalize_observations=True, log_interval=100)
    x = (np.random.RandomState(0).randn(4, 376) * 20).astype(np.float32)
    noise = np.random.RandomState(1).randn(4, 17).astype(np.float32)
    a1, v1, _, n1 = m.step(x, noise=noise)
    a2, v2, _, n2 = m.step(np.clip(x, -5, 5), noise=noise)
    np.testing.assert_array_equal(a1, a2)
    np.testing.assert_array_equal(v1, v2)
    np.testing.assert_array_equal(m.value(x), m.value(np.clip(x, -5, 5)))
    assert np.isfinite(m.get_flat_params()).all()


def test_integration_model_fn_adapter():
    """INTEGRATION.md level 1: the adapter of examples/model_fn_adapter.py plugged into learn()'s `model_fn`; the model it
    returns is driven through the REFERENCE protocol only (host arrays: step / value / train), like the reference's own
    Runner and gather loop would (here: learn() with a model that hides its fast entry points)."""
    from baselines_amd import ppo2
    from examples.model_fn_adapter import make_model_fn
    inner = make_model_fn('mlp', value_network='copy')

    class ProtocolOnly(object):      # exposes exactly model.py:115-126,133
        def __init__(self, m):
            self._m = m
            self.initial_state, self.loss_names = m.initial_state, m.loss_names
            self.step, self.value, self.train, self.save, self.load = m.step, m.value, m.train, m.save, m.load

    made = []

    def model_fn(**kw):
        made.append(inner(**kw))
        return ProtocolOnly(made[-1])

    env = _mk('mujoco', 4, 3, False)
    seen = []
    model = ppo2.learn(network='mlp', env=env, total_timesteps=2 * 4 * 16, seed=0, nsteps=16, nminibatches=2, noptepochs=2,
                       value_network='copy', model_fn=model_fn, update_fn=seen.append, log_interval=100)
    assert seen == [1, 2] and isinstance(model, ProtocolOnly) and made[0]._train_calls == 2 * 2 * 2


@pytest.mark.parametrize('space', ['multidiscrete', 'multibinary'])
def test_learn_with_multidiscrete_and_multibinary_action_spaces(space):
    """ppo2.learn on a HOST environment whose action space is MultiDiscrete([2, 3]) / MultiBinary(3) (make_pdtype,
    common/distributions.py:285-288): the Runner stores one int32 per component, every minibatch step is replayed by the
    oracle's train() (MultiCategorical / Bernoulli neglogp, entropy and their gradients) -- stats 1e-5, parameters 1e-5."""
    from baselines_amd import ppo2
    from baselines_amd.ppo2 import Model
    from baselines_amd.common.spaces import MultiBinary, MultiDiscrete
    from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnvCPU
    N, T, nmb, nep = 8, 16, 2, 2
    ac = MultiDiscrete([2, 3]) if space == 'multidiscrete' else MultiBinary(3)

    class Env(SyntheticVecEnvCPU):           # cartpole-shaped dynamics driven by the first action component
        def __init__(self):
            super().__init__('cartpole', N, seed=2)
            self.action_space = ac

        def step_async(self, actions):
            super().step_async(np.asarray(actions).reshape(N, -1)[:, 0].astype(np.int64) % 2)

    rec = {}

    class RecModel(Model):
        def train_indexed(self, lr, cliprange, rollout, idx_dev, stats_out=None):
            first = len(rec.get('calls', [])) % (nmb * nep) == 0
            rec.setdefault('calls', []).append(dict(
                lr=lr, clip=cliprange, idx=idx_dev.cpu().numpy().copy(),
                fields={k: getattr(rollout, k).cpu().numpy().copy() for k in
                        ('obs', 'actions', 'returns', 'values', 'neglogpacs')} if first else None))
            s = super().train_indexed(lr, cliprange, rollout, idx_dev, stats_out)
            rec['calls'][-1]['stats'] = s.cpu().numpy().copy()
            return s

    env = Env()
    model = ppo2.learn(network='mlp', env=env, total_timesteps=2 * N * T, seed=0, nsteps=T, nminibatches=nmb, noptepochs=nep,
                       ent_coef=0.01, lr=3e-4, cliprange=0.2, log_interval=100, model_fn=RecModel)
    assert model.pd_kind == ('multicategorical' if space == 'multidiscrete' else 'bernoulli')
    calls = rec['calls']
    assert len(calls) == 2 * nmb * nep
    a0 = calls[0]['fields']['actions']
    assert a0.dtype == np.int32 and a0.shape == (T, N, 2 if space == 'multidiscrete' else 3)
    assert a0.min() >= 0 and (a0[..., 1].max() == 2 if space == 'multidiscrete' else a0.max() == 1)
    np.random.seed(0)
    om = OracleModel(network='mlp', ob_shape=(4,), ob_dtype=np.float32, pd_kind=model.pd_kind, nact=model.nact,
                     nvec=(2, 3) if space == 'multidiscrete' else None, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5)
    fields = None
    for c in calls:
        if c['fields'] is not None:
            fields = {k: O.sf01(v) for k, v in c['fields'].items()}
        idx = c['idx']
        so = om.train(c['lr'], c['clip'], fields['obs'][idx], fields['returns'][idx], None, fields['actions'][idx],
                      fields['values'][idx], fields['neglogpacs'][idx])
        np.testing.assert_allclose(c["stats"], so, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(model.get_flat_params(), om.flat_params(), rtol=0, atol=1e-5)
    # the reference's protocol: step() returns int32 components / float bits (distributions.py:224, 273)
    a, v, s, nlp = model.step(np.zeros((N, 4), np.float32))
    assert a.shape == a0.shape[1:] and a.dtype == (np.int32 if space == 'multidiscrete' else np.float32)
