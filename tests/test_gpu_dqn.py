"""GPU: the DQN learner -- Q-networks built by `build_q_func` (deepq/models.py:5-45: mlp / cnn /
conv_only features, [dueling] heads), the TD graph and per-variable clipped Adam step (deepq/build_graph.py:380-444),
eps-greedy acting (:146-199) and the `deepq.learn` loop (deepq/deepq.py:95-333) -- against the torch-CPU restatement in
oracle/dqn_torch.py (fp32 and fp64) and, functionally, the reference's identity-env bar (common/tests/test_identity.py).
"""
import random

import numpy as np
import pytest
import torch

from oracle.dqn_torch import OracleQNet

pytestmark = pytest.mark.gpu

from baselines_amd.common.spaces import Box, Discrete          # noqa: E402

CASES = {
    'mlp_plain': dict(network='mlp', ob=Box(-1.0, 1.0, (8,), np.float32), nact=3, hiddens=(64,), dueling=False),
    'mlp_dueling': dict(network='mlp', ob=Box(-1.0, 1.0, (12,), np.float32), nact=5, hiddens=(32, 16), dueling=True),
    'mlp_dueling_layer_norm': dict(network='mlp', ob=Box(-1.0, 1.0, (12,), np.float32), nact=5, hiddens=(48, 24), dueling=True,
                                   layer_norm=True),
    # network = mlp(layer_norm=True) (common/models.py:97-98): LayerNorm variables in the BODY; heads plain / normalised too
    'mlp_body_layer_norm': dict(network='mlp', ob=Box(-1.0, 1.0, (12,), np.float32), nact=5, hiddens=(32,), dueling=True,
                                body_layer_norm=True, num_layers=3, num_hidden=40),
    'mlp_body_and_heads_layer_norm': dict(network='mlp', ob=Box(-1.0, 1.0, (10,), np.float32), nact=4, hiddens=(24,), dueling=False,
                                          layer_norm=True, body_layer_norm=True),
    'conv_only_dueling': dict(network='conv_only', ob=Box(0, 255, (84, 84, 4), np.uint8), nact=6, hiddens=(256,), dueling=True),
    'conv_only_small': dict(network='conv_only', ob=Box(0, 255, (21, 17, 4), np.uint8), nact=4, hiddens=(32,), dueling=True,
                            convs=((8, 5, 3), (12, 3, 2))),
    'nature_cnn': dict(network='cnn', ob=Box(0, 255, (84, 84, 4), np.uint8), nact=4, hiddens=(64,), dueling=True),
    # geometry the skinny-tile conv kernels meet nowhere else: 96 filters (three 32-column tiles), 96 input channels in a data gradient,
    # even receptive field with stride 2 (asymmetric SAME padding), pixel counts that are no multiple of 32
    'conv_only_wide': dict(network='conv_only', ob=Box(0, 255, (30, 26, 4), np.uint8), nact=5, hiddens=(64,), dueling=True,
                           convs=((32, 6, 2), (96, 3, 2), (64, 3, 1))),
}


def _pair(name, B, seed, double_q=True):
    from baselines_amd.deepq import QModel, build_q_func
    c = dict(CASES[name])
    ob, nact = c.pop('ob'), c.pop('nact')
    net_kw = {k: c.pop(k) for k in list(c) if k in ('convs', 'num_layers', 'num_hidden')}
    if c.get('body_layer_norm'):
        net_kw['layer_norm'] = True          # build_q_func(network, **network_kwargs) -> mlp(layer_norm=True)
        from baselines_amd.deepq.models import get_network_builder
        qf_net = get_network_builder(c['network'])(**net_kw)
    np.random.seed(seed)
    torch.manual_seed(seed)
    qf = build_q_func(qf_net if c.get('body_layer_norm') else c['network'], hiddens=c['hiddens'], dueling=c['dueling'],
                      layer_norm=c.get('layer_norm', False), **({} if c.get('body_layer_norm') else net_kw))
    qm = QModel(qf, ob, nact, lr=1e-3, gamma=0.99, grad_norm_clipping=10, double_q=double_q, max_batch=B)
    rng = np.random.RandomState(seed + 1)
    # biases are zero at init: perturb everything so every gradient path is exercised
    flat = qm.get_flat_params() + (0.02 * rng.randn(qm.P)).astype(np.float32)
    qm.params.copy_(torch.from_numpy(flat))
    tflat = flat + (0.05 * rng.randn(qm.P)).astype(np.float32)             # target != online
    qm.target.copy_(torch.from_numpy(tflat))
    kw = dict(network=c['network'], tensors=qm.tensors, nact=nact, hiddens=c['hiddens'], dueling=c['dueling'],
              convs=qf.network.kw.get('convs', ()), lr=1e-3, gamma=0.99, clip=10.0, double_q=double_q,
              layer_norm=c.get('layer_norm', False), body_layer_norm=c.get('body_layer_norm', False),
              num_layers=qf.network.kw.get('num_layers', 2))
    if c.get('body_layer_norm'):             # the body's variables sit in the q_func scope, created beta first (models.py:97-98)
        nm = [t['name'] for t in qm.tensors]
        assert nm[:4] == ['deepq/q_func/mlp_fc0/w', 'deepq/q_func/mlp_fc0/b', 'deepq/q_func/LayerNorm/beta', 'deepq/q_func/LayerNorm/gamma']
        assert 'deepq/q_func/LayerNorm_1/gamma' in nm
        g = [t for t in qm.tensors if t['name'] == 'deepq/q_func/LayerNorm/gamma'][0]
        assert g['init_kind'] == 3
    oms = []
    for dt in (torch.float32, torch.float64):
        om = OracleQNet(flat_params=flat, dtype=dt, **kw)
        om.target = {t['name']: torch.tensor(tflat[t['offset']:t['offset'] + t['size']].reshape(t['shape']), dtype=dt)
                     for t in qm.tensors}
        oms.append(om)
    if ob.dtype == np.uint8:
        mk = lambda: rng.randint(0, 256, (B,) + ob.shape).astype(np.uint8)
    else:
        mk = lambda: rng.randn(B, *ob.shape).astype(np.float32)
    batch = dict(obs_t=mk(), act=rng.randint(0, nact, B), rew=rng.randn(B).astype(np.float32), obs_tp1=mk(),
                 done=(rng.rand(B) < 0.2).astype(np.float32), w=(0.5 + rng.rand(B)).astype(np.float32))
    return qm, oms[0], oms[1], batch


@pytest.mark.parametrize('name', list(CASES))
def test_q_values_td_gradient_and_clipped_adam_vs_oracle(name):
    from baselines_amd import _lib
    B = 32
    qm, om, om64, b = _pair(name, B, 0)
    names = [t['name'] for t in qm.tensors]
    assert names[-1].startswith('deepq/q_func/state_value/' if CASES[name]['dueling'] else 'deepq/q_func/action_value/')
    # ---- q_values (online and target)
    np.testing.assert_allclose(qm.q_values(b['obs_t']), om64.q_values(b['obs_t']), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(qm.q_values(b['obs_tp1'], target=True), om64.q_values(b['obs_tp1'], target=True), rtol=1e-4, atol=2e-5)
    # ---- TD error, loss and un-clipped gradient
    o1, o2 = qm._obs(b['obs_t']), qm._obs(b['obs_tp1'])
    dev = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x)).cuda().to(dt)
    a, r, d, w = dev(b['act'], torch.int32), dev(b['rew'], torch.float32), dev(b['done'], torch.float32), dev(b['w'], torch.float32)
    td = torch.empty(B, dtype=torch.float32, device='cuda')
    _lib.check(qm.lib.mrl_qnet_td_grad(qm.handle, _lib.ptr(qm.params), _lib.ptr(qm.target), _lib.ptr(o1), _lib.ptr(a), _lib.ptr(r),
                                       _lib.ptr(o2), _lib.ptr(d), _lib.ptr(w), 0.99, 1, B, _lib.ptr(qm.grads), _lib.ptr(td),
                                       _lib.ptr(qm._loss), _lib.ptr(qm.workspace), qm.workspace.numel(), _lib.stream_ptr()),
               'mrl_qnet_td_grad')
    td_o, loss_o, g_o = om.td_and_grads(b['obs_t'], b['act'], b['rew'], b['obs_tp1'], b['done'], b['w'])
    td_64, loss_64, g_64 = om64.td_and_grads(b['obs_t'], b['act'], b['rew'], b['obs_tp1'], b['done'], b['w'])
    np.testing.assert_allclose(td.cpu().numpy(), td_64, rtol=1e-4, atol=3e-5)
    assert abs(float(qm._loss.cpu()) - loss_64) <= 1e-5 * max(1.0, abs(loss_64))
    g_d = qm.grads.cpu().numpy().astype(np.float64)
    scale = max(float(g_64[k].abs().max()) for k in names)
    for t in qm.tensors:
        ref = g_64[t['name']].numpy().reshape(-1)
        got = g_d[t['offset']:t['offset'] + t['size']]
        tol = 5e-5 * max(np.abs(ref).max(), 1e-3 * scale)
        assert np.abs(got - ref).max() <= tol, (t['name'], np.abs(got - ref).max(),
                                                np.abs(g_o[t['name']].numpy().reshape(-1) - ref).max(), tol)
    # ---- optimizer: per-variable clip_by_norm(10) + Adam (epsilon 1e-8) on IDENTICAL gradients.  Adam's first steps are
    # sign-like (|step| ~ lr whatever |g|): an absolute gradient error of 1e-7 moves entries with |g| ~ 1e-6 by 1e-5 and
    # the next gradients by a percent, so free-running trajectories of two fp32 implementations are not comparable at
    # tight tolerances; the optimizer arithmetic is, when both sides are fed the same gradient.
    flat_of = lambda dct: np.concatenate([dct[k].detach().numpy().reshape(-1) for k in names]).astype(np.float32)
    for it, gscale in enumerate((1.0, 1000.0, 1.0)):          # step 1: every variable is clipped to norm 10
        _, _, g = om.td_and_grads(b['obs_t'], b['act'], b['rew'], b['obs_tp1'], b['done'], b['w'])
        if it == 1:      # large enough that EVERY variable's norm exceeds the clip
            gscale = max(gscale, 20.0 / min(float(torch.sqrt((v * v).sum())) for v in g.values()))
        g = {k: v * gscale for k, v in g.items()}
        if it == 1:
            assert min(float(torch.sqrt((v * v).sum())) for v in g.values()) > 10.0
        qm.params.copy_(torch.from_numpy(om.flat_params()))
        qm.grads.copy_(torch.from_numpy(flat_of(g)))
        qm.adam_m.copy_(torch.from_numpy(flat_of(om.m)))
        qm.adam_v.copy_(torch.from_numpy(flat_of(om.v)))
        one = np.float32(1)
        alpha = np.float32(1e-3) * np.sqrt(one - om.beta2_power) / (one - om.beta1_power)
        _lib.check(qm.lib.mrl_qnet_adam_step(qm.handle, _lib.ptr(qm.params), _lib.ptr(qm.grads), _lib.ptr(qm.adam_m),
                                             _lib.ptr(qm.adam_v), float(alpha), None, 0.9, 0.999, 1e-8, 10.0, _lib.ptr(qm.workspace),
                                             qm.workspace.numel(), qm.max_batch, _lib.stream_ptr()), 'mrl_qnet_adam_step')
        om.apply_grads(g)
        np.testing.assert_allclose(qm.get_flat_params(), om.flat_params(), rtol=0, atol=3e-7)
        np.testing.assert_allclose(qm.adam_m.cpu().numpy(), flat_of(om.m), rtol=1e-5, atol=2e-8)
        np.testing.assert_allclose(qm.adam_v.cpu().numpy(), flat_of(om.v), rtol=1e-5, atol=1e-10)
    # ---- free-running train() calls: sanity of the composed step (loose, see above)
    qm.beta1_power, qm.beta2_power = np.float32(om.beta1_power), np.float32(om.beta2_power)
    for it in range(2):
        td_d = qm.train(b['obs_t'], b['act'], b['rew'], b['obs_tp1'], b['done'], b['w'])
        td_r, _ = om.train(b['obs_t'], b['act'], b['rew'], b['obs_tp1'], b['done'], b['w'])
        assert np.abs(td_d - td_r).max() <= 2e-2 * max(1.0, np.abs(td_r).max())
    assert np.abs(qm.get_flat_params() - om.flat_params()).mean() < 2e-5
    before = qm.target.clone()
    qm.update_target()
    assert not torch.equal(before, qm.target) and torch.equal(qm.target, qm.params)


def test_act_is_eps_greedy():
    qm, om, _, b = _pair('mlp_dueling', 256, 3)
    obs = np.random.RandomState(0).randn(256, 12).astype(np.float32)
    greedy = om.q_values(obs).argmax(axis=1)
    a = qm.act(obs, stochastic=False)
    assert a.dtype == np.int64 and np.array_equal(a, greedy)
    a0 = qm.act(obs, update_eps=1.0)                 # eps variable still 0 when this call samples (assign is an update op)
    assert np.array_equal(a0, greedy)
    a1 = qm.act(obs)                                 # eps == 1: every action uniformly random
    assert (a1 != greedy).mean() > 0.6 and set(np.unique(a1)) == set(range(5))
    qm.act(obs, update_eps=0.0)
    assert np.array_equal(qm.act(obs), greedy)
    from baselines_amd.deepq import ActWrapper
    acts, _, _, _ = ActWrapper(qm, {}).step(obs[0])
    assert acts.shape == (1,)


class DiscreteIdentityEnv(object):
    """common/tests/envs/identity_env.py: observation = uniform integer, reward = [action == observation]"""

    def __init__(self, dim, episode_len, seed):
        self.observation_space = self.action_space = Discrete(dim)
        self.dim, self.episode_len = dim, episode_len
        self._rng = np.random.RandomState(seed)

    def reset(self):
        self._t = 0
        self._state = int(self._rng.randint(self.dim))
        return self._state

    def step(self, action):
        rew = 1.0 if int(action) == self._state else 0.0
        self._t += 1
        self._state = int(self._rng.randint(self.dim))
        return self._state, rew, self._t >= self.episode_len, {}


@pytest.mark.parametrize('prioritized', [False, True])
def test_deepq_learn_identity_env(prioritized, tmp_path):
    """the reference's functional bar for deepq (common/tests/test_identity.py: gamma 0.9, network mlp, >= 90 % of the
    reward over 100 evaluation steps), here on a 5-way identity env with a shorter schedule; then save / load round trips"""
    from baselines_amd import deepq
    env = DiscreteIdentityEnv(5, 50, seed=0)
    seen = {'train': 0, 'target': 0}

    def cb(lcl, _glb):
        t = lcl['t']
        if t > 200 and t % 1 == 0:
            seen['train'] += 1
        return False

    act = deepq.learn(env, network='mlp', seed=0, gamma=0.9, total_timesteps=4000, lr=2e-3, buffer_size=2000,
                      exploration_fraction=0.3, exploration_final_eps=0.02, learning_starts=200, target_network_update_freq=100,
                      prioritized_replay=prioritized, print_freq=None, checkpoint_freq=None, callback=cb, hiddens=[32],
                      num_hidden=32)
    assert seen['train'] > 3000
    test_env = DiscreteIdentityEnv(5, 1000, seed=1)
    obs, total = test_env.reset(), 0.0
    for _ in range(100):
        a = act(np.array(obs)[None], stochastic=False)[0]
        obs, rew, _, _ = test_env.step(a)
        total += rew
    assert total >= 90, total
    # checkpoints: reference variable names, round trip through save / load_act
    p = str(tmp_path / 'model.pkl')
    act.save_act(p)
    act2 = deepq.load_act(p)
    probe = np.arange(5)
    assert np.array_equal(act(probe, stochastic=False), act2(probe, stochastic=False))
    names = set(act._model.variables())
    assert {'deepq/eps:0', 'deepq/q_func/mlp_fc0/w:0', 'deepq/target_q_func/mlp_fc0/w:0',
            'deepq/q_func/action_value/fully_connected/weights:0', 'deepq/q_func/state_value/fully_connected_1/biases:0',
            'deepq/q_func/mlp_fc0/w/Adam:0'} <= names


def test_deepq_train_step_sequence_matches_oracle_with_prioritized_replay():
    """A short seeded learner trace: prioritized replay (device trees, reference index stream from Python's `random`),
    double-Q TD step, priority update, periodic target copies -- replayed by the oracle on the SAME sampled batches:
    TD errors 1e-3, parameters 2e-5 after 30 steps."""
    from baselines_amd.deepq import PrioritizedReplayBuffer
    qm, om, _, _ = _pair('conv_only_small', 16, 5)
    rng = np.random.RandomState(9)
    random.seed(9)
    buf = PrioritizedReplayBuffer(64, alpha=0.6)
    for _ in range(64):
        buf.add(rng.randint(0, 256, (21, 17, 4)).astype(np.uint8), int(rng.randint(4)), float(rng.randn()),
                rng.randint(0, 256, (21, 17, 4)).astype(np.uint8), float(rng.rand() < 0.1))
    for step in range(30):
        o1, a, r, o2, d, w, idx = buf.sample(16, beta=0.4 + 0.02 * step)
        td_d = qm.train(o1, a, r, o2, d, w)
        td_o, _ = om.train(o1, a, r, o2, d, w)
        np.testing.assert_allclose(td_d, td_o, rtol=5e-3, atol=5e-3)
        buf.update_priorities(idx, np.abs(td_o) + 1e-6)           # same priorities on both sides keep the index streams equal
        if step % 10 == 9:
            qm.update_target()
            om.update_target()
    diff = np.abs(qm.get_flat_params() - om.flat_params())
    assert diff.mean() < 2e-5 and np.percentile(diff, 99) < 3e-4, (diff.mean(), np.percentile(diff, 99))


@pytest.mark.parametrize('name', ['mlp_dueling', 'conv_only_small'])
def test_train_dev_graph_replay_is_bit_identical_to_eager_launches(name):
    """`QModel.train_dev` replays the optimizer step as ONE captured launch graph (inputs in static device buffers, the Adam
    step size in a device word).  Same kernels, same order: TD errors, parameters and Adam slots of a graph-replayed run
    equal those of an eagerly launched run bit for bit over several steps (first step eager + capture, then replays)."""
    B = 16
    qa, _, _, batch = _pair(name, B, 31)
    qb, _, _, _ = _pair(name, B, 31)
    assert torch.equal(qa.params, qb.params) and torch.equal(qa.target, qb.target)
    rng = np.random.RandomState(2)
    dev = lambda x, dt=None: (torch.from_numpy(np.ascontiguousarray(x)).cuda() if dt is None
                              else torch.from_numpy(np.ascontiguousarray(x)).cuda().to(dt))
    for step in range(5):
        perm = rng.permutation(B)
        args = (dev(batch['obs_t'][perm]), dev(batch['act'][perm], torch.int32), dev(batch['rew'][perm]),
                dev(batch['obs_tp1'][perm]), dev(batch['done'][perm]), dev(batch['w'][perm]))
        ta = qa.train_dev(*args, graph=True).clone()
        tb = qb.train_dev(*args, graph=False).clone()
        assert torch.equal(ta, tb), step
        assert torch.equal(qa.params, qb.params) and torch.equal(qa.adam_m, qb.adam_m) and torch.equal(qa.adam_v, qb.adam_v), step
        if step == 2:
            qa.update_target()
            qb.update_target()
    assert qa._graphs[B]['graph'] not in (None, False), 'the step was captured and replayed'
    assert float((qa.params - torch.from_numpy(qa.get_flat_params()).cuda()).abs().max()) == 0.0


@pytest.mark.parametrize('name,B', [('conv_only_dueling', 32), ('conv_only_small', 48), ('mlp_plain', 7), ('nature_cnn', 100), ('conv_only_wide', 20)])
def test_small_batch_kernels_agree_with_the_tile_engines(name, B):
    """The four-kernel form of the dueling heads (csrc/qheads.hip.h: fp32 matrix pipe, K split over workgroups, direct weight-gradient
    stores), the one-launch data gradient into the latent and the skinny-tile conv kernels (csrc/convskinny.hip.h) against the
    layer-by-layer tile engines (options dqn_heads / dqn_latdgrad / conv_skinny = 0):
    the same sums in another order -- q values, TD errors and every gradient tensor within 2e-6 of the tensor's scale."""
    from baselines_amd import _lib
    qm, _, _, b = _pair(name, B, 5)
    dev = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x)).cuda().to(dt)
    o12 = torch.cat([qm._obs(b['obs_t']), qm._obs(b['obs_tp1'])], dim=0)         # back to back: the merged online pass of 2 B rows
    a, r, d, w = dev(b['act'], torch.int32), dev(b['rew'], torch.float32), dev(b['done'], torch.float32), dev(b['w'], torch.float32)
    out = {}
    try:
        for mode in (1, 0):
            for o in ('dqn_heads', 'dqn_latdgrad', 'conv_skinny'):
                _lib.set_option(o, mode)
            td = torch.empty(B, dtype=torch.float32, device='cuda')
            qm.grads.zero_()
            _lib.check(qm.lib.mrl_qnet_td_grad(qm.handle, _lib.ptr(qm.params), _lib.ptr(qm.target), _lib.ptr(o12[:B]), _lib.ptr(a), _lib.ptr(r),
                                               _lib.ptr(o12[B:]), _lib.ptr(d), _lib.ptr(w), 0.99, 1, B, _lib.ptr(qm.grads), _lib.ptr(td),
                                               _lib.ptr(qm._loss), _lib.ptr(qm.workspace), qm.workspace.numel(), _lib.stream_ptr()),
                       'mrl_qnet_td_grad')
            out[mode] = (qm.q_values(b['obs_t']).copy(), td.cpu().numpy(), qm.grads.cpu().numpy().copy(), float(qm._loss.cpu()))
    finally:
        for o in ('dqn_heads', 'dqn_latdgrad', 'conv_skinny'):
            _lib.set_option(o, 1)
    (q1, td1, g1, l1), (q0, td0, g0, l0) = out[1], out[0]
    assert np.abs(q1 - q0).max() <= 2e-6 * max(1.0, np.abs(q0).max())
    assert np.abs(td1 - td0).max() <= 4e-6 * max(1.0, np.abs(td0).max())
    assert abs(l1 - l0) <= 2e-6 * max(1.0, abs(l0))
    for t in qm.tensors:
        sl = slice(t['offset'], t['offset'] + t['size'])
        sc = max(np.abs(g0[sl]).max(), 1e-3 * np.abs(g0).max())
        assert np.abs(g1[sl] - g0[sl]).max() <= 4e-6 * sc, (t['name'], np.abs(g1[sl] - g0[sl]).max(), sc)
    assert np.abs(g1).max() > 0


def test_sampling_into_the_graph_inputs_equals_the_copying_route():
    """`sample_dev(..., out=model.graph_inputs(B))` gathers the minibatch (one launch) straight into the captured optimizer step's static
    buffers; `train_dev` then copies nothing.  Same indices, same step: parameters and priorities equal the route through fresh tensors
    bit for bit."""
    from baselines_amd.deepq import PrioritizedReplayBuffer
    B, n = 16, 200
    runs = []
    for zero_copy in (False, True):
        qm, _, _, _ = _pair('conv_only_small', B, 11)
        ob_shape = CASES['conv_only_small']['ob'].shape
        rng = np.random.RandomState(3)
        random.seed(5)
        buf = PrioritizedReplayBuffer(256, alpha=0.6)
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
        buf.add_batch(dev(rng.randint(0, 256, (n,) + ob_shape).astype(np.uint8)), dev(rng.randint(0, 4, n).astype(np.int32)),
                      dev(rng.randn(n).astype(np.float32)), dev(rng.randint(0, 256, (n,) + ob_shape).astype(np.uint8)),
                      dev((rng.rand(n) < 0.1).astype(np.float32)))
        tds = []
        for step in range(4):
            out = qm.graph_inputs(B) if zero_copy else None
            o1, a, r, o2, d, w, idx = buf.sample_dev(B, 0.4, out=out)
            if zero_copy:
                assert o1.data_ptr() == out['o1'].data_ptr() and w.data_ptr() == out['w'].data_ptr()
            td = qm.train_dev(o1, a, r, o2, d, w)
            buf.update_priorities_from_td(idx, td)
            tds.append(td.cpu().numpy().copy())
        runs.append((qm.get_flat_params(), np.stack(tds), buf.trees_numpy()))
    np.testing.assert_array_equal(runs[0][0], runs[1][0])
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
    np.testing.assert_array_equal(runs[0][2][0], runs[1][2][0])


def test_prioritized_buffer_device_path_keeps_the_running_max_on_the_device():
    """`add` after `update_priorities_from_td`: the new leaves are max_priority ** alpha with the maximum the device path
    maintains (replay_buffer.py:100-105, 191) -- no host read-back; the trees equal the host path's to 1e-12 relative."""
    from baselines_amd.deepq import PrioritizedReplayBuffer
    rng = np.random.RandomState(4)
    bufs = [PrioritizedReplayBuffer(32, alpha=0.7) for _ in range(2)]
    ob = lambda: rng.randint(0, 256, (4, 4, 1)).astype(np.uint8)
    for _ in range(20):
        o1, o2, a, r = ob(), ob(), int(rng.randint(3)), float(rng.randn())
        for b in bufs:
            b.add(o1, a, r, o2, 0.0)
    idx = np.array([3, 7, 7, 11, 19], dtype=np.int32)
    td = np.array([0.5, -2.5, 1.25, 0.01, -0.75], dtype=np.float32)
    bufs[0].update_priorities_from_td(torch.from_numpy(idx).cuda(), torch.from_numpy(td).cuda(), eps=1e-6)
    bufs[1].update_priorities([int(i) for i in idx], np.abs(td).astype(np.float64) + 1e-6)
    for _ in range(6):                                   # these slots get max_priority ** alpha = 2.500001 ** 0.7
        o1, o2 = ob(), ob()
        for b in bufs:
            b.add(o1, 1, 0.5, o2, 0.0)
    (s0, m0), (s1, m1) = bufs[0].trees_numpy(), bufs[1].trees_numpy()
    np.testing.assert_allclose(s0, s1, rtol=1e-12, atol=0)
    fin = np.isfinite(m1)
    assert (np.isfinite(m0) == fin).all()
    np.testing.assert_allclose(m0[fin], m1[fin], rtol=1e-12, atol=0)
    assert abs(s0[32 + 25] - 2.500001 ** 0.7) < 1e-12


def test_param_noise_perturbs_only_the_heads_and_adapts_its_scale_by_the_policy_kl():
    """build_act_with_param_noise (deepq/build_graph.py:202-315): noise N(0, scale) on the fully_connected variables only
    (default_param_noise_filter, :131-143), greedy actions of the perturbed copy, the copy re-drawn on `reset`, scale
    x1.01 / /1.01 by the mean KL between the unperturbed and an adaptively perturbed policy against the threshold"""
    from baselines_amd.deepq import QModel, build_q_func
    np.random.seed(5)
    torch.manual_seed(5)
    ob, nact = Box(-1.0, 1.0, (12,), np.float32), 5
    qm = QModel(build_q_func('mlp', hiddens=(32,), dueling=True), ob, nact, max_batch=64, param_noise=True)
    assert qm.param_noise_scale == np.float32(0.01) and qm.param_noise_threshold == np.float32(0.05)
    obs = np.random.RandomState(0).randn(64, 12).astype(np.float32)
    base = qm.get_flat_params()
    # before the first reset the acting copy is unperturbed; the call itself draws the perturbation (an update op)
    a0 = qm.act(obs, reset=True, update_param_noise_threshold=-1.0, stochastic=False)
    assert np.array_equal(a0, qm.q_values(obs).argmax(axis=1))
    assert qm.param_noise_threshold == np.float32(0.05)                     # a negative value leaves the threshold alone
    pert = qm.perturbed.cpu().numpy()
    noisy = np.zeros(qm.P, bool)
    for t in qm.tensors:
        sl = slice(t['offset'], t['offset'] + t['size'])
        if 'fully_connected' in t['name']:
            noisy[sl] = True
        else:
            assert 'mlp_fc' in t['name'] and np.array_equal(pert[sl], base[sl]), t['name']
    d = (pert - base)[noisy]
    assert noisy.sum() > 1000 and abs(d.std() / 0.01 - 1) < 0.1 and abs(d.mean()) < 1e-3 and np.all(d != 0)
    assert np.array_equal(qm.get_flat_params(), base)                       # the online network itself is untouched
    # the next call acts greedily on the perturbed copy (and, without reset, keeps it)
    qp = qm._values_with(qm.perturbed, qm._obs(obs), 64).cpu().numpy()
    a1 = qm.act(obs, update_param_noise_threshold=-1.0, stochastic=False)
    assert np.array_equal(a1, qp.argmax(axis=1)) and np.array_equal(qm.perturbed.cpu().numpy(), pert)
    # the KL kernel against the defining formula (float64), on a strongly perturbed adaptive copy
    qm.param_noise_scale = np.float32(0.5)
    qm.act(obs, update_param_noise_threshold=1e9, update_param_noise_scale=True, stochastic=False)
    qa = qm.q_values(obs).astype(np.float64)
    qb = qm._values_with(qm.adaptive, qm._obs(obs), 64).cpu().numpy().astype(np.float64)
    lsm = lambda q: q - q.max(1, keepdims=True) - np.log(np.exp(q - q.max(1, keepdims=True)).sum(1, keepdims=True))
    kl = (np.exp(lsm(qa)) * (lsm(qa) - lsm(qb))).sum(1).mean()
    assert kl > 1e-3 and abs(qm.last_kl - kl) < 2e-6 * max(1.0, kl)
    # that call compared the KL with the threshold of BEFORE its own assign (0.05): above -> the scale shrank;
    # the threshold is now 1e9, so the next call grows it
    assert kl > 0.05 and qm.param_noise_scale == np.float32(np.float32(0.5) / np.float32(1.01))
    assert qm.param_noise_threshold == np.float32(1e9)
    s = qm.param_noise_scale
    qm.act(obs, update_param_noise_threshold=0.0, update_param_noise_scale=True)
    assert qm.param_noise_scale == np.float32(s * np.float32(1.01)) and qm.param_noise_threshold == 0
    s = qm.param_noise_scale
    qm.act(obs, update_param_noise_threshold=-1.0, update_param_noise_scale=True)      # KL < 0 is false -> shrink
    assert qm.param_noise_scale == np.float32(s / np.float32(1.01))
    # checkpoint names of the extra graph state
    names = set(qm.variables())
    assert {'deepq/param_noise_scale:0', 'deepq/param_noise_threshold:0',
            'deepq/perturbed_q_func/action_value/fully_connected/weights:0',
            'deepq/adaptive_q_func/state_value/fully_connected_1/biases:0'} <= names
    # a model without param_noise keeps the plain signature and refuses the extra arguments
    qm2 = QModel(build_q_func('mlp', hiddens=(32,), dueling=True), ob, nact, max_batch=64)
    assert qm2.act(obs, False).shape == (64,)
    with pytest.raises(TypeError):
        qm2.act(obs, reset=True)


def test_deepq_learn_identity_env_with_param_noise():
    """deepq.learn(param_noise=True) (deepq/deepq.py:263-275): exploration by perturbed heads instead of eps-greedy, the
    KL threshold following the eps schedule; same functional bar as the eps-greedy run"""
    from baselines_amd import deepq
    env = DiscreteIdentityEnv(5, 50, seed=0)
    scales = []

    def cb(lcl, _glb):
        scales.append(float(lcl['model'].param_noise_scale))
        return False

    act = deepq.learn(env, network='mlp', seed=0, gamma=0.9, total_timesteps=4000, lr=2e-3, buffer_size=2000,
                      exploration_fraction=0.3, exploration_final_eps=0.02, learning_starts=200, target_network_update_freq=100,
                      param_noise=True, print_freq=None, checkpoint_freq=None, callback=cb, hiddens=[32], num_hidden=32)
    m = act._model
    assert m.eps == 0.0                                          # eps-greedy is off (update_eps = 0 every step)
    # at eps = 1 the threshold is -log(1/n) = 1.61: the scale climbs from 0.01; at the end (eps = 0.02) it is 0.016 and the
    # scale has come down again
    assert max(scales) > 0.05 and scales[-1] < max(scales) and abs(float(m.param_noise_threshold) + np.log(1 - 0.02 + 0.02 / 5)) < 1e-6
    test_env = DiscreteIdentityEnv(5, 1000, seed=1)
    obs, total = test_env.reset(), 0.0
    for _ in range(100):
        a = int(m.q_values(np.array(obs)[None]).argmax())        # the unperturbed greedy policy
        obs, rew, _, _ = test_env.step(a)
        total += rew
    assert total >= 90, total
