"""
TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by running the REFERENCE's
own code (imported from /root/reference, never copied) in the build container:

  * baselines.ppo2.runner.Runner.run (rollout stacking, GAE runner.py:52-65, sf01 :69-74)
    teacher-forced with a fake env + fake model that replay a seeded synthetic rollout;
  * baselines.common.segment_tree.{Sum,Min}SegmentTree and
    baselines.deepq.replay_buffer.PrioritizedReplayBuffer (loaded by file path because
    deepq/__init__.py imports tensorflow) driven by a seeded op stream;
  * baselines.common.math_util.explained_variance, baselines.common.schedules;
  * baselines/a2c/utils.py:20-35 ``ortho_init`` -- pure NumPy inside a module that imports
    tensorflow, so the FUNCTION's source is cut out of the file with ``ast`` and exec'd as it
    stands; seeded outputs -> ortho_init.npz.

The reference cannot travel to the GPU box, so the vectors are committed.
Run:  python oracle/make_golden.py      (needs /root/reference; a stub `gym` module is
injected because baselines/common/__init__.py imports gym at import time).
"""
import importlib.util
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get('BASELINES_REFERENCE', '/root/reference')
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)


def _import_reference():
    if 'gym' not in sys.modules:
        sys.modules['gym'] = types.ModuleType('gym')
    sys.path.insert(0, REF)
    from baselines.ppo2.runner import Runner, sf01  # noqa
    from baselines.common import segment_tree, math_util, schedules  # noqa
    spec = importlib.util.spec_from_file_location(
        'ref_replay_buffer', os.path.join(REF, 'baselines', 'deepq', 'replay_buffer.py'))
    rb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rb)
    return Runner, sf01, segment_tree, math_util, schedules, rb


class _Space(object):
    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)


class ReplayEnv(object):
    """Fake VecEnv replaying a pre-generated rollout (teacher forcing)."""

    def __init__(self, ro):
        self.ro = ro
        self.num_envs = ro['obs'].shape[1]
        self.observation_space = _Space(ro['obs'].shape[2:], ro['obs'].dtype)
        self.t = 0

    def reset(self):
        self.t = 0
        return self.ro['obs'][0]

    def step(self, actions):
        ro, t = self.ro, self.t
        T = ro['obs'].shape[0]
        rew = ro['rewards'][t]
        if t + 1 < T:
            obs, done = ro['obs'][t + 1], ro['dones'][t + 1]
        else:
            obs, done = ro['final_obs'], ro['last_dones']
        self.t += 1
        return obs, rew, done, [{} for _ in range(self.num_envs)]


class ReplayModel(object):
    initial_state = None

    def __init__(self, ro):
        self.ro, self.t = ro, 0

    def step(self, obs, S=None, M=None):
        ro, t = self.ro, self.t
        assert np.array_equal(obs, ro['obs'][t])
        self.t += 1
        return ro['actions'][t], ro['values'][t], None, ro['neglogpacs'][t]

    def value(self, obs, S=None, M=None):
        return self.ro['last_values']


def small_rollout(T, N, seed, ob_shape, ob_dtype, act_kind, done_p):
    rng = np.random.RandomState(seed)
    if np.dtype(ob_dtype) == np.uint8:
        obs = rng.randint(0, 256, (T, N) + ob_shape).astype(np.uint8)
        final_obs = rng.randint(0, 256, (N,) + ob_shape).astype(np.uint8)
    else:
        obs = rng.randn(*((T, N) + ob_shape)).astype(np.float32)
        final_obs = rng.randn(*((N,) + ob_shape)).astype(np.float32)
    if act_kind == 'discrete':
        actions = rng.randint(0, 6, (T, N)).astype(np.int64)
    else:
        actions = rng.randn(T, N, 3).astype(np.float32)
    rewards = (rng.randn(T, N) * 3).astype(np.float32)
    values = (rng.randn(T, N) * 2).astype(np.float32)
    neglogp = rng.rand(T, N).astype(np.float32) + 1
    dones = rng.rand(T, N) < done_p
    dones[0] = False
    last_dones = rng.rand(N) < max(done_p, 0.3)
    last_values = rng.randn(N).astype(np.float32)
    return dict(obs=obs, final_obs=final_obs, actions=actions, rewards=rewards, values=values,
                neglogpacs=neglogp, dones=dones, last_dones=last_dones, last_values=last_values)


RUNNER_CASES = [
    # name, T, N, seed, ob_shape, ob_dtype, act_kind, done_p, gamma, lam
    ('t128_n8', 128, 8, 0, (4,), 'float32', 'discrete', 0.05, 0.99, 0.95),
    ('t5_n3', 5, 3, 1, (2, 2, 1), 'uint8', 'discrete', 0.3, 0.99, 0.95),
    ('t37_n70', 37, 70, 2, (3,), 'float32', 'box', 0.1, 0.99, 0.95),
    ('t1_n4', 1, 4, 3, (3,), 'float32', 'discrete', 0.5, 0.99, 0.95),
    ('t64_n129_alldone', 64, 129, 4, (1,), 'uint8', 'discrete', 1.0, 0.99, 0.95),
    ('t64_n65_nodone', 64, 65, 5, (1,), 'float32', 'box', 0.0, 0.99, 0.95),
    ('t300_n33_g1', 300, 33, 6, (2,), 'float32', 'discrete', 0.02, 1.0, 1.0),
    ('t16_n64_lam0', 16, 64, 7, (2,), 'float32', 'discrete', 0.2, 0.9, 0.0),
]


def gen_runner(Runner):
    for name, T, N, seed, oshape, odt, akind, dp, gamma, lam in RUNNER_CASES:
        ro = small_rollout(T, N, seed, oshape, odt, akind, dp)
        env, model = ReplayEnv(ro), ReplayModel(ro)
        runner = Runner(env=env, model=model, nsteps=T, gamma=gamma, lam=lam)
        obs, returns, masks, actions, values, neglogpacs, states, epinfos = runner.run()
        assert states is None and epinfos == []
        assert returns.dtype == np.float32 and masks.dtype == np.bool_
        np.savez_compressed(os.path.join(OUT, 'runner_%s.npz' % name),
                            gamma=gamma, lam=lam,
                            **{'in_' + k: v for k, v in ro.items()},
                            out_obs=obs, out_returns=returns, out_masks=masks, out_actions=actions,
                            out_values=values, out_neglogpacs=neglogpacs)
        print('runner', name, obs.shape, obs.dtype, returns[:3])


def gen_shuffle():
    """ppo2.py:157-165 index stream for (seed, nbatch, nminibatches, noptepochs)."""
    out = {}
    for seed, nbatch, nmb, nep in [(0, 1024, 4, 4), (3, 96, 32, 2), (7, 640, 1, 3)]:
        np.random.seed(seed)
        inds = np.arange(nbatch)
        rows = []
        for _ in range(nep):
            np.random.shuffle(inds)
            for start in range(0, nbatch, nbatch // nmb):
                rows.append(inds[start:start + nbatch // nmb].copy())
        out['s%d_b%d_m%d_e%d' % (seed, nbatch, nmb, nep)] = np.stack(rows)
    np.savez_compressed(os.path.join(OUT, 'shuffle.npz'), **out)


def gen_misc(math_util, schedules):
    rng = np.random.RandomState(11)
    y = rng.randn(1000).astype(np.float32)
    yp = (y + 0.3 * rng.randn(1000)).astype(np.float32)
    ev = math_util.explained_variance(yp, y)
    ls = schedules.LinearSchedule(schedule_timesteps=1000, final_p=0.02, initial_p=1.0)
    lsv = np.array([ls.value(t) for t in (0, 1, 500, 999, 1000, 5000)], np.float64)
    np.savez_compressed(os.path.join(OUT, 'misc.npz'), ev_y=y, ev_ypred=yp, ev=np.float64(ev),
                        ev_const=np.float64(np.nan if True else 0), linsched=lsv)


def gen_replay(segment_tree, rb):
    """Drive the reference PrioritizedReplayBuffer with a seeded op stream; record every
    sampled (idxes, weights) and the final tree arrays.  Python `random` supplies the
    stratified uniforms (replay_buffer.py:112) -> we record them by reseeding."""
    cap, alpha, batch = 64, 0.6, 8
    rng = np.random.RandomState(5)
    buf = rb.PrioritizedReplayBuffer(cap, alpha)
    ops, samples = [], []
    random.seed(123)
    nadd = 150
    for i in range(nadd):
        ob = rng.randint(0, 256, (4,)).astype(np.uint8)
        ob2 = rng.randint(0, 256, (4,)).astype(np.uint8)
        a, r, d = int(rng.randint(0, 6)), float(rng.randn()), float(rng.rand() < 0.1)
        # 0-d ndarray: replay_buffer.py:39 uses np.array(action, copy=False), which NumPy 2
        # rejects for python ints (needs a copy) but accepts for ndarrays
        buf.add(ob, np.array(a), r, ob2, d)
        ops.append((ob, a, r, ob2, d))
        if i >= 20 and i % 5 == 0:
            beta = 0.4 + 0.6 * i / nadd
            st = random.getstate()
            u = [random.random() for _ in range(batch)]
            random.setstate(st)
            res = buf.sample(batch, beta)
            obs_t, act, rew, obs_tp1, done, w, idx = res
            newp = np.abs(rng.randn(batch)) + 1e-6
            buf.update_priorities(idx, newp)
            samples.append(dict(i=i, beta=beta, u=np.array(u), idx=np.array(idx), w=np.asarray(w),
                                obs_t=obs_t, act=act, rew=rew, obs_tp1=obs_tp1, done=done, newp=newp))
    flat = {}
    flat['add_obs'] = np.stack([o[0] for o in ops])
    flat['add_act'] = np.array([o[1] for o in ops])
    flat['add_rew'] = np.array([o[2] for o in ops])
    flat['add_obs2'] = np.stack([o[3] for o in ops])
    flat['add_done'] = np.array([o[4] for o in ops])
    for j, s in enumerate(samples):
        for k, v in s.items():
            flat['s%d_%s' % (j, k)] = np.asarray(v)
    flat['nsamples'] = len(samples)
    flat['cap'], flat['alpha'], flat['batch'] = cap, alpha, batch
    flat['final_sum_tree'] = np.array(buf._it_sum._value, np.float64)
    flat['final_min_tree'] = np.array(buf._it_min._value, np.float64)
    flat['final_max_priority'] = buf._max_priority
    np.savez_compressed(os.path.join(OUT, 'replay.npz'), **flat)

    # segment tree known-answer stream (generalises common/tests/test_segment_tree.py)
    st = segment_tree.SumSegmentTree(16)
    mt = segment_tree.MinSegmentTree(16)
    rng = np.random.RandomState(9)
    log = []
    for _ in range(200):
        i, v = int(rng.randint(0, 16)), float(rng.rand() * 3)
        st[i] = v
        mt[i] = v
        a = int(rng.randint(0, 16))
        b = int(rng.randint(a + 1, 17))
        ps = float(rng.rand()) * st.sum(0, 16) * 0.999
        log.append((i, v, a, b, st.sum(a, b), mt.min(a, b), ps, st.find_prefixsum_idx(ps)))
    np.savez_compressed(os.path.join(OUT, 'segment_tree.npz'), log=np.array(log, np.float64))


ORTHO_SHAPES = [((8, 8, 4, 32), 2 ** 0.5), ((3136, 512), 2 ** 0.5), ((512, 6), 0.01), ((376, 64), 2 ** 0.5)]
# TF variable-creation order of nature_cnn + heads (models.py:15-26, policies.py:49,63) and of
# mlp(2x64) with value_network='copy' on a Box(376,) / 17-dim Gaussian policy (policies.py:141-166)
ORTHO_NETS = {
    'nature_cnn_nact6': [((8, 8, 4, 32), 2 ** 0.5), ((4, 4, 32, 64), 2 ** 0.5), ((3, 3, 64, 64), 2 ** 0.5),
                         ((3136, 512), 2 ** 0.5), ((512, 6), 0.01), ((512, 1), 1.0)],
    'mlp_copy_376_17': [((376, 64), 2 ** 0.5), ((64, 64), 2 ** 0.5), ((376, 64), 2 ** 0.5), ((64, 64), 2 ** 0.5),
                        ((64, 17), 0.01), ((64, 1), 1.0)],
}
ORTHO_STRIDE = 37      # large tensors are committed as every 37th entry + float64 sums


def reference_ortho_init():
    """The reference's own ``ortho_init`` (a2c/utils.py:20-35), cut out of its source file by name and
    exec'd unmodified with only ``np`` in scope -- the enclosing module imports tensorflow and cannot
    be imported here."""
    import ast
    path = os.path.join(REF, 'baselines', 'a2c', 'utils.py')
    src = open(path).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'ortho_init']
    assert len(fn) == 1
    ns = {'np': np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, 'exec'), ns)
    return ns['ortho_init']


def ortho_digest(w):
    f = w.reshape(-1)
    return f[::ORTHO_STRIDE].copy(), np.array([f.astype(np.float64).sum(), np.abs(f.astype(np.float64)).sum()])


def gen_ortho():
    ref = reference_ortho_init()
    out = {}
    for seed in (0, 1):
        for shape, scale in ORTHO_SHAPES:
            np.random.seed(seed)
            w = ref(scale)(shape, np.float32)
            assert w.dtype == np.float32 and w.shape == tuple(shape)
            key = 's%d_%s' % (seed, 'x'.join(map(str, shape)))
            if w.size <= 20000:
                out[key + '_full'] = w
            out[key + '_sample'], out[key + '_sums'] = ortho_digest(w)
    for name, layers in ORTHO_NETS.items():
        np.random.seed(3)
        for i, (shape, scale) in enumerate(layers):
            w = ref(scale)(shape, np.float32)
            out['%s_%d_sample' % (name, i)], out['%s_%d_sums' % (name, i)] = ortho_digest(w)
        out[name + '_next_uniform'] = np.array([np.random.uniform()])     # the stream position afterwards
    np.savez_compressed(os.path.join(OUT, 'ortho_init.npz'), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    Runner, sf01, segment_tree, math_util, schedules, rb = _import_reference()
    gen_runner(Runner)
    gen_shuffle()
    gen_misc(math_util, schedules)
    gen_replay(segment_tree, rb)
    gen_ortho()
    print('golden vectors written to', OUT)


if __name__ == '__main__':
    main()
