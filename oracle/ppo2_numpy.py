"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

NumPy restatement of the NumPy / pure-Python parts of the reference PPO2 hot
path.  Every function cites the reference lines it follows
(paths relative to /root/reference/baselines/).  Pinned bit-exact against the
reference's own code by tests/test_oracle_golden.py (fixtures produced by
oracle/make_golden.py from the real ``Runner.run``).
"""
import random

import numpy as np


def set_global_seeds(i):
    """common/misc_util.py:48-62 -- ``import MPI`` can never succeed there, so
    rank is always 0 and every rank seeds NumPy / random with the raw seed."""
    np.random.seed(i)
    random.seed(i)


def ortho_init(shape, scale=1.0):
    """a2c/utils.py:20-35 (lasagne orthogonal init; consumes np.random)."""
    shape = tuple(shape)
    if len(shape) == 2:
        flat_shape = shape
    elif len(shape) == 4:  # assumes NHWC / HWIO
        flat_shape = (int(np.prod(shape[:-1])), shape[-1])
    else:
        raise NotImplementedError
    a = np.random.normal(0.0, 1.0, flat_shape)
    u, _, v = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == flat_shape else v
    q = q.reshape(shape)
    return (scale * q[:shape[0], :shape[1]]).astype(np.float32)


def gae(mb_rewards, mb_values, mb_dones, last_values, last_dones, gamma, lam):
    """ppo2/runner.py:52-65.  Inputs time-major: rewards/values f32 [T,N],
    dones bool [T,N] (done flag ENTERING step t), last_values f32 [N],
    last_dones bool [N].  Returns (mb_returns f32 [T,N], mb_advs f32 [T,N]).
    The f64 carry / f32 `gamma*nextvalues` product mix comes for free from
    NumPy's promotion rules and is what the HIP kernel must reproduce."""
    nsteps = mb_rewards.shape[0]
    mb_advs = np.zeros_like(mb_rewards)
    lastgaelam = 0
    last_dones = np.asarray(last_dones)
    for t in reversed(range(nsteps)):
        if t == nsteps - 1:
            nextnonterminal = 1.0 - last_dones
            nextvalues = last_values
        else:
            nextnonterminal = 1.0 - mb_dones[t + 1]
            nextvalues = mb_values[t + 1]
        delta = mb_rewards[t] + gamma * nextvalues * nextnonterminal - mb_values[t]
        mb_advs[t] = lastgaelam = delta + gamma * lam * nextnonterminal * lastgaelam
    mb_returns = mb_advs + mb_values
    return mb_returns, mb_advs


def sf01(arr):
    """ppo2/runner.py:69-74: swap and flatten axes 0,1 -> env-major flat i = e*T + t."""
    s = arr.shape
    return arr.swapaxes(0, 1).reshape(s[0] * s[1], *s[2:])


def minibatch_indices(nbatch, nbatch_train, noptepochs):
    """ppo2/ppo2.py:157-165: per epoch np.random.shuffle(inds) on the GLOBAL NumPy
    stream, then contiguous slices.  Yields int64 index arrays (copies)."""
    inds = np.arange(nbatch)
    for _ in range(noptepochs):
        np.random.shuffle(inds)
        for start in range(0, nbatch, nbatch_train):
            yield inds[start:start + nbatch_train].copy()


def normalize_advantages(returns, values):
    """ppo2/model.py:136-139 (per minibatch, population std, f32)."""
    advs = returns - values
    return (advs - advs.mean()) / (advs.std() + 1e-8)


def explained_variance(ypred, y):
    """common/math_util.py:25-38."""
    assert y.ndim == 1 and ypred.ndim == 1
    vary = np.var(y)
    return np.nan if vary == 0 else 1 - np.var(y - ypred) / vary


class RunningMeanStd(object):
    """common/running_mean_std.py:5-33 (f64 stats, count starts at 1e-4)."""

    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape, 'float64')
        self.var = np.ones(shape, 'float64')
        self.count = epsilon

    def update(self, x):
        batch_mean = np.mean(x, axis=0)
        batch_var = np.var(x, axis=0)
        batch_count = x.shape[0]
        self.update_from_moments(batch_mean, batch_var, batch_count)

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        tot_count = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot_count
        m_a = self.var * self.count
        m_b = batch_var * batch_count
        M2 = m_a + m_b + np.square(delta) * self.count * batch_count / tot_count
        self.mean, self.var, self.count = new_mean, M2 / tot_count, tot_count


def synthetic_rollout(kind, T, N, seed):
    """SURVEY.md 8(d) seeded synthetic rollout (time-major), identical for oracle and GPU.
    kind: 'atari' (u8 84x84x4, Discrete(6)) | 'mujoco' (f32 376, 17-dim Gaussian) |
          'cartpole' (f32 4, Discrete(2))."""
    rng = np.random.RandomState(seed)
    if kind == 'atari':
        obs = rng.randint(0, 256, (T, N, 84, 84, 4)).astype(np.uint8)
        actions = rng.randint(0, 6, (T, N)).astype(np.int64)
        rewards = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), size=(T, N), p=[.05, .9, .05]).astype(np.float32)
        dones = rng.rand(T, N) < (1.0 / 200)
        neglogp = (-np.log(1.0 / 6) + 0.1 * rng.randn(T, N)).astype(np.float32)
    elif kind == 'mujoco':
        obs = np.clip(rng.randn(T, N, 376), -10, 10).astype(np.float32)
        actions = rng.randn(T, N, 17).astype(np.float32)
        rewards = np.clip(rng.randn(T, N), -10, 10).astype(np.float32)
        dones = rng.rand(T, N) < (1.0 / 1000)
        neglogp = (17 * 0.5 * np.log(2 * np.pi) + 0.5 * rng.chisquare(17, (T, N))).astype(np.float32)
    elif kind == 'cartpole':
        obs = rng.randn(T, N, 4).astype(np.float32)
        actions = rng.randint(0, 2, (T, N)).astype(np.int64)
        rewards = np.ones((T, N), np.float32)
        dones = rng.rand(T, N) < (1.0 / 20)
        neglogp = (-np.log(0.5) + 0.05 * rng.randn(T, N)).astype(np.float32)
    else:
        raise ValueError(kind)
    values = rng.randn(T, N).astype(np.float32)
    last_values = rng.randn(N).astype(np.float32)
    last_dones = rng.rand(N) < 0.01
    dones[0] = False  # runners.py:13: initial dones are all False
    return dict(obs=obs, actions=actions, rewards=rewards, dones=dones, values=values,
                neglogpacs=neglogp, last_values=last_values, last_dones=last_dones)
