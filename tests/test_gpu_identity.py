"""GPU: functional learning tests in the spirit of the reference's common/tests/test_identity.py (identity envs: the
reward is 1 when the action repeats the observation).  Exercises Discrete observation spaces (one-hot encoding,
common/input.py:57-58), host envs through DummyVecEnv + VecMonitor, and the whole learn() loop."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from baselines_amd.common.spaces import Discrete                         # noqa: E402
from baselines_amd.common.vec_env import DummyVecEnv, VecMonitor         # noqa: E402


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


def test_discrete_identity_is_learnt():
    from baselines_amd import ppo2
    env = VecMonitor(DummyVecEnv([lambda: DiscreteIdentityEnv(6, 50, seed=0)]))
    model = ppo2.learn(network='mlp', env=env, total_timesteps=6000, seed=0, nsteps=64, lr=1e-3, ent_coef=0.0,
                       gamma=0.5, log_interval=1000)
    # evaluation as in the reference's simple_test (common/tests/util.py:12-40): 100 sampled steps, >= 90 % reward
    test_env = DummyVecEnv([lambda: DiscreteIdentityEnv(6, 1000, seed=1)])
    obs, total = test_env.reset(), 0.0
    for _ in range(100):
        a, _, _, _ = model.step(obs)
        assert a.dtype == np.int64 and a.shape == (1,)
        obs, rew, _, _ = test_env.step(a)
        total += float(rew[0])
    assert total >= 90, total
    assert env.epcount >= 100 and env.venv.buf_obs.dtype == np.int64        # VecMonitor counted the training episodes
