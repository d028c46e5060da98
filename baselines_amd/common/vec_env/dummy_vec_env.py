"""DummyVecEnv: N gym-style environments stepped one after the other inside the learner process.

Same contract as the reference's class of that name (common/vec_env/dummy_vec_env.py:5-81), which
its tests use as the equivalence oracle for every other backend (test_vec_env.py:47-111):
float32 rewards, bool dones, per-env info dicts, and AUTO-RESET -- when an env reports done the
observation handed back is already the first one of its next episode.
"""
import numpy as np

from .vec_env import VecEnv


class DummyVecEnv(VecEnv):
    def __init__(self, env_fns):
        self.envs = [make() for make in env_fns]
        first = self.envs[0]
        super().__init__(len(self.envs), first.observation_space, first.action_space)
        n, space = self.num_envs, first.observation_space
        self._obs = np.zeros((n,) + tuple(space.shape), dtype=space.dtype)
        self._rew = np.zeros(n, dtype=np.float32)
        self._done = np.zeros(n, dtype=np.bool_)
        self._info = [dict() for _ in range(n)]
        self._pending = None
        self.spec = getattr(first, 'spec', None)

    # the reference exposes these buffers under buf_* names; keep them reachable
    buf_obs = property(lambda self: self._obs)
    buf_rews = property(lambda self: self._rew)
    buf_dones = property(lambda self: self._done)
    buf_infos = property(lambda self: self._info)

    def reset(self):
        for slot, env in enumerate(self.envs):
            self._obs[slot] = env.reset()
        return self._obs.copy()

    def step_async(self, actions):
        try:
            batched = len(actions) == self.num_envs
        except TypeError:                     # a bare scalar action
            batched = False
        if not batched:
            if self.num_envs != 1:
                raise AssertionError('actions {} do not match {} environments'.format(actions, self.num_envs))
            actions = [actions]
        self._pending = actions

    def step_wait(self):
        for slot, (env, action) in enumerate(zip(self.envs, self._pending)):
            ob, rew, done, info = env.step(action)
            if done:
                ob = env.reset()
            self._obs[slot], self._rew[slot], self._done[slot], self._info[slot] = ob, rew, done, info
        return self._obs.copy(), self._rew.copy(), self._done.copy(), list(self._info)

    def close_extras(self):
        for env in self.envs:
            closer = getattr(env, 'close', None)
            if closer is not None:
                closer()
