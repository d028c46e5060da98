"""In-process VecEnv: steps the envs one after the other.  The equivalence oracle for every other
backend (reference: common/vec_env/dummy_vec_env.py:5-81, test_vec_env.py:47-111)."""
import numpy as np

from .vec_env import VecEnv


class DummyVecEnv(VecEnv):
    def __init__(self, env_fns):
        self.envs = [fn() for fn in env_fns]
        env = self.envs[0]
        VecEnv.__init__(self, len(self.envs), env.observation_space, env.action_space)
        shape, dtype = tuple(env.observation_space.shape), env.observation_space.dtype
        self.buf_obs = np.zeros((self.num_envs,) + shape, dtype=dtype)
        self.buf_dones = np.zeros((self.num_envs,), dtype=np.bool_)
        self.buf_rews = np.zeros((self.num_envs,), dtype=np.float32)
        self.buf_infos = [{} for _ in range(self.num_envs)]
        self.actions = None
        self.spec = getattr(env, 'spec', None)

    def step_async(self, actions):
        listify = True
        try:
            if len(actions) == self.num_envs:
                listify = False
        except TypeError:
            pass
        if listify:
            assert self.num_envs == 1, 'actions {} do not match {} environments'.format(actions, self.num_envs)
            actions = [actions]
        self.actions = actions

    def step_wait(self):
        for e in range(self.num_envs):
            obs, self.buf_rews[e], self.buf_dones[e], self.buf_infos[e] = self.envs[e].step(self.actions[e])
            if self.buf_dones[e]:
                obs = self.envs[e].reset()      # auto-reset
            self.buf_obs[e] = obs
        return np.copy(self.buf_obs), np.copy(self.buf_rews), np.copy(self.buf_dones), list(self.buf_infos)

    def reset(self):
        for e in range(self.num_envs):
            self.buf_obs[e] = self.envs[e].reset()
        return np.copy(self.buf_obs)

    def close_extras(self):
        for env in self.envs:
            if hasattr(env, 'close'):
                env.close()
