from abc import ABC, abstractmethod

import numpy as np


class AbstractEnvRunner(ABC):
    """Holds the environment cursor between rollouts (reference: common/runners.py:4-19): the
    current observations (`self.obs`, obs dtype of the space), `self.dones` (all False at start)
    and the recurrent `self.states` (always None on the supported hot path)."""

    def __init__(self, *, env, model, nsteps):
        self.env = env
        self.model = model
        self.nenv = nenv = env.num_envs if hasattr(env, 'num_envs') else 1
        self.batch_ob_shape = (nenv * nsteps,) + tuple(env.observation_space.shape)
        self.device_env = bool(getattr(env, 'device_resident', False))
        if self.device_env:
            self.obs = env.reset()            # device tensor [nenv, ...]
        else:
            self.obs = np.zeros((nenv,) + tuple(env.observation_space.shape), dtype=env.observation_space.dtype.name)
            self.obs[:] = env.reset()
        self.nsteps = nsteps
        self.states = model.initial_state
        self.dones = [False for _ in range(nenv)]

    @abstractmethod
    def run(self):
        raise NotImplementedError
