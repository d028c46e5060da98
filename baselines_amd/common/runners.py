"""Environment cursor shared by rollout collectors.

API kept from the reference (`baselines.common.runners.AbstractEnvRunner`, common/runners.py:4-19):
constructor keywords (env, model, nsteps) and the attributes a Runner subclass relies on --
`obs`, `dones`, `states`, `nenv`, `nsteps`, `batch_ob_shape`.  What is different here: the cursor
knows about DEVICE-RESIDENT environments (observations stay torch tensors in HBM) next to the
classic host (NumPy) ones.
"""
import abc

import numpy as np


def _first_observations(env, nenv):
    """reset() the env and return the buffer the rollout loop will keep overwriting."""
    if getattr(env, 'device_resident', False):
        return env.reset(), True                     # torch tensor [nenv, ...] living on the GPU
    space = env.observation_space
    host = np.zeros((nenv,) + tuple(space.shape), dtype=space.dtype.name)
    host[:] = env.reset()
    return host, False


class AbstractEnvRunner(abc.ABC):
    def __init__(self, *, env, model, nsteps):
        self.env, self.model, self.nsteps = env, model, nsteps
        self.nenv = getattr(env, 'num_envs', 1)
        self.batch_ob_shape = (self.nenv * nsteps,) + tuple(env.observation_space.shape)
        self.obs, self.device_env = _first_observations(env, self.nenv)
        self.dones = [False] * self.nenv             # "done entering the next step" flags
        self.states = model.initial_state            # None: recurrent policies are out of scope

    @abc.abstractmethod
    def run(self):
        """Collect nsteps transitions per env and return the reference's 8-tuple."""
