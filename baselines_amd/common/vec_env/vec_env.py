"""Vectorised-environment contract kept from the reference (common/vec_env/vec_env.py:29-138):
attributes num_envs / observation_space / action_space, reset(), step_async(actions),
step_wait() -> (obs[N,...], rews[N], dones[N] bool, infos), step(), close(), and AUTO-RESET on
done (the obs returned for a finished env is the first obs of its next episode).

Device-resident environments additionally set `device_resident = True` and exchange torch device
tensors instead of NumPy arrays, so observations never leave HBM."""
import contextlib
import os
from abc import ABC, abstractmethod


class AlreadySteppingError(Exception):
    def __init__(self):
        Exception.__init__(self, 'already running an async step')


class NotSteppingError(Exception):
    def __init__(self):
        Exception.__init__(self, 'not running an async step')


class VecEnv(ABC):
    closed = False
    viewer = None
    device_resident = False
    metadata = {'render.modes': ['human', 'rgb_array']}

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space

    @abstractmethod
    def reset(self):
        """Reset every env; returns obs [num_envs, ...]."""

    @abstractmethod
    def step_async(self, actions):
        """Start one step with a batch of actions."""

    @abstractmethod
    def step_wait(self):
        """Finish the step: (obs, rews, dones, infos)."""

    def close_extras(self):
        pass

    def close(self):
        if self.closed:
            return
        if self.viewer is not None:
            self.viewer.close()
        self.close_extras()
        self.closed = True

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def render(self, mode='human'):
        raise NotImplementedError('rendering is outside the hot path (SURVEY.md 2.1 row 3)')

    def get_images(self):
        raise NotImplementedError

    @property
    def unwrapped(self):
        if isinstance(self, VecEnvWrapper):
            return self.venv.unwrapped
        return self


class VecEnvWrapper(VecEnv):
    """Wrapper base: forwards step_async, leaves reset/step_wait to the subclass
    (reference: common/vec_env/vec_env.py:141-178)."""

    def __init__(self, venv, observation_space=None, action_space=None):
        self.venv = venv
        super().__init__(num_envs=venv.num_envs,
                         observation_space=observation_space or venv.observation_space,
                         action_space=action_space or venv.action_space)

    @property
    def device_resident(self):
        return getattr(self.venv, 'device_resident', False)

    def step_async(self, actions):
        self.venv.step_async(actions)

    def close(self):
        return self.venv.close()

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError("attempted to get missing private attribute '{}'".format(name))
        return getattr(self.venv, name)


class VecEnvObservationWrapper(VecEnvWrapper):
    @abstractmethod
    def process(self, obs):
        pass

    def reset(self):
        return self.process(self.venv.reset())

    def step_wait(self):
        obs, rews, dones, infos = self.venv.step_wait()
        return self.process(obs), rews, dones, infos


class CloudpickleWrapper(object):
    """Ships env constructors to worker processes (multiprocessing pickles with plain pickle)."""

    def __init__(self, x):
        self.x = x

    def __getstate__(self):
        import cloudpickle
        return cloudpickle.dumps(self.x)

    def __setstate__(self, ob):
        import pickle
        self.x = pickle.loads(ob)


@contextlib.contextmanager
def clear_mpi_env_vars():
    """Workers spawned from a process launched by mpirun / torchrun must not inherit the launcher's
    rendezvous variables (reference: common/vec_env/vec_env.py:207-223 for OMPI_/PMI_)."""
    removed = {}
    for k in list(os.environ.keys()):
        if any(k.startswith(p) for p in ('OMPI_', 'PMI_')):
            removed[k] = os.environ.pop(k)
    try:
        yield
    finally:
        os.environ.update(removed)
