"""The vectorised-environment contract.

Names and semantics follow the reference's `VecEnv` family (common/vec_env/vec_env.py:29-223)
because that IS the drop-in boundary on the actor side:

    num_envs, observation_space, action_space          attributes read by Runner / build_policy
    reset() -> obs[N, ...]
    step_async(actions); step_wait() -> (obs[N, ...], rews[N], dones[N] bool, infos)
    step(actions) = step_async + step_wait;  close();  unwrapped
    AUTO-RESET: a finished env's returned obs is the first obs of its next episode.

New here: `device_resident`.  An env that sets it exchanges torch DEVICE tensors (obs, rews, dones
and the actions it is given) so a rollout never crosses PCIe; Runner checks the flag and writes the
step straight into the HBM rollout buffer.
"""
import abc
import contextlib
import os


class _StepProtocolError(RuntimeError):
    message = 'vectorised env stepping protocol violated'

    def __init__(self):
        super().__init__(self.message)


class AlreadySteppingError(_StepProtocolError):
    message = 'already running an async step'


class NotSteppingError(_StepProtocolError):
    message = 'not running an async step'


class VecEnv(abc.ABC):
    metadata = {'render.modes': ['human', 'rgb_array']}
    device_resident = False
    closed = False
    viewer = None

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space

    # ---- the three methods a backend must provide
    @abc.abstractmethod
    def reset(self):
        """Restart all envs, return their first observations."""

    @abc.abstractmethod
    def step_async(self, actions):
        """Hand one action per env to the backend; must be followed by step_wait()."""

    @abc.abstractmethod
    def step_wait(self):
        """Complete the pending step -> (obs, rews, dones, infos)."""

    # ---- provided
    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        if not self.closed:
            if self.viewer is not None:
                self.viewer.close()
            self.close_extras()
            self.closed = True

    def close_extras(self):
        """Backend-specific cleanup hook (worker processes, shared memory, ...)."""

    def render(self, mode='human'):
        raise NotImplementedError('rendering is outside the supported hot path')

    def get_images(self):
        raise NotImplementedError('rendering is outside the supported hot path')

    @property
    def unwrapped(self):
        inner = self
        while isinstance(inner, VecEnvWrapper):
            inner = inner.venv
        return inner


class VecEnvWrapper(VecEnv):
    """Decorator base class: owns an inner `venv`, forwards what it does not override."""

    def __init__(self, venv, observation_space=None, action_space=None):
        self.venv = venv
        VecEnv.__init__(self, venv.num_envs,
                        venv.observation_space if observation_space is None else observation_space,
                        venv.action_space if action_space is None else action_space)

    device_resident = property(lambda self: bool(getattr(self.venv, 'device_resident', False)))

    def step_async(self, actions):
        self.venv.step_async(actions)

    def close(self):
        return self.venv.close()

    def __getattr__(self, name):
        # only reached for attributes the wrapper itself lacks
        if name.startswith('_'):
            raise AttributeError("attempted to get missing private attribute '{}'".format(name))
        return getattr(self.venv, name)


class VecEnvObservationWrapper(VecEnvWrapper):
    """Wrapper that only transforms observations: implement process(obs)."""

    @abc.abstractmethod
    def process(self, obs):
        """Map a batch of inner observations to outer ones."""

    def reset(self):
        return self.process(self.venv.reset())

    def step_wait(self):
        obs, rews, dones, infos = self.venv.step_wait()
        return self.process(obs), rews, dones, infos


class CloudpickleWrapper(object):
    """Lets closures (env constructors) cross a multiprocessing boundary: serialised with
    cloudpickle on the way out, rebuilt with the standard pickle loader on the way in."""

    def __init__(self, x):
        self.x = x

    def __getstate__(self):
        import cloudpickle
        return cloudpickle.dumps(self.x)

    def __setstate__(self, blob):
        import pickle
        self.x = pickle.loads(blob)


_LAUNCHER_PREFIXES = ('OMPI_', 'PMI_')


@contextlib.contextmanager
def clear_mpi_env_vars():
    """Temporarily hide the parallel launcher's rendezvous variables while env worker processes are
    spawned, so the children do not believe they are ranks themselves (mpirun in the reference,
    vec_env.py:207-223; the same prefixes are honoured here)."""
    hidden = {k: os.environ.pop(k) for k in [k for k in os.environ if k.startswith(_LAUNCHER_PREFIXES)]}
    try:
        yield
    finally:
        os.environ.update(hidden)
