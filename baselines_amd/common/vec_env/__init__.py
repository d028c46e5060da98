from .vec_env import (AlreadySteppingError, NotSteppingError, VecEnv, VecEnvWrapper,  # noqa: F401
                      VecEnvObservationWrapper, CloudpickleWrapper, clear_mpi_env_vars)
from .dummy_vec_env import DummyVecEnv   # noqa: F401
