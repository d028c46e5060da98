from .vec_env import (AlreadySteppingError, NotSteppingError, VecEnv, VecEnvWrapper,  # noqa: F401
                      VecEnvObservationWrapper, CloudpickleWrapper, clear_mpi_env_vars)
from .dummy_vec_env import DummyVecEnv   # noqa: F401
from .vec_frame_stack import VecFrameStack   # noqa: F401,E402
from .vec_normalize import VecNormalize     # noqa: F401,E402
from .shmem_vec_env import ShmemVecEnv, SubprocVecEnv   # noqa: F401,E402
from .vec_monitor import VecMonitor   # noqa: F401,E402
