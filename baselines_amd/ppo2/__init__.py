from .ppo2 import learn   # noqa: F401
from .runner import Runner, Rollout, RolloutField, sf01   # noqa: F401
from .model import Model   # noqa: F401
from .microbatched_model import MicrobatchedModel   # noqa: F401
