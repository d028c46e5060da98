"""DQN on the device: HBM-resident (prioritized) replay buffer, device segment trees,
TD-target / Huber kernel, the Q-network (build_q_func: mlp / cnn / conv_only features, dueling heads), and the
reference's `deepq.learn` loop around them."""
from ..common.schedules import LinearSchedule  # noqa: F401
from .replay_buffer import PrioritizedReplayBuffer, ReplayBuffer, dqn_td_loss  # noqa: F401
from .models import build_q_func  # noqa: F401
from .qmodel import QModel  # noqa: F401
from .deepq import ActWrapper, learn, load_act  # noqa: F401
