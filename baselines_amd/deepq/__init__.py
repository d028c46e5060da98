"""DQN replay slice: HBM-resident (prioritized) replay buffer, device
segment trees, TD-target / Huber kernel and the schedules that drive them.  The training loop
`deepq.learn` itself (d6) and the Q-network builders (d7) are outside this round's scope."""
from ..common.schedules import LinearSchedule  # noqa: F401
from .replay_buffer import PrioritizedReplayBuffer, ReplayBuffer, dqn_td_loss  # noqa: F401
