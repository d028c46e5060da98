"""`deepq.learn` with the reference's signature and loop (deepq/deepq.py:95-333): one host environment, eps-greedy
acting, (prioritized) replay in HBM, a train step every `train_freq` steps after `learning_starts`, target-network copy
every `target_network_update_freq` steps, ActWrapper with save / load.

What runs where: the Q-networks, the TD target / Huber loss / per-variable clip / Adam, the replay ring and the segment
trees are HIP kernels behind the C ABI (QModel, replay_buffer.py); the loop below is the reference's control flow."""
import os

import numpy as np

from .. import logger
from ..common import set_global_seeds
from ..common.schedules import LinearSchedule
from .models import build_q_func
from .qmodel import QModel
from .replay_buffer import PrioritizedReplayBuffer, ReplayBuffer


class ActWrapper(object):
    """deepq/deepq.py:23-78"""

    def __init__(self, model, act_params):
        self._model = model
        self._act_params = act_params
        self.initial_state = None

    def __call__(self, *args, **kwargs):
        return self._model.act(*args, **kwargs)

    def step(self, observation, **kwargs):
        # DQN doesn't use RNNs so we ignore states and masks
        kwargs.pop('S', None)
        kwargs.pop('M', None)
        return self._model.act([observation], **kwargs), None, None, None

    def save(self, path):
        """tf_util.save_variables: joblib dict {variable name: ndarray}"""
        import joblib
        dirname = os.path.dirname(path)
        if dirname:
            os.makedirs(dirname, exist_ok=True)
        joblib.dump(self._model.variables(), path)

    def save_act(self, path=None):
        """pickle of (variables, act_params) -- the reference zips a TF checkpoint next to the cloudpickled act_params"""
        import cloudpickle
        if path is None:
            path = os.path.join(logger.get_dir(), 'model.pkl')
        with open(path, 'wb') as f:
            cloudpickle.dump((self._model.variables(), self._act_params), f)

    @staticmethod
    def load_act(path):
        import cloudpickle
        with open(path, 'rb') as f:
            variables, act_params = cloudpickle.load(f)
        model = QModel(**act_params)
        model.load_variables(variables)
        return ActWrapper(model, act_params)


def load_act(path):
    return ActWrapper.load_act(path)


class _Checkpointer(object):
    """Best-mean-reward snapshot of the reference's loop (deepq.py:237-258, 315-333): every `checkpoint_freq` steps the
    variables are saved when the 100-episode mean improved; at the end the best snapshot is restored.  The reference
    round-trips through a file in a temporary directory; here the snapshot also stays in memory, and a model file found in
    `checkpoint_path` at start-up counts as a saved model (it is what gets reloaded at the end unless a better one is
    written, like the reference's `load_variables(model_file)`)."""

    def __init__(self, model, act, checkpoint_path, verbose):
        self.model, self.act, self.verbose = model, act, verbose
        self.model_file = os.path.join(checkpoint_path, 'model') if checkpoint_path else None
        self.best_mean_reward = None
        self.snapshot = None

    def load_existing(self, load_path):
        import joblib
        if self.model_file and os.path.exists(self.model_file):
            self.model.load_variables(joblib.load(self.model_file))
            self.snapshot = self.model.variables()
            logger.info('Loaded model from {}'.format(self.model_file))
        elif load_path is not None:
            self.model.load_variables(joblib.load(os.path.expanduser(load_path)))
            logger.info('Loaded model from {}'.format(load_path))

    def maybe_save(self, mean_reward):
        if self.best_mean_reward is not None and not mean_reward > self.best_mean_reward:
            return
        if self.verbose:
            logger.info('Saving model due to mean reward increase: {} -> {}'.format(self.best_mean_reward, mean_reward))
        self.snapshot = self.model.variables()
        if self.model_file:
            self.act.save(self.model_file)
        self.best_mean_reward = mean_reward

    def restore_best(self):
        if self.snapshot is None:
            return
        if self.verbose:
            logger.info('Restored model with mean reward: {}'.format(self.best_mean_reward))
        self.model.load_variables(self.snapshot)


def learn(env, network, seed=None, lr=5e-4, total_timesteps=100000, buffer_size=50000, exploration_fraction=0.1,
          exploration_final_eps=0.02, train_freq=1, batch_size=32, print_freq=100, checkpoint_freq=10000,
          checkpoint_path=None, learning_starts=1000, gamma=1.0, target_network_update_freq=500, prioritized_replay=False,
          prioritized_replay_alpha=0.6, prioritized_replay_beta0=0.4, prioritized_replay_beta_iters=None,
          prioritized_replay_eps=1e-6, param_noise=False, callback=None, load_path=None, **network_kwargs):
    """deepq/deepq.py:95-333.  The local names the reference exposes to `callback(locals(), globals())` (t, obs, action,
    rew, new_obs, done, episode_rewards, replay_buffer, exploration, ...) keep their names.

    Data path of a train step: everything stays in HBM -- `sample_dev` gathers the minibatch on the device,
    `QModel.train_dev` replays the captured optimizer step and leaves the TD errors on the device, and
    `update_priorities_from_td` turns them into priorities there (deepq.py:291-303 without the reference's host arrays)."""
    set_global_seeds(seed)
    q_func = build_q_func(network, **network_kwargs)
    act_params = dict(q_func=q_func, observation_space=env.observation_space, num_actions=env.action_space.n)
    # deepq.py:205-213: AdamOptimizer(learning_rate=lr) (epsilon 1e-8), gamma, grad_norm_clipping=10, double_q (default)
    model = QModel(lr=lr, gamma=gamma, grad_norm_clipping=10, max_batch=batch_size, param_noise=param_noise, **act_params)
    act = ActWrapper(model, act_params)

    if prioritized_replay:
        replay_buffer = PrioritizedReplayBuffer(buffer_size, alpha=prioritized_replay_alpha)
        if prioritized_replay_beta_iters is None:
            prioritized_replay_beta_iters = total_timesteps
        beta_schedule = LinearSchedule(prioritized_replay_beta_iters, initial_p=prioritized_replay_beta0, final_p=1.0)
    else:
        replay_buffer = ReplayBuffer(buffer_size)
        beta_schedule = None
    exploration = LinearSchedule(schedule_timesteps=int(exploration_fraction * total_timesteps), initial_p=1.0,
                                 final_p=exploration_final_eps)
    model.update_target()

    episode_rewards = [0.0]
    obs = env.reset()
    reset = True
    checkpoints = _Checkpointer(model, act, checkpoint_path, print_freq is not None)
    checkpoints.load_existing(load_path)

    def train_step(t):
        # the minibatch is gathered straight into the captured step's static input buffers when the replay stores observations
        # in the form the network takes them (not for one-hot encoded Discrete observations)
        gin = model.graph_inputs(batch_size)
        if tuple(gin['o1'].shape[1:]) != tuple(replay_buffer._ob_shape) or gin['o1'].dtype != replay_buffer._ob_dtype:
            gin = None
        if prioritized_replay:
            o1, a, r, o2, d, w, idx = replay_buffer.sample_dev(batch_size, beta=beta_schedule.value(t), out=gin)
            td = model.train_dev(o1, a, r, o2, d, w)
            replay_buffer.update_priorities_from_td(idx, td, eps=prioritized_replay_eps)
        else:
            o1, a, r, o2, d = replay_buffer.sample_dev(batch_size, out=gin)
            model.train_dev(o1, a, r, o2, d, model.ones(batch_size))

    n_actions = float(env.action_space.n)

    def exploration_arguments(t, reset):
        """what `act` is told at step t (deepq.py:263-275).  eps-greedy: the schedule's eps.  Parameter noise: eps-greedy off,
        and the KL threshold that sizes the perturbation is the KL between a greedy policy and its eps-greedy version at the
        schedule's eps, -log(1 - eps + eps / |A|) (Plappert et al. 2017, appendix C.1); the acting copy is re-drawn at the
        start of every episode."""
        eps = exploration.value(t)
        if not param_noise:
            return eps, {}
        return 0.0, dict(reset=reset, update_param_noise_threshold=-np.log(1.0 - eps + eps / n_actions),
                         update_param_noise_scale=True)

    for t in range(total_timesteps):
        if callback is not None and callback(locals(), globals()):
            break
        update_eps, kwargs = exploration_arguments(t, reset)
        action = act(np.array(obs)[None], update_eps=update_eps, **kwargs)[0]
        reset = False
        new_obs, rew, done, _ = env.step(action)
        replay_buffer.add(obs, action, rew, new_obs, float(done))
        obs = new_obs
        episode_rewards[-1] += rew
        if done:
            obs = env.reset()
            episode_rewards.append(0.0)
            reset = True

        learning = t > learning_starts
        if learning and t % train_freq == 0:
            train_step(t)
        if learning and t % target_network_update_freq == 0:
            model.update_target()

        num_episodes = len(episode_rewards)
        mean_100ep_reward = round(np.mean(episode_rewards[-101:-1]), 1) if num_episodes > 1 else float('nan')
        if done and print_freq is not None and num_episodes % print_freq == 0:
            for key, value in (('steps', t), ('episodes', num_episodes), ('mean 100 episode reward', mean_100ep_reward),
                               ('% time spent exploring', int(100 * exploration.value(t)))):
                logger.record_tabular(key, value)
            logger.dump_tabular()
        if checkpoint_freq is not None and learning and num_episodes > 100 and t % checkpoint_freq == 0:
            checkpoints.maybe_save(mean_100ep_reward)
    checkpoints.restore_best()
    return act
