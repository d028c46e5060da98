"""build_policy with the reference's signature (common/policies.py:121-179).  Returns a
PolicySpec: everything the learner needs to lay the model out on the device (network description,
spaces, value-network mode, probability-distribution type = common/distributions.py:278-290)."""
import numpy as np

from .models import get_network_builder, NetworkDesc
from .spaces import is_box, is_discrete, is_multibinary, is_multidiscrete


class PolicySpec(object):
    def __init__(self, ob_space, ac_space, network, value_network):
        self.ob_space, self.ac_space = ob_space, ac_space
        self.network = network
        if value_network not in (None, 'shared', 'copy'):
            raise NotImplementedError("value_network must be None, 'shared' or 'copy' on the HIP path")
        self.value_copy = (value_network == 'copy')
        self.nvec = None
        self.recurrent = network.kind in ('lstm', 'cnn_lstm')
        self.nlstm = int(network.kw.get('nlstm', 0)) if self.recurrent else 0
        if self.recurrent and self.value_copy:
            # policies.py:162-165 "recurrent architectures are not supported with value_network=copy yet"
            raise NotImplementedError('recurrent architectures are not supported with value_network=copy')
        if is_discrete(ac_space):
            self.pd_kind, self.nact = 'categorical', int(ac_space.n)
        elif is_box(ac_space):
            assert len(ac_space.shape) == 1
            self.pd_kind, self.nact = 'gaussian', int(ac_space.shape[0])
        elif is_multidiscrete(ac_space):              # MultiCategoricalPdType(ac_space.nvec), distributions.py:285-286
            self.nvec = tuple(int(v) for v in np.asarray(ac_space.nvec).reshape(-1))
            if len(self.nvec) > 16:
                raise NotImplementedError('MultiDiscrete with more than 16 components')
            self.pd_kind, self.nact = 'multicategorical', int(sum(self.nvec))
        elif is_multibinary(ac_space):                # BernoulliPdType(ac_space.n), distributions.py:287-288
            self.pd_kind, self.nact = 'bernoulli', int(ac_space.n)
        else:
            raise NotImplementedError('action space {} is outside the supported hot path'.format(ac_space))
        # common/input.py:43-63 encode_observation: Discrete observations become one-hot float32 rows (the device model
        # then sees a Box of n floats); Box observations pass through (uint8 pixels are scaled inside the first conv layer)
        self.ob_onehot = int(ob_space.n) if is_discrete(ob_space) else 0
        self.ob_clip = None                 # set by build_policy(normalize_observations=True)
        if self.ob_onehot:
            self.ob_shape, self.ob_dtype = (self.ob_onehot,), np.dtype(np.float32)
        elif is_box(ob_space):
            self.ob_shape = tuple(ob_space.shape)
            self.ob_dtype = np.dtype(ob_space.dtype)
        else:
            raise NotImplementedError('observation space {} is outside the supported hot path'.format(ob_space))

    def device_model_kwargs(self):
        kw = dict(network=self.network.kind, ob_shape=self.ob_shape, ob_dtype=self.ob_dtype, pd_kind=self.pd_kind,
                  nact=self.nact, value_copy=self.value_copy)
        if self.nvec:
            kw['nvec'] = self.nvec
        kw.update(self.network.kw)
        return kw


def build_policy(env, policy_network, value_network=None, normalize_observations=False, estimate_q=False,
                 **policy_kwargs):
    if estimate_q:
        raise NotImplementedError('estimate_q (Q-value heads) is outside the PPO2 hot path')
    if isinstance(policy_network, str):
        policy_network = get_network_builder(policy_network)(**policy_kwargs)
    if not isinstance(policy_network, NetworkDesc):
        raise NotImplementedError('custom TF network functions cannot run on the HIP path; register a NetworkDesc')
    spec = PolicySpec(env.observation_space, env.action_space, policy_network, value_network)
    if normalize_observations:
        # policies.py:133-135,182-185: clip((x - rms.mean) / rms.std, -5, 5) with a RunningMeanStd that ppo2 never
        # updates (only trpo_mpi / ppo1 call rms.update): mean stays 0 and std 1, i.e. the observation is clipped to
        # [-5, 5].  Float Box observations only, like the reference (`X.dtype == tf.float32`).
        if spec.ob_dtype != np.dtype(np.float32) or spec.ob_onehot:
            raise NotImplementedError('normalize_observations applies to float32 Box observations')
        spec.ob_clip = 5.0
    return spec
