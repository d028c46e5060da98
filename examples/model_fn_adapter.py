"""The adapter of INTEGRATION.md level 1: run the REFERENCE's `baselines.ppo2.ppo2.learn` (and its Runner, logger,
schedules ...) on top of the MI355X learner through the reference's own plug point `model_fn` (ppo2/ppo2.py:103-109).

    from baselines.ppo2 import ppo2 as ref_ppo2
    from examples.model_fn_adapter import make_model_fn
    ref_ppo2.learn(network='mlp', env=env, total_timesteps=..., model_fn=make_model_fn('mlp', value_network='copy'))

`make_model_fn` is exercised by tests/test_gpu_ppo2.py::test_integration_model_fn_adapter (with this package's own
`learn` standing in for the reference's, whose TensorFlow dependency is not installed here)."""
from baselines_amd.common.policies import build_policy
from baselines_amd.ppo2 import Model


class Spaces(object):
    """the two attributes build_policy reads from an env (common/policies.py:127)"""

    def __init__(self, ob_space, ac_space):
        self.observation_space, self.action_space = ob_space, ac_space


def make_model_fn(network, **network_kwargs):
    """-> a callable with the signature ppo2.learn calls model_fn with (ppo2/ppo2.py:105-107); the reference's `policy`
    argument is a TF graph builder and is ignored: the spaces and the network NAME are all the device model needs"""
    def model_fn(*, policy, ob_space, ac_space, nbatch_act, nbatch_train, nsteps, ent_coef, vf_coef, max_grad_norm,
                 comm=None, mpi_rank_weight=1):
        amd_policy = build_policy(Spaces(ob_space, ac_space), network, **network_kwargs)
        return Model(policy=amd_policy, ob_space=ob_space, ac_space=ac_space, nbatch_act=nbatch_act,
                     nbatch_train=nbatch_train, nsteps=nsteps, ent_coef=ent_coef, vf_coef=vf_coef,
                     max_grad_norm=max_grad_norm, comm=comm, mpi_rank_weight=mpi_rank_weight)
    return model_fn
