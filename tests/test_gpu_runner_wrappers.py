"""GPU: device-env wrappers driven through Runner / learn(): `VecEnvWrapper.__getattr__` forwards
unknown attributes to the wrapped env, so a `hasattr(env, 'step_into')` fast path would step the INNER env and hand the
learner raw observations / rewards.  The Runner now takes the zero-copy path only when the env's own class implements
it; behind a wrapper it calls the wrapper chain's step_wait() and copies the results into the HBM rollout.

Check: a second, identical wrapper chain is stepped by hand with the actions the Runner recorded -- observations,
rewards and dones in the rollout must be exactly what that chain returns (and differ from the inner env's raw values).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _chain(kind, N, seed):
    from baselines_amd.common.vec_env import VecFrameStack, VecNormalize
    from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv
    if kind == 'normalize':
        return VecNormalize(SyntheticVecEnv('mujoco', N, seed=seed))
    return VecFrameStack(SyntheticVecEnv('atari', N, seed=seed, ob_shape=(84, 84, 1)), 4)


def _model(env, N, T):
    from baselines_amd.common import set_global_seeds
    from baselines_amd.common.policies import build_policy
    from baselines_amd.ppo2 import Model
    set_global_seeds(0)
    net, vn = ('mlp', 'copy') if env.observation_space.dtype == np.float32 else ('cnn', None)
    policy = build_policy(env, net, value_network=vn)
    return Model(policy=policy, ob_space=env.observation_space, ac_space=env.action_space, nbatch_act=N,
                 nbatch_train=N * T // 2, nsteps=T, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5)


@pytest.mark.parametrize('kind', ['normalize', 'framestack'])
def test_runner_steps_the_wrapper_not_the_inner_env(kind):
    from baselines_amd.ppo2 import Runner
    N, T = 8, 6
    env, twin = _chain(kind, N, 9), _chain(kind, N, 9)
    assert env.device_resident
    model = _model(env, N, T)
    runner = Runner(env=env, model=model, nsteps=T, gamma=0.99, lam=0.95, return_host=False)
    assert not runner._env_step_into                       # the wrapper's class has no step_into of its own
    first = twin.reset().clone()
    runner.run()
    ro = runner.rollout
    np.testing.assert_array_equal(ro.obs[0].cpu().numpy(), first.cpu().numpy().reshape(ro.obs[0].shape))
    prev_done = torch.zeros(N, dtype=torch.uint8, device='cuda')
    for t in range(T):
        np.testing.assert_array_equal(ro.dones[t].cpu().numpy(), prev_done.cpu().numpy())
        o, r, d, _ = twin.step(ro.actions[t])
        nxt = ro.obs[t + 1] if t + 1 < T else runner.obs
        np.testing.assert_array_equal(nxt.cpu().numpy(), o.cpu().numpy().reshape(nxt.shape))
        np.testing.assert_array_equal(ro.rewards[t].cpu().numpy(), r.cpu().numpy())
        prev_done = (d.view(torch.uint8) if d.dtype == torch.bool else d).clone()
    if kind == 'framestack':
        assert ro.obs.shape[2:] == (84, 84, 4)             # stacked frames, not the inner env's single channel
        # the newest frame sits in the last channel, older ones slide towards channel 0 (envs that did not just restart)
        alive = ro.dones[2] == 0
        assert bool(alive.any()) and torch.equal(ro.obs[2][alive][..., 2], ro.obs[1][alive][..., 3])
    else:
        o = ro.obs[1:].float()
        assert float(o.abs().max()) <= 10.0                # VecNormalize's clip range
        assert float(o.std()) > 0.8                        # normalised to ~unit scale (the raw uniform[-1,1] obs have std 0.58)


@pytest.mark.parametrize('kind', ['normalize', 'framestack'])
def test_learn_runs_through_device_wrappers(kind):
    from baselines_amd.ppo2 import learn
    N, T = 8, 8
    env = _chain(kind, N, 4)
    net = 'mlp' if kind == 'normalize' else 'cnn'
    seen = []
    model = learn(network=net, env=env, total_timesteps=2 * N * T, nsteps=T, nminibatches=2, noptepochs=2, seed=0,
                  log_interval=100, lr=3e-4, update_fn=lambda u: seen.append(u),
                  **({'value_network': 'copy'} if net == 'mlp' else {}))
    assert seen == [1, 2]
    assert np.all(np.isfinite(model.get_flat_params()))
    if kind == 'normalize':
        assert env.ob_rms.count > N * T                    # the wrapper's statistics were updated by the rollouts
