#!/usr/bin/env python
"""Phase timestamps (shader clock) of one workgroup of the fused MLP step kernel (experiment builds, -DMRL_X6_EXPERIMENTS).
MRL_MLP_DBG=<1 + workgroup> python scripts/mlp_phases.py      (sliced launches: even workgroups carry the policy net)"""
import os
import sys
os.environ.setdefault('MRL_MLP_DBG', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa
import torch  # noqa
from baselines_amd import ops  # noqa
from baselines_amd.common import set_global_seeds  # noqa
from baselines_amd.common.policies import build_policy  # noqa
from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv  # noqa
from baselines_amd.ppo2 import Model, Runner  # noqa

N, T, M = 1024, 128, 32
torch.cuda.set_device(0)
set_global_seeds(0)
env = SyntheticVecEnv('mujoco', N, seed=1)
model = Model(policy=build_policy(env, 'mlp', value_network='copy'), ob_space=env.observation_space, ac_space=env.action_space,
              nbatch_act=N, nbatch_train=N * T // M, nsteps=T, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5)
runner = Runner(env=env, model=model, nsteps=T, gamma=0.99, lam=0.95, return_host=False)
runner.run()
ro = runner.rollout
ro.returns = ops.gae(ro.rewards, ro.values, ro.dones, model.value_dev(runner.obs), runner._dones_dev, 0.99, 0.95)
inds = torch.from_numpy(np.random.permutation(N * T)).to(model.device)
B = N * T // M
for it in range(3):
    model.train_indexed(3e-4, 0.2, ro, inds[it * B:(it + 1) * B])
torch.cuda.synchronize()
ws = model.dm.workspace
stamps = ws[-2048 + 512:-2048 + 512 + 64].view(torch.int64).cpu().numpy()
d = np.diff(stamps)
names = ['P0 gather', 'P1 fc0 fwd', 'P2 fc1 fwd', 'P3 heads+loss', 'P4 head grads/dz1', 'P5 fc1 bwd', 'P6 fc0 wgrad']
# the stamps are shader-clock cycles (s_memtime via __builtin_readcyclecounter): ~2.4 GHz when nothing else loads the chip
for n, x in zip(names, d):
    print('%-20s %8d cycles  %6.2f us at 2.4 GHz' % (n, x, x / 2400.0))
print('workgroup %d: total %d cycles = %.2f us at 2.4 GHz' % (int(os.environ['MRL_MLP_DBG']) - 1, stamps[-1] - stamps[0],
                                                          (stamps[-1] - stamps[0]) / 2400.0))
