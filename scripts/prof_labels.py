#!/usr/bin/env python
"""Per-label HIP-event times of one PPO2 epoch with the default kernel choices (env knobs apply).
   python scripts/prof_labels.py [num_envs] [label=variant ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + sys.argv[1:]
import numpy as np  # noqa
import torch  # noqa
from baselines_amd import _lib, ops  # noqa
from baselines_amd.common import set_global_seeds  # noqa
from baselines_amd.common.policies import build_policy  # noqa
from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv  # noqa
from baselines_amd.ppo2 import Model, Runner  # noqa

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for kv in sys.argv[2:]:
    k, v = kv.split('=')
    _lib.tune_set(k, int(v))
T, M = 128, 4
torch.cuda.set_device(0)
set_global_seeds(0)
env = SyntheticVecEnv('atari', N, seed=1)
model = Model(policy=build_policy(env, 'cnn'), ob_space=env.observation_space, ac_space=env.action_space, nbatch_act=N,
              nbatch_train=N * T // M, nsteps=T, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5)
runner = Runner(env=env, model=model, nsteps=T, gamma=0.99, lam=0.95, return_host=False)
runner.run()
ro = runner.rollout
ro.returns = ops.gae(ro.rewards, ro.values, ro.dones, model.value_dev(runner.obs), runner._dones_dev, 0.99, 0.95)
inds = np.arange(N * T)
np.random.shuffle(inds)
inds_dev = torch.from_numpy(inds).to(model.device)
B = N * T // M


def epoch():
    for s in range(0, N * T, B):
        model.train_indexed(2.5e-4, 0.1, ro, inds_dev[s:s + B])


epoch()
torch.cuda.synchronize()
_lib.prof_enable(True)
epoch()
torch.cuda.synchronize()
_lib.prof_enable(False)
rep = _lib.prof_report()
tot = sum(d['ms'] for d in rep.values())
print(' '.join('%s=%.2f/%.0fT' % (k, d['ms'], d['flops'] / d['ms'] / 1e9) for k, d in sorted(rep.items()) if d['flops']), 'TOTAL=%.1f' % tot)
