#!/usr/bin/env python
"""Timing experiments on the position-major data-gradient engine (dgradx6.hip.h): where does a tile's time go?
MRL_DGX6_DBG bits: 1 = no mask loads, 2 = no stores, 4 = no main loop.   python scripts/dgx6_experiments.py [num_envs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from baselines_amd import _lib, ops  # noqa: E402
from baselines_amd.common import set_global_seeds  # noqa: E402
from baselines_amd.common.policies import build_policy  # noqa: E402
from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv  # noqa: E402
from baselines_amd.ppo2 import Model, Runner  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T, M = 128, 4
torch.cuda.set_device(0)
set_global_seeds(0)
env = SyntheticVecEnv('atari', N, seed=1)
policy = build_policy(env, 'cnn')
model = Model(policy=policy, ob_space=env.observation_space, ac_space=env.action_space, nbatch_act=N,
              nbatch_train=N * T // M, nsteps=T, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5)
runner = Runner(env=env, model=model, nsteps=T, gamma=0.99, lam=0.95, return_host=False)
runner.run()
ro = runner.rollout
ro.returns = ops.gae(ro.rewards, ro.values, ro.dones, model.value_dev(runner.obs), runner._dones_dev, 0.99, 0.95)
inds = np.arange(N * T)
np.random.shuffle(inds)
inds_dev = torch.from_numpy(inds).to(model.device)
B = N * T // M
grads = torch.empty_like(model.params)
stats = torch.empty(5, device='cuda')


def epoch():
    for s in range(0, N * T, B):
        model.dm.grad(model.params, ro.obs, ro.actions, ro.returns, ro.values, ro.neglogpacs, inds_dev[s:s + B], B, T, N, 0.1,
                      0.01, 0.5, grads, stats)
    torch.cuda.synchronize()


epoch()
for dbg in (0, 1, 2, 3, 4, 6, 7):
    _lib.set_option('dgx6_dbg', dbg)
    _lib.prof_enable(True)
    epoch()
    _lib.prof_enable(False)
    r = _lib.prof_report()
    print('dbg %d: c2.dgrad %.3f ms  c3.dgrad %.3f ms   (c2.fwd %.3f c3.fwd %.3f)' % (
        dbg, r['c2.dgrad']['ms'] / r['c2.dgrad']['count'], r['c3.dgrad']['ms'] / r['c3.dgrad']['count'],
        r['c2.fwd']['ms'] / r['c2.fwd']['count'], r['c3.fwd']['ms'] / r['c3.fwd']['count']), flush=True)
_lib.set_option('dgx6_dbg', 0)
