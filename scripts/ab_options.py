#!/usr/bin/env python
"""A/B timing of engine options inside ONE process on ONE box (box-to-box clocks differ by several percent): one PPO2
epoch at the bench shape per setting, per-kernel HIP-event times.
    python scripts/ab_options.py [num_envs] name=v0,v1[,v2] [name2=...]     (the cross product is NOT taken: one knob at a time)"""
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

args = sys.argv[1:]
N = int(args.pop(0)) if args and args[0].isdigit() else 4096
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


def measure(tag):
    epoch()
    _lib.prof_enable(True)
    epoch()
    epoch()
    _lib.prof_enable(False)
    r = _lib.prof_report()
    tot = sum(v['ms'] for v in r.values()) / 2
    keys = ['c1.fwd', 'c2.fwd', 'c3.fwd', 'fc1.fwd', 'fc1.dgrad', 'c3.dgrad', 'c2.dgrad', 'c1.wgrad', 'c2.wgrad', 'c3.wgrad', 'fc1.wgrad', 'heads_loss']
    import hashlib
    sha = hashlib.sha1(grads.cpu().numpy().tobytes() + stats.cpu().numpy().tobytes()).hexdigest()[:12]     # same bits <=> same digest (builds / options)
    print('%-22s grad+stats sha1 %s' % (tag, sha))
    print('%-22s epoch %.2f ms | ' % (tag, tot) + ' '.join('%s %.2f' % (k, r[k]['ms'] / r[k]['count']) for k in keys), flush=True)


measure('defaults')
for kv in args:
    name, vals = kv.split('=')
    default = _lib.get_option(name)
    for v in vals.split(','):
        _lib.set_option(name, int(v))
        model.dm.set_chunk(model.dm.chunk)          # some options change the workspace layout (plane tensors)
        measure('%s=%s' % (name, v))
    _lib.set_option(name, default)
    model.dm.set_chunk(model.dm.chunk)
measure('defaults (again)')
