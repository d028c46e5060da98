#!/usr/bin/env python
"""Phase timestamps of the LDS-resident data-gradient kernels (MRL_DGRAD_DBG=<layer index 1|2>; conv2 runs the
asynchronous variant unless MRL_DGRAD_ASYNC=0), workgroup 0,
first and last wave, first 6 image groups.   python scripts/dgrad_phases.py [num_envs]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa
import torch  # noqa
from baselines_amd import ops  # noqa
from baselines_amd.common import set_global_seeds  # noqa
from baselines_amd.common.policies import build_policy  # noqa
from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv  # noqa
from baselines_amd.ppo2 import Model, Runner  # noqa

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
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
inds = torch.from_numpy(np.random.permutation(N * T)).to(model.device)
B = N * T // M
for it in range(2):
    model.train_indexed(2.5e-4, 0.1, ro, inds[it * B:(it + 1) * B])
torch.cuda.synchronize()
ws = model.dm.workspace
st = ws[-2048 + 512:-2048 + 512 + 96 * 8].view(torch.int64).cpu().numpy().reshape(2, 6, 8)
names = (['offsets', 'dma-wait+barrier', 'masks+MFMA', 'stores', 'barrier', 'dma-issue'] if os.environ.get('MRL_DGRAD_ASYNC', '1') != '0' and os.environ.get('MRL_DGRAD_DBG') == '1'
         else ['wait prev (barrier)', 'stage', 'barrier', 'MFMA stream', 'epi loads issued', 'epi stores issued'])
for w, wn in enumerate(('wave 0', 'last wave')):
    print(wn)
    for it in range(6):
        d = np.diff(st[w, it, :7])
        nxt = (st[w, it + 1, 0] - st[w, it, 6]) if it < 5 else 0
        print('  group %d: ' % it + '  '.join('%s=%d' % (n.split()[0], x) for n, x in zip(names, d)) + '  total=%d' % (st[w, it, 6] - st[w, it, 0]))
