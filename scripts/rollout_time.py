#!/usr/bin/env python
"""Act side at shard sizes: wall time of one device rollout (nsteps act forwards + synthetic env step + rollout
store, replayed as a hipGraph) next to one update, Atari-shaped NatureCNN.
    python scripts/rollout_time.py [num_envs] [nsteps] [atari|mujoco]     (under rocprofv3 --kernel-trace: per-kernel times of the act path)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from baselines_amd.common import set_global_seeds  # noqa: E402
from baselines_amd.common.policies import build_policy  # noqa: E402
from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv  # noqa: E402
from baselines_amd.ppo2 import Model, Runner  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
KIND = sys.argv[3] if len(sys.argv) > 3 else 'atari'
torch.cuda.set_device(0)
set_global_seeds(0)
env = SyntheticVecEnv(KIND, N, seed=1)
policy = build_policy(env, 'cnn') if KIND == 'atari' else build_policy(env, 'mlp', value_network='copy')
model = Model(policy=policy, ob_space=env.observation_space, ac_space=env.action_space, nbatch_act=N,
              nbatch_train=N * T // 4, nsteps=T, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5)
runner = Runner(env=env, model=model, nsteps=T, gamma=0.99, lam=0.95, return_host=False)
for _ in range(2):
    runner.run()
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    runner.run()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print('num_envs %d nsteps %d: rollout %.2f ms (%.1f us per env step of the vector env), %.0f env-steps/s act-side only'
      % (N, T, min(ts) * 1e3, min(ts) / T * 1e6, N * T / min(ts)))
