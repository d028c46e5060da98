"""PCIe-inclusive rate of the PPO2 update when the caller hands over HOST buffers (the reference's own hand-over:
`runner.run()` returns NumPy arrays, `ppo2.py:157-166` gathers `arr[mbinds]` on the host and feeds `model.train` through
feed_dict).  The C boundary (`include/mrl.h`) takes device pointers and `learn()` keeps the rollout in HBM, so this is NOT the
bench metric -- it is the figure DESIGN.md 5 quotes next to it for callers that stay on the reference's array interface.

  python scripts/pcie_rate.py [num_envs] [nsteps]       -> one JSON line

Measured on one GPU: (a) raw H2D / D2H bandwidth, pinned and pageable; (b) `Runner.run()` with `return_host=True` (the rollout
leaves HBM as fresh host arrays, `runner.py:60-67`); (c) one epoch of the reference loop -- host gather + `Model.train` on host
arrays (pageable upload of every minibatch + the device step); (d) the same minibatches already on the device (`train_indexed`).
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from baselines_amd.common import set_global_seeds                                   # noqa: E402
from baselines_amd.common.policies import build_policy                              # noqa: E402
from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv          # noqa: E402
from baselines_amd.ppo2 import Model, Runner                                        # noqa: E402


def bandwidth(nbytes=1 << 30, reps=3):
    dev = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    out = {}
    for name, host in (('pinned', torch.empty(nbytes, dtype=torch.uint8).pin_memory()),
                       ('pageable', torch.empty(nbytes, dtype=torch.uint8))):
        host.fill_(1)
        for direction in ('h2d', 'd2h'):
            best = 1e9
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if direction == 'h2d':
                    dev.copy_(host, non_blocking=True)
                else:
                    host.copy_(dev, non_blocking=True)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            out['%s_%s_GBps' % (name, direction)] = round(nbytes / best / 1e9, 1)
    return out


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    nminibatches, lr, cliprange = 4, 2.5e-4, 0.1
    nbatch = N * T
    B = nbatch // nminibatches
    set_global_seeds(0)
    env = SyntheticVecEnv('atari', N, seed=1000)
    policy = build_policy(env, 'cnn', value_network=None)
    model = Model(policy=policy, ob_space=env.observation_space, ac_space=env.action_space, nbatch_act=N, nbatch_train=B,
                  nsteps=T, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5)
    out = {'num_envs': N, 'nsteps': T, 'minibatch': B, 'bandwidth': bandwidth()}

    # (b) the rollout handed back as host arrays, as the reference's runner does
    runner = Runner(env=env, model=model, nsteps=T, gamma=0.99, lam=0.95, return_host=True)
    for _ in range(2):
        res = runner.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = runner.run()
    torch.cuda.synchronize()
    out['rollout_to_host_s'] = round(time.perf_counter() - t0, 3)
    obs, returns, masks, actions, values, neglogpacs = res[:6]
    out['rollout_host_GB'] = round(sum(np.asarray(a).nbytes for a in res[:6]) / 1e9, 2)

    # (c) one epoch of ppo2.py:157-166 on host arrays
    inds = np.arange(nbatch)
    np.random.shuffle(inds)
    t_gather = t_train = 0.0
    for start in range(0, nbatch, B):
        mbinds = inds[start:start + B]
        t0 = time.perf_counter()
        slices = tuple(arr[mbinds] for arr in (obs, returns, masks, actions, values, neglogpacs))
        t1 = time.perf_counter()
        model.train(lr, cliprange, *slices)               # returns Python floats: synchronous
        t2 = time.perf_counter()
        t_gather += t1 - t0
        t_train += t2 - t1
    out['epoch_host_gather_s'] = round(t_gather, 3)
    out['epoch_train_on_host_arrays_s'] = round(t_train, 3)

    # (d) the same epoch on the resident rollout
    runner_dev = Runner(env=env, model=model, nsteps=T, gamma=0.99, lam=0.95, return_host=False)
    for _ in range(2):
        runner_dev.run()
    inds_dev = model.indices_to_device(inds)
    model.train_epoch(lr, cliprange, runner_dev.rollout, inds_dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.train_epoch(lr, cliprange, runner_dev.rollout, inds_dev)
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    out['epoch_resident_s'] = round(t_dev, 4)
    noptepochs = 4
    out['env_steps_per_s'] = {
        'resident (the bench metric, update only)': round(nbatch / (noptepochs * t_dev)),
        'host arrays into Model.train, gather excluded (PCIe-inclusive)': round(nbatch / (noptepochs * t_train)),
        'host arrays, host gather included (the reference loop as written)': round(nbatch / (noptepochs * (t_train + t_gather))),
    }
    print(json.dumps(out))


if __name__ == '__main__':
    main()
