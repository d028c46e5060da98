#!/usr/bin/env python
"""Per-launch-site tile-variant sweep of the fp32-MFMA implicit GEMM on one MI355X.
Runs one PPO2 update (Atari-shaped NatureCNN) per variant with the HIP-event profiler on and
prints TFLOP/s per label and variant.   python scripts/tune_gemm.py [num_envs] [chunk]"""
import json
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

VARIANTS = [('128x32', 0), ('128x64w41', 2), ('128x128', 5), ('wres16', 100), ('imgres', 102)]
LABELS = ['c1.fwd', 'c2.fwd', 'c3.fwd', 'fc1.fwd', 'c1.wgrad', 'c2.wgrad', 'c3.wgrad', 'fc1.wgrad', 'c2.dgrad',
          'c3.dgrad', 'fc1.dgrad']


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else None
    T, M = 128, 4
    torch.cuda.set_device(0)
    set_global_seeds(0)
    env = SyntheticVecEnv('atari', N, seed=1)
    policy = build_policy(env, 'cnn')
    model = Model(policy=policy, ob_space=env.observation_space, ac_space=env.action_space, nbatch_act=N,
                  nbatch_train=N * T // M, nsteps=T, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, chunk=chunk)
    runner = Runner(env=env, model=model, nsteps=T, gamma=0.99, lam=0.95, return_host=False)
    runner.run()
    ro = runner.rollout
    ro.returns = ops.gae(ro.rewards, ro.values, ro.dones, model.value_dev(runner.obs), runner._dones_dev, 0.99, 0.95)
    inds = np.arange(N * T)
    np.random.shuffle(inds)
    inds_dev = torch.from_numpy(inds).to(model.device)
    B = N * T // M

    def one_epoch():
        for s in range(0, N * T, B):
            model.train_indexed(2.5e-4, 0.1, ro, inds_dev[s:s + B])

    res = {}
    for name, v in [('default', -1)] + VARIANTS:
        for lab in LABELS:
            _lib.tune_set(lab, v)
        one_epoch()
        torch.cuda.synchronize()
        _lib.prof_enable(True)
        one_epoch()
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        rep = _lib.prof_report()
        res[name] = {k: dict(ms=d['ms'], tflops=(d['flops'] / d['ms'] / 1e9 if d['flops'] else None)) for k, d in rep.items()}
    labs = sorted({k for r in res.values() for k in r})
    print('%-12s' % 'label' + ''.join('%13s' % n for n in res))
    for k in labs:
        row = '%-12s' % k
        for n in res:
            d = res[n].get(k)
            row += '%13s' % ('%.1f/%5.1fT' % (d['ms'], d['tflops']) if d and d['tflops'] else ('%.2fms' % d['ms'] if d else '-'))
        print(row)
    print('TOTAL_MS    ' + ''.join('%13.1f' % sum(d['ms'] for d in res[n].values()) for n in res))
    best = {k: max((n for n in res if n != 'default' and k in res[n]), key=lambda n: -res[n][k]['ms']) for k in LABELS}
    print('best per label:', json.dumps(best))
    print('sum of best ms: %.1f' % (sum(res[best[k]][k]['ms'] for k in LABELS)))
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'tune_N%d.json' % N), 'w'))


if __name__ == '__main__':
    main()
