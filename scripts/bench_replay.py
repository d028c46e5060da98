#!/usr/bin/env python
"""Config 5 of BASELINE.json (the DQN slice): a PongNoFrameskip-shaped replay buffer of 10^6 transitions resident in HBM
(2 x 28,224-byte frames stacks per transition = 56.5 GB), prioritized sampling and the TD-target kernel.
Per-kernel times are HIP events on the launch stream (mrl_prof_*); bytes are the algorithmic traffic each ProfScope
declares.  Prints one JSON line.      python scripts/bench_replay.py [capacity]"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from baselines_amd import _lib  # noqa: E402
from baselines_amd.deepq import PrioritizedReplayBuffer, dqn_td_loss  # noqa: E402

CAP = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
SHAPE = (84, 84, 4)
NA = 6
torch.cuda.set_device(0)
random.seed(0)
buf = PrioritizedReplayBuffer(CAP, alpha=0.6)
gen = torch.Generator(device='cuda').manual_seed(0)


def batch(n):
    o = torch.randint(0, 256, (n,) + SHAPE, dtype=torch.uint8, device='cuda', generator=gen)
    return (o, torch.randint(0, NA, (n,), device='cuda', generator=gen).int(), torch.rand(n, device='cuda', generator=gen),
            o.flip(0), (torch.rand(n, device='cuda', generator=gen) < 0.01).float())


# fill: 4096 transitions per insert (one per env of a vectorised actor)
fill = batch(4096)
t0 = time.perf_counter()
while len(buf) < CAP:
    n = min(4096, CAP - len(buf))
    buf.add_batch(*(x[:n] for x in fill))
torch.cuda.synchronize()
fill_s = time.perf_counter() - t0


def learner_step(B, beta=0.4):
    o1, a, r, o2, d, w, idx = buf.sample_dev(B, beta)
    q = torch.randn((B, NA), device='cuda', generator=gen)          # stands in for the Q-networks' outputs
    td, loss, dq = dqn_td_loss(q, q.roll(1, 0), q.roll(2, 0), a, r, d, w, 0.99)
    buf.update_priorities_from_td(idx, td)
    return loss


res = {'capacity': CAP, 'hbm_resident_GB': round(2 * CAP * int(np.prod(SHAPE)) / 1e9, 1), 'fill_seconds': round(fill_s, 2)}
for B in (32, 4096):
    for _ in range(3):
        learner_step(B)
    buf.add_batch(*fill)
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    t0 = time.perf_counter()
    iters = 200 if B == 32 else 20
    for _ in range(iters):
        learner_step(B)
        if B == 4096:
            buf.add_batch(*fill)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    _lib.prof_enable(False)
    rep = _lib.prof_report()
    out = {'wall_us_per_learner_step': round(wall / iters * 1e6, 1)}
    for k, v in rep.items():
        us = v['ms'] / v['count'] * 1e3
        out[k] = {'us': round(us, 2), 'GB/s': round(v['bytes'] / v['count'] / (us * 1e-6) / 1e9, 1) if v['bytes'] else None,
                  'calls': v['count']}
    res['batch_%d' % B] = out
# ---- the whole DQN learner step of the reference (deepq.py:291-303) around the replay slice: prioritized sample ->
# Q-network TD step (conv_only + dueling, double-Q, per-variable clip, Adam; csrc/qnet.hip.h) -> priority update, batch 32
from baselines_amd.common.spaces import Box  # noqa: E402
from baselines_amd.deepq import QModel, build_q_func  # noqa: E402

np.random.seed(0)
qm = QModel(build_q_func('conv_only'), Box(0, 255, SHAPE, np.uint8), NA, lr=1e-4, gamma=0.99, max_batch=32)


def dqn_step(B=32, beta=0.4, graph=True):
    o1, a, r, o2, d, w, idx = buf.sample_dev(B, beta, out=qm.graph_inputs(B))   # gathered into the captured step's static inputs
    td = qm.train_dev(o1, a, r, o2, d, w, graph=graph)       # device TD errors, no host round trip
    buf.update_priorities_from_td(idx, td)


def timed(n, **kw):
    for _ in range(5):
        dqn_step(**kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        dqn_step(**kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


# the optimizer step replayed as one hipGraph (the default), then launched kernel by kernel with HIP events around each
res['dqn_learner_step_batch32_us'] = round(timed(200), 1)
res['dqn_learner_step_batch32_eager_us'] = round(timed(100, graph=False), 1)
_lib.prof_enable(True)
for _ in range(20):
    dqn_step(graph=False)
torch.cuda.synchronize()
_lib.prof_enable(False)
rep = _lib.prof_report()
res['dqn_learner_step_batch32_kernels_us'] = {k: {'us': round(v['ms'] / v['count'] * 1e3, 2), 'calls_per_step': v['count'] / 20.0}
                                              for k, v in sorted(rep.items(), key=lambda kv: -kv[1]['ms'])}
res['dqn_learner_step_batch32_kernel_sum_us'] = round(sum(v['ms'] for v in rep.values()) / 20 * 1e3, 1)
print(json.dumps(res))
