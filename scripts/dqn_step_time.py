#!/usr/bin/env python
"""Config 5's learner step alone (prioritized sample -> Q-network TD step replayed as a hipGraph -> priority update) at batch 32 out of
a small resident replay: wall time per step; under `rocprofv3 --kernel-trace` (scripts/gpu_trace.sh, GAPS=.) the per-kernel times and
whether the branches of the replayed graph overlap (kernels busy > 100 % of the window).
    python scripts/dqn_step_time.py [steps] [capacity] [graph|eager]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from baselines_amd.common.spaces import Box  # noqa: E402
from baselines_amd.deepq import PrioritizedReplayBuffer, QModel, build_q_func  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
CAP = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
SHAPE, NA = (84, 84, 4), 6
ZERO_COPY = os.environ.get('ZERO_COPY', '1') != '0'     # sample into the captured step's static input buffers (deepq.learn does)
torch.cuda.set_device(0)
gen = torch.Generator(device='cuda').manual_seed(0)
buf = PrioritizedReplayBuffer(CAP, alpha=0.6)
o = torch.randint(0, 256, (4096,) + SHAPE, dtype=torch.uint8, device='cuda', generator=gen)
while len(buf) < CAP:
    buf.add_batch(o, torch.randint(0, NA, (4096,), device='cuda', generator=gen).int(), torch.rand(4096, device='cuda', generator=gen),
                  o.flip(0), (torch.rand(4096, device='cuda', generator=gen) < 0.01).float())
np.random.seed(0)
qm = QModel(build_q_func('conv_only'), Box(0, 255, SHAPE, np.uint8), NA, lr=1e-4, gamma=0.99, max_batch=32)


def step(graph=True):
    o1, a, r, o2, d, w, idx = buf.sample_dev(32, 0.4, out=qm.graph_inputs(32) if ZERO_COPY else None)
    td = qm.train_dev(o1, a, r, o2, d, w, graph=graph)
    buf.update_priorities_from_td(idx, td)


MODES = {'graph': (True,), 'eager': (False,)}.get(sys.argv[3] if len(sys.argv) > 3 else '', (True, False))
for g in MODES:
    for _ in range(5):
        step(g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        step(g)
    torch.cuda.synchronize()
    print('learner step, batch 32, %s: %.1f us' % ('hipGraph replay' if g else 'eager launches', (time.perf_counter() - t0) / STEPS * 1e6))
