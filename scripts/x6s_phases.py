#!/usr/bin/env python
"""Phase timestamps of the wave-specialised split engine (gemmx6s.hip.h), workgroup 0, steps 16..23:
consumer wave 0 (barrier wait, fragment reads issued, first / second MFMA half-step) and producer wave 4 (split + LDS
writes, next loads issued, barrier wait).    python scripts/x6s_phases.py <MRL_X6_DBG value: 11 = c2.fwd, 12 = c3.fwd, 1 = fc1.fwd> [B]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['MRL_X6_DBG'] = sys.argv[1] if len(sys.argv) > 1 else '11'
import numpy as np  # noqa
import torch  # noqa
from baselines_amd import ops  # noqa

B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
dm = ops.DeviceModel(network='cnn', ob_shape=(84, 84, 4), ob_dtype=np.uint8, pd_kind='categorical', nact=6, value_copy=False, chunk=B)
r = np.random.RandomState(1)
params = torch.from_numpy((r.randn(dm.P) * 0.05).astype(np.float32)).cuda()
obs = torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device='cuda')
noise = torch.rand((B, 6), device='cuda')
for _ in range(2):
    dm.act(params, obs, noise)
torch.cuda.synchronize()
st = dm.workspace[-2048 + 512:-2048 + 512 + 8 * 16 * 8].view(torch.int64).cpu().numpy().reshape(8, 16)
for s in range(8):
    c, p = st[s, 0:5], st[s, 8:12]
    nxt = (st[s + 1, 0] - c[4]) if s < 7 else 0
    print('step %2d consumer: barrier=%5d frags=%5d mma0=%5d mma1=%5d  -> next %5d | producer: swrite=%5d fetch=%5d barrier=%5d  (p0-c0 %6d)' % (
        s + 16, c[1] - c[0], c[2] - c[1], c[3] - c[2], c[4] - c[3], nxt, p[1] - p[0], p[2] - p[1], p[3] - p[2], p[0] - c[0]))
