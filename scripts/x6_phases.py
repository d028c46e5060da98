#!/usr/bin/env python
"""Phase timestamps of the tiled bf16x6 GEMM (fc1 forward, MRL_X6_DBG=1): workgroup 0, wave 0, tiles 8..13."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MRL_X6_DBG', '1')      # 1: fc1 forward; 10 + layer index: that conv layer's forward
import numpy as np  # noqa
import torch  # noqa
from baselines_amd import ops, _lib  # noqa

for _o in ('x6_il', 'act_planes'):
    if os.environ.get('OPT_' + _o.upper()):
        _lib.set_option(_o, int(os.environ['OPT_' + _o.upper()]))

B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
dm = ops.DeviceModel(network='cnn', ob_shape=(84, 84, 4), ob_dtype=np.uint8, pd_kind='categorical', nact=6, value_copy=False, chunk=B)
r = np.random.RandomState(1)
params = torch.from_numpy((r.randn(dm.P) * 0.05).astype(np.float32)).cuda()
obs = torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device='cuda')
noise = torch.rand((B, 6), device='cuda')
for _ in range(2):
    dm.act(params, obs, noise)
torch.cuda.synchronize()
raw = dm.workspace[-2048 + 512:-2048 + 512 + 7 * 8 * 8].view(torch.int64).cpu().numpy()
st = raw[:48].reshape(6, 8)
ts = raw[48:52]
print('tile of workgroup 0: prologue %d  k loop %d  epilogue %d cycles' % (ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2]))
names = ['barrier1', 'swrite', 'barrier2', 'fetch', 'mfma']
rt = (st[5, 6] - st[0, 6]) * 10e-9         # s_memrealtime ticks (100 MHz) between the first stamps of k tiles 8 and 13
print('shader clock over k tiles 8..13: %.0f MHz' % ((st[5, 0] - st[0, 0]) / rt / 1e6))
for t in range(6):
    d = np.diff(st[t, :6])
    nxt = st[t + 1, 0] - st[t, 5] if t < 5 else 0
    print('tile %d: ' % (t + 8) + '  '.join('%s=%d' % (n, x) for n, x in zip(names, d)) + '  total=%d  loop=%d' % (st[t, 5] - st[t, 0], nxt))
