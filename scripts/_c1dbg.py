import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baselines_amd import _lib as L, ops
r = np.random.RandomState(1)
for B in (64, 300, 1152):
    dm = ops.DeviceModel(network='cnn', ob_shape=(84, 84, 4), ob_dtype=np.uint8, pd_kind='categorical', nact=6, value_copy=False, chunk=B)
    params = torch.from_numpy((r.randn(dm.P) * 0.05).astype(np.float32)).cuda()
    obs = torch.from_numpy(r.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)).cuda()
    outs = {}
    for v in (1, 2, 0):
        L.set_option('c1_lds', v)
        a, val, nlp, pd = dm.act(params, obs, want_actions=False, want_pdparam=True)
        outs[v] = (val.cpu().numpy().copy(), pd.cpu().numpy().copy())
    for v in (2, 0):
        dv = np.abs(outs[v][0] - outs[1][0]); dp = np.abs(outs[v][1] - outs[1][1]).max(axis=1)
        bad = np.where(dv > 1e-4)[0]
        print('B', B, 'variant', v, 'max dv', dv.max(), 'max dpd', dp.max(), 'bad samples', bad[:20], len(bad))
L.set_option('c1_lds', 1)
