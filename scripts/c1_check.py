#!/usr/bin/env python
"""conv1 engine variants against each other on the device: forward (logits / values) and the whole gradient (ReLU bit masks,
weight gradient) must be BIT-identical between option values -- same products, same accumulation order.
    python scripts/c1_check.py c1_lds=2,3 [B ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from baselines_amd import _lib, ops  # noqa: E402

opt, vals = sys.argv[1].split('=')
vals = [int(v) for v in vals.split(',')]
sizes = [int(x) for x in sys.argv[2:]] or [1, 7, 255, 256, 257, 1000, 4096 + 13]
old = _lib.get_option(opt)
ok = True
for B in sizes:
    r = np.random.RandomState(B)
    outs = []
    for v in vals:
        _lib.set_option(opt, v)
        dm = ops.DeviceModel(network='cnn', ob_shape=(84, 84, 4), ob_dtype=np.uint8, pd_kind='categorical', nact=6, chunk=B)
        rr = np.random.RandomState(B)
        params = torch.from_numpy((rr.randn(dm.P) * 0.05).astype(np.float32)).cuda()
        obs = torch.from_numpy(rr.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)).cuda()
        act = torch.from_numpy(rr.randint(0, 6, B).astype(np.int32)).cuda()
        ret, val_, nlp = (torch.from_numpy(rr.randn(B).astype(np.float32)).cuda() for _ in range(3))
        nlp = nlp.abs() + 1.0
        _, vv, _, pd = dm.act(params, obs, None, want_actions=False, want_pdparam=True)
        g = torch.empty(dm.P, dtype=torch.float32, device='cuda')
        st = torch.empty(5, dtype=torch.float32, device='cuda')
        dm.grad(params, obs, act, ret, val_, nlp, None, B, 1, 1, 0.2, 0.01, 0.5, g, st)
        torch.cuda.synchronize()
        outs.append((vv.cpu(), pd.cpu(), g.cpu(), st.cpu()))
    for v, o in zip(vals[1:], outs[1:]):
        same = all(torch.equal(a, b) for a, b in zip(outs[0], o))
        ok &= same
        print('B=%d %s=%d vs %d: %s   (|g|max %.3e, max diff %.3e)' % (B, opt, v, vals[0], 'bit-identical' if same else 'DIFFERENT',
                                                                   float(outs[0][2].abs().max()), float((outs[0][2] - o[2]).abs().max())), flush=True)
_lib.set_option(opt, old)
sys.exit(0 if ok else 1)
