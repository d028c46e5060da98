import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baselines_amd import _lib as L, ops
def run(B, u8, x6):
    L.set_option('u8_bf16x3', u8); L.set_option('f32_bf16x6', x6)
    dm = ops.DeviceModel(network='cnn', ob_shape=(84,84,4), ob_dtype=np.uint8, pd_kind='categorical', nact=6, value_copy=False, chunk=B)
    dm.workspace.zero_()
    r = np.random.RandomState(1)
    params = torch.from_numpy((r.randn(dm.P) * 0.05).astype(np.float32)).cuda()
    obs = torch.from_numpy(r.randint(0, 256, (B,84,84,4)).astype(np.uint8)).cuda()
    act = torch.from_numpy(r.randint(0, 6, B).astype(np.int32)).cuda()
    ret, val_, nlp = (torch.from_numpy(r.randn(B).astype(np.float32)).cuda() for _ in range(3))
    nlp = nlp.abs() + 1.0
    g = torch.empty(dm.P, dtype=torch.float32, device='cuda'); st = torch.empty(5, dtype=torch.float32, device='cuda')
    dm.grad(params, obs, act, ret, val_, nlp, None, B, 1, 1, 0.2, 0.01, 0.5, g, st)
    torch.cuda.synchronize()
    bufs = {}
    off = 0
    al = lambda n: (n + 255) // 256 * 256
    for name, oe in (('c1', 12800), ('c2', 5184), ('c3', 3136), ('fc1', 512)):
        n = B * oe * 4
        bufs['h_' + name] = dm.workspace[off:off + n].view(torch.float32).cpu().numpy().copy(); off += al(n)
        bufs['dz_' + name] = dm.workspace[off:off + n].view(torch.float32).cpu().numpy().copy(); off += al(n)
    return bufs, g.cpu().numpy()
B = 1152
a, ga = run(B, 1, 1)
c, gc = run(B, 0, 1)
for k in a:
    d = np.abs(a[k] - c[k]); 
    bad = np.argwhere(d > 1e-4 * np.abs(a[k]).max()).ravel()
    print(k, 'max|x|=%.3e maxdiff=%.3e nbad=%d' % (np.abs(a[k]).max(), d.max(), bad.size), 'first bad idx', bad[:5], 'rows', (bad[:5] // (a[k].size // B)))
