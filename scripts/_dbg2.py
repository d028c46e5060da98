import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baselines_amd import _lib as L, ops
def grads(B, u8, x6):
    L.set_option('u8_bf16x3', u8); L.set_option('f32_bf16x6', x6)
    dm = ops.DeviceModel(network='cnn', ob_shape=(84,84,4), ob_dtype=np.uint8, pd_kind='categorical', nact=6, value_copy=False, chunk=B)
    r = np.random.RandomState(1)
    params = torch.from_numpy((r.randn(dm.P) * 0.05).astype(np.float32)).cuda()
    obs = torch.from_numpy(r.randint(0, 256, (B,84,84,4)).astype(np.uint8)).cuda()
    act = torch.from_numpy(r.randint(0, 6, B).astype(np.int32)).cuda()
    ret, val_, nlp = (torch.from_numpy(r.randn(B).astype(np.float32)).cuda() for _ in range(3))
    nlp = nlp.abs() + 1.0
    g = torch.empty(dm.P, dtype=torch.float32, device='cuda'); st = torch.empty(5, dtype=torch.float32, device='cuda')
    dm.grad(params, obs, act, ret, val_, nlp, None, B, 1, 1, 0.2, 0.01, 0.5, g, st)
    return dm, g.cpu().numpy()
B = 1152
res = {}
for name, (u8, x6) in {'A11': (1, 1), 'B10': (1, 0), 'C01': (0, 1), 'D00': (0, 0), 'D00b': (0, 0), 'C01b': (0, 1), 'A11b': (1, 1)}.items():
    dm, res[name] = grads(B, u8, x6)
def cmp(a, b):
    out = []
    for t in dm.tensors:
        x = res[a][t['offset']:t['offset']+t['size']]; y = res[b][t['offset']:t['offset']+t['size']]
        out.append('%s=%.1e' % (t['name'].split('/')[-2] + '/' + t['name'].split('/')[-1], np.abs(x-y).max()/np.abs(y).max()))
    print(a, 'vs', b, ' '.join(out))
for a, b in [('A11', 'A11b'), ('D00', 'D00b'), ('C01', 'C01b'), ('A11', 'B10'), ('A11', 'D00'), ('A11', 'C01'), ('C01', 'D00')]:
    cmp(a, b)
