import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baselines_amd import _lib as L, ops
def grads(B, val, opt='f32_bf16x6'):
    L.set_option(opt, val)
    dm = ops.DeviceModel(network='cnn', ob_shape=(84,84,4), ob_dtype=np.uint8, pd_kind='categorical', nact=6, value_copy=False, chunk=B)
    r = np.random.RandomState(1)
    params = torch.from_numpy((r.randn(dm.P) * 0.05).astype(np.float32)).cuda()
    obs = torch.from_numpy(r.randint(0, 256, (B,84,84,4)).astype(np.uint8)).cuda()
    act = torch.from_numpy(r.randint(0, 6, B).astype(np.int32)).cuda()
    ret, val_, nlp = (torch.from_numpy(r.randn(B).astype(np.float32)).cuda() for _ in range(3))
    nlp = nlp.abs() + 1.0
    g = torch.empty(dm.P, dtype=torch.float32, device='cuda'); st = torch.empty(5, dtype=torch.float32, device='cuda')
    dm.grad(params, obs, act, ret, val_, nlp, None, B, 1, 1, 0.2, 0.01, 0.5, g, st)
    return dm, g.cpu().numpy(), st.cpu().numpy()
for B in (160, 1152):
    dm, g1, s1 = grads(B, 1)
    dm, g0, s0 = grads(B, 0)
    print('B', B, 'stats', s1, s0)
    for t in dm.tensors:
        a = g1[t['offset']:t['offset']+t['size']]; b = g0[t['offset']:t['offset']+t['size']]
        print('  %-28s max|g|=%.3e maxdiff=%.3e rel=%.2e' % (t['name'], np.abs(b).max(), np.abs(a-b).max(), np.abs(a-b).max()/np.abs(b).max()))
print('--- with u8_bf16x3=0')
L.set_option('u8_bf16x3', 0)
dm, g1, s1 = grads(1152, 1)
dm, g0, s0 = grads(1152, 0)
for t in dm.tensors:
    a = g1[t['offset']:t['offset']+t['size']]; b = g0[t['offset']:t['offset']+t['size']]
    print('  %-28s max|g|=%.3e maxdiff=%.3e rel=%.2e' % (t['name'], np.abs(b).max(), np.abs(a-b).max(), np.abs(a-b).max()/np.abs(b).max()))
print('--- u8 on/off at 1152, x6=1')
L.set_option('f32_bf16x6', 1)
dm, g1, s1 = grads(1152, 1, 'u8_bf16x3')
dm, g0, s0 = grads(1152, 0, 'u8_bf16x3')
for t in dm.tensors:
    a = g1[t['offset']:t['offset']+t['size']]; b = g0[t['offset']:t['offset']+t['size']]
    print('  %-28s max|g|=%.3e maxdiff=%.3e rel=%.2e' % (t['name'], np.abs(b).max(), np.abs(a-b).max(), np.abs(a-b).max()/np.abs(b).max()))
