#!/usr/bin/env python
"""bench.py's self-check, taken apart: the gradient of one minibatch of the benched shape computed in ONE chunk and in 8192-sample
chunks, per tensor, each against a referee (all engines on the fp32 MFMA pipe, 8192-sample chunks), with one engine option toggled
at a time for the one-chunk path.
    python scripts/chunk_diff.py [num_envs] [nsteps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from baselines_amd import _lib as L, ops  # noqa: E402
from baselines_amd.common import set_global_seeds  # noqa: E402
from baselines_amd.common.policies import build_policy  # noqa: E402
from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv  # noqa: E402
from baselines_amd.ppo2 import Model, Runner  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nbatch = N * T
B = nbatch // 4
set_global_seeds(0)
env = SyntheticVecEnv('atari', N, seed=1000)
policy = build_policy(env, 'cnn', value_network=None)
model = Model(policy=policy, ob_space=env.observation_space, ac_space=env.action_space, nbatch_act=N, nbatch_train=B, nsteps=T,
              ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5)
runner = Runner(env=env, model=model, nsteps=T, gamma=0.99, lam=0.95, return_host=False)
for _ in range(3):
    runner.run()
ro = runner.rollout
last_values = model.value_dev(runner.obs)
ops.gae(ro.rewards, ro.values, ro.dones, last_values, runner._dones_dev, 0.99, 0.95, out=ro.returns)
idx = torch.from_numpy(np.random.RandomState(7).permutation(nbatch)[:B]).to(model.device)


def grad(chunk, opts=None):
    opts = opts or {}
    old = {k: L.get_option(k) for k in opts}
    for k, v in opts.items():
        L.set_option(k, v)
    dm = ops.DeviceModel(chunk=chunk, device=model.device, **policy.device_model_kwargs())
    g = torch.empty(dm.P, dtype=torch.float32, device=model.device)
    st = torch.empty(5, dtype=torch.float32, device=model.device)
    dm.grad(model.params, ro.obs, ro.actions, ro.returns, ro.values, ro.neglogpacs, idx, B, T, N, 0.1, 0.01, 0.5, g, st)
    torch.cuda.synchronize()
    for k, v in old.items():
        L.set_option(k, v)
    return g.double().cpu().numpy(), st.double().cpu().numpy()


ref_opts = dict(f32_bf16x6=0, x6_dither=0, dgrad_x6=0, relu_bits=0, c1_lds=0, wgrad_x8=0, c1_wgrad2=0, tr_epilogue=0, wgrad_tr=0, u8_bf16x3=0)
g_ref, st_ref = grad(8192, ref_opts)
g_ref2, _ = grad(B, ref_opts)
g_big, st_big = grad(B)
g_small, st_small = grad(8192)
tensors = model.dm.tensors
gs = np.abs(g_ref).max()
print('num_envs %d  minibatch %d  global grad scale %.4g' % (N, B, gs))
print('stats  big-small %.3g   big-ref %.3g   small-ref %.3g' % (np.abs(st_big - st_small).max(), np.abs(st_big - st_ref).max(), np.abs(st_small - st_ref).max()))
print('whole vector, max |diff| / global scale:  big-small %.3g  big-ref %.3g  small-ref %.3g  ref(one chunk)-ref(8192) %.3g' % (
    np.abs(g_big - g_small).max() / gs, np.abs(g_big - g_ref).max() / gs, np.abs(g_small - g_ref).max() / gs, np.abs(g_ref2 - g_ref).max() / gs))
print('%-28s %10s %12s %12s %12s %12s' % ('tensor', 'scale', 'big-small', 'big-ref', 'small-ref', 'ref1-ref'))
for t in tensors:
    sl = slice(t['offset'], t['offset'] + t['size'])
    sc = np.abs(g_ref[sl]).max()
    print('%-28s %10.3g %12.2e %12.2e %12.2e %12.2e' % (t['name'], sc, np.abs(g_big[sl] - g_small[sl]).max() / sc, np.abs(g_big[sl] - g_ref[sl]).max() / sc,
                                                     np.abs(g_small[sl] - g_ref[sl]).max() / sc, np.abs(g_ref2[sl] - g_ref[sl]).max() / sc))


def worst(g):
    out = []
    for t in tensors:
        sl = slice(t['offset'], t['offset'] + t['size'])
        out.append((np.abs(g[sl] - g_ref[sl]).max() / np.abs(g_ref[sl]).max(), t['name'].split('/', 1)[-1]))
    w = max(out)
    return '%.2e (%s)  whole / global scale %.2e' % (w[0], w[1], np.abs(g - g_ref).max() / gs)


print('one engine option at a time (vs referee, worst tensor):')
for chunk in (B, 8192):
    for name, opts in (('defaults', {}), ('conv_x6c=0', dict(conv_x6c=0)), ('x6_frag=0', dict(x6_frag=0)), ('x6_dither=0', dict(x6_dither=0)),
                       ('wgrad_pipe=0', dict(wgrad_pipe=0)), ('dgrad_x6=0', dict(dgrad_x6=0)), ('wgrad_tr=0', dict(wgrad_tr=0)),
                       ('f32_bf16x6=0', dict(f32_bf16x6=0)), ('u8_bf16x3=0', dict(u8_bf16x3=0, c1_lds=0)), ('relu_bits=0', dict(relu_bits=0)),
                       ('tr_epilogue=0', dict(tr_epilogue=0))):
        try:
            g, _ = grad(chunk, opts)
            print('  chunk %6d  %-14s %s' % (chunk, name, worst(g)))
        except Exception as exc:
            print('  chunk %6d  %-14s failed: %r' % (chunk, name, exc))


# ---- is the one-chunk / 8192-chunk difference the forward's?  (round 5: the class-resident conv forward alternates signs over the rows of
# whole-image tiles, whose positions depend on the chunk size; a ReLU unit whose pre-activation is zero to rounding then takes a different
# side in the two chunkings.)  With the tiled conv forward the forward pass is bit-identical in any chunking.
print('one chunk vs 8192-sample chunks under the same options (whole / global scale; stats):')
for name, opts in (('defaults', {}), ('conv_x6c=0', dict(conv_x6c=0)), ('conv_x6c=0 x6_frag=0', dict(conv_x6c=0, x6_frag=0)),
                   ('x6_dither=0', dict(x6_dither=0)), ('x6_dither=0 conv_x6c=0', dict(x6_dither=0, conv_x6c=0))):
    gb, sb = grad(B, opts)
    gsm, ss = grad(8192, opts)
    d = np.abs(gb - gsm)
    per = []
    for t in tensors:
        sl = slice(t['offset'], t['offset'] + t['size'])
        per.append('%s %.1e' % (t['name'].split('/', 1)[-1], d[sl].max() / np.abs(g_ref[sl]).max()))
    print('  %-24s %.3g   stats %.3g   rel L2 %.3g   | %s' % (name, d.max() / gs, np.abs(sb - ss).max(), np.linalg.norm(gb - gsm) / np.linalg.norm(gsm), '  '.join(per)))
# where the default difference sits in fc1/w: per output unit (column)
t = [t for t in tensors if t['name'].endswith('fc1/w')][0]
sl = slice(t['offset'], t['offset'] + t['size'])
dm_ = np.abs(g_big[sl] - g_small[sl]).reshape(t['shape'])
sc = np.abs(g_ref[sl]).max()
col = dm_.max(axis=0) / sc
print('fc1/w %s: columns with a difference above 1e-5 / 1e-6 of the tensor scale: %d / %d of %d; largest %s' % (
    t['shape'], int((col > 1e-5).sum()), int((col > 1e-6).sum()), col.size, np.sort(col)[-5:]))
