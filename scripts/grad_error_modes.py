#!/usr/bin/env python
"""Per-tensor gradient error of the NatureCNN minibatch gradient against the fp64 oracle (tests/test_gpu_large_batch.py's screened
problem) for: the product arithmetic (six exact bf16 products per multiply), the all-fp32-MFMA engines, and the fp32 CPU oracle.
    python scripts/grad_error_modes.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from baselines_amd import _lib as L  # noqa: E402
from tests.test_gpu_large_batch import _device_grad, _device_model, _problem, dev  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
om, om64, mb = _problem(B, 11)
_, g64 = om64.compute_grads(0.2, mb['obs'], mb['returns'], mb['actions'], mb['values'], mb['neglogpacs'])
_, g32 = om.compute_grads(0.2, mb['obs'], mb['returns'], mb['actions'], mb['values'], mb['neglogpacs'])
g64, g32 = g64.numpy(), g32.numpy().astype(np.float64)
params = dev(om.flat_params().astype(np.float32))
res = {}
ref_opts = dict(f32_bf16x6=0, x6_dither=0, dgrad_x6=0, relu_bits=0, c1_lds=0, wgrad_x8=0, c1_wgrad2=0, tr_epilogue=0, wgrad_tr=0, u8_bf16x3=0)
old = {k: L.get_option(k) for k in ref_opts}
dm = _device_model(B)
res['device (product)'] = _device_grad(dm, params, mb, 0.2)[0]
for k, v in ref_opts.items():
    L.set_option(k, v)
dm = _device_model(B)
res['device (fp32 MFMA)'] = _device_grad(dm, params, mb, 0.2)[0]
for k, v in old.items():
    L.set_option(k, v)
res['fp32 CPU oracle'] = g32
print('%-28s' % ('B = %d' % B) + ''.join('%22s' % k for k in res))
for t in dm.tensors:
    sl = slice(t['offset'], t['offset'] + t['size'])
    sc = np.abs(g64[sl]).max()
    print('%-28s' % t['name'] + ''.join('%22.2e' % (np.abs(g[sl] - g64[sl]).max() / sc) for g in res.values()))


# ---- forward only: value / logit error vs fp64, one engine family at a time
with torch.no_grad():
    pi64, v64 = om64.forward(mb['obs'])
    pi32, v32 = om.forward(mb['obs'])
print('forward: max |v - v64| / max |v64|, max |logit - logit64|')
dv = (v32.double() - v64) / v64.abs().max()
print('  %-44s %.2e %.2e   v error / max |v64|: mean %+.2e rms %.2e' % ('fp32 CPU oracle', float(dv.abs().max()), float((pi32.double() - pi64).abs().max()),
                                                                     float(dv.mean()), float(dv.pow(2).mean().sqrt())))
cases = [('product arithmetic (defaults)', {}), ('conv1 on the fp32 pipe (u8_bf16x3=0)', dict(u8_bf16x3=0, c1_lds=0)),
         ('conv2/conv3/fc1 on the fp32 pipe (f32_bf16x6=0)', dict(f32_bf16x6=0)), ('all fp32 MFMA', ref_opts)]
for name, opts in cases:
    old = {k: L.get_option(k) for k in opts}
    for k, v in opts.items():
        L.set_option(k, v)
    dmx = _device_model(B)
    _, vd, _, pdd = dmx.act(params, dev(mb['obs']), None, want_actions=False, want_pdparam=True)
    for k, v in old.items():
        L.set_option(k, v)
    dv = (vd.cpu().double() - v64) / v64.abs().max()
    print('  %-44s %.2e %.2e   v error / max |v64|: mean %+.2e rms %.2e' % (name, float(dv.abs().max()), float((pdd.cpu().double() - pi64).abs().max()),
                                                                         float(dv.mean()), float(dv.pow(2).mean().sqrt())))


# ---- which option moves the gradient error?  one toggle at a time from the product defaults
names = {t['name']: slice(t['offset'], t['offset'] + t['size']) for t in dm.tensors}
def errs(g):
    return ' '.join('%s %.2e' % (k.split('/')[-2] + '/' + k.split('/')[-1], np.abs(g[sl] - g64[sl]).max() / np.abs(g64[sl]).max())
                    for k, sl in names.items() if k.endswith(('c1/w', 'c3/w', 'fc1/w', 'vf/b', 'pi/b')))
for k, v in ref_opts.items():
    oldv = L.get_option(k)
    L.set_option(k, v)
    dmx = _device_model(B)
    g = _device_grad(dmx, params, mb, 0.2)[0]
    L.set_option(k, oldv)
    print('  %-14s -> %d : %s' % (k, v, errs(g)))
L.set_option('heads_wave', 0)
g = _device_grad(_device_model(B), params, mb, 0.2)[0]
L.set_option('heads_wave', 1)
print('  %-14s -> %d : %s' % ('heads_wave', 0, errs(g)))
