#!/usr/bin/env python
"""Teacher-forced config-3 update under each arithmetic mode: where does the per-step evaluation error of the device come from?
    python scripts/teacher_forced_modes.py [steps] [mode ...]
For every mode a fresh Model is built with the options set, driven through the fp64 teacher's steps (tests/_teacher_forced.py)
and compared: per step the worst of the 5 statistics over (1 + |s|), per statistic the worst over steps, post-step parameter
error.  The fp32 CPU restatement started from the same states is the yardstick."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from baselines_amd import _lib as L  # noqa: E402
from tests import _teacher_forced as TF  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
MODES = {
    'default': {},
    'c1_fp32': dict(u8_bf16x3=0, c1_lds=0),
    'c1wgrad_fp32': dict(c1_wgrad2=0),
    'x6_fp32': dict(f32_bf16x6=0),
    'no_dither': dict(x6_dither=0),
    'no_x6c': dict(conv_x6c=0),
    'heads0': dict(heads_wave=0),
    'all_fp32': dict(f32_bf16x6=0, x6_dither=0, dgrad_x6=0, relu_bits=0, c1_lds=0, wgrad_x8=0, c1_wgrad2=0, tr_epilogue=0,
                     wgrad_tr=0, u8_bf16x3=0),
}
want = sys.argv[2:] or list(MODES)
model, ro = TF.make_model_and_rollout()
traj = TF.teacher_trajectory(model, ro, steps=steps)
names = ['pg', 'vf', 'ent', 'kl', 'clipfrac']
e32 = np.array([np.abs(r['s32'] - r['s64']) / (1 + np.abs(r['s64'])) for r in traj])
print('fp32 CPU restatement: worst stat err/(1+|s|) per statistic', dict(zip(names, ['%.2e' % x for x in e32.max(0)])),
      'param err after step: max %.2e' % max(float(np.abs(r['p32_after'] - r['p64_after']).max()) for r in traj))
out = {}
for mode in want:
    opts = MODES[mode]
    old = {k: L.get_option(k) for k in opts}
    for k, v in opts.items():
        L.set_option(k, v)
    m2, _ = TF.make_model_and_rollout() if opts else (model, None)
    res = TF.device_teacher_forced(m2, ro, traj)
    for k, v in old.items():
        L.set_option(k, v)
    er = TF.errors(traj, res)
    se = np.array([e['stat_err'] for e in er])
    sa = np.array([e['stat_abs'] for e in er])
    signed = np.array([r['stats'] - t['s64'] for r, t in zip(res, traj)])
    print('%-14s worst/(1+|s|) per stat %s | per step %s | param err max %.2e' % (
        mode, ' '.join('%s %.2e' % (n, x) for n, x in zip(names, se.max(0))), ' '.join('%.1e' % x for x in se.max(1)),
        max(e['param_err'] for e in er)))
    print('%-14s signed vf err per step %s' % ('', ' '.join('%+.1e' % x for x in signed[:, 1])))
    print('%-14s signed pg err per step %s' % ('', ' '.join('%+.1e' % x for x in signed[:, 0])))
    out[mode] = dict(stat_err=se.tolist(), stat_abs=sa.tolist(), param_err=[e['param_err'] for e in er])
out['fp32_cpu'] = dict(stat_err=e32.tolist(), param_err=[float(np.abs(r['p32_after'] - r['p64_after']).max()) for r in traj])
out['s64'] = [r['s64'].tolist() for r in traj]
path = os.environ.get('MRL_TF_REPORT')
if path:
    json.dump(out, open(path, 'w'))
