#!/usr/bin/env python
"""How far do FREE-RUNNING fp32 trajectories of the config-3 update spread from the fp64 trajectory (and from each other)?
Companion of scripts/teacher_forced_modes.py: that one shows the per-step evaluation error of the device is fp32-class; this one
runs the same 16 steps without teacher forcing for (a) the device in product arithmetic, (b) the device on the all-fp32-MFMA
engines, (c) the device with the row alternation off, (d) the fp32 CPU restatement, each against the fp64 oracle.  Any two fp32
implementations differ in rounding only; Adam's step alpha * m / (sqrt(v) + eps) has sensitivity up to 1 / eps = 1e5 to an entry's
gradient, so the trajectories separate -- the spread between (a) ... (d) is the size of that effect, not an arithmetic defect.
    python scripts/free_running_spread.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from baselines_amd import _lib as L  # noqa: E402
from oracle import ppo2_numpy as O  # noqa: E402
from oracle.ppo2_torch import OracleModel  # noqa: E402
from tests import _teacher_forced as TF  # noqa: E402

MODES = {
    'device product': {},
    'device all-fp32-MFMA': dict(f32_bf16x6=0, x6_dither=0, dgrad_x6=0, relu_bits=0, c1_lds=0, wgrad_x8=0, c1_wgrad2=0, tr_epilogue=0,
                                 wgrad_tr=0, u8_bf16x3=0),
    'device no alternation': dict(x6_dither=0),
    'device conv1 fp32': dict(u8_bf16x3=0, c1_lds=0),
}
torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
model, ro = TF.make_model_and_rollout()
p0 = model.get_flat_params()
np.random.seed(0)
om = OracleModel(**TF.CNN_KW)
om64 = OracleModel(dtype=torch.float64, params=om.params_numpy(), **TF.CNN_KW)
f = {k: O.sf01(getattr(ro, k).cpu().numpy()) for k in ('obs', 'actions', 'returns', 'values', 'neglogpacs')}
perm = []
inds = np.arange(TF.N * TF.T)
for e in range(TF.E):
    np.random.shuffle(inds)
    for lo in range(0, TF.N * TF.T, TF.B):
        perm.append(inds[lo:lo + TF.B].copy())
s64, p64, s32, p32 = [], [], [], []
for idx in perm:
    args = (TF.LR, TF.CLIP, f['obs'][idx], f['returns'][idx], None, f['actions'][idx], f['values'][idx], f['neglogpacs'][idx])
    s32.append(np.array(om.train(*args)))
    s64.append(np.array(om64.train(*args)))
    p32.append(om.flat_params().astype(np.float64))
    p64.append(om64.flat_params())
s64 = np.array(s64)
runs = {'fp32 CPU restatement': (np.array(s32), p32)}
for name, opts in MODES.items():
    old = {k: L.get_option(k) for k in opts}
    for k, v in opts.items():
        L.set_option(k, v)
    m2, _ = TF.make_model_and_rollout()
    assert np.array_equal(m2.get_flat_params(), p0)
    st, ps = [], []
    for idx in perm:
        st.append(m2.train_indexed(TF.LR, TF.CLIP, ro, m2.indices_to_device(idx)).cpu().numpy().astype(np.float64))
        ps.append(m2.get_flat_params().astype(np.float64))
    for k, v in old.items():
        L.set_option(k, v)
    runs[name] = (np.array(st), ps)
out = {}
print('per step: max over the 5 statistics of |s - s64| / (1 + |s64|)   ||   max |p - p64| after the step')
for name, (st, ps) in runs.items():
    e = (np.abs(st - s64) / (1 + np.abs(s64))).max(1)
    d = [float(np.abs(a - b).max()) for a, b in zip(ps, p64)]
    out[name] = dict(stat_err=e.tolist(), param_drift=d, vf_abs_err=np.abs(st[:, 1] - s64[:, 1]).tolist())
    print('%-24s %s' % (name, ' '.join('%.1e' % x for x in e)))
    print('%-24s %s' % ('  value-loss abs err', ' '.join('%.1e' % x for x in np.abs(st[:, 1] - s64[:, 1]))))
    print('%-24s %s' % ('  param drift', ' '.join('%.1e' % x for x in d)))
names = list(runs)
print('pairwise max |p_a - p_b| after step 16:')
for i, a in enumerate(names):
    for b in names[i + 1:]:
        print('  %-24s vs %-24s %.2e' % (a, b, float(np.abs(runs[a][1][-1] - runs[b][1][-1]).max())))
path = os.environ.get('MRL_TF_REPORT')
if path:
    json.dump(out, open(path, 'w'))
