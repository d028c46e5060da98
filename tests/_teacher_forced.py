"""Teacher-forced config-3 update: separates per-step EVALUATION error of the device update
from accumulated TRAJECTORY drift.

The teacher is the fp64 oracle (oracle/ppo2_torch.py, a restatement of ppo2/model.py:57-114,133-158).  Before every one of
the 16 minibatch steps of a config-3 update (ppo2/ppo2.py:154-166: nature_cnn, num_envs = 256, nsteps = 128, 4 epochs x 4
minibatches of 8192 out of the rollout the device Runner stored) the teacher's parameters and Adam slots are rounded to
fp32; the teacher itself continues FROM THE ROUNDED STATE (so device, fp32 restatement and teacher all start the step at
bit-identical fp32-representable values and the comparison is exact: what differs afterwards is what the step's own
arithmetic did).  Shared by tests/test_gpu_benched_shapes.py and scripts/teacher_forced_modes.py."""
import os

import numpy as np
import torch

from oracle import ppo2_numpy as O
from oracle.ppo2_torch import OracleModel

CNN_KW = dict(network='cnn', ob_shape=(84, 84, 4), ob_dtype=np.uint8, pd_kind='categorical', nact=6, value_network=None,
              ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5)
N, T, M, E = 256, 128, 4, 4
B = N * T // M
LR, CLIP = 2.5e-4, 0.1


def make_model_and_rollout(seed=1003):
    from baselines_amd.common import set_global_seeds
    from baselines_amd.common.policies import build_policy
    from baselines_amd.common.vec_env.synthetic_vec_env import SyntheticVecEnv
    from baselines_amd.ppo2 import Model, Runner
    env = SyntheticVecEnv('atari', N, seed=seed)
    set_global_seeds(0)
    policy = build_policy(env, 'cnn')
    model = Model(policy=policy, ob_space=env.observation_space, ac_space=env.action_space, nbatch_act=N, nbatch_train=B,
                  nsteps=T, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5)
    runner = Runner(env=env, model=model, nsteps=T, gamma=0.99, lam=0.95, return_host=False)
    runner.run()
    runner.run()                                       # second rollout: episodes have ended, dones are mixed in
    return model, runner.rollout


def _state_flat(om, which):
    d = {'p': om.p, 'm': om.m, 'v': om.v}[which]
    return np.concatenate([d[k].detach().numpy().reshape(-1) for k in om.names])


def _set_state(om, which, flat):
    d = {'p': om.p, 'm': om.m, 'v': om.v}[which]
    off = 0
    with torch.no_grad():
        for k in om.names:
            n = d[k].numel()
            d[k].copy_(torch.from_numpy(np.asarray(flat[off:off + n]).reshape(tuple(d[k].shape))).to(d[k].dtype))
            off += n


def teacher_trajectory(model, ro, steps=E * M, with_fp32=True, seed=0):
    """The teacher's 16 steps.  Returns a list of dicts, one per step: idx (env-major minibatch indices), p32 / m32 / v32
    (the fp32 state every implementation starts the step from), s64 (the 5 statistics, fp64 arithmetic at that state),
    p64_after (the fp64 post-step parameters), and -- the yardstick of what 'fp32-class' means -- s32 / p32_after of the
    fp32 CPU restatement started from the same state."""
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    np.random.seed(seed)
    om = OracleModel(**CNN_KW)                         # same ortho-init stream as Model (a2c/utils.py:20-35)
    np.testing.assert_array_equal(model.get_flat_params(), om.flat_params())
    om64 = OracleModel(dtype=torch.float64, params=om.params_numpy(), **CNN_KW)
    f = {k: O.sf01(getattr(ro, k).cpu().numpy()) for k in ('obs', 'actions', 'returns', 'values', 'neglogpacs')}
    inds = np.arange(N * T)
    out = []
    for epoch in range(E):
        np.random.shuffle(inds)                        # ppo2.py:157-158
        for lo in range(0, N * T, B):
            if len(out) >= steps:
                return out
            idx = inds[lo:lo + B].copy()
            st = {w: _state_flat(om64, w).astype(np.float32) for w in 'pmv'}
            for w in 'pmv':                            # the teacher continues from the rounded state
                _set_state(om64, w, st[w].astype(np.float64))
            args = (LR, CLIP, f['obs'][idx], f['returns'][idx], None, f['actions'][idx], f['values'][idx], f['neglogpacs'][idx])
            rec = dict(idx=idx, p32=st['p'], m32=st['m'], v32=st['v'], beta1_power=np.float32(om64.beta1_power),
                       beta2_power=np.float32(om64.beta2_power))
            if with_fp32:
                for w in 'pmv':
                    _set_state(om, w, st[w])
                om.beta1_power, om.beta2_power = np.float32(om64.beta1_power), np.float32(om64.beta2_power)
                rec['s32'] = np.array(om.train(*args), dtype=np.float64)
                rec['p32_after'] = om.flat_params().astype(np.float64)
            rec['s64'] = np.array(om64.train(*args), dtype=np.float64)
            rec['p64_after'] = om64.flat_params().astype(np.float64)
            out.append(rec)
    return out


def device_teacher_forced(model, ro, traj):
    """Runs the device through the teacher's steps (state overwritten before each).  Returns per step: stats (5,), params after."""
    res = []
    for rec in traj:
        model.set_flat_params(rec['p32'])
        model.adam_m.copy_(torch.from_numpy(rec['m32']))
        model.adam_v.copy_(torch.from_numpy(rec['v32']))
        model.beta1_power, model.beta2_power = np.float32(rec['beta1_power']), np.float32(rec['beta2_power'])
        idx_dev = model.indices_to_device(rec['idx'])
        st = model.train_indexed(LR, CLIP, ro, idx_dev).cpu().numpy().astype(np.float64)
        res.append(dict(stats=st, p_after=model.get_flat_params().astype(np.float64)))
    return res


def errors(traj, res):
    """per step: max over the 5 statistics of |device - fp64| / (1 + |fp64|), the same for the fp32 restatement, and the
    post-step parameter error max |p - p64|."""
    out = []
    for rec, r in zip(traj, res):
        e = dict(stat_err=np.abs(r['stats'] - rec['s64']) / (1.0 + np.abs(rec['s64'])),
                 stat_abs=np.abs(r['stats'] - rec['s64']),
                 param_err=float(np.abs(r['p_after'] - rec['p64_after']).max()))
        if 's32' in rec:
            e['stat_err_fp32'] = np.abs(rec['s32'] - rec['s64']) / (1.0 + np.abs(rec['s64']))
            e['param_err_fp32'] = float(np.abs(rec['p32_after'] - rec['p64_after']).max())
        out.append(e)
    return out
