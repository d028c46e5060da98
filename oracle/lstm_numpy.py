"""
TEST INFRASTRUCTURE ONLY -- groundwork for SURVEY.md 8 (f4): the recurrent cell of the reference's `lstm` /
`cnn_lstm` policies, restated in NumPy with its closed-form backward pass (the equations a HIP kernel will implement).

Follows baselines/a2c/utils.py:81-102 `lstm(xs, ms, s, scope, nh)`:
    c = c * (1 - m);  h = h * (1 - m)                      (mask = "episode ended before this step")
    z = x @ wx + h @ wh + b;   i, f, o, u = split(z, 4)     (gate order i, f, o, u)
    i, f, o = sigmoid(.);  u = tanh(u);   c = f * c + i * u;   h = o * tanh(c)
state s = concat([c, h], axis=1) ([nenv, 2*nh], utils.py:87,101); parameter shapes wx [nin, 4nh], wh [nh, 4nh], b [4nh].
Parity unpinned at the TF boundary (tensorflow 1.x is not installed): pinned instead by torch autograd of the same
forward in float64 (tests/test_oracle_golden.py::test_lstm_closed_form_backward_matches_autograd).
"""
import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_forward(xs, ms, s0, wx, wh, b):
    """xs [T, E, nin], ms [T, E] (1.0 where the env was reset before step t), s0 [E, 2nh]
    -> hs [T, E, nh], s_last [E, 2nh], cache for the backward pass"""
    T, E, _ = xs.shape
    nh = wh.shape[0]
    c, h = s0[:, :nh].copy(), s0[:, nh:].copy()
    hs = np.zeros((T, E, nh), xs.dtype)
    cache = []
    for t in range(T):
        keep = (1.0 - ms[t])[:, None].astype(xs.dtype)
        cm, hm = c * keep, h * keep
        z = xs[t] @ wx + hm @ wh + b
        i, f, o, u = _sigmoid(z[:, :nh]), _sigmoid(z[:, nh:2 * nh]), _sigmoid(z[:, 2 * nh:3 * nh]), np.tanh(z[:, 3 * nh:])
        c = f * cm + i * u
        tc = np.tanh(c)
        h = o * tc
        hs[t] = h
        cache.append((keep, cm, hm, i, f, o, u, tc))
    return hs, np.concatenate([c, h], axis=1), cache


def lstm_backward(dhs, ds_last, xs, wx, wh, cache):
    """dhs [T, E, nh] = dL/dh_t from the heads, ds_last [E, 2nh] = dL/d(final state) (zeros in PPO2)
    -> dxs [T, E, nin], ds0 [E, 2nh], dwx, dwh, db"""
    T, E, nin = xs.shape
    nh = wh.shape[0]
    dc, dh = ds_last[:, :nh].copy(), ds_last[:, nh:].copy()
    dxs = np.zeros_like(xs)
    dwx, dwh, db = np.zeros_like(wx), np.zeros_like(wh), np.zeros(4 * nh, wx.dtype)
    for t in range(T - 1, -1, -1):
        keep, cm, hm, i, f, o, u, tc = cache[t]
        dh = dh + dhs[t]
        do = dh * tc
        dc = dc + dh * o * (1.0 - tc * tc)
        di, df, du, dcm = dc * u, dc * cm, dc * i, dc * f
        dz = np.concatenate([di * i * (1.0 - i), df * f * (1.0 - f), do * o * (1.0 - o), du * (1.0 - u * u)], axis=1)
        dwx += xs[t].T @ dz
        dwh += hm.T @ dz
        db += dz.sum(axis=0)
        dxs[t] = dz @ wx.T
        dh = (dz @ wh.T) * keep            # through h * (1 - m)
        dc = dcm * keep                    # through c * (1 - m)
    return dxs, np.concatenate([dc, dh], axis=1), dwx, dwh, db


def recurrent_minibatches(nenvs, nsteps, nminibatches, rng=np.random):
    """ppo2.py:167-180: env-wise minibatches for recurrent policies -- shuffle ENV indices, each minibatch takes
    nenvs // nminibatches whole env trajectories; yields (mbenvinds, mbflatinds) with flat index e * nsteps + t."""
    assert nenvs % nminibatches == 0
    envsperbatch = nenvs // nminibatches
    envinds = np.arange(nenvs)
    flatinds = np.arange(nenvs * nsteps).reshape(nenvs, nsteps)
    rng.shuffle(envinds)
    for start in range(0, nenvs, envsperbatch):
        mbenvinds = envinds[start:start + envsperbatch]
        yield mbenvinds, flatinds[mbenvinds].ravel()
