"""Seeded synthetic vectorised environments with Atari-shaped / MuJoCo-shaped / CartPole-shaped
spaces (BASELINE.json configs).

  SyntheticVecEnv     device-resident: state and observations live in HBM, stepped by the HIP
                      kernels in csrc/envs.hip; exchanges torch device tensors.
  SyntheticVecEnvCPU  NumPy twin with the identical counter-hash dynamics (an ordinary host VecEnv,
                      like any gym env): feeds the oracle the same rollouts and exercises the
                      host-env path of the Runner.
"""
import time

import numpy as np

from .vec_env import VecEnv
from ..spaces import Box, Discrete

KINDS = {
    # kind: (ob_shape, ob_dtype, action space factory, reward_kind, lmin, lspan)
    'atari': ((84, 84, 4), np.uint8, lambda: Discrete(6), 0, 150, 100),
    'mujoco': ((376,), np.float32, lambda: Box(-1.0, 1.0, (17,), np.float32), 1, 600, 800),
    'cartpole': ((4,), np.float32, lambda: Discrete(2), 2, 8, 40),
}


def _mix32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over='ignore'):
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x85ebca6b)
        x ^= x >> np.uint32(13)
        x *= np.uint32(0xc2b2ae35)
        x ^= x >> np.uint32(16)
    return x


def _u32(x):
    return np.asarray(x).astype(np.uint64).astype(np.uint32) if np.asarray(x).dtype != np.uint32 else np.asarray(x)


class _SynthBase(VecEnv):
    def __init__(self, kind, num_envs, seed=0, ob_shape=None, nact=None, lmin=None, lspan=None):
        self.tstart = time.time()
        shape, dtype, acf, self.reward_kind, lm, ls = KINDS[kind]
        if ob_shape is not None:
            shape = tuple(ob_shape)
        self.kind = kind
        self.seed_value = int(seed) & 0xffffffff
        self.lmin, self.lspan = int(lmin or lm), int(lspan or ls)
        ac = acf()
        if nact is not None:
            ac = Discrete(nact) if isinstance(ac, Discrete) else Box(-1.0, 1.0, (nact,), np.float32)
        if np.dtype(dtype) == np.uint8:
            ob = Box(0, 255, shape, np.uint8)
        else:
            ob = Box(-1.0, 1.0, shape, np.float32)
        VecEnv.__init__(self, num_envs, ob, ac)
        self.ob_elems = int(np.prod(shape))
        self.ob_u8 = int(np.dtype(dtype) == np.uint8)
        self.discrete = int(isinstance(ac, Discrete))
        self.spec = None


class SyntheticVecEnvCPU(_SynthBase):
    """Host twin (pure NumPy).  infos carry {'episode': {'r','l'}} when an episode ends, like
    bench.Monitor (bench/monitor.py:58-77)."""

    def __init__(self, kind, num_envs, seed=0, **kw):
        super().__init__(kind, num_envs, seed, **kw)
        N = num_envs
        self.ep = np.zeros(N, np.uint32)
        self.st = np.zeros(N, np.int32)
        self.ep_ret = np.zeros(N, np.float32)
        with np.errstate(over='ignore'):
            self.key = _mix32(np.uint32(self.seed_value) * np.uint32(0x9E3779B1) + np.arange(N, dtype=np.uint32))
        self._actions = None

    def _k2(self):
        with np.errstate(over='ignore'):
            return _mix32(self.key ^ _mix32(self.ep * np.uint32(0x632BE5AB) + self.st.astype(np.uint32)))

    def _obs(self):
        k2 = self._k2()
        words = (self.ob_elems + 3) // 4 if self.ob_u8 else self.ob_elems
        with np.errstate(over='ignore'):
            h = _mix32(k2[:, None] + np.arange(words, dtype=np.uint32)[None, :] * np.uint32(0x9E3779B9))
        if self.ob_u8:
            b = h.view(np.uint8).reshape(self.num_envs, words * 4)[:, :self.ob_elems]   # little endian
            return b.reshape((self.num_envs,) + self.observation_space.shape).copy()
        f = (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 8388608.0) - np.float32(1.0)
        return f.reshape((self.num_envs,) + self.observation_space.shape)

    def reset(self):
        self.ep[:] = 0
        self.st[:] = 0
        self.ep_ret[:] = 0
        return self._obs()

    def step_async(self, actions):
        self._actions = actions

    def step_wait(self):
        N = self.num_envs
        k2 = self._k2()
        if self.discrete:
            a = np.asarray(self._actions).astype(np.int64).astype(np.uint32).reshape(N)
        else:
            a = np.zeros(N, np.uint32)
        with np.errstate(over='ignore'):
            hr = _mix32((k2 ^ np.uint32(0xA5A5A5A5)) + a * np.uint32(0x27D4EB2F))
            L = self.lmin + (_mix32(self.key ^ (self.ep * np.uint32(0x85EBCA77) + np.uint32(0x1234567)))
                             % np.uint32(self.lspan)).astype(np.int64)
        if self.reward_kind == 0:
            r = np.where(hr < np.uint32(214748365), -1.0, np.where(hr >= np.uint32(4080218931), 1.0, 0.0))
            r = r.astype(np.float32)
        elif self.reward_kind == 1:
            r = (hr >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 8388608.0) - np.float32(1.0)
        else:
            r = np.ones(N, np.float32)
        ret = (self.ep_ret + r).astype(np.float32)
        length = self.st + 1
        done = length >= L
        infos = [{} for _ in range(N)]
        for e in np.nonzero(done)[0]:
            infos[e] = {'episode': {'r': float(ret[e]), 'l': int(length[e]), 't': round(time.time() - self.tstart, 6)}}
        self.ep = np.where(done, self.ep + np.uint32(1), self.ep).astype(np.uint32)
        self.st = np.where(done, 0, length).astype(np.int32)
        self.ep_ret = np.where(done, np.float32(0), ret).astype(np.float32)
        return self._obs(), r, done.astype(np.bool_), infos


class SyntheticVecEnv(_SynthBase):
    """Device-resident env: reset()/step_wait() return torch device tensors; actions are device
    tensors (int32 [N] for Discrete).  Finished-episode statistics are returned as two extra device
    arrays in `infos` = {'fin_r': f32 [N], 'fin_l': i32 [N]} (non-zero length where an episode ended)
    so that no host synchronisation happens inside the rollout loop."""
    device_resident = True

    def __init__(self, kind, num_envs, seed=0, device='cuda', **kw):
        super().__init__(kind, num_envs, seed, **kw)
        import torch
        from ... import _lib
        _lib.require_gpu()
        self._lib, self._torch = _lib, torch
        self.device = torch.device(device)
        N = num_envs
        self.ep = torch.zeros(N, dtype=torch.int32, device=self.device)     # bit pattern of u32
        self.st = torch.zeros(N, dtype=torch.int32, device=self.device)
        self.ep_ret = torch.zeros(N, dtype=torch.float32, device=self.device)
        self.ob_torch_dtype = torch.uint8 if self.ob_u8 else torch.float32
        self._actions = None

    def _new_obs(self):
        return self._torch.empty((self.num_envs,) + self.observation_space.shape, dtype=self.ob_torch_dtype,
                                 device=self.device)

    def reset(self, out=None):
        L = self._lib
        self.ep.zero_()
        self.st.zero_()
        self.ep_ret.zero_()
        obs = out if out is not None else self._new_obs()
        L.check(L.load().mrl_synth_env_obs(self.seed_value, self.ob_elems, self.ob_u8, self.num_envs, L.ptr(self.ep),
                                           L.ptr(self.st), L.ptr(obs), L.stream_ptr()), 'mrl_synth_env_obs')
        return obs

    def step_async(self, actions):
        self._actions = actions

    def step_wait(self, out=None):
        return self.step_into(self._actions, obs_out=out)

    def step_into(self, actions, obs_out=None, rew_out=None, done_out=None):
        """Zero-copy step: results are written into caller-provided device buffers (e.g. slot t+1 of
        the rollout buffer)."""
        L, torch = self._lib, self._torch
        N = self.num_envs
        obs = obs_out if obs_out is not None else self._new_obs()
        rew = rew_out if rew_out is not None else torch.empty(N, dtype=torch.float32, device=self.device)
        done = done_out if done_out is not None else torch.empty(N, dtype=torch.uint8, device=self.device)
        fin_r = torch.empty(N, dtype=torch.float32, device=self.device)
        fin_l = torch.empty(N, dtype=torch.int32, device=self.device)
        a = None
        if self.discrete:
            a = actions
            assert a.is_cuda and a.dtype == torch.int32 and a.numel() == N
        L.check(L.load().mrl_synth_env_step(self.seed_value, self.ob_elems, self.ob_u8, self.discrete,
                                            self.reward_kind, self.lmin, self.lspan, N, L.ptr(self.ep), L.ptr(self.st),
                                            L.ptr(self.ep_ret), L.ptr(a), L.ptr(obs), L.ptr(rew), L.ptr(done),
                                            L.ptr(fin_r), L.ptr(fin_l), L.stream_ptr()), 'mrl_synth_env_step')
        return obs, rew, done, {'fin_r': fin_r, 'fin_l': fin_l}
