"""VecFrameStack for device-resident environments (reference: common/vec_env/vec_frame_stack.py:6-30).

The stacked observation buffer lives in HBM and is updated in place by one HIP kernel per step
(`mrl_framestack_step`): roll, zero finished envs, append the newest frame.  Like the reference, `step_wait`
returns the INTERNAL buffer (aliased across steps; the Runner snapshots it into the rollout).  The
reference rolls the stack axis by one ELEMENT (np.roll shift=-1), which is a frame shift for the
single-channel Atari frames it is used with; that behaviour is kept bit for bit.
"""
import numpy as np

from ._lazy import torch     # resolved on first use: env worker processes import this package without paying for torch

from ... import _lib
from ..spaces import Box
from .vec_env import VecEnvWrapper


class VecFrameStack(VecEnvWrapper):
    def __init__(self, venv, nstack):
        if not getattr(venv, 'device_resident', False):
            raise ValueError('VecFrameStack here wraps device-resident VecEnvs (observations are device tensors)')
        self.venv = venv
        self.nstack = int(nstack)
        wos = venv.observation_space
        low = np.repeat(np.asarray(wos.low), self.nstack, axis=-1)
        high = np.repeat(np.asarray(wos.high), self.nstack, axis=-1)
        self._c = int(wos.shape[-1])
        self._pix = int(np.prod(wos.shape[:-1], dtype=np.int64))
        tdtype = {np.dtype(np.uint8): torch.uint8, np.dtype(np.float32): torch.float32}[np.dtype(wos.dtype)]
        self._esize = np.dtype(wos.dtype).itemsize
        self.stackedobs = torch.zeros((venv.num_envs,) + low.shape, dtype=tdtype, device='cuda')
        VecEnvWrapper.__init__(self, venv, observation_space=Box(low=low, high=high, dtype=wos.dtype))

    def _kernel(self, obs, news, reset):
        obs = obs.contiguous()
        n = None if news is None else (news.view(torch.uint8) if news.dtype == torch.bool else news).contiguous()
        _lib.check(_lib.load().mrl_framestack_step(_lib.ptr(self.stackedobs), _lib.ptr(obs), _lib.ptr(n), self.num_envs,
                                                   self._pix, self._c * self.nstack, self._c, self._esize,
                                                   1 if reset else 0, _lib.stream_ptr()), 'mrl_framestack_step')

    def step_wait(self):
        obs, rews, news, infos = self.venv.step_wait()
        self._kernel(obs, news, False)
        return self.stackedobs, rews, news, infos

    def reset(self):
        obs = self.venv.reset()
        self._kernel(obs, None, True)
        return self.stackedobs
