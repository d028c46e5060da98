"""VecNormalize for device-resident environments (reference: common/vec_env/vec_normalize.py:4-47 with
common/running_mean_std.py:5-33, use_tf=False).

Running statistics are float64 device buffers (`count` starts at 1e-4, `var` at 1, like the reference);
per step two HIP kernels normalise + clip the observations (`mrl_vecnorm_ob`: batch moments in float32 row
after row == np.mean/np.var(axis=0), merge and normalisation in float64) and one does the return-based
reward scaling (`mrl_vecnorm_rew`).  Observations come back as float32 device tensors (the reference
returns float64 arrays that the Runner casts to the float32 observation dtype anyway).
"""
import torch

from ... import _lib
from .vec_env import VecEnvWrapper


class _DeviceRunningMeanStd(object):
    def __init__(self, shape, device, epsilon=1e-4):
        self.mean = torch.zeros(shape, dtype=torch.float64, device=device)
        self.var = torch.ones(shape, dtype=torch.float64, device=device)
        self.count = epsilon


class _DeviceReturnStats(object):
    """RunningMeanStd(shape=()) of the discounted return: stats = [mean, var] (float64, device)"""

    def __init__(self, device, epsilon=1e-4):
        self.stats = torch.tensor([0.0, 1.0], dtype=torch.float64, device=device)
        self.count = epsilon

    mean = property(lambda self: float(self.stats[0].item()))
    var = property(lambda self: float(self.stats[1].item()))


class VecNormalize(VecEnvWrapper):
    def __init__(self, venv, ob=True, ret=True, clipob=10., cliprew=10., gamma=0.99, epsilon=1e-8, use_tf=False):
        if not getattr(venv, 'device_resident', False):
            raise ValueError('VecNormalize here wraps device-resident VecEnvs (observations are device tensors)')
        VecEnvWrapper.__init__(self, venv)
        dev = 'cuda'
        self.ob_rms = _DeviceRunningMeanStd(tuple(self.observation_space.shape), dev) if ob else None
        self.ret_rms = _DeviceReturnStats(dev) if ret else None
        self.clipob, self.cliprew, self.gamma, self.epsilon = float(clipob), float(cliprew), float(gamma), float(epsilon)
        self.ret = torch.zeros(self.num_envs, dtype=torch.float64, device=dev)

    def _obfilt(self, obs, want_f64=False):
        if self.ob_rms is None:
            return obs
        x = obs.reshape(self.num_envs, -1).to(torch.float32).contiguous()
        out = torch.empty_like(x)
        out64 = torch.empty(x.shape, dtype=torch.float64, device=x.device) if want_f64 else None
        rms = self.ob_rms
        _lib.check(_lib.load().mrl_vecnorm_ob(_lib.ptr(x), x.shape[0], x.shape[1], _lib.ptr(rms.mean.view(-1)),
                                              _lib.ptr(rms.var.view(-1)), float(rms.count), self.epsilon, self.clipob,
                                              _lib.ptr(out), _lib.ptr(out64), _lib.stream_ptr()), 'mrl_vecnorm_ob')
        rms.count = rms.count + x.shape[0]
        self._last_obs64 = out64
        return out.reshape(obs.shape)

    def step_wait(self, want_f64=False):
        obs, rews, news, infos = self.venv.step_wait()
        obs = self._obfilt(obs, want_f64)
        n8 = (news.view(torch.uint8) if news.dtype == torch.bool else news).contiguous()
        r32 = rews.to(torch.float32).contiguous()
        if self.ret_rms is not None:
            out = torch.empty_like(r32)
            out64 = torch.empty(r32.shape, dtype=torch.float64, device=r32.device) if want_f64 else None
            _lib.check(_lib.load().mrl_vecnorm_rew(_lib.ptr(r32), _lib.ptr(n8), self.num_envs, _lib.ptr(self.ret),
                                                   _lib.ptr(self.ret_rms.stats), float(self.ret_rms.count), self.gamma,
                                                   self.epsilon, self.cliprew, _lib.ptr(out), _lib.ptr(out64),
                                                   _lib.stream_ptr()), 'mrl_vecnorm_rew')
            self.ret_rms.count = self.ret_rms.count + self.num_envs
            self._last_rews64 = out64
            rews = out
        else:
            self.ret = self.ret * self.gamma + r32.to(torch.float64)
            self.ret[news.bool()] = 0.
        return obs, rews, news, infos

    def reset(self, want_f64=False):
        self.ret.zero_()
        return self._obfilt(self.venv.reset(), want_f64)
