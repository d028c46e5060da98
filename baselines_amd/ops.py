"""Thin Python wrappers over the C ABI (include/mrl.h): torch tensors are only the owners of the
device memory and of the stream; every op below is a hand-written HIP kernel in libmrl.so."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import c_void_p, check, ptr, stream_ptr


def _dev(t):
    assert isinstance(t, torch.Tensor) and t.is_cuda, 'device tensor required (no CPU path)'
    return t if t.is_contiguous() else t.contiguous()


def gae(rewards, values, dones, last_values, last_dones, gamma, lam, want_advs=False, out=None):
    """ppo2/runner.py:52-65 on device.  rewards/values f32 [T,N], dones u8/bool [T,N],
    last_values f32 [N], last_dones u8/bool [N] -> returns f32 [T,N] (and advs).  `out`: write the returns into this
    tensor (a rollout buffer whose address launch graphs have captured) instead of a new one."""
    _lib.require_gpu()
    rewards, values = _dev(rewards), _dev(values)
    T, N = rewards.shape
    dones = _dev(dones.view(torch.uint8) if dones.dtype == torch.bool else dones)
    last_dones = _dev(last_dones.view(torch.uint8) if last_dones.dtype == torch.bool else last_dones)
    last_values = _dev(last_values)
    assert rewards.dtype == values.dtype == last_values.dtype == torch.float32
    assert dones.dtype == torch.uint8 and dones.shape == (T, N) and values.shape == (T, N)
    if out is not None:
        assert out.shape == rewards.shape and out.dtype == torch.float32 and out.is_contiguous() and out.device == rewards.device
    ret = out if out is not None else torch.empty_like(rewards)
    adv = torch.empty_like(rewards) if want_advs else None
    check(_lib.load().mrl_gae(ptr(rewards), ptr(values), ptr(dones), ptr(last_values), ptr(last_dones),
                              float(gamma), float(lam), ptr(adv), ptr(ret), T, N, stream_ptr()), 'mrl_gae')
    return (ret, adv) if want_advs else ret


def gather_rows(src, idx, T, N):
    """`arr[mbinds]` of the reference (ppo2.py:162-164) on a time-major device field.
    src [T*N, ...] (time-major rows), idx int64 env-major flat indices [B] -> [B, ...]."""
    _lib.require_gpu()
    src, idx = _dev(src), _dev(idx)
    assert idx.dtype == torch.int64
    assert src.dim() >= 2 and src.shape[0] == T and src.shape[1] == N, 'expected a [T, N, ...] field'
    trailing = tuple(src.shape[2:])
    row_bytes = int(np.prod(trailing, dtype=np.int64)) * src.element_size()
    rows = src
    out = torch.empty((idx.numel(),) + trailing, dtype=src.dtype, device=src.device)
    check(_lib.load().mrl_gather_rows(ptr(rows), ptr(idx), ptr(out), idx.numel(), T, N, row_bytes, stream_ptr()),
          'mrl_gather_rows')
    return out


def sf01(src):
    """runner.py:69-74 as a device op: [T,N,...] -> env-major [N*T,...]."""
    _lib.require_gpu()
    src = _dev(src)
    T, N = src.shape[:2]
    row_bytes = int(np.prod(src.shape[2:], dtype=np.int64)) * src.element_size()
    out = torch.empty((T * N,) + tuple(src.shape[2:]), dtype=src.dtype, device=src.device)
    check(_lib.load().mrl_sf01(ptr(src), ptr(out), T, N, row_bytes, stream_ptr()), 'mrl_sf01')
    return out


def adam_clip_step(params, grads, m, v, alpha, beta1, beta2, eps, max_grad_norm, total_weight, scratch,
                   gnorm_out=None):
    """model.py:105-114 (+ mpi_adam_optimizer.py:40 division) as two launches."""
    _lib.require_gpu()
    P = params.numel()
    mgn = -1.0 if max_grad_norm is None else float(max_grad_norm)
    check(_lib.load().mrl_adam_clip_step(ptr(params), ptr(grads), ptr(m), ptr(v), P, float(alpha), float(beta1),
                                         float(beta2), float(eps), mgn, float(total_weight), ptr(gnorm_out),
                                         ptr(scratch), stream_ptr()), 'mrl_adam_clip_step')


def clip_accumulate(grads, acc, max_grad_norm, total_weight, first, scratch):
    """acc (+)= clip_by_global_norm(grads / total_weight): one MicrobatchedModel slice (microbatched_model.py:57-64)."""
    _lib.require_gpu()
    mgn = -1.0 if max_grad_norm is None else float(max_grad_norm)
    check(_lib.load().mrl_clip_accumulate(ptr(grads), ptr(acc), grads.numel(), mgn, float(total_weight), int(bool(first)),
                                          ptr(scratch), stream_ptr()), 'mrl_clip_accumulate')


class DeviceModel(object):
    """Handle on an `mrl_model` layout object + the device buffers it works on."""

    def __init__(self, *, network, ob_shape, ob_dtype, pd_kind, nact, value_copy=False, num_layers=2,
                 num_hidden=64, activation='tanh', nlstm=128, layer_norm=False, convs=None, fc_hidden=512, pad='VALID',
                 nvec=None, chunk=None, device=None):
        _lib.require_gpu()
        lib = _lib.load()
        d = _lib.ModelDesc()
        d.network = {'mlp': _lib.NET_MLP, 'cnn': _lib.NET_NATURE_CNN, 'lstm': _lib.NET_LSTM,
                     'cnn_lstm': _lib.NET_CNN_LSTM}[network]
        ob_shape = tuple(int(s) for s in ob_shape)
        if d.network in (_lib.NET_MLP, _lib.NET_LSTM):
            ob_shape = (int(np.prod(ob_shape)),)
        d.nlstm = int(nlstm)
        d.layer_norm = 1 if layer_norm else 0
        # conv stack other than nature_cnn's (cnn_small; models.py:117-129) / cnn(pad='SAME')
        if convs is not None and d.network in (_lib.NET_NATURE_CNN, _lib.NET_CNN_LSTM):
            d.nconv, d.fc_hidden = len(convs), int(fc_hidden)
            for i, c in enumerate(convs):
                for k in range(3):
                    d.convs[i][k] = int(c[k])
        d.conv_pad = {'VALID': 0, 'SAME': 1}[pad]
        d.ob_ndim = len(ob_shape)
        for i, s in enumerate(ob_shape):
            d.ob_shape[i] = s
        ob_dtype = np.dtype(ob_dtype)
        if ob_dtype == np.int8:   # common/input.py:27-29
            ob_dtype = np.dtype(np.uint8)
        d.ob_dtype = _lib.OB_U8 if ob_dtype == np.uint8 else _lib.OB_F32
        d.num_layers, d.num_hidden = int(num_layers), int(num_hidden)
        d.activation = {'tanh': _lib.ACT_TANH, 'relu': _lib.ACT_RELU}[activation]
        d.value_copy = 1 if value_copy else 0
        d.pd_kind = {'categorical': _lib.PD_CATEGORICAL, 'gaussian': _lib.PD_DIAG_GAUSSIAN,
                     'multicategorical': _lib.PD_MULTICATEGORICAL, 'bernoulli': _lib.PD_BERNOULLI}[pd_kind]
        self.nvec = tuple(int(v) for v in nvec) if nvec is not None else None
        if pd_kind == 'multicategorical':
            assert self.nvec and sum(self.nvec) == int(nact), 'multicategorical: nact is the width of the flat logits = sum(nvec)'
            d.nsub = len(self.nvec)
            for i, v in enumerate(self.nvec):
                d.nvec[i] = v
        d.nact = int(nact)
        self.desc = d
        h = c_void_p()
        check(lib.mrl_model_create(ctypes.byref(d), ctypes.byref(h)), 'mrl_model_create')
        self.handle = h
        self.lib = lib
        self.network, self.pd_kind, self.nact = network, pd_kind, int(nact)
        self.ob_shape, self.ob_dtype = ob_shape, ob_dtype
        self.torch_ob_dtype = torch.uint8 if d.ob_dtype == _lib.OB_U8 else torch.float32
        self.P = int(lib.mrl_model_num_params(h))
        self.state_size = int(lib.mrl_model_state_size(h))          # 2*nlstm for recurrent networks, else 0
        self.recurrent = self.state_size > 0
        self.tensors = []
        name = ctypes.create_string_buffer(128)
        for i in range(lib.mrl_model_num_tensors(h)):
            nd, shp, off, sc = ctypes.c_int(), (ctypes.c_int * 4)(), ctypes.c_long(), ctypes.c_double()
            check(lib.mrl_model_tensor_info(h, i, name, 128, ctypes.byref(nd), ctypes.byref(shp), ctypes.byref(off),
                                            ctypes.byref(sc)), 'mrl_model_tensor_info')
            shape = tuple(shp[k] for k in range(nd.value))
            self.tensors.append(dict(name=name.value.decode(), shape=shape, offset=off.value,
                                     size=int(np.prod(shape)), init_scale=(None if sc.value < 0 else sc.value),
                                     init_const=(1.0 if sc.value == -2.0 else 0.0)))     # -2: ones (layer-norm gamma)
        self.device = torch.device(device or 'cuda')
        self.chunk = None
        self.workspace = None
        if chunk:
            self.set_chunk(chunk)

    @property
    def action_dtype(self):
        """device dtype of the actions: int32 for the discrete distributions (Categorical / MultiCategorical / Bernoulli)"""
        return torch.float32 if self.pd_kind == 'gaussian' else torch.int32

    def action_shape(self, n):
        """[n] (Categorical), [n, len(nvec)] (MultiCategorical), [n, nact] (Bernoulli bits, DiagGaussian)"""
        if self.pd_kind == 'categorical':
            return (n,)
        if self.pd_kind == 'multicategorical':
            return (n, len(self.nvec))
        return (n, self.nact)

    def set_chunk(self, chunk):
        chunk = int(chunk)
        nbytes = int(self.lib.mrl_model_workspace_bytes(self.handle, chunk))
        # zero-initialised once, as include/mrl.h asks (the carve holds a page of zeros that gathers read for
        # out-of-map taps and that no kernel ever writes)
        self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self.chunk = chunk

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.mrl_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _obs(self, obs):
        obs = _dev(obs)
        assert obs.dtype == self.torch_ob_dtype, (obs.dtype, self.torch_ob_dtype)
        return obs

    def act(self, params, obs, noise=None, want_actions=True, want_pdparam=False):
        """policies.py:77-113.  obs [n, ...] device; noise f32 [n, nact]."""
        obs = self._obs(obs)
        n = obs.shape[0]
        dev = obs.device
        values = torch.empty(n, dtype=torch.float32, device=dev)
        actions = neglogp = pdparam = None
        if want_actions:
            assert noise is not None and noise.shape == (n, self.nact) and noise.dtype == torch.float32
            noise = _dev(noise)
            actions = torch.empty(self.action_shape(n), dtype=self.action_dtype, device=dev)
            neglogp = torch.empty(n, dtype=torch.float32, device=dev)
        if want_pdparam:
            pdparam = torch.empty((n, self.nact), dtype=torch.float32, device=dev)
        self.act_into(params, obs, noise, actions, values, neglogp, pdparam)
        return actions, values, neglogp, pdparam

    def act_rnn_into(self, params, obs, noise, state_in, mask, state_out, actions, values, neglogp, pdparam=None):
        """recurrent policies: one step of n sequences (state_in may alias state_out)"""
        n = obs.shape[0]
        check(self.lib.mrl_model_act_rnn(self.handle, ptr(params), ptr(obs), ptr(noise), n, ptr(state_in), ptr(mask),
                                         ptr(state_out), ptr(actions), ptr(values), ptr(neglogp), ptr(pdparam),
                                         ptr(self.workspace), self.workspace.numel(), self.chunk, stream_ptr()),
              'mrl_model_act_rnn')

    def grad_rnn(self, params, obs, actions, returns, values, neglogpacs, masks, states, nseq, idx, B, T, N, cliprange,
                 ent_coef, vf_coef, grads_out, stats_out):
        """recurrent policies: gradient over a minibatch of nseq whole trajectories (back-propagation through time)"""
        check(self.lib.mrl_model_grad_rnn(self.handle, ptr(params), ptr(obs), ptr(actions), ptr(returns), ptr(values),
                                          ptr(neglogpacs), ptr(masks), ptr(states), int(nseq), ptr(idx), int(B), int(T),
                                          int(N), float(cliprange), float(ent_coef), float(vf_coef), ptr(grads_out),
                                          ptr(stats_out), ptr(self.workspace), self.workspace.numel(), self.chunk,
                                          stream_ptr()), 'mrl_model_grad_rnn')

    def act_into(self, params, obs, noise, actions, values, neglogp, pdparam=None):
        n = obs.shape[0]
        check(self.lib.mrl_model_act(self.handle, ptr(params), ptr(obs), ptr(noise), n, ptr(actions), ptr(values),
                                     ptr(neglogp), ptr(pdparam), ptr(self.workspace), self.workspace.numel(),
                                     self.chunk, stream_ptr()), 'mrl_model_act')

    def grad(self, params, obs, actions, returns, values, neglogpacs, idx, B, T, N, cliprange, ent_coef, vf_coef,
             grads_out, stats_out):
        """model.py:133-158 minus the optimizer: flat dloss/dparams + the 5 stats."""
        check(self.lib.mrl_model_grad(self.handle, ptr(params), ptr(obs), ptr(actions), ptr(returns), ptr(values),
                                      ptr(neglogpacs), ptr(idx), int(B), int(T), int(N), float(cliprange),
                                      float(ent_coef), float(vf_coef), ptr(grads_out), ptr(stats_out),
                                      ptr(self.workspace), self.workspace.numel(), self.chunk, stream_ptr()),
              'mrl_model_grad')

    def attach_comm(self, native_comm, rank_weight=1.0):
        """data parallel: gradients returned by grad/grad_micro/train_step become rank-weighted sums over ranks (RCCL
        all-reduce issued from inside the backward pass); None detaches."""
        check(self.lib.mrl_model_attach_comm(self.handle, native_comm, float(rank_weight)), 'mrl_model_attach_comm')

    def grad_micro(self, params, obs, actions, returns, values, neglogpacs, idx, B, mb0, mbn, T, N, cliprange, ent_coef,
                   vf_coef, grads_out, stats_out):
        """one MicrobatchedModel slice: advantage statistics over all B samples, loss/gradient over [mb0, mb0+mbn)."""
        check(self.lib.mrl_model_grad_micro(self.handle, ptr(params), ptr(obs), ptr(actions), ptr(returns), ptr(values),
                                            ptr(neglogpacs), ptr(idx), int(B), int(mb0), int(mbn), int(T), int(N),
                                            float(cliprange), float(ent_coef), float(vf_coef), ptr(grads_out),
                                            ptr(stats_out), ptr(self.workspace), self.workspace.numel(), self.chunk,
                                            stream_ptr()), 'mrl_model_grad_micro')
