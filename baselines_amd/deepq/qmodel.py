"""Device Q-network pair (online + target) with the three functions `deepq.build_train` returns
(deepq/build_graph.py:317-449):

    act(ob, stochastic=True, update_eps=-1)            eps-greedy actions (:146-199)
    act(ob, reset, update_param_noise_threshold, update_param_noise_scale, ...)   with param_noise=True (:202-315)
    train(obs_t, action, reward, obs_tp1, done, weight) -> td_errors   one Adam step on the weighted Huber TD loss (:380-444)
    update_target()                                    target <- online (:423-428)
    debug['q_values'](obs)

Parameters, Adam slots and the target copy are flat fp32 device buffers in TF variable-creation order; the work is done
by libmrl's C ABI (mrl_qnet_values / mrl_qnet_act / mrl_qnet_td_grad / mrl_qnet_adam_step, include/mrl.h)."""
import ctypes
import os

import numpy as np
import torch

from .. import _lib
from .._lib import c_void_p, check, ptr, stream_ptr
from ..ppo2.model import ortho_init


def default_param_noise_filter(name):
    """build_graph.py:131-143 on variable NAMES (every variable of the Q-network is trainable): only the
    tf.contrib.layers.fully_connected layers -- the action-value / state-value heads -- are perturbed"""
    return 'fully_connected' in name


class QModel(object):
    def __init__(self, q_func, observation_space, num_actions, lr=5e-4, gamma=1.0, grad_norm_clipping=10, double_q=True,
                 max_batch=32, adam_epsilon=1e-8, device=None, param_noise=False, param_noise_filter_func=None):
        _lib.require_gpu()
        lib = self.lib = _lib.load()
        self.device = torch.device(device or ('cuda:%d' % torch.cuda.current_device()))
        self.num_actions, self.gamma, self.double_q = int(num_actions), float(gamma), bool(double_q)
        self.grad_norm_clipping = -1.0 if grad_norm_clipping is None else float(grad_norm_clipping)
        self.lr = float(lr)
        d = _lib.QNetDesc()
        net = q_func.network
        if net.kind == 'cnn' and (net.kw.get('convs') or net.kw.get('pad')):
            raise NotImplementedError('Q-networks take nature_cnn or conv_only(convs=...) bodies; cnn_small / cnn(pad=...) are built for ppo2 only')
        d.network = {'mlp': _lib.NET_MLP, 'cnn': _lib.NET_NATURE_CNN, 'conv_only': _lib.NET_CONV_ONLY}[net.kind]
        # deepq/utils.py ObservationInput -> common/input.py:43-63: Discrete observations are fed one-hot encoded
        self.ob_onehot = int(observation_space.n) if type(observation_space).__name__ == 'Discrete' else 0
        if self.ob_onehot:
            ob_shape, ob_dtype = (self.ob_onehot,), np.dtype(np.float32)
        else:
            ob_shape = tuple(int(s) for s in observation_space.shape)
            ob_dtype = np.dtype(observation_space.dtype)
        if net.kind == 'mlp':
            ob_shape = (int(np.prod(ob_shape)),)
        d.ob_ndim = len(ob_shape)
        for i, s in enumerate(ob_shape):
            d.ob_shape[i] = s
        d.ob_dtype = _lib.OB_U8 if ob_dtype in (np.dtype(np.uint8), np.dtype(np.int8)) else _lib.OB_F32
        d.num_layers, d.num_hidden = int(net.kw.get('num_layers', 2)), int(net.kw.get('num_hidden', 64))
        d.activation = {'tanh': _lib.ACT_TANH, 'relu': _lib.ACT_RELU}[net.kw.get('activation', 'tanh')]
        convs = net.kw.get('convs', ())
        d.nconv = len(convs)
        for i, c in enumerate(convs):
            for k in range(3):
                d.convs[i][k] = int(c[k])
        d.nhidden = len(q_func.hiddens)
        for i, h in enumerate(q_func.hiddens):
            d.hiddens[i] = int(h)
        d.dueling = 1 if q_func.dueling else 0
        d.nact = self.num_actions
        d.layer_norm = 1 if getattr(q_func, 'layer_norm', False) else 0
        # network = mlp(layer_norm=True): the body's own LayerNorm variables (common/models.py:97-98), independent of the heads'
        d.body_layer_norm = 1 if (net.kind == 'mlp' and net.kw.get('layer_norm', False)) else 0
        self.ob_shape, self.torch_ob_dtype = ob_shape, (torch.uint8 if d.ob_dtype == _lib.OB_U8 else torch.float32)
        h = c_void_p()
        check(lib.mrl_qnet_create(ctypes.byref(d), ctypes.byref(h)), 'mrl_qnet_create')
        self.handle = h
        self.P = int(lib.mrl_qnet_num_params(h))
        self.tensors = []
        name = ctypes.create_string_buffer(160)
        for i in range(lib.mrl_qnet_num_tensors(h)):
            nd, shp, off, kind, sc = ctypes.c_int(), (ctypes.c_int * 4)(), ctypes.c_long(), ctypes.c_int(), ctypes.c_double()
            check(lib.mrl_qnet_tensor_info(h, i, name, 160, ctypes.byref(nd), ctypes.byref(shp), ctypes.byref(off),
                                           ctypes.byref(kind), ctypes.byref(sc)), 'mrl_qnet_tensor_info')
            shape = tuple(shp[k] for k in range(nd.value))
            self.tensors.append(dict(name=name.value.decode(), shape=shape, offset=off.value, size=int(np.prod(shape)),
                                     init_kind=kind.value, init_scale=sc.value))
        self.params = torch.from_numpy(self._initial_parameters()).to(self.device)
        self.target = self.params.clone()               # deepq.py:236-237 U.initialize(); update_target()
        self.grads = torch.zeros_like(self.params)
        self.adam_m = torch.zeros_like(self.params)
        self.adam_v = torch.zeros_like(self.params)
        self.beta1, self.beta2, self.epsilon = np.float32(0.9), np.float32(0.999), np.float32(adam_epsilon)
        self.beta1_power, self.beta2_power = np.float32(0.9), np.float32(0.999)
        self.eps = 0.0                                   # the graph's `eps` variable (build_graph.py:177)
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(int(torch.initial_seed()) & 0x7fffffff)
        self.max_batch = 0
        self._set_batch(max_batch)
        self._td = torch.empty(self.max_batch, dtype=torch.float32, device=self.device)
        self._loss = torch.empty(1, dtype=torch.float32, device=self.device)
        self._graphs = {}                                # batch size -> captured optimizer step (train_dev)
        self._ones = {}
        self.param_noise = bool(param_noise)
        if self.param_noise:
            # build_graph.py:241-262: two more copies of q_func -- the one that acts (re-perturbed on `reset`) and the one
            # the scale adaptation probes; noise goes to the variables the filter accepts (default_param_noise_filter,
            # :131-143: the tf.contrib 'fully_connected' layers, i.e. the Q heads)
            flt = param_noise_filter_func or default_param_noise_filter
            mask = np.zeros(self.P, np.float32)
            for t in self.tensors:
                if flt(t['name']):
                    mask[t['offset']:t['offset'] + t['size']] = 1.0
            self._noise_mask = torch.from_numpy(mask).to(self.device)
            # TF initialises the copies with their own random draws; they are overwritten by the first reset=True call
            # (deepq.py:223 starts with reset=True), so they start as plain copies here
            self.perturbed = self.params.clone()
            self.adaptive = self.params.clone()
            self.param_noise_scale = np.float32(0.01)        # build_graph.py:238
            self.param_noise_threshold = np.float32(0.05)    # build_graph.py:239
            self._kl = torch.zeros(1, dtype=torch.float32, device=self.device)

    def _initial_parameters(self):
        """common.models networks: orthogonal init from the global NumPy stream like the reference (a2c/utils.py:20-35);
        tf.contrib layers (conv_only, the Q heads): xavier-uniform weights U(+-sqrt(6 / (fan_in + fan_out))), zero biases.
        TF draws those from its own generator, which cannot be reproduced: the NumPy global stream stands in."""
        flat = np.zeros(self.P, np.float32)
        for t in self.tensors:
            sl = slice(t['offset'], t['offset'] + t['size'])
            if t['init_kind'] == 1:
                flat[sl] = ortho_init(t['shape'], t['init_scale']).reshape(-1)
            elif t['init_kind'] == 2:
                shape = t['shape']
                receptive = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
                fan_in, fan_out = shape[-2] * receptive, shape[-1] * receptive
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                flat[sl] = np.random.uniform(-lim, lim, size=t['size']).astype(np.float32)
            elif t['init_kind'] == 3:
                flat[sl] = 1.0                                    # LayerNorm gamma
        return flat

    def _set_batch(self, n):
        if n > self.max_batch:
            self.max_batch = int(n)
            # room for the learner step's merged online pass (2 B rows) plus a second workspace for the target network's pass on
            # its side stream (csrc/qnet.hip.h, mrl_qnet_td_grad)
            nbytes = (int(self.lib.mrl_qnet_workspace_bytes(self.handle, 2 * self.max_batch))
                      + int(self.lib.mrl_qnet_workspace_bytes(self.handle, self.max_batch)) + 1024)
            self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.mrl_qnet_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _obs(self, ob):
        if self.ob_onehot:
            t = ob.to(self.device) if isinstance(ob, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(ob))).to(self.device)
            if t.dim() < 2 or t.shape[-1] != self.ob_onehot or not t.is_floating_point():
                t = torch.nn.functional.one_hot(t.reshape(-1).long(), self.ob_onehot)
            return t.to(torch.float32).reshape(-1, self.ob_onehot).contiguous()
        if isinstance(ob, torch.Tensor):
            t = ob.to(self.device)
        else:
            a = np.asarray(ob)
            if a.dtype == np.int8:
                a = a.view(np.uint8)
            t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        if t.dtype != self.torch_ob_dtype:
            t = t.to(self.torch_ob_dtype)
        return t.reshape((-1,) + tuple(self.ob_shape)).contiguous()

    # ---- the reference's three functions ----------------------------------------------------------------------------
    def q_values(self, ob, target=False):
        obs = self._obs(ob)
        n = obs.shape[0]
        self._set_batch(min(n, 4096))
        out = torch.empty((n, self.num_actions), dtype=torch.float32, device=self.device)
        check(self.lib.mrl_qnet_values(self.handle, ptr(self.target if target else self.params), ptr(obs), n, ptr(out),
                                       ptr(self.workspace), self.workspace.numel(), self.max_batch, stream_ptr()),
              'mrl_qnet_values')
        return out.cpu().numpy()

    def _perturb(self, dst):
        """perturb_vars (build_graph.py:264-277): dst <- params + N(0, scale) on the filtered variables, a copy elsewhere"""
        noise = torch.randn(self.P, generator=self._gen, device=self.device, dtype=torch.float32)
        torch.addcmul(self.params, noise, self._noise_mask, value=float(self.param_noise_scale), out=dst)

    def _values_with(self, params, obs, n):
        out = torch.empty((n, self.num_actions), dtype=torch.float32, device=self.device)
        check(self.lib.mrl_qnet_values(self.handle, ptr(params), ptr(obs), n, ptr(out), ptr(self.workspace),
                                       self.workspace.numel(), self.max_batch, stream_ptr()), 'mrl_qnet_values')
        return out

    def _act_param_noise(self, ob, reset, update_param_noise_threshold, update_param_noise_scale, stochastic, update_eps):
        """build_graph.py:202-315.  One session.run of the reference: the outputs (greedy actions of the PERTURBED network,
        eps-random otherwise) are computed from the variables as they were before the run's update ops; the updates are
        eps.assign, the re-perturbation on `reset`, the scale adaptation and the threshold assign (:300-307)."""
        obs = self._obs(ob)
        n = obs.shape[0]
        self._set_batch(n)
        u = rnd = None
        if stochastic:
            u = torch.rand(n, generator=self._gen, device=self.device, dtype=torch.float32)
            rnd = torch.randint(0, self.num_actions, (n,), generator=self._gen, device=self.device, dtype=torch.int32)
        a = torch.empty(n, dtype=torch.int32, device=self.device)
        check(self.lib.mrl_qnet_act(self.handle, ptr(self.perturbed), ptr(obs), n, float(self.eps), ptr(u), ptr(rnd), ptr(a),
                                    None, ptr(self.workspace), self.workspace.numel(), self.max_batch, stream_ptr()),
              'mrl_qnet_act')
        if update_eps >= 0:
            self.eps = float(update_eps)
        if reset:
            self._perturb(self.perturbed)
        if update_param_noise_scale:
            # update_scale (:279-291): perturb the adaptive copy, measure the action-space distance to the unperturbed
            # policy on this observation batch, grow the scale by 1 % when it is below the threshold, shrink it otherwise
            self._perturb(self.adaptive)
            qa, qb = self._values_with(self.params, obs, n), self._values_with(self.adaptive, obs, n)
            check(self.lib.mrl_qnet_policy_kl(ptr(qa), ptr(qb), n, self.num_actions, ptr(self._kl), stream_ptr()),
                  'mrl_qnet_policy_kl')
            self.last_kl = float(self._kl.item())
            if self.last_kl < float(self.param_noise_threshold):
                self.param_noise_scale = np.float32(self.param_noise_scale * np.float32(1.01))
            else:
                self.param_noise_scale = np.float32(self.param_noise_scale / np.float32(1.01))
        # :305: tf.cond(ph >= 0, assign ph, keep) -- a Python False arrives as 0.0 and IS assigned, like the reference
        if float(update_param_noise_threshold) >= 0:
            self.param_noise_threshold = np.float32(update_param_noise_threshold)
        return a.cpu().numpy().astype(np.int64)

    def act(self, ob, *args, **kwargs):
        """act(ob, stochastic=True, update_eps=-1) -- build_graph.py:146-199; returns int64 actions [n] like tf.argmax.
        On a model built with param_noise=True the signature is the one of build_act_with_param_noise (:202-315):
        act(ob, reset=False, update_param_noise_threshold=False, update_param_noise_scale=False, stochastic=True, update_eps=-1)"""
        if self.param_noise:
            names = ('reset', 'update_param_noise_threshold', 'update_param_noise_scale', 'stochastic', 'update_eps')
            kw = dict(reset=False, update_param_noise_threshold=False, update_param_noise_scale=False, stochastic=True, update_eps=-1)
        else:
            names, kw = ('stochastic', 'update_eps'), dict(stochastic=True, update_eps=-1)
        if len(args) > len(names):
            raise TypeError('act() takes at most %d arguments after ob' % len(names))
        given = dict(zip(names, args))
        for k, v in kwargs.items():
            if k not in kw or k in given:
                raise TypeError('act() got an unexpected or repeated argument %r' % k)
            given[k] = v
        kw.update(given)
        if self.param_noise:
            return self._act_param_noise(ob, **kw)
        stochastic, update_eps = kw['stochastic'], kw['update_eps']
        obs = self._obs(ob)
        n = obs.shape[0]
        self._set_batch(n)
        u = rnd = None
        if stochastic:
            # the chose_random draw uses the eps of the PREVIOUS call: eps.assign is an update op of the same run whose
            # read is not ordered after it (build_graph.py:191-196)
            u = torch.rand(n, generator=self._gen, device=self.device, dtype=torch.float32)
            rnd = torch.randint(0, self.num_actions, (n,), generator=self._gen, device=self.device, dtype=torch.int32)
        a = torch.empty(n, dtype=torch.int32, device=self.device)
        check(self.lib.mrl_qnet_act(self.handle, ptr(self.params), ptr(obs), n, float(self.eps), ptr(u), ptr(rnd), ptr(a),
                                    None, ptr(self.workspace), self.workspace.numel(), self.max_batch, stream_ptr()),
              'mrl_qnet_act')
        if update_eps >= 0:
            self.eps = float(update_eps)
        return a.cpu().numpy().astype(np.int64)

    def _next_alpha(self):
        """TF-1 Adam step size of this step + the beta-power update (host f32 arithmetic like the slot variables)"""
        one = np.float32(1)
        alpha = np.float32(self.lr) * np.sqrt(one - self.beta2_power) / (one - self.beta1_power)
        self.beta1_power = np.float32(self.beta1_power * self.beta1)
        self.beta2_power = np.float32(self.beta2_power * self.beta2)
        return alpha

    def _step_launches(self, o1, a, r, o2, d, w, td, B, alpha, alpha_dev):
        """the launches of one optimizer step (build_graph.py:380-421) on the current stream"""
        check(self.lib.mrl_qnet_td_grad(self.handle, ptr(self.params), ptr(self.target), ptr(o1), ptr(a), ptr(r), ptr(o2),
                                        ptr(d), ptr(w), self.gamma, int(self.double_q), B, ptr(self.grads), ptr(td),
                                        ptr(self._loss), ptr(self.workspace), self.workspace.numel(), stream_ptr()),
              'mrl_qnet_td_grad')
        check(self.lib.mrl_qnet_adam_step(self.handle, ptr(self.params), ptr(self.grads), ptr(self.adam_m), ptr(self.adam_v),
                                          float(alpha), ptr(alpha_dev), float(self.beta1), float(self.beta2),
                                          float(self.epsilon), self.grad_norm_clipping, ptr(self.workspace),
                                          self.workspace.numel(), self.max_batch, stream_ptr()), 'mrl_qnet_adam_step')

    def train_dev(self, obs_t, action, reward, obs_tp1, done, weight, graph=True):
        """One optimizer step on DEVICE tensors (obs in the network's dtype, action int32, the rest f32); returns the device
        td_errors f32 [B] without a host round trip -- `PrioritizedReplayBuffer.update_priorities_from_td` consumes them.

        The ~70 small launches of a batch-32 step (three forward passes, one backward, per-variable clip, Adam) are
        launch-latency-bound, so by default the step is captured ONCE per batch size as a hipGraph that reads its inputs
        from static device buffers and its Adam step size from a device word, and is replayed afterwards."""
        obs_t, obs_tp1 = self._obs(obs_t), self._obs(obs_tp1)     # one-hot / dtype / shape as the network wants them (no-ops else)
        B = int(obs_t.shape[0])
        self._set_batch(B)
        if not graph or os.environ.get('MRL_DQN_GRAPH', '1') == '0' or _lib.prof_enabled():
            td = torch.empty(B, dtype=torch.float32, device=self.device)
            o12 = torch.cat([obs_t, obs_tp1], dim=0)     # back to back: the same merged online pass as the replayed graph (bit-identical)
            self._step_launches(o12[:B], action, reward, o12[B:], done, weight, td, B, self._next_alpha(), None)
            self.last_td = td
            return td
        g = self.graph_inputs(B, tuple(obs_t.shape[1:]), obs_t.dtype)
        for k, src in (('o1', obs_t), ('o2', obs_tp1), ('a', action), ('r', reward), ('d', done), ('w', weight)):
            if src.data_ptr() != g[k].data_ptr():        # a minibatch sampled straight into the static buffers needs no copy
                g[k].copy_(src)
        alpha = float(self._next_alpha())
        if alpha != g['alpha_host']:                     # constant after Adam's bias correction has saturated in fp32
            g['alpha'].fill_(alpha)
            g['alpha_host'] = alpha
        if g['graph'] is None or g['ws'] != self.workspace.data_ptr():
            # first step at this batch size: run it eagerly from the static buffers (also warms every kernel up), capture the
            # launch sequence for the following steps
            self._step_launches(g['o1'], g['a'], g['r'], g['o2'], g['d'], g['w'], g['td'], B, 0.0, g['alpha'])
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            try:                                         # (stream capture records the launches, it does not execute them)
                with _lib.capture_graph(graph):
                    self._step_launches(g['o1'], g['a'], g['r'], g['o2'], g['d'], g['w'], g['td'], B, 0.0, g['alpha'])
                g['graph'], g['ws'] = graph, self.workspace.data_ptr()
            except Exception as exc:                     # capture unsupported here: stay eager
                import warnings
                warnings.warn('DQN step graph capture failed (%s); using eager launches' % (exc,))
                g['graph'], g['ws'] = False, self.workspace.data_ptr()
        elif g['graph'] is False:
            self._step_launches(g['o1'], g['a'], g['r'], g['o2'], g['d'], g['w'], g['td'], B, 0.0, g['alpha'])
        else:
            g['graph'].replay()
        self.last_td = g['td']
        return g['td']

    def graph_inputs(self, B, ob_shape=None, ob_dtype=None):
        """The static input buffers of the captured step at batch size B (o1 | o2 back to back: one online pass of 2 B rows).
        `ReplayBuffer.sample_dev(..., out=model.graph_inputs(B))` gathers the minibatch straight into them."""
        g = self._graphs.get(B)
        if g is None:
            if ob_shape is None:
                ob_shape, ob_dtype = ((self.ob_onehot,), torch.float32) if self.ob_onehot else (tuple(self.ob_shape), self.torch_ob_dtype)
            o12 = torch.empty((2 * B,) + tuple(ob_shape), dtype=ob_dtype, device=self.device)
            g = dict(o1=o12[:B], o2=o12[B:],
                     a=torch.empty(B, dtype=torch.int32, device=self.device),
                     r=torch.empty(B, dtype=torch.float32, device=self.device),
                     d=torch.empty(B, dtype=torch.float32, device=self.device),
                     w=torch.empty(B, dtype=torch.float32, device=self.device),
                     td=torch.empty(B, dtype=torch.float32, device=self.device),
                     alpha=torch.zeros(1, dtype=torch.float32, device=self.device), alpha_host=0.0, graph=None,
                     ws=self.workspace.data_ptr())
            self._graphs[B] = g
        return g

    def train(self, obs_t, action, reward, obs_tp1, done, weight):
        """one optimizer step; returns td_errors f32 [B] on the host (build_graph.py:430-441)"""
        f = lambda x, dt: (x.to(self.device, dt) if isinstance(x, torch.Tensor)
                           else torch.from_numpy(np.ascontiguousarray(np.asarray(x))).to(self.device).to(dt)).contiguous()
        a, r, d, w = f(action, torch.int32), f(reward, torch.float32), f(done, torch.float32), f(weight, torch.float32)
        return self.train_dev(obs_t, a, r, obs_tp1, d, w).cpu().numpy()

    def ones(self, n):
        """importance weights of uniform replay (deepq.py:297: np.ones_like(rewards)) as a cached device tensor"""
        t = self._ones.get(n)
        if t is None:
            t = self._ones[n] = torch.ones(n, dtype=torch.float32, device=self.device)
        return t

    def update_target(self):
        self.target.copy_(self.params)

    # ---- checkpoints: {tf_variable_name: ndarray} like tf_util.save_variables (tf_util.py:345-372) ----------------
    def variables(self):
        out = {'deepq/eps:0': np.float32(self.eps)}
        for buf, scope, suffix in ((self.params, 'deepq/q_func', ''), (self.target, 'deepq/target_q_func', ''),
                                   (self.adam_m, 'deepq/q_func', '/Adam'), (self.adam_v, 'deepq/q_func', '/Adam_1')):
            host = buf.detach().cpu().numpy()
            for t in self.tensors:
                nm = t['name'].replace('deepq/q_func', scope, 1) + suffix + ':0'
                out[nm] = host[t['offset']:t['offset'] + t['size']].reshape(t['shape']).copy()
        out['beta1_power:0'], out['beta2_power:0'] = np.float32(self.beta1_power), np.float32(self.beta2_power)
        if self.param_noise:
            out['deepq/param_noise_scale:0'] = np.float32(self.param_noise_scale)
            out['deepq/param_noise_threshold:0'] = np.float32(self.param_noise_threshold)
            for buf, scope in ((self.perturbed, 'deepq/perturbed_q_func'), (self.adaptive, 'deepq/adaptive_q_func')):
                host = buf.detach().cpu().numpy()
                for t in self.tensors:
                    nm = t['name'].replace('deepq/q_func', scope, 1) + ':0'
                    out[nm] = host[t['offset']:t['offset'] + t['size']].reshape(t['shape']).copy()
        return out

    def load_variables(self, d):
        for buf, scope, suffix, required in ((self.params, 'deepq/q_func', '', True), (self.target, 'deepq/target_q_func', '', False),
                                             (self.adam_m, 'deepq/q_func', '/Adam', False),
                                             (self.adam_v, 'deepq/q_func', '/Adam_1', False)):
            host = buf.detach().cpu().numpy()
            for t in self.tensors:
                nm = t['name'].replace('deepq/q_func', scope, 1) + suffix + ':0'
                if nm in d:
                    host[t['offset']:t['offset'] + t['size']] = np.asarray(d[nm], np.float32).reshape(-1)
                elif required:
                    raise KeyError(nm)
            buf.copy_(torch.from_numpy(host))
        if 'deepq/eps:0' in d:
            self.eps = float(d['deepq/eps:0'])
        if self.param_noise:
            for buf, scope in ((self.perturbed, 'deepq/perturbed_q_func'), (self.adaptive, 'deepq/adaptive_q_func')):
                host = self.params.detach().cpu().numpy().copy()         # absent in the file: unperturbed
                for t in self.tensors:
                    nm = t['name'].replace('deepq/q_func', scope, 1) + ':0'
                    if nm in d:
                        host[t['offset']:t['offset'] + t['size']] = np.asarray(d[nm], np.float32).reshape(-1)
                buf.copy_(torch.from_numpy(host))
            if 'deepq/param_noise_scale:0' in d:
                self.param_noise_scale = np.float32(d['deepq/param_noise_scale:0'])
            if 'deepq/param_noise_threshold:0' in d:
                self.param_noise_threshold = np.float32(d['deepq/param_noise_threshold:0'])
        if 'beta1_power:0' in d:
            self.beta1_power, self.beta2_power = np.float32(d['beta1_power:0']), np.float32(d['beta2_power:0'])

    def get_flat_params(self):
        return self.params.detach().cpu().numpy().copy()
