"""PPO2 learner object with the reference's protocol (ppo2/model.py:27-158):

    Model(policy=, ob_space=, ac_space=, nbatch_act=, nbatch_train=, nsteps=, ent_coef=, vf_coef=,
          max_grad_norm=, mpi_rank_weight=1, comm=None, microbatch_size=None)
    .step(obs, S=, M=) -> (actions, values, states, neglogpacs)     .value(obs, S=, M=)
    .train(lr, cliprange, obs, returns, masks, actions, values, neglogpacs, states=None) -> 5 stats
    .loss_names  .initial_state  .save(path)  .load(path)

Where the reference builds a TF graph and calls sess.run, this object owns flat fp32 device
buffers (params / grads / Adam m, v) and drives the HIP kernels through the C ABI:
    mrl_model_act   (policy forward + sampling)
    mrl_model_train_step  (gather + fwd + loss + bwd + [RCCL all-reduce inside the backward] + avg -> clip -> Adam)
    (pieces: mrl_model_grad, mrl_allreduce_grads, mrl_adam_clip_step)
Extra, faster entry points used by our Runner / learn (no host round trips):
    .step_into(...)   .train_indexed(lr, cliprange, rollout, idx_dev)
"""
import os

import numpy as np
import torch

from .. import _lib, ops
from ..common.dist import default_comm


def ortho_init(shape, scale):
    """Orthogonal init drawn from the GLOBAL NumPy stream exactly like the reference
    (a2c/utils.py:20-35: np.random.normal -> SVD -> pick the factor with the right shape), so
    `set_global_seeds(seed)` reproduces the reference's weights.  Host code by design: LAPACK sign
    conventions make a device re-implementation non-reproducible."""
    shape = tuple(shape)
    if len(shape) == 2:
        flat_shape = shape
    elif len(shape) == 4:
        flat_shape = (int(np.prod(shape[:-1])), shape[-1])
    else:
        raise NotImplementedError
    a = np.random.normal(0.0, 1.0, flat_shape)
    u, _, v = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == flat_shape else v
    q = q.reshape(shape)
    return (scale * q[:shape[0], :shape[1]]).astype(np.float32)


def checkpoint_to_flat(loaded, tensors):
    """Host-only half of `Model.load` (no device needed): maps what `joblib.load` returned from a checkpoint written by
    the reference's `save_variables` (tf_util.py:345-355) onto the flat parameter / Adam-slot buffers.

    tensors: [(tf_variable_name, shape, offset, size)] in variable-creation order (`mrl_model_tensor_info`).
    loaded:  * dict {var.name: ndarray} -- `ppo2_model/.../w:0`, `.../w/Adam:0`, `.../w/Adam_1:0`, `beta1_power:0`,
               `beta2_power:0` (tf_util.py:367-369 assigns by name; here parameters are required, optimizer state optional);
             * list (legacy format, tf_util.py:362-366) -- one array per GLOBAL variable in creation order: the n model
               variables, then what `AdamOptimizer._create_slots` adds when `apply_gradients` runs outside the
               `ppo2_model` scope (model.py:100-114): beta1_power, beta2_power, then (m, v) for each variable in turn.
               A list of only the n model variables (saved with `variables=trainable`) is accepted too; any other length
               fails with the reference's message.
    Returns {'params' | 'adam_m' | 'adam_v': [(offset, flat float32 array)], 'beta_powers': (b1, b2) | None}."""
    out = {'params': [], 'adam_m': [], 'adam_v': [], 'beta_powers': None}

    def put(key, t, arr):
        name, shape, off, size = t
        a = np.asarray(arr, np.float32)
        # the variable's own shape, or -- only where that cannot be mistaken for another layout -- a flat array of a variable
        # that is itself 1-D or scalar (tf_util.py:369 assigns through tf.assign, which requires the exact shape)
        flat_ok = a.ndim == 1 and len([d for d in shape if d != 1]) <= 1
        if a.size != size or (tuple(a.shape) != tuple(shape) and not flat_ok and a.ndim != 0):
            raise ValueError('checkpoint variable %s has shape %s, the model expects %s' % (name, a.shape, tuple(shape)))
        out[key].append((off, a.reshape(-1)))

    n = len(tensors)
    if isinstance(loaded, list):
        # (the params-only list is an extension: tf_util.load_variables demands len == number of global variables)
        if len(loaded) not in (n, 3 * n + 2):
            raise ValueError('number of variables loaded mismatches len(variables)')           # tf_util.py:363's message, -O-proof
        for t, a in zip(tensors, loaded[:n]):
            put('params', t, a)
        if len(loaded) > n:
            out['beta_powers'] = (np.float32(loaded[n]), np.float32(loaded[n + 1]))
            for i, t in enumerate(tensors):
                put('adam_m', t, loaded[n + 2 + 2 * i])
                put('adam_v', t, loaded[n + 3 + 2 * i])
        return out
    for t in tensors:
        put('params', t, loaded[t[0] + ':0'])          # KeyError for a missing model variable, like tf_util.py:369
        if t[0] + '/Adam:0' in loaded:
            put('adam_m', t, loaded[t[0] + '/Adam:0'])
        if t[0] + '/Adam_1:0' in loaded:
            put('adam_v', t, loaded[t[0] + '/Adam_1:0'])
    if 'beta1_power:0' in loaded:
        out['beta_powers'] = (np.float32(loaded['beta1_power:0']), np.float32(loaded['beta2_power:0']))
    return out


class Model(object):
    loss_names = ['policy_loss', 'value_loss', 'policy_entropy', 'approxkl', 'clipfrac']

    def __init__(self, *, policy, ob_space, ac_space, nbatch_act, nbatch_train, nsteps, ent_coef, vf_coef,
                 max_grad_norm, mpi_rank_weight=1, comm=None, microbatch_size=None, chunk=None, device=None):
        _lib.require_gpu()
        self.device = torch.device(device or ('cuda:%d' % torch.cuda.current_device()))
        if comm is None:
            comm = default_comm()
        self.comm = comm
        self.mpi_rank_weight = mpi_rank_weight
        self.policy = policy
        self.nbatch_act, self.nbatch_train, self.nsteps = nbatch_act, nbatch_train, nsteps
        self.ent_coef, self.vf_coef, self.max_grad_norm = float(ent_coef), float(vf_coef), max_grad_norm
        self.microbatch_size = microbatch_size
        kw = policy.device_model_kwargs()
        if chunk is None:
            # samples processed per pass of the layer kernels.  Activations and their gradients live in
            # the workspace (~173 KB/sample for NatureCNN): with 288 GB of HBM a whole 131072-sample
            # minibatch (22.7 GB) fits, which means ONE split-K reduction per layer per minibatch step.
            chunk = 131072 if kw['network'] == 'cnn' else 1 << 20
        chunk = int(max(1, min(chunk, max(nbatch_train or 1, nbatch_act or 1))))
        self.dm = ops.DeviceModel(chunk=chunk, device=self.device, **kw)
        self.pd_kind, self.nact = self.dm.pd_kind, self.dm.nact
        P = self.dm.P
        # ---- parameters: reference init order and RNG stream ----
        flat = np.zeros(P, np.float32)
        for t in self.dm.tensors:
            if t['init_scale'] is not None:
                flat[t['offset']:t['offset'] + t['size']] = ortho_init(t['shape'], t['init_scale']).reshape(-1)
            elif t.get('init_const'):
                flat[t['offset']:t['offset'] + t['size']] = t['init_const']      # layer-norm gamma: ones
        self.params = torch.from_numpy(flat).to(self.device)
        self.grads = torch.zeros(P, dtype=torch.float32, device=self.device)
        self.adam_m = torch.zeros(P, dtype=torch.float32, device=self.device)
        self.adam_v = torch.zeros(P, dtype=torch.float32, device=self.device)
        self._scratch = torch.empty(int(_lib.load().mrl_adam_scratch_bytes(P)), dtype=torch.uint8, device=self.device)
        self._gnorm = torch.zeros(1, dtype=torch.float32, device=self.device)
        # tf.train.AdamOptimizer(learning_rate=LR, epsilon=1e-5) (model.py:98-100): f32 slot variables
        self.beta1, self.beta2, self.epsilon = np.float32(0.9), np.float32(0.999), np.float32(1e-5)
        self.beta1_power, self.beta2_power = np.float32(0.9), np.float32(0.999)
        # recurrent policies: the LSTM state is managed outside the policy (common/models.py:132-176); initial_state is
        # np.zeros([nenv, 2*nlstm], dtype=float) of the act model (models.py:171)
        self.recurrent = self.dm.recurrent
        self.initial_state = np.zeros((nbatch_act, self.dm.state_size), dtype=float) if self.recurrent else None
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(int(torch.initial_seed()) & 0x7fffffff)
        self._train_calls = 0
        self._fast = None
        self._epoch_graph = None
        self._rings = {}                                  # pinned staging rings of the asynchronous uploads (_upload)
        # ---- multi-rank: total weight (mpi_adam_optimizer.py:25-27), sync_from_root (model.py:129-131) ----
        self.total_weight = 1.0
        if self.comm is not None and self.comm.Get_size() > 1:
            self.total_weight = self.comm.total_weight(mpi_rank_weight)
            for t in (self.params, self.adam_m, self.adam_v):
                self.comm.bcast_(t, 0)
        self.multi = self.comm is not None and self.comm.Get_size() > 1
        # in-library RCCL communicator: the all-reduce is issued by libmrl.so from inside the backward pass
        self.native_dp = self.multi and getattr(self.comm, 'native', None) is not None
        if self.native_dp:
            self.dm.attach_comm(self.comm.native, float(mpi_rank_weight))

    # ------------------------------------------------------------------ act side
    def _to_dev_obs(self, obs):
        if self.policy.ob_onehot:                   # Discrete observations: tf.one_hot of input.py:57-58
            t = obs.to(self.device) if isinstance(obs, torch.Tensor) else torch.from_numpy(np.asarray(obs)).to(self.device)
            if t.dim() < 2 or t.shape[-1] != self.policy.ob_onehot or not t.is_floating_point():
                t = torch.nn.functional.one_hot(t.reshape(-1).long(), self.policy.ob_onehot)
            return t.to(torch.float32).reshape(-1, self.policy.ob_onehot).contiguous()
        if isinstance(obs, torch.Tensor):
            t = obs.to(self.device)
        else:
            a = np.asarray(obs)
            if a.dtype == np.int8:
                a = a.view(np.uint8)
            t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        if t.dtype != self.dm.torch_ob_dtype:
            t = t.to(self.dm.torch_ob_dtype)
        if self.policy.ob_clip:                     # normalize_observations: see build_policy
            t = t.clamp(-self.policy.ob_clip, self.policy.ob_clip)
        # tf_util.adjust_shape (tf_util.py:377-401): anything reshapeable to [-1, *ob_shape] is accepted
        return t.reshape((-1,) + tuple(self.policy.ob_shape)).contiguous()

    def make_noise(self, n):
        """uniform(0,1) for the Gumbel-max sampler (distributions.py:199-201), N(0,1) for the Gaussian
        (:247-248); torch's Philox generator on the device stands in for TF's."""
        if self.pd_kind != 'gaussian':              # Gumbel-max per (slice of the) logits / u < sigmoid(logit) (:271-273)
            return torch.rand((n, self.nact), generator=self._gen, device=self.device, dtype=torch.float32)
        return torch.randn((n, self.nact), generator=self._gen, device=self.device, dtype=torch.float32)

    def _dev_state(self, S, n):
        """S of the reference's feed (host [n, 2*nlstm], float64 zeros initially) -> contiguous f32 device tensor"""
        if isinstance(S, torch.Tensor):
            t = S.to(self.device, torch.float32)
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(S, dtype=np.float32))).to(self.device)
        return t.reshape(n, self.dm.state_size).contiguous()

    def _dev_mask(self, M, n):
        """M: done flags entering the step (bools / 0-1 floats / uint8) -> u8 device tensor [n]"""
        if isinstance(M, torch.Tensor):
            t = M.to(self.device)
            t = t.view(torch.uint8) if t.dtype == torch.bool else (t != 0).to(torch.uint8)
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(M).astype(np.bool_)).view(np.uint8)).to(self.device)
        return t.reshape(n).contiguous()

    def step_into(self, obs_dev, actions_out, values_out, neglogp_out, noise=None, states=None, masks=None):
        """Device fast path: obs [n, ...] device tensor; results written into the given device
        buffers (e.g. slot t of the rollout SoA) -- K1 'rollout store' is therefore zero-copy.
        Recurrent policies: `states` (f32 device [n, 2*nlstm]) is advanced IN PLACE, `masks` u8 device [n]."""
        if noise is None:
            noise = self.make_noise(obs_dev.shape[0])
        if self.recurrent:
            self.dm.act_rnn_into(self.params, obs_dev, noise, states, masks, states, actions_out, values_out, neglogp_out)
        else:
            self.dm.act_into(self.params, obs_dev, noise, actions_out, values_out, neglogp_out)

    def step(self, observation, S=None, M=None, noise=None, **_):
        """policies.py:77-96: returns host arrays (actions int64 [n] | f32 [n, nact], values f32,
        next state f32 [n, 2*nlstm] | None, neglogpacs f32)."""
        obs = self._to_dev_obs(observation)
        n = obs.shape[0]
        if noise is None:
            noise = self.make_noise(n)
        elif not isinstance(noise, torch.Tensor):
            noise = torch.from_numpy(np.ascontiguousarray(noise, dtype=np.float32)).to(self.device)
        state = None
        if self.recurrent:
            assert S is not None and M is not None, 'recurrent policy: step(obs, S=states, M=dones)'
            st = self._dev_state(S, n).clone()
            a = torch.empty(self.dm.action_shape(n), dtype=self.dm.action_dtype, device=self.device)
            v = torch.empty(n, dtype=torch.float32, device=self.device)
            nlp = torch.empty(n, dtype=torch.float32, device=self.device)
            self.dm.act_rnn_into(self.params, obs, noise.contiguous(), st, self._dev_mask(M, n), st, a, v, nlp)
            state = st.cpu().numpy()
        else:
            a, v, nlp, _ = self.dm.act(self.params, obs, noise)
        a = a.cpu().numpy()
        if self.pd_kind == 'categorical':
            a = a.astype(np.int64)       # tf.argmax dtype
        elif self.pd_kind == 'bernoulli':
            a = a.astype(np.float32)     # tf.to_float(u < p), distributions.py:273 (MultiCategorical: tf.int32, :224)
        return a, v.cpu().numpy(), state, nlp.cpu().numpy()

    def value(self, ob, *args, S=None, M=None, **kwargs):
        obs = self._to_dev_obs(ob)
        if self.recurrent:
            assert S is not None and M is not None, 'recurrent policy: value(obs, S=states, M=dones)'
            return self.value_dev(obs, self._dev_state(S, obs.shape[0]), self._dev_mask(M, obs.shape[0])).cpu().numpy()
        return self.dm.act(self.params, obs, None, want_actions=False)[1].cpu().numpy()

    def value_dev(self, obs_dev, states=None, masks=None):
        if self.policy.ob_clip:
            obs_dev = obs_dev.clamp(-self.policy.ob_clip, self.policy.ob_clip)
        if self.recurrent:
            v = torch.empty(obs_dev.shape[0], dtype=torch.float32, device=self.device)
            self.dm.act_rnn_into(self.params, obs_dev, None, states, masks, torch.empty_like(states), None, v, None)
            return v
        return self.dm.act(self.params, obs_dev, None, want_actions=False)[1]

    # ------------------------------------------------------------------ learner
    def _apply_gradients(self, lr):
        """[RCCL all-reduce] -> / total weight -> clip_by_global_norm -> Adam   (model.py:102-114,
        mpi_adam_optimizer.py:21,39-40).  Everything stays on the device / on the stream."""
        if self.multi:
            if not self.native_dp:                    # else: mrl_model_grad already returned the weighted sum over ranks
                if self.mpi_rank_weight != 1:
                    self.grads.mul_(float(self.mpi_rank_weight))
                self.comm.allreduce_sum_(self.grads)
            if self._train_calls % 100 == 0:          # mpi_adam_optimizer.py:41-43
                self.comm.check_synced(self.params[:1024].sum().reshape(1))
        one = np.float32(1)
        alpha = np.float32(lr) * np.sqrt(one - self.beta2_power) / (one - self.beta1_power)
        ops.adam_clip_step(self.params, self.grads, self.adam_m, self.adam_v, alpha, self.beta1, self.beta2,
                           self.epsilon, self.max_grad_norm, self.total_weight, self._scratch, self._gnorm)
        self.beta1_power = np.float32(self.beta1_power * self.beta1)
        self.beta2_power = np.float32(self.beta2_power * self.beta2)
        self._train_calls += 1

    def train_indexed(self, lr, cliprange, rollout, idx_dev, stats_out=None, states=None):
        """One minibatch step reading the device rollout in place: `idx_dev` (int64 device tensor) holds
        the reference's env-major flat indices (ppo2.py:160-162); the gather is fused into the
        first-layer loaders.  Returns a device tensor [5] (no host sync).

        This is the per-step host path of the launch-bound MLP configs (320 steps per update), so the
        device pointers of the long-lived buffers are cached and the two C calls are made directly."""
        stats = stats_out if stats_out is not None else torch.empty(5, dtype=torch.float32, device=self.device)
        if self.recurrent:
            # env-wise minibatch (ppo2.py:167-180): idx_dev = whole trajectories of the chosen envs (flat index e*T + t, env
            # after env), states f32 device [nseq, 2*nlstm] = their states when the rollout started; masks = rollout.dones
            assert states is not None, 'recurrent policy: train_indexed(..., states=mbstates)'
            self.dm.grad_rnn(self.params, rollout.obs, rollout.actions, rollout.returns, rollout.values, rollout.neglogpacs,
                             rollout.dones, states.contiguous(), states.shape[0], idx_dev, idx_dev.numel(), rollout.T,
                             rollout.N, cliprange, self.ent_coef, self.vf_coef, self.grads, stats)
            self._apply_gradients(lr)
            return stats
        c = self._fast
        if c is None or c['ro'] is not rollout or c['obs_ptr'] != rollout.obs.data_ptr():
            vp = _lib.c_void_p
            lib = _lib.load()
            c = self._fast = dict(
                ro=rollout, obs_ptr=rollout.obs.data_ptr(), lib=lib, h=self.dm.handle, params=_lib.ptr(self.params),
                grads=_lib.ptr(self.grads), m=_lib.ptr(self.adam_m), v=_lib.ptr(self.adam_v),
                ws=_lib.ptr(self.dm.workspace), wsn=self.dm.workspace.numel(), chunk=self.dm.chunk,
                obs=_lib.ptr(rollout.obs), act=_lib.ptr(rollout.actions), val=_lib.ptr(rollout.values),
                nlp=_lib.ptr(rollout.neglogpacs), scratch=_lib.ptr(self._scratch), gnorm=_lib.ptr(self._gnorm),
                T=int(rollout.T), N=int(rollout.N), P=self.params.numel())
        if self.multi and not self.native_dp:          # collectives through torch.distributed (gloo / MRL_NATIVE_COMM=0)
            self.dm.grad(self.params, rollout.obs, rollout.actions, rollout.returns, rollout.values, rollout.neglogpacs,
                         idx_dev, idx_dev.numel(), rollout.T, rollout.N, cliprange, self.ent_coef, self.vf_coef,
                         self.grads, stats)
            self._apply_gradients(lr)
            return stats
        if self.multi and self._train_calls % 100 == 0:          # mpi_adam_optimizer.py:41-43
            self.comm.check_synced(self.params[:1024].sum().reshape(1))
        st = _lib.stream_ptr()
        lib = c['lib']
        one = np.float32(1)
        alpha = np.float32(lr) * np.sqrt(one - self.beta2_power) / (one - self.beta1_power)
        mgn = -1.0 if self.max_grad_norm is None else float(self.max_grad_norm)
        # Model.train as ONE C call: gather -> fwd -> loss -> bwd -> [RCCL all-reduce inside the backward] -> clip -> Adam
        rc = lib.mrl_model_train_step(c['h'], c['params'], c['grads'], c['m'], c['v'], c['obs'], c['act'],
                                      _lib.ptr(rollout.returns), c['val'], c['nlp'], _lib.ptr(idx_dev), idx_dev.numel(),
                                      c['T'], c['N'], float(cliprange), self.ent_coef, self.vf_coef, float(alpha), None,
                                      float(self.beta1), float(self.beta2), float(self.epsilon), mgn,
                                      float(self.total_weight), _lib.ptr(stats), c['gnorm'], c['ws'], c['wsn'], c['chunk'], st)
        if rc:
            _lib.check(rc, 'mrl_model_train_step')
        self.beta1_power = np.float32(self.beta1_power * self.beta1)
        self.beta2_power = np.float32(self.beta2_power * self.beta2)
        self._train_calls += 1
        return stats

    # ------------------------------------------------------------------ asynchronous host -> device uploads
    def _upload(self, host_array, key, slots=4):
        """Copy a host array to the device WITHOUT blocking the host: through a ring of pinned staging buffers, each guarded
        by an event (a slot is reused only after the copy that read it has run), into a ring of device buffers.
        A pageable `tensor.to(device)` makes the host wait for everything queued before it -- with one launch graph per epoch
        that left the GPU idle between epochs while the next permutation was drawn and shipped (0.5 ms of every 2.7 ms in the
        MuJoCo-shaped configuration).  The returned tensor is valid until `slots` further uploads with the same key."""
        a = np.ascontiguousarray(host_array)
        ring = self._rings.get(key)
        if ring is None or ring['shape'] != a.shape or ring['dtype'] != a.dtype:
            tdt = torch.from_numpy(a[:0].copy()).dtype
            ring = self._rings[key] = dict(
                shape=a.shape, dtype=a.dtype, n=0,
                host=[torch.empty(a.shape, dtype=tdt).pin_memory() for _ in range(slots)],
                dev=[torch.empty(a.shape, dtype=tdt, device=self.device) for _ in range(slots)],
                ev=[torch.cuda.Event() for _ in range(slots)])
        k = ring['n'] % len(ring['host'])
        ring['n'] += 1
        if ring['n'] > len(ring['host']):
            ring['ev'][k].synchronize()
        ring['host'][k].numpy()[...] = a
        ring['dev'][k].copy_(ring['host'][k], non_blocking=True)
        ring['ev'][k].record()
        return ring['dev'][k]

    def indices_to_device(self, inds):
        """an epoch's permutation (or a minibatch's index list) as a device tensor, uploaded asynchronously"""
        return self._upload(np.asarray(inds, dtype=np.int64), 'inds')

    # ------------------------------------------------------------------ one epoch as a replayable launch graph
    def _next_alpha(self, lr):
        """TF-1 Adam step size for the next step + the beta-power update (host f32 arithmetic, model.py:98-100)"""
        one = np.float32(1)
        alpha = np.float32(lr) * np.sqrt(one - self.beta2_power) / (one - self.beta1_power)
        self.beta1_power = np.float32(self.beta1_power * self.beta1)
        self.beta2_power = np.float32(self.beta2_power * self.beta2)
        self._train_calls += 1
        return alpha

    def train_epoch(self, lr, cliprange, rollout, inds_dev):
        """All minibatch steps of one epoch (`inds_dev`: the epoch's permutation, nminibatches * nbatch_train env-major
        indices on the device).  Returns a device tensor [nminibatches, 5] of loss stats.

        The launch-bound MLP configurations (320 steps of ~7 small kernels per update) run the epoch as ONE hipGraph:
        the step sequence is captured once (again only when the clip range or a rollout buffer changes: the rollout fields
        are written in place, the learning rate enters through the step sizes), reads its indices and its Adam step sizes
        from device buffers that are refreshed before every replay, and is replayed once per epoch -- HIP graphs instead of ~2,000 individual launches per update.  Large-kernel
        configurations (NatureCNN), multi-rank runs (the all-reduce sits between the two halves of a step) and
        per-kernel profiling use the plain step loop."""
        B = self.nbatch_train
        M = inds_dev.numel() // B
        graphable = (self._epoch_graph is not False and not self.multi and self.dm.network == 'mlp'
                     and self._train_calls > 0 and not _lib.prof_enabled()
                     and type(self).train_indexed is Model.train_indexed          # subclasses that hook the step keep it
                     and os.environ.get('MRL_EPOCH_GRAPH', '1') != '0')
        if not graphable:
            return torch.stack([self.train_indexed(lr, cliprange, rollout, inds_dev[k * B:(k + 1) * B]) for k in range(M)])
        # (the learning rate is not part of the captured sequence: the Adam step sizes come from the `alpha` buffer)
        # everything the captured launches bake in: rollout pointers, shapes, the workspace (set_chunk re-allocates it) and the
        # engine options read at capture time -- a set_option / set_chunk after the first capture re-captures
        key = (float(cliprange), rollout.obs.data_ptr(), rollout.actions.data_ptr(), rollout.returns.data_ptr(),
               rollout.values.data_ptr(), rollout.neglogpacs.data_ptr(), M, B, self.dm.workspace.data_ptr(), self.dm.chunk,
               tuple(_lib.get_option(o) for o in ('mlp_fused', 'mlp_waves', 'mlp_slice')))
        g = self._epoch_graph
        if g is None or g['key'] != key:
            try:
                g = self._epoch_graph = self._capture_epoch(key, cliprange, rollout, M, B)
            except Exception as exc:                     # capture unsupported here: keep the step loop from now on
                import warnings
                warnings.warn('epoch graph capture failed (%s); using the step loop' % (exc,))
                self._epoch_graph = False
                torch.cuda.synchronize()
                return self.train_epoch(lr, cliprange, rollout, inds_dev)
        g['idx'].copy_(inds_dev.view(M, B))
        g['alpha'].copy_(self._upload(np.array([self._next_alpha(lr) for _ in range(M)], dtype=np.float32), 'alpha'))
        g['graph'].replay()
        return g['stats'].clone()

    def _capture_epoch(self, key, cliprange, rollout, M, B):
        lib, st_dtype = _lib.load(), torch.float32
        idx = torch.zeros((M, B), dtype=torch.int64, device=self.device)
        alpha = torch.zeros(M, dtype=torch.float32, device=self.device)
        stats = torch.zeros((M, 5), dtype=st_dtype, device=self.device)
        adv = torch.zeros((M, 2), dtype=torch.float32, device=self.device)     # (mean, std) of every minibatch of the epoch
        mgn = -1.0 if self.max_grad_norm is None else float(self.max_grad_norm)
        P = self.params.numel()
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        try:
            return self._capture_epoch_into(graph, key, cliprange, rollout, M, B, idx, alpha, stats, adv, mgn)
        finally:
            lib.mrl_model_set_advstat(self.dm.handle, None)

    def _capture_epoch_into(self, graph, key, cliprange, rollout, M, B, idx, alpha, stats, adv, mgn):
        lib = _lib.load()
        with _lib.capture_graph(graph):
            st = _lib.stream_ptr()                         # the capturing stream
            # the epoch's permutation is known up front: the advantage statistics (model.py:136-139) of all its minibatches
            # in ONE launch, instead of every workgroup of every fused step gathering its whole minibatch again
            fused = self.dm.network == 'mlp' and _lib.get_option('mlp_fused')
            if fused:
                _lib.check(lib.mrl_advstat_minibatches(_lib.ptr(rollout.returns), _lib.ptr(rollout.values), _lib.ptr(idx), M, B,
                                                       int(rollout.T), int(rollout.N), _lib.ptr(adv), st), 'mrl_advstat_minibatches')
            for k in range(M):
                if fused:
                    _lib.check(lib.mrl_model_set_advstat(self.dm.handle, _lib.ptr(adv[k])), 'mrl_model_set_advstat')
                _lib.check(lib.mrl_model_train_step(
                    self.dm.handle, _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v),
                    _lib.ptr(rollout.obs), _lib.ptr(rollout.actions), _lib.ptr(rollout.returns), _lib.ptr(rollout.values),
                    _lib.ptr(rollout.neglogpacs), _lib.ptr(idx[k]), B, int(rollout.T), int(rollout.N), float(cliprange),
                    self.ent_coef, self.vf_coef, 0.0, _lib.ptr(alpha[k:k + 1]), float(self.beta1), float(self.beta2),
                    float(self.epsilon), mgn, float(self.total_weight), _lib.ptr(stats[k]), _lib.ptr(self._gnorm),
                    _lib.ptr(self.dm.workspace), self.dm.workspace.numel(), self.dm.chunk, st), 'mrl_model_train_step')
        return dict(key=key, graph=graph, idx=idx, alpha=alpha, stats=stats, adv=adv)

    def _field(self, x, dtype):
        if isinstance(x, torch.Tensor):
            return x.to(self.device, dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(x))).to(self.device).to(dtype).contiguous()

    def train(self, lr, cliprange, obs, returns, masks, actions, values, neglogpacs, states=None):
        """Reference signature (model.py:133-158): arrays of one already-gathered minibatch (host
        NumPy or device tensors).  Returns the 5 stats as Python floats."""
        obs = self._to_dev_obs(obs)
        B = obs.shape[0]
        act = self._field(actions, self.dm.action_dtype)
        ret, val, nlp = (self._field(x, torch.float32) for x in (returns, values, neglogpacs))
        stats = torch.empty(5, dtype=torch.float32, device=self.device)
        if self.recurrent:             # model.py:153-155: td_map[S] = states, td_map[M] = masks
            assert states is not None, 'recurrent policy: train(..., states=mbstates)'
            nseq = np.asarray(states).shape[0] if not isinstance(states, torch.Tensor) else states.shape[0]
            self.dm.grad_rnn(self.params, obs, act, ret, val, nlp, self._dev_mask(masks, B), self._dev_state(states, nseq),
                             nseq, None, B, 1, 1, cliprange, self.ent_coef, self.vf_coef, self.grads, stats)
        else:
            assert states is None, 'states given to a non-recurrent policy'
            self.dm.grad(self.params, obs, act, ret, val, nlp, None, B, 1, 1, cliprange, self.ent_coef, self.vf_coef,
                         self.grads, stats)
        self._apply_gradients(lr)
        return [float(x) for x in stats.cpu().numpy()]

    # ------------------------------------------------------------------ checkpoints
    def _named(self, flat):
        host = flat.detach().cpu().numpy()
        return {t['name']: host[t['offset']:t['offset'] + t['size']].reshape(t['shape']).copy() for t in self.dm.tensors}

    def save(self, save_path):
        """joblib dict {tf_variable_name: ndarray} over ALL global variables -- parameters, Adam
        slots and beta powers -- the reference's checkpoint format (tf_util.py:345-355; names per
        SURVEY.md App. A.6)."""
        import joblib
        d = {}
        for k, v in self._named(self.params).items():
            d[k + ':0'] = v
        for k, v in self._named(self.adam_m).items():
            d[k + '/Adam:0'] = v
        for k, v in self._named(self.adam_v).items():
            d[k + '/Adam_1:0'] = v
        d['beta1_power:0'] = np.float32(self.beta1_power)
        d['beta2_power:0'] = np.float32(self.beta2_power)
        dirname = os.path.dirname(save_path)
        if dirname:
            os.makedirs(dirname, exist_ok=True)
        joblib.dump(d, save_path)

    def load(self, load_path):
        """tf_util.py:357-372: a dict is assigned by variable name (missing Adam slots keep their values), the legacy
        LIST format (`:362-366`) by position in GLOBAL_VARIABLES order -- see `checkpoint_to_flat`."""
        import joblib
        loaded = joblib.load(os.path.expanduser(load_path))
        tensors = [(t['name'], tuple(t['shape']), t['offset'], t['size']) for t in self.dm.tensors]
        out = checkpoint_to_flat(loaded, tensors)
        for dst, key in ((self.params, 'params'), (self.adam_m, 'adam_m'), (self.adam_v, 'adam_v')):
            host = dst.detach().cpu().numpy()
            for off, arr in out[key]:
                host[off:off + arr.size] = arr
            dst.copy_(torch.from_numpy(host))
        if out['beta_powers'] is not None:
            self.beta1_power, self.beta2_power = out['beta_powers']

    # helpers for tests / users
    def get_flat_params(self):
        return self.params.detach().cpu().numpy().copy()

    def set_flat_params(self, flat):
        self.params.copy_(torch.from_numpy(np.ascontiguousarray(flat, dtype=np.float32)))
