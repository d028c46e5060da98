"""PPO2 rollout collector with the reference's contract (ppo2/runner.py:13-74):

    Runner(env=, model=, nsteps=, gamma=, lam=).run()
        -> (obs, returns, masks, actions, values, neglogpacs, states, epinfos)
    first six: env-major flat [N*T, ...] (index i = e*T + t), dtypes obs-dtype / f32 / bool /
    int64-or-f32 / f32 / f32.

MI355X design: the rollout lives in HBM as time-major SoA `Rollout` ([T][N] fields; coalesced for
the act kernels, the GAE scan and the gather loaders).  The per-step Python lists + np.asarray +
sf01 copies of the reference (3 full copies of the observations) do not exist: the act kernel
writes slot t in place, a device env writes slot t+1 in place, GAE is one HIP kernel, and env-major
order is an index translation inside the consumers.  `run()` still hands back env-major host arrays
when the caller wants them (`return_host=True`, default for host environments); with
`return_host=False` it returns `RolloutField` handles that behave like those arrays under
`arr[mbinds]` but gather on the device.
"""
import os
import time

import numpy as np
import torch

from .. import _lib, ops
from ..common.runners import AbstractEnvRunner


def _defines(obj, name):
    """True when obj's OWN class (or a base of it) implements `name` -- as opposed to a VecEnvWrapper forwarding the
    attribute lookup to the env it wraps (`VecEnvWrapper.__getattr__`), which would silently bypass the wrapper."""
    return any(name in vars(k) for k in type(obj).__mro__)


class Rollout(object):
    """Device-resident SoA rollout buffer (time-major)."""

    def __init__(self, T, N, ob_shape, ob_dtype, pd_kind, nact, device, nsub=None):
        self.T, self.N = T, N
        f32 = dict(dtype=torch.float32, device=device)
        self.obs = torch.empty((T, N) + tuple(ob_shape), dtype=ob_dtype, device=device)
        if pd_kind == 'categorical':
            self.actions = torch.empty((T, N), dtype=torch.int32, device=device)
        elif pd_kind == 'multicategorical':           # one int32 per component of the MultiDiscrete space
            self.actions = torch.empty((T, N, nsub), dtype=torch.int32, device=device)
        elif pd_kind == 'bernoulli':                  # one bit per component of the MultiBinary space
            self.actions = torch.empty((T, N, nact), dtype=torch.int32, device=device)
        else:
            self.actions = torch.empty((T, N, nact), **f32)
        self.rewards = torch.empty((T, N), **f32)
        self.values = torch.empty((T, N), **f32)
        self.neglogpacs = torch.empty((T, N), **f32)
        self.returns = torch.empty((T, N), **f32)
        self.dones = torch.empty((T, N), dtype=torch.uint8, device=device)   # done flag ENTERING step t
        self.pd_kind = pd_kind


class RolloutField(object):
    """Env-major view of a time-major device field: `field[mbinds]` == the reference's
    `sf01(arr)[mbinds]` (runner.py:69-74 + ppo2.py:162), computed by the gather kernel.
    It ALIASES the runner's resident rollout buffer: the next Runner.run() overwrites what it shows."""

    def __init__(self, tensor, T, N, host_dtype):
        self.tensor, self.T, self.N, self.host_dtype = tensor, T, N, host_dtype

    def __len__(self):
        return self.T * self.N

    @property
    def shape(self):
        return (self.T * self.N,) + tuple(self.tensor.shape[2:])

    def __getitem__(self, idx):
        if not isinstance(idx, torch.Tensor):
            idx = torch.from_numpy(np.ascontiguousarray(np.asarray(idx, dtype=np.int64)))
        idx = idx.to(self.tensor.device, torch.int64)
        return ops.gather_rows(self.tensor, idx, self.T, self.N)

    def to_numpy(self):
        return ops.sf01(self.tensor).cpu().numpy().astype(self.host_dtype, copy=False)

    def __array__(self, dtype=None, copy=None):
        a = self.to_numpy()
        return a.astype(dtype) if dtype is not None else a


class Runner(AbstractEnvRunner):
    def __init__(self, *, env, model, nsteps, gamma, lam, return_host=None):
        self._tstart = time.time()
        super().__init__(env=env, model=model, nsteps=nsteps)
        self.lam = lam
        self.gamma = gamma
        self.device = getattr(model, 'device', torch.device('cuda'))
        ob_space = env.observation_space
        ob_np = np.dtype(ob_space.dtype)
        ob_t = torch.uint8 if ob_np in (np.dtype(np.uint8), np.dtype(np.int8)) else torch.float32
        # Discrete observations live in the HBM rollout in their encoded (one-hot float32) form; `run()` hands that form
        # back (the reference returns the raw integers -- its own train() re-encodes them; ours accepts both)
        self._onehot = int(getattr(getattr(model, 'policy', None), 'ob_onehot', 0) or 0)
        ob_shape = (self._onehot,) if self._onehot else ob_space.shape
        if self._onehot:
            ob_np, ob_t = np.dtype(np.float32), torch.float32
        pd_kind = getattr(model, 'pd_kind', None)
        if pd_kind is None:
            pd_kind = {'Discrete': 'categorical', 'MultiDiscrete': 'multicategorical',
                       'MultiBinary': 'bernoulli'}.get(type(env.action_space).__name__, 'gaussian')
        nact = getattr(model, 'nact', None) or (env.action_space.n if pd_kind in ('categorical', 'bernoulli')
                                                else int(np.sum(env.action_space.nvec)) if pd_kind == 'multicategorical'
                                                else env.action_space.shape[0])
        nsub = len(np.asarray(env.action_space.nvec).reshape(-1)) if pd_kind == 'multicategorical' else None
        self.rollout = Rollout(nsteps, self.nenv, ob_shape, ob_t, pd_kind, nact, self.device, nsub=nsub)
        self.return_host = (not self.device_env) if return_host is None else return_host
        self.fast_step = hasattr(model, 'step_into')
        self._dones_dev = torch.zeros(self.nenv, dtype=torch.uint8, device=self.device)
        # host-env bridge only for an env whose own class provides it (an unwrapped ShmemVecEnv: its observations already
        # sit in a staging slot); a wrapper stacked on top must see -- and may transform -- the observations itself
        self._bridge = _defines(env, 'obs_to_device') and not self._onehot
        # zero-copy device stepping likewise only when the env's own class writes into caller-provided buffers
        self._env_step_into = _defines(env, 'step_into')
        self._ob_np = ob_np
        self._graph, self._graph_out, self._eager_rollouts = None, None, 0
        # recurrent policies (runner.py:23,28): the LSTM state travels with the env cursor; on the device for our Model
        self.recurrent = self.states is not None
        self._states_dev = None
        if self.recurrent and self.fast_step:
            self._states_dev = torch.from_numpy(np.ascontiguousarray(np.asarray(self.states, dtype=np.float32))).to(self.device)
        self._mb_states = None
        self._ob_clip = getattr(getattr(model, 'policy', None), 'ob_clip', None)

    # ------------------------------------------------------------------
    def _rollout_steps(self, ro):
        """the T act + env steps of one rollout, all on the device, written straight into the HBM rollout
        -> (last_values [N], fin_r [T, N], fin_l [T, N])"""
        T = self.nsteps
        fin_r, fin_l = [], []
        ro.obs[0].copy_(self.obs)                      # cursor -> slot 0
        ro.dones[0].copy_(self._dones_dev)             # ... the dones that entered the rollout likewise; the env then writes the
        nxt_last = self.obs                            # dones of step t straight into slot t + 1 (no copy per step)
        if self.recurrent:
            self._mb_states = self._states_dev.clone()     # runner.py:23 mb_states = self.states (state BEFORE the rollout)
        # the stock Model's sampling noise for all T steps in ONE generator call (its per-step launch was a tenth of an MLP env step;
        # same distribution, the Philox stream is consumed in one piece instead of T); wrapped / subclassed models keep their own
        from .model import Model
        noise_all = None
        if (isinstance(self.model, Model) and not self.recurrent and type(self.model).step_into is Model.step_into
                and type(self.model).make_noise is Model.make_noise):       # (MicrobatchedModel etc.: the same act side)
            noise_all = self.model.make_noise(T * self.nenv).reshape(T, self.nenv, -1)
        for t in range(T):
            if self._ob_clip:                          # normalize_observations: the rollout holds the clipped observations
                ro.obs[t].clamp_(-self._ob_clip, self._ob_clip)
            if self.recurrent:                         # runner.py:28 step(obs, S=self.states, M=self.dones)
                self.model.step_into(ro.obs[t], ro.actions[t], ro.values[t], ro.neglogpacs[t], states=self._states_dev,
                                     masks=ro.dones[t])
            elif noise_all is not None:
                self.model.step_into(ro.obs[t], ro.actions[t], ro.values[t], ro.neglogpacs[t], noise=noise_all[t])
            else:
                self.model.step_into(ro.obs[t], ro.actions[t], ro.values[t], ro.neglogpacs[t])
            obs_out = ro.obs[t + 1] if t + 1 < T else nxt_last
            done_out = ro.dones[t + 1] if t + 1 < T else self._dones_dev
            if self._env_step_into:
                _, _, _, info = self.env.step_into(ro.actions[t], obs_out=obs_out, rew_out=ro.rewards[t],
                                                   done_out=done_out)
            else:
                # device-resident env behind wrappers (VecNormalize, VecFrameStack, ...): the wrapper chain runs its own
                # kernels in step_wait(); its results are copied (device to device) into the rollout slot
                o, r, d, info = self.env.step(ro.actions[t])
                obs_out.copy_(o.reshape(obs_out.shape))
                ro.rewards[t].copy_(r)
                done_out.copy_(d.view(torch.uint8) if d.dtype == torch.bool else d)
                if not isinstance(info, dict) or 'fin_r' not in info:
                    info = {'fin_r': torch.zeros_like(ro.rewards[t]), 'fin_l': torch.zeros_like(ro.actions[t], dtype=torch.int32).reshape(-1)[:self.nenv]}
            fin_r.append(info['fin_r'])
            fin_l.append(info['fin_l'])
        self.obs = nxt_last
        if self.recurrent:                             # runner.py:50 value(obs, S=self.states, M=self.dones)
            return self.model.value_dev(self.obs, self._states_dev, self._dones_dev), torch.stack(fin_r), torch.stack(fin_l)
        return self.model.value_dev(self.obs), torch.stack(fin_r), torch.stack(fin_l)

    def _graphable(self):
        """A rollout is one static launch sequence when both sides are ours: the stock Model (its noise comes from a
        torch generator that can be registered with the graph) and the stock device env (its state lives in device
        arrays).  Wrappers, subclasses, teacher-forced test models and multi-rank runs (other threads of the
        process talk to the runtime while we would be capturing) keep the eager loop."""
        from .model import Model
        from ..common.vec_env.synthetic_vec_env import SyntheticVecEnv
        return (type(self.model) is Model and not self.model.multi and not self.recurrent
                and type(self.env) is SyntheticVecEnv and not _lib.prof_enabled()
                and os.environ.get('MRL_ROLLOUT_GRAPH', '1') != '0' and self._graph is not False)

    def _run_device_env(self, ro):
        """The whole rollout (T x ~12 small launches) is captured once as a hipGraph and replayed per update: every
        buffer it touches is static (rollout SoA, observation cursor, env state, parameters updated in place)."""
        if self._graphable():
            if self._graph is None and self._eager_rollouts >= 1:        # capture after one eager warm-up rollout
                try:
                    g = torch.cuda.CUDAGraph()
                    g.register_generator_state(self.model._gen)
                    torch.cuda.synchronize()
                    with _lib.capture_graph(g):
                        out = self._rollout_steps(ro)
                    self._graph, self._graph_out = g, out
                except Exception as exc:                                    # capture unsupported here: stay eager
                    import warnings
                    warnings.warn('rollout graph capture failed (%s); using the step loop' % (exc,))
                    self._graph = False
                    torch.cuda.synchronize()
                    return self._run_device_env(ro)
            if self._graph:
                self._graph.replay()
                last_values, fr, fl = self._graph_out
                return last_values, self._epinfos(fr, fl)
        self._eager_rollouts += 1
        last_values, fr, fl = self._rollout_steps(ro)
        return last_values, self._epinfos(fr, fl)

    def _epinfos(self, fr, fl):
        """episode records of a device rollout in the shape of bench/monitor.py:58-77 ({'r', 'l', 't'}).  The device env
        reports finished episodes as arrays and the host looks at them once per rollout, so 't' (seconds since the
        runner was built) is the time the rollout ended, not the step the episode ended in."""
        mask = fl > 0
        if not bool(mask.any()):                       # one host sync per rollout, not per step
            return []
        rs = fr[mask].cpu().numpy()
        ls = fl[mask].cpu().numpy()
        t = round(time.time() - self._tstart, 6)
        return [{'r': float(r), 'l': int(l), 't': t} for r, l in zip(rs, ls)]

    def _run_host_env(self, ro):
        T = self.nsteps
        epinfos = []
        rewards_host = np.zeros((T, self.nenv), np.float32)
        dones_host = np.zeros((T, self.nenv), np.bool_)
        if self.recurrent:
            self._mb_states = self._states_dev.clone() if self._states_dev is not None else np.array(self.states, copy=True)
        for t in range(T):
            if self._onehot:
                ro.obs[t].copy_(self.model._to_dev_obs(self.obs))
            elif self._bridge:
                # host-env bridge (ShmemVecEnv): asynchronous DMA from the page-locked staging slot the workers wrote
                self.env.obs_to_device(ro.obs[t])
            else:
                obs_np = self.obs.view(np.uint8) if self.obs.dtype == np.int8 else self.obs
                ro.obs[t].copy_(torch.from_numpy(obs_np))      # runner.py:30 snapshot, straight into HBM
            if self._ob_clip:
                ro.obs[t].clamp_(-self._ob_clip, self._ob_clip)
            if self.fast_step:
                if self.recurrent:
                    m8 = torch.from_numpy(np.ascontiguousarray(np.asarray(self.dones, np.bool_)).view(np.uint8)).to(self.device)
                    self.model.step_into(ro.obs[t], ro.actions[t], ro.values[t], ro.neglogpacs[t], states=self._states_dev,
                                         masks=m8)
                else:
                    self.model.step_into(ro.obs[t], ro.actions[t], ro.values[t], ro.neglogpacs[t])
                actions = ro.actions[t].cpu().numpy()
                if ro.pd_kind == 'categorical':
                    actions = actions.astype(np.int64)
                elif ro.pd_kind == 'bernoulli':
                    actions = actions.astype(np.float32)
            else:
                actions, values, self.states, neglogpacs = self.model.step(self.obs, S=self.states, M=self.dones)
                ro.actions[t].copy_(torch.from_numpy(np.asarray(actions)).to(ro.actions.dtype))
                ro.values[t].copy_(torch.from_numpy(np.asarray(values, np.float32)))
                ro.neglogpacs[t].copy_(torch.from_numpy(np.asarray(neglogpacs, np.float32)))
            dones_host[t] = self.dones
            self.obs[:], rewards, self.dones, infos = self.env.step(actions)
            for info in infos:
                maybeepinfo = info.get('episode')
                if maybeepinfo:
                    epinfos.append(maybeepinfo)
            rewards_host[t] = rewards
        ro.rewards.copy_(torch.from_numpy(rewards_host))
        ro.dones.copy_(torch.from_numpy(dones_host.view(np.uint8)))
        self._dones_dev.copy_(torch.from_numpy(np.asarray(self.dones, np.bool_).view(np.uint8)))
        if self.recurrent and self._states_dev is not None:
            last_values = self.model.value_dev(self.model._to_dev_obs(self.obs), self._states_dev, self._dones_dev)
        elif hasattr(self.model, 'value_dev') and self._onehot:
            last_values = self.model.value_dev(self.model._to_dev_obs(self.obs))
        elif hasattr(self.model, 'value_dev'):
            obs_np = self.obs.view(np.uint8) if self.obs.dtype == np.int8 else self.obs
            last_values = self.model.value_dev(torch.from_numpy(obs_np).to(self.device))
        else:
            last_values = torch.from_numpy(np.asarray(self.model.value(self.obs, S=self.states, M=self.dones),
                                                      np.float32)).to(self.device)
        return last_values, epinfos

    def run(self):
        """runner.py:20-67.  With return_host=True the eight results are fresh host arrays like the reference's.  With
        return_host=False (the device fast path learn() uses) the fields are `RolloutField` VIEWS of the ONE resident HBM
        rollout -- obs, returns, dones, actions, values, neglogpacs are overwritten in place by the next run(); a caller that
        wants to keep a rollout across run() calls must copy it first (`field.to_numpy()` or `field.tensor.clone()`)."""
        ro = self.rollout
        if self.device_env and self.fast_step:
            last_values, epinfos = self._run_device_env(ro)
        else:
            last_values, epinfos = self._run_host_env(ro)
        # GAE(lambda) + returns: one HIP kernel, bit-exact vs runner.py:52-65
        ops.gae(ro.rewards, ro.values, ro.dones, last_values, self._dones_dev, self.gamma, self.lam, out=ro.returns)
        T, N = ro.T, ro.N
        act_dtype = {'categorical': np.int64, 'multicategorical': np.int32}.get(ro.pd_kind, np.float32)
        fields = (RolloutField(ro.obs, T, N, self._ob_np), RolloutField(ro.returns, T, N, np.float32),
                  RolloutField(ro.dones, T, N, np.bool_), RolloutField(ro.actions, T, N, act_dtype),
                  RolloutField(ro.values, T, N, np.float32), RolloutField(ro.neglogpacs, T, N, np.float32))
        if self.return_host:
            fields = tuple(f.to_numpy() for f in fields)
        states = self._mb_states if self.recurrent else self.states
        if self.recurrent and self._states_dev is not None:
            self.states = self._states_dev                 # the cursor's live state (device)
            if self.return_host:
                states = states.cpu().numpy()
        return (*fields, states, epinfos)


def sf01(arr):
    """swap and then flatten axes 0 and 1 (runner.py:69-74); host helper kept for API parity."""
    s = arr.shape
    return arr.swapaxes(0, 1).reshape(s[0] * s[1], *s[2:])
