"""Host-environment bridge: real gym-style environments step in worker processes and hand their
observations to the learner through ONE shared, page-locked staging block, from which the device rollout is fed by
asynchronous DMA.  API-compatible with the reference's `ShmemVecEnv(env_fns, spaces=None, context='spawn')`
(common/vec_env/shmem_vec_env.py:20-141) and `SubprocVecEnv(env_fns, spaces=None, context='spawn', in_series=1)`
(subproc_vec_env.py:36-138): lock-step `step_async` / `step_wait`, float32 rewards, bool dones, per-env info
dicts, auto-reset on done (the observation returned with done=True already belongs to the next episode).

What is different (MI355X-first):
  * a single POSIX shared-memory block holds a RING of observation slots [ring][N][ob...] plus rewards and dones;
    workers write their rows in place -- observations are never pickled and never copied on the host;
  * the learner page-locks that block once (hipHostRegister through torch's runtime binding) so that
    `obs_to_device(dst)` is a true asynchronous DMA into the HBM rollout slot: the copy of step t overlaps the
    policy forward, and the next `step_wait` writes the OTHER ring slot;
  * `in_series` environments share one worker process (fewer processes than envs for cheap simulators).
Everything else -- infos, actions -- crosses a pipe, as in the reference.
"""
import multiprocessing as mp
from multiprocessing import shared_memory

import numpy as np

from .vec_env import AlreadySteppingError, CloudpickleWrapper, NotSteppingError, VecEnv, clear_mpi_env_vars


def _layout(num_envs, ob_shape, ob_dtype, ring):
    """byte offsets of the three arrays inside the shared block (64-byte aligned)"""
    ob_bytes = int(np.prod((ring, num_envs) + tuple(ob_shape), dtype=np.int64)) * np.dtype(ob_dtype).itemsize
    up = lambda n: (n + 63) // 64 * 64
    off_rew = up(ob_bytes)
    off_done = off_rew + up(4 * num_envs)
    return off_rew, off_done, off_done + up(num_envs)


def _views(buf, num_envs, ob_shape, ob_dtype, ring):
    off_rew, off_done, _ = _layout(num_envs, ob_shape, ob_dtype, ring)
    obs = np.ndarray((ring, num_envs) + tuple(ob_shape), dtype=ob_dtype, buffer=buf, offset=0)
    rew = np.ndarray((num_envs,), dtype=np.float32, buffer=buf, offset=off_rew)
    done = np.ndarray((num_envs,), dtype=np.bool_, buffer=buf, offset=off_done)
    return obs, rew, done


def _worker(pipe, parent_pipe, env_fns_wrapped, shm_name, first, num_envs, ob_shape, ob_dtype, ring):
    """serves envs [first, first+len) of the vector: commands ('reset', slot) / ('step', slot, actions) / ('close',)"""
    parent_pipe.close()
    shm = None
    obs = rew = done = None
    envs = []
    try:
        shm = shared_memory.SharedMemory(name=shm_name)      # inside the try: a failed attach is reported, not a hang
        obs, rew, done = _views(shm.buf, num_envs, ob_shape, ob_dtype, ring)
        envs = [make() for make in env_fns_wrapped.x]
        while True:
            msg = pipe.recv()
            if msg[0] == 'reset':
                for j, env in enumerate(envs):
                    obs[msg[1], first + j] = env.reset()
                pipe.send(('ok', None))
            elif msg[0] == 'step':
                infos = []
                for j, (env, action) in enumerate(zip(envs, msg[2])):
                    ob, r, d, info = env.step(action)
                    if d:
                        ob = env.reset()
                    obs[msg[1], first + j] = ob
                    rew[first + j] = r
                    done[first + j] = d
                    infos.append(info)
                pipe.send(('ok', infos))
            elif msg[0] == 'render':
                pipe.send(('ok', [env.render(mode='rgb_array') for env in envs]))
            elif msg[0] == 'close':
                pipe.send(('ok', None))
                break
            else:
                raise RuntimeError('unknown command %r' % (msg[0],))
    except KeyboardInterrupt:
        pass
    except Exception as exc:                     # surface the failure in the learner instead of hanging it
        import traceback
        try:
            pipe.send(('error', '%s\n%s' % (exc, traceback.format_exc())))
        except Exception:
            pass
    finally:
        obs = rew = done = None                   # drop the views before the mapping goes away
        for env in envs:
            closer = getattr(env, 'close', None)
            if closer is not None:
                closer()
        if shm is not None:
            shm.close()


class ShmemVecEnv(VecEnv):
    def __init__(self, env_fns, spaces=None, context='spawn', in_series=1, ring=2, pin=None):
        """env_fns: callables creating gym-style envs.  spaces=(observation_space, action_space) skips the probe
        environment the reference also builds otherwise.  in_series: envs per worker process.  ring: observation
        slots in the staging block (>= 2 lets a DMA of the previous step overlap the next step).  pin: page-lock the
        block for asynchronous device copies (default: when a GPU is visible)."""
        n = len(env_fns)
        if n % in_series != 0:
            raise AssertionError('Number of envs must be divisible by number of envs to run in series')
        if spaces is None:
            probe = env_fns[0]()
            spaces = (probe.observation_space, probe.action_space)
            closer = getattr(probe, 'close', None)
            if closer is not None:
                closer()
            del probe
        ob_space, ac_space = spaces
        VecEnv.__init__(self, n, ob_space, ac_space)
        if not hasattr(ob_space, 'shape') or ob_space.shape is None:
            raise ValueError('ShmemVecEnv stages array observations; dict/tuple spaces are outside the hot path')
        self._ob_shape, self._ob_dtype = tuple(ob_space.shape), np.dtype(ob_space.dtype)
        self.ring = max(1, int(ring))
        self._slot = 0
        total = _layout(n, self._ob_shape, self._ob_dtype, self.ring)[2]
        self._shm = shared_memory.SharedMemory(create=True, size=total)
        self._obs, self._rew, self._done = _views(self._shm.buf, n, self._ob_shape, self._ob_dtype, self.ring)
        self._rew[:] = 0
        self._done[:] = False
        self.in_series = int(in_series)
        ctx = mp.get_context(context)
        self._pipes, self._procs = [], []
        groups = [env_fns[i:i + self.in_series] for i in range(0, n, self.in_series)]
        with clear_mpi_env_vars():
            for g, fns in enumerate(groups):
                parent, child = ctx.Pipe()
                proc = ctx.Process(target=_worker, daemon=True,
                                   args=(child, parent, CloudpickleWrapper(list(fns)), self._shm.name, g * self.in_series, n,
                                         self._ob_shape, self._ob_dtype, self.ring))
                proc.start()
                child.close()
                self._pipes.append(parent)
                self._procs.append(proc)
        self._stepping = False
        self._pinned_t = None
        self._registered = False
        if pin is None:
            pin = self._gpu_visible()
        if pin:
            self._pin()

    # ------------------------------------------------------------------ staging block <-> device
    @staticmethod
    def _gpu_visible():
        try:
            import torch
            return torch.cuda.is_available()
        except Exception:
            return False

    def _pin(self):
        """page-lock the shared block in place (hipHostRegister) and keep a torch view of the observation ring"""
        import torch
        flat = np.ndarray((self._shm.size,), dtype=np.uint8, buffer=self._shm.buf)
        t = torch.from_numpy(flat)
        rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), t.numel(), 0)
        if int(rc) != 0:
            raise RuntimeError('hipHostRegister failed with %s' % (rc,))
        self._registered = True
        self._flat_t = t
        self._pinned_t = torch.from_numpy(self._obs.view(np.uint8) if self._ob_dtype == np.int8 else self._obs)

    @property
    def staging(self):
        """torch view [N, ob...] of the current observation slot in the shared block (page-locked when `pin`)"""
        import torch
        if self._pinned_t is None:
            self._pinned_t = torch.from_numpy(self._obs.view(np.uint8) if self._ob_dtype == np.int8 else self._obs)
        return self._pinned_t[self._slot]

    def obs_to_device(self, dst, non_blocking=True):
        """DMA the current observations into the device tensor `dst` (e.g. slot t of the HBM rollout); asynchronous on
        the current stream when the block is page-locked.  The slot is not rewritten before `ring - 1` further steps."""
        dst.copy_(self.staging, non_blocking=bool(non_blocking and self._registered))
        return dst

    # ------------------------------------------------------------------ VecEnv contract
    def _collect(self):
        out = []
        for pipe in self._pipes:
            status, payload = pipe.recv()
            if status != 'ok':
                raise RuntimeError('environment worker failed: %s' % (payload,))
            out.append(payload)
        return out

    def reset(self):
        if self._stepping:
            self.step_wait()                      # like the reference: drain the pending step first
        for pipe in self._pipes:
            pipe.send(('reset', self._slot))
        self._collect()
        return self._obs[self._slot].copy()

    def step_async(self, actions):
        if self._stepping:
            raise AlreadySteppingError()
        actions = list(actions) if not isinstance(actions, np.ndarray) else actions
        if len(actions) != self.num_envs:
            raise AssertionError('%d actions for %d environments' % (len(actions), self.num_envs))
        nxt = (self._slot + 1) % self.ring
        for g, pipe in enumerate(self._pipes):
            pipe.send(('step', nxt, [actions[g * self.in_series + j] for j in range(self.in_series)]))
        self._next_slot = nxt
        self._stepping = True

    def step_wait(self):
        if not self._stepping:
            raise NotSteppingError()
        infos = [info for group in self._collect() for info in group]
        self._stepping = False
        self._slot = self._next_slot
        # fresh arrays owned by the caller, as the reference returns (np.stack / np.array copies)
        return self._obs[self._slot].copy(), self._rew.copy(), self._done.copy(), infos

    def get_images(self):
        for pipe in self._pipes:
            pipe.send(('render',))
        return [img for group in self._collect() for img in group]

    def close_extras(self):
        if self._stepping:
            try:
                self._collect()
            except Exception:
                pass
            self._stepping = False
        for pipe in self._pipes:
            try:
                pipe.send(('close',))
            except Exception:
                pass
        for pipe in self._pipes:
            try:
                pipe.recv()
            except Exception:
                pass
            pipe.close()
        for proc in self._procs:
            proc.join(timeout=5)
        if self._registered:
            import torch
            torch.cuda.cudart().cudaHostUnregister(self._flat_t.data_ptr())
            self._registered = False
        self._pinned_t = None
        self._flat_t = None
        del self._obs, self._rew, self._done
        self._shm.close()
        try:
            self._shm.unlink()
        except FileNotFoundError:
            pass


class SubprocVecEnv(ShmemVecEnv):
    """The reference's pipe-based class of this name (subproc_vec_env.py:36-138) has the same lock-step contract; here
    it is the shared-staging bridge with its `in_series` argument in the reference's position."""

    def __init__(self, env_fns, spaces=None, context='spawn', in_series=1, **kw):
        ShmemVecEnv.__init__(self, env_fns, spaces=spaces, context=context, in_series=in_series, **kw)
