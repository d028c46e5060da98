"""Device-resident replay buffers with the reference's interface (deepq/replay_buffer.py:7-191):

    ReplayBuffer(size)                       .add(obs_t, action, reward, obs_tp1, done)  .sample(batch_size)  len()
    PrioritizedReplayBuffer(size, alpha)     .add(...)  .sample(batch_size, beta) -> (..., weights, idxes)
                                             .update_priorities(idxes, priorities)

Where the reference keeps a Python list of tuples and Python-float segment trees, these classes keep
SoA ring buffers in HBM (obs_t / obs_tp1 raw bytes, int32 actions, f32 rewards / dones -- 56.5 GB for
10^6 Pong-shaped transitions, which is what the 288 GB are for) and two float64 device trees with the
same heap layout, driven through libmrl's C ABI (mrl_replay_*, mrl_segtree_*, mrl_per_*).

Parity notes
  * the stratified uniforms come from Python's global `random.random()` exactly like
    replay_buffer.py:112, so a seeded run samples the same indices as the reference (bit-exact);
  * `add` / `update_priorities` compute `priority ** alpha` with Python floats on the host (libm pow,
    like the reference) and upload the leaf values; `update_priorities_from_td` is the device fast
    path (|td| + eps -> pow on the GPU, <= 2 ulp from libm);
  * importance weights are float64 like the reference's np.array of Python floats (device pow:
    agree to ~1e-15 relative).
Extra entry points (no host round trips): add_batch, sample_dev, update_priorities_from_td.
"""
import random

import numpy as np
import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr


def _dev_tensor(x, dtype, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x))).to(device).to(dtype).contiguous()


class ReplayBuffer(object):
    def __init__(self, size, device=None):
        _lib.require_gpu()
        self._maxsize = int(size)
        self._next_idx = 0
        self._len = 0
        self.device = torch.device(device or ('cuda:%d' % torch.cuda.current_device()))
        self._obs_t = self._obs_tp1 = self._act = self._rew = self._done = None
        self._ob_shape = self._ob_dtype = None

    def __len__(self):
        return self._len

    # ---- storage ---------------------------------------------------------------------------
    def _allocate(self, ob_shape, ob_dtype):
        self._ob_shape, self._ob_dtype = tuple(ob_shape), ob_dtype
        n = self._maxsize
        self._obs_t = torch.empty((n,) + self._ob_shape, dtype=ob_dtype, device=self.device)
        self._obs_tp1 = torch.empty((n,) + self._ob_shape, dtype=ob_dtype, device=self.device)
        self._act = torch.zeros(n, dtype=torch.int32, device=self.device)
        self._rew = torch.zeros(n, dtype=torch.float32, device=self.device)
        self._done = torch.zeros(n, dtype=torch.float32, device=self.device)
        self._ob_bytes = int(np.prod(self._ob_shape, dtype=np.int64)) * self._obs_t.element_size()

    def add_batch(self, obs_t, actions, rewards, obs_tp1, dones):
        """n transitions at once (one per env of a vectorised actor); host arrays or device tensors."""
        obs_t = obs_t if isinstance(obs_t, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(obs_t))
        obs_tp1 = obs_tp1 if isinstance(obs_tp1, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(obs_tp1))
        n = int(obs_t.shape[0])
        assert 0 < n <= self._maxsize
        if self._obs_t is None:
            self._allocate(obs_t.shape[1:], obs_t.dtype)
        o1 = obs_t.to(self.device).contiguous()
        o2 = obs_tp1.to(self.device).contiguous()
        a = _dev_tensor(actions, torch.int32, self.device)
        r = _dev_tensor(rewards, torch.float32, self.device)
        d = _dev_tensor(dones, torch.float32, self.device)
        start = self._next_idx
        check(_lib.load().mrl_replay_insert(ptr(self._obs_t), ptr(self._obs_tp1), ptr(self._act), ptr(self._rew),
                                            ptr(self._done), self._maxsize, start, n, self._ob_bytes, ptr(o1), ptr(o2),
                                            ptr(a), ptr(r), ptr(d), stream_ptr()), 'mrl_replay_insert')
        self._len = min(self._maxsize, max(self._len, start + n))
        self._next_idx = (start + n) % self._maxsize
        return start, n

    def add(self, obs_t, action, reward, obs_tp1, done):
        """replay_buffer.py:24-31"""
        self.add_batch(np.asarray(obs_t)[None], np.asarray([action]), np.asarray([reward], np.float32),
                       np.asarray(obs_tp1)[None], np.asarray([float(done)], np.float32))

    # ---- sampling --------------------------------------------------------------------------
    def _gather_dev(self, idx_dev, out=None):
        """`out`: a dict of preallocated destination tensors o1 / o2 / a / r / d (`QModel.graph_inputs(B)`: the minibatch then lands in
        the captured optimizer step's static input buffers and `train_dev` has nothing to copy)"""
        B = int(idx_dev.numel())
        if out is not None:
            o1, o2, a, r, d = out['o1'], out['o2'], out['a'], out['r'], out['d']
            assert tuple(o1.shape) == (B,) + tuple(self._ob_shape) and o1.dtype == self._ob_dtype and o1.is_contiguous() and o2.is_contiguous()
        else:
            o1 = torch.empty((B,) + self._ob_shape, dtype=self._ob_dtype, device=self.device)
            o2 = torch.empty_like(o1)
            a = torch.empty(B, dtype=torch.int32, device=self.device)
            r = torch.empty(B, dtype=torch.float32, device=self.device)
            d = torch.empty(B, dtype=torch.float32, device=self.device)
        check(_lib.load().mrl_replay_gather(ptr(self._obs_t), ptr(self._obs_tp1), ptr(self._act), ptr(self._rew),
                                            ptr(self._done), ptr(idx_dev), B, self._ob_bytes, ptr(o1), ptr(o2), ptr(a),
                                            ptr(r), ptr(d), stream_ptr()), 'mrl_replay_gather')
        return o1, a, r, o2, d

    def _encode_sample(self, idxes):
        idx_dev = torch.as_tensor(np.asarray(idxes, dtype=np.int32), device=self.device)
        o1, a, r, o2, d = self._gather_dev(idx_dev)
        # dtypes of the reference's np.array(...) over python values: actions int64, rewards/dones f64
        return (o1.cpu().numpy(), a.cpu().numpy().astype(np.int64), r.cpu().numpy().astype(np.float64),
                o2.cpu().numpy(), d.cpu().numpy().astype(np.float64))

    def sample(self, batch_size):
        """replay_buffer.py:45-68: uniform indices from Python's `random`"""
        idxes = [random.randint(0, self._len - 1) for _ in range(batch_size)]
        return self._encode_sample(idxes)

    def sample_dev(self, batch_size, out=None):
        """the same draw (same `random` stream, same indices) with the minibatch left on the device:
        -> (obs_t, actions int32, rewards f32, obs_tp1, dones f32) device tensors (`out`: see _gather_dev)"""
        idxes = [random.randint(0, self._len - 1) for _ in range(batch_size)]
        return self._gather_dev(torch.as_tensor(np.asarray(idxes, dtype=np.int32), device=self.device), out)


class PrioritizedReplayBuffer(ReplayBuffer):
    def __init__(self, size, alpha, device=None):
        super().__init__(size, device=device)
        assert alpha >= 0
        self._alpha = alpha
        cap = 1
        while cap < size:
            cap *= 2
        self._capacity = cap
        self._sum = torch.empty(2 * cap, dtype=torch.float64, device=self.device)
        self._min = torch.empty(2 * cap, dtype=torch.float64, device=self.device)
        check(_lib.load().mrl_segtree_init(ptr(self._sum), ptr(self._min), cap, stream_ptr()), 'mrl_segtree_init')
        self._max_priority = 1.0
        self._max_priority_dev = None        # used by the device fast path only

    # ---- insertion -------------------------------------------------------------------------
    def _current_max_priority(self):
        if self._max_priority_dev is not None:
            self._max_priority = max(self._max_priority, float(self._max_priority_dev.item()))
        return self._max_priority

    def add_batch(self, *args):
        start, n = super().add_batch(*args)
        if self._max_priority_dev is not None:
            # device fast path in use: the running maximum lives in device memory (update_priorities_from_td); the new
            # leaves are computed from it there -- no host read-back per `add`
            check(_lib.load().mrl_segtree_set_ring_dev(ptr(self._sum), ptr(self._min), self._capacity, start, self._maxsize,
                                                       n, ptr(self._max_priority_dev), float(self._alpha), stream_ptr()),
                  'mrl_segtree_set_ring_dev')
            return start, n
        leaf = self._max_priority ** self._alpha                      # replay_buffer.py:100-105
        check(_lib.load().mrl_segtree_set_ring(ptr(self._sum), ptr(self._min), self._capacity, start, self._maxsize, n,
                                               float(leaf), stream_ptr()), 'mrl_segtree_set_ring')
        return start, n

    # ---- sampling --------------------------------------------------------------------------
    def sample_dev(self, batch_size, beta, uniforms=None, out=None):
        """-> (obs_t, actions int32, rewards, obs_tp1, dones, weights f32, idxes int32), all device tensors
        (`out`: destination tensors o1 / o2 / a / r / d / w, see _gather_dev)"""
        assert beta > 0
        if uniforms is None:
            uniforms = [random.random() for _ in range(batch_size)]       # replay_buffer.py:112
        u = torch.as_tensor(np.asarray(uniforms, dtype=np.float64), device=self.device)
        idx = torch.empty(batch_size, dtype=torch.int32, device=self.device)
        w64 = torch.empty(batch_size, dtype=torch.float64, device=self.device)
        w32 = out['w'] if out is not None else torch.empty(batch_size, dtype=torch.float32, device=self.device)
        check(_lib.load().mrl_per_sample(ptr(self._sum), ptr(self._min), self._capacity, self._len, batch_size, ptr(u),
                                         float(beta), ptr(idx), ptr(w64), ptr(w32), stream_ptr()), 'mrl_per_sample')
        o1, a, r, o2, d = self._gather_dev(idx, out)
        self._last_w64 = w64
        return o1, a, r, o2, d, w32, idx

    def sample(self, batch_size, beta):
        """replay_buffer.py:117-167"""
        o1, a, r, o2, d, _, idx = self.sample_dev(batch_size, beta)
        idxes = [int(i) for i in idx.cpu().numpy()]
        return (o1.cpu().numpy(), a.cpu().numpy().astype(np.int64), r.cpu().numpy().astype(np.float64), o2.cpu().numpy(),
                d.cpu().numpy().astype(np.float64), self._last_w64.cpu().numpy(), idxes)

    # ---- priorities ------------------------------------------------------------------------
    def update_priorities(self, idxes, priorities):
        """replay_buffer.py:169-191 (host pow for bit parity; sequential duplicate semantics on the device)"""
        assert len(idxes) == len(priorities)
        leaves = []
        for i, p in zip(idxes, priorities):
            p = float(p)
            assert p > 0
            assert 0 <= i < self._len
            leaves.append(p ** self._alpha)
            self._max_priority = max(self._max_priority, p)
        if self._max_priority_dev is not None:                         # both paths in one run: one running maximum
            self._max_priority = max(self._max_priority, float(self._max_priority_dev.item()))
            self._max_priority_dev.fill_(self._max_priority)
        idx = torch.as_tensor(np.asarray(idxes, dtype=np.int32), device=self.device)
        leaf = torch.as_tensor(np.asarray(leaves, dtype=np.float64), device=self.device)
        check(_lib.load().mrl_segtree_set(ptr(self._sum), ptr(self._min), self._capacity, ptr(idx), ptr(leaf),
                                          len(leaves), stream_ptr()), 'mrl_segtree_set')

    def update_priorities_from_td(self, idx_dev, td_dev, eps=1e-6):
        """device fast path of `new_priorities = |td| + eps; update_priorities(...)` (deepq.py:300-303)"""
        if self._max_priority_dev is None:
            self._max_priority_dev = torch.tensor([self._max_priority], dtype=torch.float64, device=self.device)
        check(_lib.load().mrl_per_update_from_td(ptr(self._sum), ptr(self._min), self._capacity, ptr(idx_dev),
                                                 ptr(td_dev), float(eps), float(self._alpha),
                                                 ptr(self._max_priority_dev), int(idx_dev.numel()), stream_ptr()),
              'mrl_per_update_from_td')

    # test helpers
    def trees_numpy(self):
        return self._sum.cpu().numpy(), self._min.cpu().numpy()


def dqn_td_loss(q_t, q_tp1_target, q_tp1_online, actions, rewards, dones, weights, gamma, want_grad=True):
    """build_graph.py:396-413 on the device: -> (td_error [B], weighted Huber loss [1], dloss/dq_t [B, nA])"""
    _lib.require_gpu()
    B, nA = q_t.shape
    dev = q_t.device
    td = torch.empty(B, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dq = torch.empty((B, nA), dtype=torch.float32, device=dev) if want_grad else None
    lib = _lib.load()
    scratch = torch.empty(int(lib.mrl_dqn_td_scratch_bytes(B)), dtype=torch.uint8, device=dev)
    check(lib.mrl_dqn_td(ptr(q_t.contiguous()), ptr(q_tp1_target.contiguous()),
                         ptr(q_tp1_online.contiguous() if q_tp1_online is not None else None), ptr(actions), ptr(rewards),
                         ptr(dones), ptr(weights), float(gamma), B, nA, ptr(td), ptr(loss), ptr(dq), ptr(scratch),
                         stream_ptr()), 'mrl_dqn_td')
    return td, loss, dq
