"""
TEST INFRASTRUCTURE ONLY -- CPU restatement of the DQN replay slice.

Array-based restatement (NumPy float64 arrays instead of Python lists) of
  * baselines/common/segment_tree.py:4-145   SegmentTree / SumSegmentTree / MinSegmentTree
  * baselines/deepq/replay_buffer.py:7-191   ReplayBuffer / PrioritizedReplayBuffer
  * baselines/deepq/build_graph.py:396-413 + baselines/common/tf_util.py:39-45   double-Q TD target,
    Huber loss, importance-weighted mean (the TF part: float32, "parity unpinned" at the TF boundary)
  * baselines/common/schedules.py:76-99      LinearSchedule

Pinned by tests/test_oracle_golden.py against tests/golden/{replay,segment_tree,misc}.npz, which
oracle/make_golden.py produced by running the reference's own classes in the build container.
All priority arithmetic is float64 with Python-float semantics (`**` == libm pow), like the reference.
"""
import numpy as np


class SegmentTree(object):
    """Heap-ordered complete binary tree over `capacity` leaves (power of two): node 1 is the root,
    leaves live at [capacity, 2*capacity).  segment_tree.py:4-86."""

    def __init__(self, capacity, op, neutral):
        assert capacity > 0 and capacity & (capacity - 1) == 0
        self.capacity = capacity
        self.value = np.full(2 * capacity, neutral, dtype=np.float64)
        self.op = op

    def __setitem__(self, idx, val):
        # segment_tree.py:76-86: write the leaf, then refresh every ancestor from its two children
        node = idx + self.capacity
        self.value[node] = val
        node //= 2
        while node >= 1:
            self.value[node] = self.op(self.value[2 * node], self.value[2 * node + 1])
            node //= 2

    def __getitem__(self, idx):
        assert 0 <= idx < self.capacity
        return float(self.value[self.capacity + idx])

    def reduce(self, start=0, end=None):
        """op over leaves [start, end) with the reference's association order (segment_tree.py:36-74):
        the recursion splits at node midpoints and combines left-part OP right-part."""
        if end is None:
            end = self.capacity
        if end < 0:
            end += self.capacity
        end -= 1

        def rec(start, end, node, lo, hi):
            if start == lo and end == hi:
                return float(self.value[node])
            mid = (lo + hi) // 2
            if end <= mid:
                return rec(start, end, 2 * node, lo, mid)
            if mid + 1 <= start:
                return rec(start, end, 2 * node + 1, mid + 1, hi)
            return self.op(rec(start, mid, 2 * node, lo, mid), rec(mid + 1, end, 2 * node + 1, mid + 1, hi))

        return rec(start, end, 1, 0, self.capacity - 1)


class SumSegmentTree(SegmentTree):
    def __init__(self, capacity):
        SegmentTree.__init__(self, capacity, lambda a, b: float(a) + float(b), 0.0)

    def sum(self, start=0, end=None):
        return self.reduce(start, end)

    def find_prefixsum_idx(self, prefixsum):
        """segment_tree.py:105-131: descend from the root, left if value[2i] > prefixsum, else subtract
        the left mass and go right."""
        node = 1
        prefixsum = float(prefixsum)
        while node < self.capacity:
            left = float(self.value[2 * node])
            if left > prefixsum:
                node = 2 * node
            else:
                prefixsum -= left
                node = 2 * node + 1
        return node - self.capacity


class MinSegmentTree(SegmentTree):
    def __init__(self, capacity):
        SegmentTree.__init__(self, capacity, lambda a, b: min(float(a), float(b)), float('inf'))

    def min(self, start=0, end=None):
        return self.reduce(start, end)


class ReplayBuffer(object):
    """Ring of transitions (replay_buffer.py:7-68) held as preallocated arrays."""

    def __init__(self, size, ob_shape=None, ob_dtype=np.uint8):
        self.maxsize = int(size)
        self.next_idx = 0
        self.length = 0
        self.ob_shape, self.ob_dtype = ob_shape, ob_dtype
        self.obs_t = self.obs_tp1 = self.act = self.rew = self.done = None

    def __len__(self):
        return self.length

    def _alloc(self, obs_t, action):
        shape = tuple(np.shape(obs_t))
        self.obs_t = np.zeros((self.maxsize,) + shape, np.asarray(obs_t).dtype)
        self.obs_tp1 = np.zeros_like(self.obs_t)
        self.act = np.zeros((self.maxsize,) + tuple(np.shape(action)), np.asarray(action).dtype)
        self.rew = np.zeros(self.maxsize, np.float64)
        self.done = np.zeros(self.maxsize, np.float64)

    def add(self, obs_t, action, reward, obs_tp1, done):
        # replay_buffer.py:24-31
        if self.obs_t is None:
            self._alloc(obs_t, action)
        i = self.next_idx
        self.obs_t[i], self.act[i], self.rew[i], self.obs_tp1[i], self.done[i] = obs_t, action, reward, obs_tp1, done
        self.length = max(self.length, i + 1)
        self.next_idx = (i + 1) % self.maxsize

    def encode_sample(self, idxes):
        idxes = np.asarray(idxes, dtype=np.int64)
        return self.obs_t[idxes], self.act[idxes], self.rew[idxes], self.obs_tp1[idxes], self.done[idxes]


class PrioritizedReplayBuffer(ReplayBuffer):
    def __init__(self, size, alpha, **kw):
        ReplayBuffer.__init__(self, size, **kw)
        assert alpha >= 0
        self.alpha = alpha
        cap = 1
        while cap < size:                     # replay_buffer.py:92-94
            cap *= 2
        self.it_sum = SumSegmentTree(cap)
        self.it_min = MinSegmentTree(cap)
        self.max_priority = 1.0

    def add(self, *args):
        i = self.next_idx
        ReplayBuffer.add(self, *args)
        leaf = self.max_priority ** self.alpha        # replay_buffer.py:100-105
        self.it_sum[i] = leaf
        self.it_min[i] = leaf

    def sample_proportional(self, batch_size, uniforms):
        """replay_buffer.py:107-115 with the `random.random()` draws passed in.  NOTE the quirk:
        sum(0, len-1) excludes the newest element's mass (segment_tree.py:69-74 decrements `end`)."""
        p_total = self.it_sum.sum(0, self.length - 1)
        every = p_total / batch_size
        return [self.it_sum.find_prefixsum_idx(float(u) * every + i * every) for i, u in enumerate(uniforms)]

    def sample(self, batch_size, beta, uniforms):
        assert beta > 0
        idxes = self.sample_proportional(batch_size, uniforms)
        total = self.it_sum.sum()
        p_min = self.it_min.min() / total                                   # replay_buffer.py:157-158
        max_weight = (p_min * self.length) ** (-beta)
        weights = np.array([((self.it_sum[i] / total) * self.length) ** (-beta) / max_weight for i in idxes])
        return tuple(list(self.encode_sample(idxes)) + [weights, idxes])

    def update_priorities(self, idxes, priorities):
        # replay_buffer.py:169-191 (sequential: with duplicate indices the LAST one wins)
        for i, p in zip(idxes, priorities):
            p = float(p)
            assert p > 0 and 0 <= i < self.length
            leaf = p ** self.alpha
            self.it_sum[int(i)] = leaf
            self.it_min[int(i)] = leaf
            self.max_priority = max(self.max_priority, p)


def linear_schedule(t, schedule_timesteps, final_p, initial_p=1.0):
    """schedules.py:76-99"""
    fraction = min(float(t) / schedule_timesteps, 1.0)
    return initial_p + fraction * (final_p - initial_p)


def huber_loss(x, delta=1.0):
    """tf_util.py:39-45 (float32)"""
    x = np.asarray(x, np.float32)
    d = np.float32(delta)
    return np.where(np.abs(x) < d, np.float32(0.5) * x * x, d * (np.abs(x) - np.float32(0.5) * d)).astype(np.float32)


def dqn_td(q_t, q_tp1_target, q_tp1_online, actions, rewards, dones, weights, gamma, double_q=True):
    """build_graph.py:396-413 in float32.  Returns (td_error [B], weighted_error scalar,
    d weighted_error / d q_t [B, nA])."""
    q_t = np.asarray(q_t, np.float32)
    B, nA = q_t.shape
    ar = np.arange(B)
    q_sel = q_t[ar, actions]
    if double_q:
        best = np.argmax(np.asarray(q_tp1_online, np.float32), axis=1)       # first max wins (tf.argmax)
        q_best = np.asarray(q_tp1_target, np.float32)[ar, best]
    else:
        q_best = np.asarray(q_tp1_target, np.float32).max(axis=1)
    one = np.float32(1.0)
    q_best_masked = (one - np.asarray(dones, np.float32)) * q_best
    target = np.asarray(rewards, np.float32) + np.float32(gamma) * q_best_masked
    td = (q_sel - target).astype(np.float32)
    w = np.asarray(weights, np.float32)
    err = huber_loss(td)
    weighted = np.float32((w * err).mean(dtype=np.float32))
    dtd = np.where(np.abs(td) < one, td, np.sign(td)).astype(np.float32)      # Huber'(td), delta = 1
    dq = np.zeros_like(q_t)
    dq[ar, actions] = w * dtd / np.float32(B)
    return td, weighted, dq


# ---------------------------------------------------------------------------------------------
# actor-side wrappers: CPU restatement of
#   baselines/common/vec_env/vec_frame_stack.py:6-30, vec_normalize.py:4-47,
#   baselines/common/running_mean_std.py:5-33
# pinned by tests/golden/wrappers.npz (reference classes run verbatim, oracle/make_golden_wrappers.py)
# ---------------------------------------------------------------------------------------------
def framestack_reset(obs, nstack):
    """vec_frame_stack.py:26-30"""
    stacked = np.zeros(obs.shape[:-1] + (obs.shape[-1] * nstack,), obs.dtype)
    stacked[..., -obs.shape[-1]:] = obs
    return stacked


def framestack_step(stacked, obs, news):
    """vec_frame_stack.py:17-24.  NOTE the reference rolls the channel axis by ONE ELEMENT
    (`np.roll(..., shift=-1, axis=-1)`), not by one frame: identical for single-channel frames (the Atari
    case), but for C > 1 the history slides by a single channel per step.  Restated as is."""
    c = obs.shape[-1]
    out = np.empty_like(stacked)
    out[..., :-1] = stacked[..., 1:]
    out[..., -1] = stacked[..., 0]           # wrapped element (overwritten below since c >= 1)
    out[np.asarray(news, bool)] = 0
    out[..., -c:] = obs
    return out


class RunningMeanStdOracle(object):
    """running_mean_std.py:5-33 (float64 state, count starts at 1e-4, var at 1)"""

    def __init__(self, shape=(), epsilon=1e-4):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = epsilon

    def update(self, x):
        bm, bv, bc = np.mean(x, axis=0), np.var(x, axis=0), x.shape[0]
        delta = bm - self.mean
        tot = self.count + bc
        new_mean = self.mean + delta * bc / tot
        m2 = self.var * self.count + bv * bc + np.square(delta) * self.count * bc / tot
        self.mean, self.var, self.count = new_mean, m2 / tot, tot


class VecNormalizeOracle(object):
    """vec_normalize.py:4-47 with ob=True, ret=True, use_tf=False"""

    def __init__(self, num_envs, ob_shape, clipob=10., cliprew=10., gamma=0.99, epsilon=1e-8):
        self.ob_rms, self.ret_rms = RunningMeanStdOracle(ob_shape), RunningMeanStdOracle(())
        self.clipob, self.cliprew, self.gamma, self.epsilon = clipob, cliprew, gamma, epsilon
        self.ret = np.zeros(num_envs)

    def obfilt(self, obs):
        self.ob_rms.update(obs)
        return np.clip((obs - self.ob_rms.mean) / np.sqrt(self.ob_rms.var + self.epsilon), -self.clipob, self.clipob)

    def reset(self, obs):
        self.ret = np.zeros_like(self.ret)
        return self.obfilt(obs)

    def step(self, obs, rews, news):
        self.ret = self.ret * self.gamma + rews
        obs = self.obfilt(obs)
        self.ret_rms.update(self.ret)
        rews = np.clip(rews / np.sqrt(self.ret_rms.var + self.epsilon), -self.cliprew, self.cliprew)
        self.ret[np.asarray(news, bool)] = 0.
        return obs, rews
