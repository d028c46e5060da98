"""Minimal gym.spaces stand-ins (gym is not a dependency of the hot path: the reference only reads
`.shape`, `.dtype`, `.n` of the spaces -- common/runners.py:7-9, common/policies.py:47,127,
common/input.py:24-31, common/distributions.py:278-290).  Real gym spaces are accepted too
(duck-typed by class name)."""
import numpy as np


class Space(object):
    def __init__(self, shape, dtype):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)
        return [seed]


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.asarray(low).shape
        super().__init__(shape, dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape)
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape)

    def sample(self):
        if self.dtype.kind == 'f':
            return self._rng.uniform(self.low, self.high, self.shape).astype(self.dtype)
        return self._rng.randint(self.low, self.high.astype(np.int64) + 1, self.shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high)

    def __repr__(self):
        return 'Box%s' % (self.shape,)

    def __eq__(self, other):
        return isinstance(other, Box) and self.shape == other.shape and self.dtype == other.dtype \
            and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high)


class Discrete(Space):
    def __init__(self, n):
        super().__init__((), np.int64)
        self.n = int(n)

    def sample(self):
        return int(self._rng.randint(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n

    def __repr__(self):
        return 'Discrete(%d)' % self.n

    def __eq__(self, other):
        return isinstance(other, Discrete) and self.n == other.n


class MultiDiscrete(Space):
    """gym.spaces.MultiDiscrete: a vector of len(nvec) categorical choices (common/distributions.py:285-286)"""

    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        super().__init__(self.nvec.shape, np.int64)

    def sample(self):
        return (self._rng.random_sample(self.nvec.shape) * self.nvec).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and np.all(x >= 0) and np.all(x < self.nvec)

    def __repr__(self):
        return 'MultiDiscrete(%s)' % (self.nvec.tolist(),)

    def __eq__(self, other):
        return isinstance(other, MultiDiscrete) and np.array_equal(self.nvec, other.nvec)


class MultiBinary(Space):
    """gym.spaces.MultiBinary(n) (common/distributions.py:287-288)"""

    def __init__(self, n):
        self.n = int(n)
        super().__init__((self.n,), np.int8)

    def sample(self):
        return self._rng.randint(0, 2, self.n).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and np.all((x == 0) | (x == 1))

    def __repr__(self):
        return 'MultiBinary(%d)' % self.n

    def __eq__(self, other):
        return isinstance(other, MultiBinary) and self.n == other.n


def is_multidiscrete(space):
    return type(space).__name__ == 'MultiDiscrete'


def is_multibinary(space):
    return type(space).__name__ == 'MultiBinary'


def is_discrete(space):
    return type(space).__name__ == 'Discrete'


def is_box(space):
    return type(space).__name__ == 'Box'
