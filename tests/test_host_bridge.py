"""Host-environment bridge: ShmemVecEnv / SubprocVecEnv against DummyVecEnv, the equivalence oracle the
reference's own tests use (common/vec_env/test_vec_env.py:14-111)."""
import numpy as np
import pytest

from baselines_amd.common.spaces import Box, Discrete
from baselines_amd.common.vec_env import DummyVecEnv, ShmemVecEnv, SubprocVecEnv
from baselines_amd.common.vec_env.vec_env import AlreadySteppingError, NotSteppingError


def make_fn(seed, shape, dtype, episode_len=7):
    """a deterministic env built inside a closure, so that cloudpickle ships it to spawned workers by value
    (same idea as the reference's SimpleEnv, test_vec_env.py:114-141)"""
    def make():
        class SimpleEnv(object):
            def __init__(self):
                rng = np.random.RandomState(seed)
                self._dtype, self._shape = np.dtype(dtype), tuple(shape)
                self._start = np.array(rng.randint(0, 100, size=shape), dtype=dtype)
                self._max, self._cur, self._steps = episode_len + seed % 3, None, 0
                hi = 255 if self._dtype == np.uint8 else 1e6
                self.observation_space = Box(low=0, high=hi, shape=self._shape, dtype=self._dtype)
                self.action_space = Discrete(5)

            def reset(self):
                self._steps, self._cur = 0, self._start.copy()
                return self._cur

            def step(self, action):
                self._cur = (self._cur + np.asarray(action).astype(self._dtype)).astype(self._dtype)
                self._steps += 1
                done = self._steps >= self._max
                info = {'episode': {'r': float(self._steps), 'l': self._steps}} if done else {'foo': 'bar%d' % self._steps}
                return self._cur, float(self._steps) / self._max, done, info

            def close(self):
                pass
        return SimpleEnv()
    return make


def assert_same_rollout(venv1, venv2, num_steps=40):
    try:
        o1, o2 = venv1.reset(), venv2.reset()
        assert o1.shape == o2.shape and o1.dtype == o2.dtype
        np.testing.assert_array_equal(o1, o2)
        rng = np.random.RandomState(0)
        for _ in range(num_steps):
            actions = rng.randint(0, 5, venv1.num_envs)
            for v in (venv1, venv2):
                v.step_async(actions)
            (o1, r1, d1, i1), (o2, r2, d2, i2) = venv1.step_wait(), venv2.step_wait()
            for a, b in ((o1, o2), (r1, r2), (d1, d2)):
                assert a.shape == b.shape and a.dtype == b.dtype
                np.testing.assert_array_equal(a, b)
            assert list(i1) == list(i2)
    finally:
        venv1.close()
        venv2.close()


@pytest.mark.parametrize('cls,kw', [(ShmemVecEnv, {}), (SubprocVecEnv, {'in_series': 3}), (ShmemVecEnv, {'ring': 1})])
@pytest.mark.parametrize('dtype,shape', [('uint8', (5, 4, 2)), ('float32', (7,))])
def test_bridge_equals_dummy(cls, kw, dtype, shape):
    fns = [make_fn(i, shape, dtype) for i in range(6)]
    assert_same_rollout(DummyVecEnv(fns), cls(fns, context='fork', pin=False, **kw))


def test_bridge_spawn_context_and_given_spaces():
    """the reference's default context; spaces passed in so no probe env is built in the learner"""
    fns = [make_fn(i, (3, 3), 'uint8') for i in range(2)]
    spaces = (Box(low=0, high=255, shape=(3, 3), dtype=np.uint8), Discrete(5))
    assert_same_rollout(DummyVecEnv(fns), ShmemVecEnv(fns, spaces=spaces, context='spawn', pin=False), num_steps=12)


def test_bridge_protocol_errors_and_staging_view():
    fns = [make_fn(i, (4,), 'float32') for i in range(4)]
    with pytest.raises(AssertionError):
        SubprocVecEnv(fns, context='fork', in_series=3, pin=False)
    venv = ShmemVecEnv(fns, context='fork', in_series=2, pin=False)
    try:
        obs0 = venv.reset()
        np.testing.assert_array_equal(venv.staging.numpy(), obs0)           # zero-copy view of the current slot
        with pytest.raises(NotSteppingError):
            venv.step_wait()
        venv.step_async([1, 2, 3, 4])
        with pytest.raises(AlreadySteppingError):
            venv.step_async([1, 2, 3, 4])
        obs1, rew, done, infos = venv.step_wait()
        assert rew.dtype == np.float32 and done.dtype == np.bool_ and len(infos) == 4
        np.testing.assert_array_equal(venv.staging.numpy(), obs1)
        obs1[:] = -1                                                        # returned arrays are the caller's own
        assert (venv.staging.numpy() >= 0).all()
        with pytest.raises(AssertionError):
            venv.step_async([1, 2])
    finally:
        venv.close()
    venv.close()                                                            # idempotent


def test_bridge_worker_failure_is_reported():
    def bad():
        class Bad(object):
            observation_space = Box(low=0, high=1, shape=(2,), dtype=np.float32)
            action_space = Discrete(2)

            def reset(self):
                return np.zeros(2, np.float32)

            def step(self, a):
                raise ValueError('simulator exploded')
        return Bad()
    venv = ShmemVecEnv([bad], context='fork', pin=False)
    try:
        venv.reset()
        venv.step_async([0])
        with pytest.raises(RuntimeError, match='simulator exploded'):
            venv.step_wait()
    finally:
        venv.close()
