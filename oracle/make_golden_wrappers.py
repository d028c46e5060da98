"""
TEST INFRASTRUCTURE ONLY.  Golden vectors for the actor-side wrappers, produced by running the
REFERENCE's own classes (imported from /root/reference, never copied):
  * baselines.common.vec_env.vec_frame_stack.VecFrameStack   (np.roll shift, zero on done, newest frame last)
  * baselines.common.vec_env.vec_normalize.VecNormalize + running_mean_std.RunningMeanStd (use_tf=False)
  * baselines.common.vec_env.vec_monitor.VecMonitor + bench.monitor.ResultsWriter (episode records, monitor.csv)
`gym` and `tensorflow` are not installed here: minimal stub modules are injected (only attribute containers --
no arithmetic of the wrappers lives in them).   Run:  python oracle/make_golden_wrappers.py
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('BASELINES_REFERENCE', '/root/reference')
OUT = os.path.join(ROOT, 'tests', 'golden')


def _stubs():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    gym, spaces, core = mod('gym'), mod('gym.spaces'), mod('gym.core')
    gym.__path__ = []

    class Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high = np.asarray(low), np.asarray(high)
            self.shape = self.low.shape if shape is None else tuple(shape)
            self.dtype = np.dtype(dtype)

    class Wrapper(object):
        def __init__(self, env):
            self.env = env

    spaces.Box, spaces.Dict, spaces.Tuple = Box, dict, tuple
    spaces.Discrete = type('Discrete', (), {})
    gym.spaces, gym.core = spaces, core
    gym.Env = core.Env = type('Env', (), {})
    gym.Wrapper = core.Wrapper = gym.ObservationWrapper = gym.RewardWrapper = gym.ActionWrapper = Wrapper

    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith('__'):
                raise AttributeError(k)
            return _Any(self.__name__ + '.' + k)

        def __call__(self, *a, **kw):
            return _Any(self.__name__ + '()')

    for name in ('tensorflow', 'tensorflow.python', 'tensorflow.python.client', 'tensorflow.core', 'tensorflow.core.util'):
        sys.modules[name] = _Any(name)
    return Box


class FakeVenv(object):
    """Replays pre-generated (obs, rews, dones) batches through the VecEnv interface."""

    def __init__(self, space, obs_seq, rew_seq, done_seq):
        self.num_envs = obs_seq.shape[1]
        self.observation_space, self.action_space = space, None
        self.obs_seq, self.rew_seq, self.done_seq = obs_seq, rew_seq, done_seq
        self.t = 0

    def reset(self):
        self.t = 0
        return self.obs_seq[0].copy()

    def step_async(self, actions):
        pass

    def step_wait(self):
        self.t += 1
        return self.obs_seq[self.t].copy(), self.rew_seq[self.t].copy(), self.done_seq[self.t].copy(), [{}] * self.num_envs


def main():
    Box = _stubs()
    sys.path.insert(0, REF)
    from baselines.common.vec_env.vec_frame_stack import VecFrameStack
    from baselines.common.vec_env.vec_normalize import VecNormalize
    out = {}
    # ---- VecFrameStack: Atari-shaped single frames (84, 84, 1) uint8, nstack = 4, small map variant too
    for tag, (N, H, W, C, nstack, steps) in {'fs_atari': (2, 84, 84, 1, 4, 5), 'fs_small': (3, 6, 5, 2, 3, 7)}.items():
        rng = np.random.RandomState(len(tag))
        obs = rng.randint(0, 256, (steps + 1, N, H, W, C)).astype(np.uint8)
        dones = rng.rand(steps + 1, N) < 0.3
        rews = rng.randn(steps + 1, N).astype(np.float32)
        space = Box(low=np.zeros((H, W, C), np.uint8), high=np.full((H, W, C), 255, np.uint8), dtype=np.uint8)
        fs = VecFrameStack(FakeVenv(space, obs, rews, dones), nstack)
        outs = [fs.reset().copy()]
        for _ in range(steps):
            o, _, _, _ = fs.step_wait()
            outs.append(o.copy())
        out[tag + '_in_obs'], out[tag + '_in_dones'], out[tag + '_nstack'] = obs, dones, nstack
        out[tag + '_out'] = np.stack(outs)
    # ---- VecNormalize: MuJoCo-shaped float32 observations, default clip/gamma/epsilon
    for tag, (N, D, steps) in {'vn_mujoco': (16, 376, 8), 'vn_small': (5, 3, 9)}.items():
        rng = np.random.RandomState(len(tag) + 7)
        obs = (rng.randn(steps + 1, N, D) * 3 + 1).astype(np.float32)
        rews = (rng.randn(steps + 1, N) * 2).astype(np.float32)
        dones = rng.rand(steps + 1, N) < 0.15
        space = Box(low=-np.inf * np.ones(D, np.float32), high=np.inf * np.ones(D, np.float32), dtype=np.float32)
        vn = VecNormalize(FakeVenv(space, obs, rews, dones))
        o_list, r_list = [vn.reset().copy()], []
        for _ in range(steps):
            o, r, _, _ = vn.step_wait()
            o_list.append(np.asarray(o).copy())
            r_list.append(np.asarray(r).copy())
        out[tag + '_in_obs'], out[tag + '_in_rews'], out[tag + '_in_dones'] = obs, rews, dones
        out[tag + '_out_obs'], out[tag + '_out_rews'] = np.stack(o_list), np.stack(r_list)
        out[tag + '_ob_mean'], out[tag + '_ob_var'], out[tag + '_ob_count'] = vn.ob_rms.mean, vn.ob_rms.var, vn.ob_rms.count
        out[tag + '_ret_mean'], out[tag + '_ret_var'], out[tag + '_ret_count'] = vn.ret_rms.mean, vn.ret_rms.var, vn.ret_rms.count
        out[tag + '_ret'] = vn.ret
    # ---- VecMonitor (+ its monitor.csv writer): episode records and the csv text, wall-clock column masked
    import tempfile
    from baselines.common.vec_env.vec_monitor import VecMonitor
    rng = np.random.RandomState(11)
    N, steps = 5, 40
    rews = (rng.randn(steps + 1, N) * 2).astype(np.float32)
    dones = rng.rand(steps + 1, N) < 0.2
    obs = rng.randn(steps + 1, N, 3).astype(np.float32)
    space = Box(low=-np.ones(3, np.float32), high=np.ones(3, np.float32), dtype=np.float32)
    with tempfile.TemporaryDirectory() as tmp:
        vm = VecMonitor(FakeVenv(space, obs, rews, dones), filename=os.path.join(tmp, 'run'), keep_buf=7)
        vm.reset()
        recs = []
        for t in range(1, steps + 1):
            _, _, d, infos = vm.step_wait()
            for e in range(N):
                if d[e]:
                    recs.append((t, e, infos[e]['episode']['r'], infos[e]['episode']['l']))
                else:
                    assert 'episode' not in infos[e]
        vm.results_writer.f.close()
        text = open(os.path.join(tmp, 'run.monitor.csv')).read().splitlines()
    out['vm_in_rews'], out['vm_in_dones'], out['vm_in_obs'] = rews, dones, obs
    out['vm_records'] = np.array(recs, dtype=np.float64)             # (step, env, r as f32 value, l)
    out['vm_csv_rl'] = np.array([ln.rsplit(',', 1)[0] for ln in text[2:]])     # "r,l" of every row, 't' dropped
    out['vm_csv_header'] = np.array(text[1])
    out['vm_keep_r'], out['vm_keep_l'] = np.array(vm.epret_buf, np.float64), np.array(vm.eplen_buf, np.int64)
    np.savez_compressed(os.path.join(OUT, 'wrappers.npz'), **out)
    print('written', os.path.join(OUT, 'wrappers.npz'), {k: (v.shape, v.dtype) for k, v in out.items() if hasattr(v, 'shape')})


if __name__ == '__main__':
    main()
