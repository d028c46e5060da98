"""GPU: device VecFrameStack / VecNormalize against the golden run of the reference's own
classes (tests/golden/wrappers.npz, oracle/make_golden_wrappers.py) and the oracle restatement."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from baselines_amd.common.spaces import Box                               # noqa: E402
from baselines_amd.common.vec_env import VecEnv, VecFrameStack, VecNormalize  # noqa: E402
from oracle import replay_numpy as R                                      # noqa: E402


class ReplayDeviceVenv(VecEnv):
    """device-resident VecEnv replaying pre-generated batches"""
    device_resident = True

    def __init__(self, space, obs_seq, rew_seq, done_seq):
        VecEnv.__init__(self, obs_seq.shape[1], space, None)
        self.obs, self.rew, self.done = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (obs_seq, rew_seq, done_seq))
        self.t = 0

    def reset(self):
        self.t = 0
        return self.obs[0].clone()

    def step_async(self, actions):
        pass

    def step_wait(self):
        self.t += 1
        return self.obs[self.t].clone(), self.rew[self.t].clone(), self.done[self.t].clone(), [{}] * self.num_envs


@pytest.mark.parametrize('tag', ['fs_atari', 'fs_small'])
def test_framestack_matches_reference_run(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, 'wrappers.npz'))
    obs, dones, nstack, want = g[tag + '_in_obs'], g[tag + '_in_dones'], int(g[tag + '_nstack']), g[tag + '_out']
    shape = obs.shape[2:]
    space = Box(low=np.zeros(shape, np.uint8), high=np.full(shape, 255, np.uint8), dtype=np.uint8)
    fs = VecFrameStack(ReplayDeviceVenv(space, obs, np.zeros(obs.shape[:2], np.float32), dones), nstack)
    assert fs.observation_space.shape == want.shape[2:]
    np.testing.assert_array_equal(fs.reset().cpu().numpy(), want[0])
    for t in range(1, len(obs)):
        o, _, _, _ = fs.step_wait()
        np.testing.assert_array_equal(o.cpu().numpy(), want[t])               # bit-exact


@pytest.mark.parametrize('tag', ['vn_mujoco', 'vn_small'])
def test_vecnormalize_matches_reference_run(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, 'wrappers.npz'))
    obs, rews, dones = g[tag + '_in_obs'], g[tag + '_in_rews'], g[tag + '_in_dones']
    D = obs.shape[2]
    space = Box(low=-np.inf * np.ones(D, np.float32), high=np.inf * np.ones(D, np.float32), dtype=np.float32)
    vn = VecNormalize(ReplayDeviceVenv(space, obs, rews, dones))
    o0 = vn.reset(want_f64=True)
    np.testing.assert_array_equal(vn._last_obs64.cpu().numpy().reshape(obs[0].shape), g[tag + '_out_obs'][0])   # f64 bit-exact
    np.testing.assert_array_equal(o0.cpu().numpy(), g[tag + '_out_obs'][0].astype(np.float32))
    for t in range(1, len(obs)):
        o, r, _, _ = vn.step_wait(want_f64=True)
        np.testing.assert_array_equal(vn._last_obs64.cpu().numpy().reshape(obs[t].shape), g[tag + '_out_obs'][t])
        np.testing.assert_array_equal(o.cpu().numpy(), g[tag + '_out_obs'][t].astype(np.float32))
        # returns: float64 tree reduction vs NumPy pairwise -> 1e-12
        np.testing.assert_allclose(vn._last_rews64.cpu().numpy(), g[tag + '_out_rews'][t - 1], rtol=1e-12, atol=0)
    np.testing.assert_array_equal(vn.ob_rms.mean.cpu().numpy(), g[tag + '_ob_mean'])
    np.testing.assert_array_equal(vn.ob_rms.var.cpu().numpy(), g[tag + '_ob_var'])
    assert vn.ob_rms.count == pytest.approx(float(g[tag + '_ob_count']), rel=1e-15)
    np.testing.assert_allclose(vn.ret_rms.var, float(g[tag + '_ret_var']), rtol=1e-12)
    np.testing.assert_allclose(vn.ret.cpu().numpy(), g[tag + '_ret'], rtol=1e-15, atol=0)


def test_wrappers_full_size_properties():
    """config-sized batches: frame stack history invariants at N=512 Atari frames, normalised MuJoCo-shaped
    observations stay inside the clip range with ~zero mean / unit variance after a few batches"""
    N = 512
    rng = np.random.RandomState(0)
    obs = rng.randint(0, 256, (6, N, 84, 84, 1)).astype(np.uint8)
    dones = rng.rand(6, N) < 0.2
    space = Box(low=np.zeros((84, 84, 1), np.uint8), high=np.full((84, 84, 1), 255, np.uint8), dtype=np.uint8)
    fs = VecFrameStack(ReplayDeviceVenv(space, obs, np.zeros((6, N), np.float32), dones), 4)
    prev = fs.reset().cpu().numpy().copy()
    for t in range(1, 6):
        cur = fs.step_wait()[0].cpu().numpy()
        np.testing.assert_array_equal(cur[..., -1], obs[t][..., 0])                       # newest frame last
        keep = ~dones[t]
        np.testing.assert_array_equal(cur[keep][..., :-1], prev[keep][..., 1:])          # history slides
        assert not cur[dones[t]][..., :-1].any()                                         # finished envs restart blank
        prev = cur.copy()
    N, D = 1024, 376
    ob = (rng.randn(5, N, D) * 4 + 2).astype(np.float32)
    sp = Box(low=-np.inf * np.ones(D, np.float32), high=np.inf * np.ones(D, np.float32), dtype=np.float32)
    vn = VecNormalize(ReplayDeviceVenv(sp, ob, rng.randn(5, N).astype(np.float32), rng.rand(5, N) < 0.01))
    vn.reset()
    for _ in range(4):
        o, r, _, _ = vn.step_wait()
    o = o.cpu().numpy()
    assert np.abs(o).max() <= 10.0 and abs(o.mean()) < 0.05 and abs(o.std() - 1.0) < 0.05
    assert np.abs(r.cpu().numpy()).max() <= 10.0
