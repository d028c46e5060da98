"""GPU: the DQN replay slice through the C ABI vs the oracle (oracle/replay_numpy.py) and vs the golden
run of the reference's own PrioritizedReplayBuffer (tests/golden/replay.npz, segment_tree.npz)."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from baselines_amd import _lib                                    # noqa: E402
from baselines_amd._lib import check, ptr, stream_ptr             # noqa: E402
from oracle import replay_numpy as R                              # noqa: E402


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_golden_prioritized_replay_run(golden_dir):
    """replays the reference's op stream: identical sampled indices / transitions, weights to 1e-14,
    final trees identical"""
    from baselines_amd.deepq import PrioritizedReplayBuffer
    g = np.load(os.path.join(golden_dir, 'replay.npz'))
    buf = PrioritizedReplayBuffer(int(g['cap']), float(g['alpha']))
    batch = int(g['batch'])
    by_i = {int(g['s%d_i' % j]): j for j in range(int(g['nsamples']))}
    random.seed(123)                                   # the generator's Python-random seed
    for i in range(len(g['add_act'])):
        buf.add(g['add_obs'][i], int(g['add_act'][i]), float(g['add_rew'][i]), g['add_obs2'][i], float(g['add_done'][i]))
        if i in by_i:
            j = by_i[i]
            obs_t, act, rew, obs_tp1, done, w, idx = buf.sample(batch, float(g['s%d_beta' % j]))
            np.testing.assert_array_equal(np.asarray(idx), g['s%d_idx' % j])
            np.testing.assert_allclose(w, g['s%d_w' % j], rtol=1e-14, atol=0)
            np.testing.assert_array_equal(obs_t, g['s%d_obs_t' % j])
            np.testing.assert_array_equal(obs_tp1, g['s%d_obs_tp1' % j])
            np.testing.assert_array_equal(act, g['s%d_act' % j])
            np.testing.assert_allclose(rew, g['s%d_rew' % j], rtol=1e-7)          # f32 storage of python floats
            np.testing.assert_array_equal(done, g['s%d_done' % j])
            buf.update_priorities(idx, g['s%d_newp' % j])
    s, m = buf.trees_numpy()
    np.testing.assert_array_equal(s, g['final_sum_tree'])
    np.testing.assert_array_equal(m, g['final_min_tree'])
    assert buf._max_priority == float(g['final_max_priority'])
    assert len(buf) == int(g['cap'])


@pytest.mark.parametrize('cap,n', [(16, 200), (1024, 3000), (1 << 20, 4096)])
def test_segtree_batched_update_equals_sequential(cap, n):
    """random point updates WITH duplicates in a batch: device batch == oracle sequential, bit-exact;
    prefix-sum descent and range sums agree"""
    lib = _lib.load()
    rng = np.random.RandomState(cap % 97)
    st, mt = R.SumSegmentTree(cap), R.MinSegmentTree(cap)
    ds = torch.empty(2 * cap, dtype=torch.float64, device='cuda')
    dm = torch.empty(2 * cap, dtype=torch.float64, device='cuda')
    check(lib.mrl_segtree_init(ptr(ds), ptr(dm), cap, stream_ptr()))
    done = 0
    while done < n:
        b = int(min(n - done, rng.choice([1, 7, 32, 300, 1500])))
        idx = rng.randint(0, min(cap, 5000), b).astype(np.int32)
        if b > 4:
            idx[-1] = idx[0]                          # force duplicates: the last one must win
        val = rng.rand(b) * 3 + 1e-3
        for i, v in zip(idx, val):
            st[int(i)] = float(v)
            mt[int(i)] = float(v)
        idx_d, val_d = _dev(idx), _dev(val)            # keep the temporaries alive across the launch
        check(lib.mrl_segtree_set(ptr(ds), ptr(dm), cap, ptr(idx_d), ptr(val_d), b, stream_ptr()))
        done += b
    np.testing.assert_array_equal(ds.cpu().numpy(), st.value)
    np.testing.assert_array_equal(dm.cpu().numpy(), mt.value)
    # sampling on top of it: indices bit-exact for several lengths (prefix query order), weights ~1e-14
    B = 64
    for length in (2, 3, min(cap, 5000) - 1, min(cap, 5000)):
        u = rng.rand(B)
        p_total = st.sum(0, length - 1)
        want = [st.find_prefixsum_idx(float(x) * (p_total / B) + i * (p_total / B)) for i, x in enumerate(u)]
        idx = torch.empty(B, dtype=torch.int32, device='cuda')
        w = torch.empty(B, dtype=torch.float64, device='cuda')
        u_d = _dev(u)
        check(lib.mrl_per_sample(ptr(ds), ptr(dm), cap, length, B, ptr(u_d), 0.7, ptr(idx), ptr(w), ptr(None), stream_ptr()))
        np.testing.assert_array_equal(idx.cpu().numpy(), np.asarray(want, np.int32))
        tot = st.sum()
        maxw = ((mt.min() / tot) * length) ** (-0.7)
        wwant = np.array([((st[i] / tot) * length) ** (-0.7) / maxw for i in want])
        np.testing.assert_allclose(w.cpu().numpy(), wwant, rtol=1e-14)


def test_ring_insert_wraps_and_gathers():
    from baselines_amd.deepq import ReplayBuffer
    rng = np.random.RandomState(3)
    size, shape = 37, (84, 84, 4)
    buf, ref = ReplayBuffer(size), R.ReplayBuffer(size)
    for it in range(9):
        n = int(rng.randint(1, 12))
        o1 = rng.randint(0, 256, (n,) + shape).astype(np.uint8)
        o2 = rng.randint(0, 256, (n,) + shape).astype(np.uint8)
        a = rng.randint(0, 6, n)
        r = rng.randn(n).astype(np.float32)
        d = (rng.rand(n) < 0.3).astype(np.float32)
        buf.add_batch(o1, a, r, o2, d)
        for k in range(n):
            ref.add(o1[k], np.array(a[k]), float(r[k]), o2[k], float(d[k]))
        assert len(buf) == len(ref) and buf._next_idx == ref.next_idx
    idx = rng.randint(0, len(ref), 50)
    got = buf._encode_sample(list(idx))
    want = ref.encode_sample(idx)
    for x, y in zip(got, want):
        np.testing.assert_array_equal(x, y)
    random.seed(5)
    s = buf.sample(16)
    assert s[0].shape == (16,) + shape and s[0].dtype == np.uint8 and s[1].dtype == np.int64


def test_per_device_fast_path_and_full_size_properties():
    """config 5 sized trees (2^20 leaves), priorities from TD errors on the device: tree invariants
    (every inner node = op(children)), max_priority, sampled indices within the filled range."""
    from baselines_amd.deepq import PrioritizedReplayBuffer
    cap = 1 << 20
    buf = PrioritizedReplayBuffer(cap, 0.6)
    rng = np.random.RandomState(0)
    shape = (8,)                                         # small obs: the trees are what is being sized here
    n = 4096
    for _ in range(3):
        buf.add_batch(rng.randint(0, 256, (n,) + shape).astype(np.uint8), rng.randint(0, 6, n),
                      rng.randn(n).astype(np.float32), rng.randint(0, 256, (n,) + shape).astype(np.uint8),
                      np.zeros(n, np.float32))
    assert len(buf) == 3 * n
    random.seed(0)
    o1, a, r, o2, d, w, idx = buf.sample_dev(512, 0.4)
    assert int(idx.min()) >= 0 and int(idx.max()) < len(buf)
    assert float(w.max()) <= 1.0 + 1e-6 and float(w.min()) > 0
    td = torch.randn(512, device='cuda') * 3
    buf.update_priorities_from_td(idx, td, eps=1e-6)
    s, m = buf.trees_numpy()
    inner = np.arange(1, cap)
    np.testing.assert_array_equal(s[inner], s[2 * inner] + s[2 * inner + 1])
    np.testing.assert_array_equal(m[inner], np.minimum(m[2 * inner], m[2 * inner + 1]))
    # leaves: last duplicate wins, value = (|td| + eps) ** alpha.  The device computes it correctly rounded (csrc/powcr.hip.h, round 6;
    # ROCm's own pow was one ulp off Python's `**` in 16 % of the inputs): equal to Python's unless glibc's pow itself misrounds (about
    # one input in a thousand, tests/test_powcr.py), never more than one ulp apart
    idx_h, td_h = idx.cpu().numpy(), td.cpu().numpy().astype(np.float64)
    last = {}
    for i, t in zip(idx_h, td_h):
        last[int(i)] = (abs(t) + 1e-6) ** 0.6
    got = s[cap + np.array(list(last.keys()))]
    want = np.array(list(last.values()))
    ulp = np.abs(got.view(np.int64) - want.view(np.int64))
    assert ulp.max() <= 1 and int((ulp != 0).sum()) <= 3, (int(ulp.max()), int((ulp != 0).sum()))
    assert buf._current_max_priority() == pytest.approx(max(1.0, float(np.abs(td_h).max() + 1e-6)), rel=1e-12)


@pytest.mark.parametrize('B,nA,double_q', [(32, 6, True), (1000, 18, True), (257, 4, False)])
def test_dqn_td_huber_vs_oracle(B, nA, double_q):
    from baselines_amd.deepq import dqn_td_loss
    rng = np.random.RandomState(B)
    q_t, q1, q2 = (rng.randn(B, nA).astype(np.float32) * 2 for _ in range(3))
    a = rng.randint(0, nA, B).astype(np.int32)
    r = rng.randn(B).astype(np.float32)
    d = (rng.rand(B) < 0.2).astype(np.float32)
    w = (rng.rand(B) + 0.1).astype(np.float32)
    td_o, loss_o, dq_o = R.dqn_td(q_t, q1, q2 if double_q else None, a, r, d, w, 0.99, double_q=double_q)
    td, loss, dq = dqn_td_loss(_dev(q_t), _dev(q1), _dev(q2) if double_q else None, _dev(a), _dev(r), _dev(d), _dev(w), 0.99)
    np.testing.assert_allclose(td.cpu().numpy(), td_o, rtol=0, atol=1e-6)
    np.testing.assert_allclose(float(loss.item()), float(loss_o), rtol=1e-5)        # tolerance of the float path: 1e-5
    np.testing.assert_allclose(dq.cpu().numpy(), dq_o, rtol=1e-6, atol=1e-9)


def test_device_priorities_are_correctly_rounded_powers():
    """mrl_per_update_from_td over 2^17 random TD errors: every leaf is the NEAREST double to (|td| + eps) ** alpha (70-digit decimal
    reference on a sample), bit-equal to the host build of the same source, and equal to Python's `**` except for glibc's own
    misroundings (<= 0.3 %; deepq/replay_buffer.py:169-191, deepq/deepq.py:302)."""
    import ctypes
    import subprocess
    import tempfile
    from decimal import Decimal, getcontext
    from baselines_amd.deepq import PrioritizedReplayBuffer
    cap = 1 << 17
    buf = PrioritizedReplayBuffer(cap, 0.6)
    rng = np.random.RandomState(5)
    n = 4096
    ob = rng.randint(0, 256, (n, 4)).astype(np.uint8)
    for _ in range(cap // n):
        buf.add_batch(ob, rng.randint(0, 6, n), rng.randn(n).astype(np.float32), ob, np.zeros(n, np.float32))
    td = (torch.randn(cap, device='cuda', generator=torch.Generator(device='cuda').manual_seed(3)) * 3).float()
    idx = torch.arange(cap, device='cuda', dtype=torch.int32)
    buf.update_priorities_from_td(idx, td, eps=1e-6)
    s, _ = buf.trees_numpy()
    got = s[cap:2 * cap]
    p = np.abs(td.cpu().numpy().astype(np.float64)) + 1e-6
    ref = np.array([float(v) ** 0.6 for v in p])
    ulp = np.abs(got.view(np.int64) - ref.view(np.int64))
    assert ulp.max() <= 1 and (ulp != 0).mean() <= 3e-3, (int(ulp.max()), float((ulp != 0).mean()))
    getcontext().prec = 70
    a = Decimal(0.6)
    for v, g in zip(p[:3000], got[:3000]):
        assert float((Decimal(float(v)).ln() * a).exp()) == g
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 'h.cpp'), 'w').write('#include "powcr.hip.h"\nextern "C" void f(const double* x, double a, double* o, long n) '
                                                   '{ for (long i = 0; i < n; ++i) o[i] = mrl::pow_cr(x[i], a); }\n')
        subprocess.check_call(['g++', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-I' + os.path.join(root, 'baselines_amd', 'csrc'),
                               os.path.join(d, 'h.cpp'), '-o', os.path.join(d, 'h.so')])
        L = ctypes.CDLL(os.path.join(d, 'h.so'))
        host = np.empty_like(p)
        L.f(p.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(0.6), host.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(p.size))
    np.testing.assert_array_equal(got, host)
