"""GPU parity at the batch sizes the benchmark actually runs.

The engines `pick_variant` / `layer_forward` / `net_backward` select depend on the batch: the tiled split engines
(bf16 planes of fp32 operands, gemmx6.hip.h) take fc1 forward / data gradient only for B >= 1024, conv data gradients
use the position-major tiled engine, image-resident weight gradients run one persistent workgroup per CU.  The small
shapes of test_gpu_kernels.py never reach those paths, so here the NatureCNN update is compared with the oracle
(torch-CPU fp32 AND fp64 restatement of ppo2/model.py:57-114, common/models.py:15-26) at
  * B = 2048  (gradients per tensor, max-error bound on every entry),
  * B = 8192  (config 3's real minibatch: one whole train step -- stats 1e-5, parameters 5e-6 after the step),
  * B = 131072 with chunk 131072 (the metric's minibatch): act-side logits / values of 512 rows -- first, last, and
    around the 2^31-byte offsets of the observation and activation tensors -- against the oracle on just those rows.

ReLU kinks: a pre-activation within float round-off of zero can switch on in one implementation and off in the other,
which changes that sample's gradient by a finite amount -- a property of the function, not of either implementation.
The minibatches are therefore screened with the fp64 oracle: samples with any pre-activation closer to zero than
MARGIN are replaced by spare ones, after which every entry of the gradient has to agree.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.ppo2_torch import OracleModel

pytestmark = pytest.mark.gpu

MARGIN = 1e-5       # ~50x the round-off of a pre-activation; keeps ~84 % of random samples
KW = dict(network='cnn', ob_shape=(84, 84, 4), ob_dtype=np.uint8, pd_kind='categorical', nact=6, value_network=None,
          ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5)


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def _min_abs_preact(om64, obs):
    """smallest |pre-activation| per sample over all four ReLU layers (fp64 restatement of OracleModel._net)"""
    p = om64.p
    with torch.no_grad():
        h = torch.as_tensor(obs).to(torch.float64) / 255.
        h = h.permute(0, 3, 1, 2)
        lo = torch.full((obs.shape[0],), np.inf, dtype=torch.float64)
        for name, stride in (('c1', 4), ('c2', 2), ('c3', 1)):
            w = p['ppo2_model/pi/%s/w' % name].permute(3, 2, 0, 1)
            z = F.conv2d(h, w, p['ppo2_model/pi/%s/b' % name].reshape(-1), stride=stride)
            lo = torch.minimum(lo, z.abs().reshape(z.shape[0], -1).min(dim=1)[0])
            h = F.relu(z)
        h = h.permute(0, 2, 3, 1).reshape(h.shape[0], -1)
        z = h @ p['ppo2_model/pi/fc1/w'] + p['ppo2_model/pi/fc1/b']
        lo = torch.minimum(lo, z.abs().min(dim=1)[0])
    return lo.numpy()


def _problem(B, seed):
    """(fp32 oracle, fp64 oracle, minibatch dict) with B screened samples; actions / values / neglogpacs come from a
    slightly older version of the policy so that ratios != 1 and some samples are clipped"""
    rng = np.random.RandomState(seed)
    np.random.seed(seed)
    om = OracleModel(**KW)
    with torch.no_grad():
        for k in om.names:
            om.p[k] += torch.tensor(0.02 * rng.randn(*om.p[k].shape), dtype=torch.float32)
    pool = rng.randint(0, 256, (B + B // 2 + 64, 84, 84, 4)).astype(np.uint8)
    a_old, v_old, _, nlp_old = om.step(pool, rng.rand(pool.shape[0], 6).astype(np.float32))
    with torch.no_grad():
        for k in om.names:
            om.p[k] += torch.tensor(0.003 * rng.randn(*om.p[k].shape), dtype=torch.float32)
    om64 = OracleModel(dtype=torch.float64, params=om.params_numpy(), **KW)
    keep = np.nonzero(_min_abs_preact(om64, pool) > MARGIN)[0]
    assert keep.size >= B, 'screening left %d of %d samples' % (keep.size, B)
    keep = keep[:B]
    mb = dict(obs=pool[keep], actions=a_old[keep], values=v_old[keep], neglogpacs=nlp_old[keep],
              returns=(v_old[keep] + 0.5 * rng.randn(B)).astype(np.float32))
    return om, om64, mb


def _report(name, dm, g_d, g32, g64):
    """one JSON line per call into $MRL_PARITY_REPORT: per tensor, max error over the tensor's scale against the fp64 oracle"""
    path = os.environ.get('MRL_PARITY_REPORT')
    if not path:
        return
    import json
    errs = {}
    for t in dm.tensors:
        sl = slice(t['offset'], t['offset'] + t['size'])
        sc = float(np.abs(g64[sl]).max())
        errs[t['name']] = dict(err_device=float(np.abs(g_d[sl] - g64[sl]).max()) / sc, err_fp32_oracle=float(np.abs(g32[sl] - g64[sl]).max()) / sc)
    with open(path, 'a') as fh:
        fh.write(json.dumps(dict(test=name, per_tensor=errs)) + '\n')


def _device_model(B):
    from baselines_amd import ops
    return ops.DeviceModel(network='cnn', ob_shape=(84, 84, 4), ob_dtype=np.uint8, pd_kind='categorical', nact=6, chunk=B)


def _device_grad(dm, params, mb, cliprange):
    B = mb['obs'].shape[0]
    grads = torch.empty(dm.P, dtype=torch.float32, device='cuda')
    stats = torch.empty(5, dtype=torch.float32, device='cuda')
    dm.grad(params, dev(mb['obs']), dev(mb['actions'].astype(np.int32)), dev(mb['returns']), dev(mb['values']),
            dev(mb['neglogpacs']), None, B, 1, 1, cliprange, 0.01, 0.5, grads, stats)
    return grads.cpu().numpy().astype(np.float64), stats.cpu().numpy()


def test_nature_cnn_gradient_at_b2048_vs_oracle_every_entry():
    """fc1 forward + data gradient on the tiled split engine (B >= 1024), conv data gradients on the position-major
    tiled engine: per-tensor max error over ALL entries against the fp64 oracle, next to the fp32 oracle's own error."""
    B, clip = 2048, 0.2
    om, om64, mb = _problem(B, 11)
    dm = _device_model(B)
    g_d, s_d = _device_grad(dm, dev(om.flat_params().astype(np.float32)), mb, clip)
    s_o, g_o = om.compute_grads(clip, mb['obs'], mb['returns'], mb['actions'], mb['values'], mb['neglogpacs'])
    s_64, g_64 = om64.compute_grads(clip, mb['obs'], mb['returns'], mb['actions'], mb['values'], mb['neglogpacs'])
    np.testing.assert_allclose(s_d, np.array(s_o), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(s_d, np.array(s_64), rtol=1e-5, atol=1e-5)
    g64 = g_64.numpy()
    g32 = g_o.numpy().astype(np.float64)
    scale = np.abs(g64).max()
    for t in dm.tensors:
        sl = slice(t['offset'], t['offset'] + t['size'])
        ref = g64[sl]
        tol = 5e-5 * max(np.abs(ref).max(), 1e-3 * scale)          # the bar of test_model_act_grad_train_vs_oracle
        err_d = np.abs(g_d[sl] - ref).max()
        err_o = np.abs(g32[sl] - ref).max()
        assert err_d <= tol, (t['name'], err_d, err_o, tol)          # every entry, no percentile
        # the same class as the fp32 CPU restatement (measured: 0.9-1.1e-6 of a tensor's scale, oracle 0.2-1.3e-6; without the sign
        # alternation of the staged rows -- option x6_dither, DESIGN.md 3.1 -- the split engines sit at 3-4.6e-6)
        assert err_d <= max(8.0 * err_o, 2.5e-6 * max(np.abs(ref).max(), 1e-3 * scale)), (t['name'], err_d, err_o)
    _report('b2048_backward', dm, g_d, g32, g64)
    # the common part of the split engines' error: with every row staged as is the matrix instruction's bias toward -inf has the same
    # sign for all samples and survives the sums over the minibatch
    from baselines_amd import _lib
    dither = _lib.get_option('x6_dither')
    _lib.set_option('x6_dither', dither & ~1)
    try:
        g_0, _ = _device_grad(_device_model(B), dev(om.flat_params().astype(np.float32)), mb, clip)
    finally:
        _lib.set_option('x6_dither', dither)
    _report('b2048_backward_x6_dither_0', dm, g_0, g32, g64)
    for t in dm.tensors:
        if t['name'].endswith(('c1/w', 'vf/b')):
            sl = slice(t['offset'], t['offset'] + t['size'])
            assert np.abs(g_d[sl] - g64[sl]).max() < 0.6 * np.abs(g_0[sl] - g64[sl]).max(), t['name']


def test_nature_cnn_train_step_at_b8192_vs_oracle():
    """config 3's minibatch (num_envs=256, nsteps=128, 4 minibatches): one Model.train step through
    mrl_model_train_step -- stats within 1e-5, parameters within 5e-6 after the optimizer step (north-star bar)."""
    from baselines_amd import _lib
    B, clip, lr = 8192, 0.1, 2.5e-4
    om, om64, mb = _problem(B, 12)
    dm = _device_model(B)
    params = dev(om.flat_params().astype(np.float32))
    p0 = params.clone()
    grads, m, v = (torch.zeros(dm.P, dtype=torch.float32, device='cuda') for _ in range(3))
    stats = torch.empty(5, dtype=torch.float32, device='cuda')
    gn = torch.empty(1, dtype=torch.float32, device='cuda')
    one = np.float32(1)
    alpha = np.float32(lr) * np.sqrt(one - np.float32(0.999)) / (one - np.float32(0.9))
    d_obs, d_act = dev(mb['obs']), dev(mb['actions'].astype(np.int32))          # keep the owners alive across the call
    d_ret, d_val, d_nlp = dev(mb['returns']), dev(mb['values']), dev(mb['neglogpacs'])
    _lib.check(dm.lib.mrl_model_train_step(
        dm.handle, _lib.ptr(params), _lib.ptr(grads), _lib.ptr(m), _lib.ptr(v), _lib.ptr(d_obs),
        _lib.ptr(d_act), _lib.ptr(d_ret), _lib.ptr(d_val),
        _lib.ptr(d_nlp), None, B, 1, 1, clip, 0.01, 0.5, float(alpha), None, 0.9, 0.999, 1e-5, 0.5, 1.0,
        _lib.ptr(stats), _lib.ptr(gn), _lib.ptr(dm.workspace), dm.workspace.numel(), dm.chunk, _lib.stream_ptr()),
        'mrl_model_train_step')
    s_o = om.train(lr, clip, mb['obs'], mb['returns'], None, mb['actions'], mb['values'], mb['neglogpacs'])
    s_64 = om64.train(lr, clip, mb['obs'], mb['returns'], None, mb['actions'], mb['values'], mb['neglogpacs'])
    s_d = stats.cpu().numpy()
    np.testing.assert_allclose(s_d, np.array(s_o), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(s_d, np.array(s_64), rtol=1e-5, atol=1e-5)
    assert abs(float(gn.cpu()) - om64.last_gnorm) <= 2e-5 * om64.last_gnorm
    pd_ = params.cpu().numpy()
    assert np.abs(pd_ - p0.cpu().numpy()).max() > 1e-5                  # the step did move the parameters
    np.testing.assert_allclose(pd_, om.flat_params(), rtol=0, atol=5e-6)
    np.testing.assert_allclose(pd_, om64.flat_params(), rtol=0, atol=5e-6)
    # the clipped, averaged gradient left in `grads` (model.py:112 self.grads)
    np.testing.assert_allclose(grads.cpu().numpy(), om64.last_grads.numpy(), rtol=0,
                               atol=2e-5 * float(om64.last_grads.abs().max()))


def test_full_size_minibatch_act_rows_vs_oracle():
    """B = 131072 in ONE chunk (the benched configuration: 22.7 GB of activations, tensors past 2^31 and 2^32 bytes):
    logits and values of 512 rows against the oracle evaluated on just those rows."""
    B = 131072
    rng = np.random.RandomState(5)
    np.random.seed(5)
    om = OracleModel(**KW)
    with torch.no_grad():
        for k in om.names:
            om.p[k] += torch.tensor(0.02 * rng.randn(*om.p[k].shape), dtype=torch.float32)
    om64 = OracleModel(dtype=torch.float64, params=om.params_numpy(), **KW)
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1234)
    obs = torch.randint(0, 256, (B, 84, 84, 4), generator=gen, device='cuda', dtype=torch.uint8)
    dm = _device_model(B)
    params = dev(om.flat_params().astype(np.float32))
    _, v_d, _, pd_d = dm.act(params, obs, None, want_actions=False, want_pdparam=True)
    # rows whose observation / conv1 / conv2 activations straddle 2^31 and 2^32 bytes, plus the ends and random ones
    edges = [0, 1, B - 2, B - 1]
    for per_sample_bytes in (28224, 20 * 20 * 32 * 4, 9 * 9 * 64 * 4, 7 * 7 * 64 * 4, 512 * 4):
        for lim in (2 ** 31, 2 ** 32):
            r = lim // per_sample_bytes
            edges += [x for x in (r - 1, r, r + 1) if 0 <= x < B]
    rows = np.unique(np.concatenate([np.array(edges), rng.randint(0, B, 512)]))[:512 + len(edges)]
    sub = obs[torch.as_tensor(rows).cuda()].cpu().numpy()
    with torch.no_grad():
        pi_o, v_o = om.forward(sub)
        pi_64, v_64 = om64.forward(sub)
    got_pi, got_v = pd_d.cpu().numpy()[rows], v_d.cpu().numpy()[rows]
    np.testing.assert_allclose(got_pi, pi_64.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(got_v, v_64.numpy(), rtol=1e-4, atol=2e-5)
    # as close to fp64 as the fp32 CPU restatement is
    e_d = np.abs(got_v - v_64.numpy()).max()
    e_o = np.abs(v_o.numpy() - v_64.numpy()).max()
    assert e_d <= max(8 * e_o, 2e-6), (e_d, e_o)
    assert np.isfinite(pd_d.cpu().numpy()).all() and np.isfinite(v_d.cpu().numpy()).all()


def _sliced(n, step):
    return [slice(s, min(s + step, n)) for s in range(0, n, step)]


def test_full_size_minibatch_backward_vs_oracle_every_entry():
    """THE benched configuration: `mrl_model_grad` at B = 131072 in ONE 131072-sample chunk -- the
    weight- and data-gradient kernels walk 512 images per persistent workgroup, activation / dz / bit-mask tensors are
    0.27-6.7 GB and cross 2^31 and 2^32 bytes.  Every entry of every gradient tensor and the 5 statistics are compared with
    the fp64 oracle (ppo2/model.py:57-114, common/models.py:15-26); the oracle and the fp64 ReLU-margin screening run over
    8192-sample slices with the advantage statistics of the WHOLE minibatch (model.py:136-139), which is the same sum.
    Second check, GPU only: the sum of `mrl_model_grad_micro` over 16 slices of 8192 (the size the other tests verify
    against the oracle) has to reproduce the one-chunk gradient to 3e-6 of the gradient scale, every entry."""
    import os
    B, S, clip = 131072, 8192, 0.1
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    rng = np.random.RandomState(21)
    np.random.seed(21)
    om = OracleModel(**KW)
    with torch.no_grad():
        for k in om.names:
            om.p[k] += torch.tensor(0.02 * rng.randn(*om.p[k].shape), dtype=torch.float32)
    npool = B + B // 4 + 2048
    pool = rng.randint(0, 256, (npool, 84, 84, 4), dtype=np.uint8)
    a_old = np.empty(npool, np.int64)
    v_old = np.empty(npool, np.float32)
    nlp_old = np.empty(npool, np.float32)
    for sl in _sliced(npool, S):
        a_old[sl], v_old[sl], _, nlp_old[sl] = om.step(pool[sl], rng.rand(sl.stop - sl.start, 6).astype(np.float32))
    with torch.no_grad():
        for k in om.names:
            om.p[k] += torch.tensor(0.003 * rng.randn(*om.p[k].shape), dtype=torch.float32)
    om64 = OracleModel(dtype=torch.float64, params=om.params_numpy(), **KW)
    margin = np.concatenate([_min_abs_preact(om64, pool[sl]) for sl in _sliced(npool, S)])
    keep = np.nonzero(margin > MARGIN)[0]
    assert keep.size >= B, 'screening left %d of %d samples' % (keep.size, B)
    keep = keep[:B]
    obs = pool[keep]
    del pool
    acts, vals, nlps = a_old[keep], v_old[keep], nlp_old[keep]
    rets = (vals + 0.5 * rng.randn(B)).astype(np.float32)

    # ---- device: one call, one chunk
    dm = _device_model(B)
    assert dm.chunk == B
    params = dev(om.flat_params().astype(np.float32))
    d_obs, d_act = dev(obs), dev(acts.astype(np.int32))
    d_ret, d_val, d_nlp = dev(rets), dev(vals), dev(nlps)
    grads = torch.empty(dm.P, dtype=torch.float32, device='cuda')
    stats = torch.empty(5, dtype=torch.float32, device='cuda')
    dm.grad(params, d_obs, d_act, d_ret, d_val, d_nlp, None, B, 1, 1, clip, 0.01, 0.5, grads, stats)
    g_d = grads.cpu().numpy().astype(np.float64)
    s_d = stats.cpu().numpy()
    assert np.isfinite(g_d).all() and np.isfinite(s_d).all()

    # ---- device: 16 microbatch slices of 8192 (advantage statistics over all B samples), averaged
    gm = torch.empty(dm.P, dtype=torch.float32, device='cuda')
    sm = torch.empty(5, dtype=torch.float32, device='cuda')
    acc = torch.zeros(dm.P, dtype=torch.float64, device='cuda')
    sacc = torch.zeros(5, dtype=torch.float64, device='cuda')
    for sl in _sliced(B, S):
        dm.grad_micro(params, d_obs, d_act, d_ret, d_val, d_nlp, None, B, sl.start, S, 1, 1, clip, 0.01, 0.5, gm, sm)
        acc += gm.double()
        sacc += sm.double()
    g_m = (acc / (B // S)).cpu().numpy()
    s_m = (sacc / (B // S)).cpu().numpy()
    scale = np.abs(g_d).max()
    assert np.abs(g_m - g_d).max() <= 3e-6 * scale, (np.abs(g_m - g_d).max(), scale)
    np.testing.assert_allclose(s_d, s_m, rtol=1e-5, atol=1e-5)

    # ---- fp64 oracle over the same slices ($MRL_PARITY_REPORT set: the fp32 CPU restatement too, so that the device's error
    #      can be quoted next to the error of an fp32 implementation, per tensor -- VERDICT r03 weak 3)
    import json
    report_path = os.environ.get('MRL_PARITY_REPORT')
    advs = om64._normalised_advs(rets, vals)
    g64 = np.zeros(dm.P, np.float64)
    s64 = np.zeros(5, np.float64)
    g32 = np.zeros(dm.P, np.float64) if report_path else None
    advs32 = om._normalised_advs(rets, vals) if report_path else None
    for sl in _sliced(B, S):
        st, fl = om64.compute_grads(clip, obs[sl], rets[sl], acts[sl], vals[sl], nlps[sl], advs=advs[sl])
        g64 += fl.numpy()
        s64 += np.array(st)
        if report_path:
            _, fl32 = om.compute_grads(clip, obs[sl], rets[sl], acts[sl], vals[sl], nlps[sl], advs=advs32[sl])
            g32 += fl32.numpy().astype(np.float64)
    g64 /= B // S
    s64 /= B // S
    if report_path:
        g32 /= B // S
        errs = {}
        for t in dm.tensors:
            sl = slice(t['offset'], t['offset'] + t['size'])
            sc = float(np.abs(g64[sl]).max())
            errs[t['name']] = dict(scale=sc, err_device=float(np.abs(g_d[sl] - g64[sl]).max()) / sc,
                                   err_fp32_oracle_summed_over_16_slices=float(np.abs(g32[sl] - g64[sl]).max()) / sc)
        with open(report_path, 'a') as fh:
            fh.write(json.dumps(dict(test='full_size_minibatch_backward', B=B, chunk=B,
                                     stats_abs_diff_vs_fp64=[float(x) for x in np.abs(s_d - s64)],
                                     grad_errors_over_tensor_scale=errs)) + '\n')
    np.testing.assert_allclose(s_d, s64, rtol=1e-5, atol=1e-5)
    scale = np.abs(g64).max()
    for t in dm.tensors:
        sl = slice(t['offset'], t['offset'] + t['size'])
        ref = g64[sl]
        # 5e-6 of the tensor's scale, every entry (measured 0.9e-7 .. 1.2e-6 with the sign alternation of x6_dither; 1.35e-5 on the
        # first layer's weights before it, which is why this bar used to be 5e-5)
        tol = 5e-6 * max(np.abs(ref).max(), 1e-3 * scale)
        err = np.abs(g_d[sl] - ref).max()
        assert err <= tol, (t['name'], err, tol)
