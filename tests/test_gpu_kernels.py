"""GPU parity tests (run with -m gpu on an MI355X): every HIP entry point of libmrl.so, called
through the C ABI, against the oracle (NumPy restatement pinned by the reference's golden
vectors; torch-CPU restatement of the TF graph)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import ppo2_numpy as O
from oracle.ppo2_torch import OracleModel, build_param_specs, init_params

pytestmark = pytest.mark.gpu


def _ops():
    from baselines_amd import ops
    return ops


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['t128_n8', 't5_n3', 't37_n70', 't1_n4', 't64_n129_alldone', 't64_n65_nodone',
                                  't300_n33_g1', 't16_n64_lam0'])
def test_gae_bit_exact_vs_reference_golden(golden_dir, name):
    from baselines_amd import _lib as L
    ops = _ops()
    g = np.load(os.path.join(golden_dir, 'runner_%s.npz' % name))
    ret, adv = ops.gae(dev(g['in_rewards']), dev(g['in_values']), dev(g['in_dones']), dev(g['in_last_values']),
                       dev(g['in_last_dones']), float(g['gamma']), float(g['lam']), want_advs=True)
    got = ops.sf01(ret).cpu().numpy()
    np.testing.assert_array_equal(got, g['out_returns'])      # bit-exact vs the reference's Runner.run
    # option gae_lane = 0: the LDS-staged kernel at every size -- the same bits (at N >= 64 the default is the lane kernel)
    assert L.get_option('gae_lane') == 1
    L.set_option('gae_lane', 0)
    try:
        ret0, adv0 = ops.gae(dev(g['in_rewards']), dev(g['in_values']), dev(g['in_dones']), dev(g['in_last_values']),
                             dev(g['in_last_dones']), float(g['gamma']), float(g['lam']), want_advs=True)
    finally:
        L.set_option('gae_lane', 1)
    assert torch.equal(ret0, ret) and torch.equal(adv0, adv)
    ret_o, adv_o = O.gae(g['in_rewards'], g['in_values'], g['in_dones'], g['in_last_values'], g['in_last_dones'],
                         float(g['gamma']), float(g['lam']))
    np.testing.assert_array_equal(adv.cpu().numpy(), adv_o)


def test_gae_full_size_properties():
    """BASELINE size (T=128, N=4096): bit-exact vs the oracle + all-done rows reduce to r - v."""
    ops = _ops()
    ro = O.synthetic_rollout('cartpole', 128, 4096, 3)
    rew = (np.random.RandomState(0).randn(128, 4096) * 2).astype(np.float32)
    ret = ops.gae(dev(rew), dev(ro['values']), dev(ro['dones']), dev(ro['last_values']), dev(ro['last_dones']),
                  0.99, 0.95).cpu().numpy()
    ret_o, _ = O.gae(rew, ro['values'], ro['dones'], ro['last_values'], ro['last_dones'], 0.99, 0.95)
    np.testing.assert_array_equal(ret, ret_o)
    # terminal transitions: adv = r - v exactly (f64 then f32) -> ret = f32(f32(r - v) + v)
    nxt = np.concatenate([ro['dones'][1:], ro['last_dones'][None]], 0)
    expect = ((rew.astype(np.float64) - ro['values']).astype(np.float32) + ro['values'])
    np.testing.assert_array_equal(ret[nxt], expect[nxt])


@pytest.mark.parametrize('shape,dtype', [((84, 84, 4), np.uint8), ((376,), np.float32), ((), np.float32),
                                         ((3,), np.uint8), ((17,), np.float32)])
def test_gather_and_sf01(shape, dtype):
    ops = _ops()
    T, N = 7, 13
    rng = np.random.RandomState(1)
    arr = (rng.randint(0, 255, (T, N) + shape)).astype(dtype)
    flat = O.sf01(arr)
    idx = rng.permutation(T * N)[:40].astype(np.int64)
    got = ops.gather_rows(dev(arr), dev(idx), T, N).cpu().numpy()
    np.testing.assert_array_equal(got, flat[idx])
    np.testing.assert_array_equal(ops.sf01(dev(arr)).cpu().numpy(), flat)


# ------------------------------------------------------------------------------------------------
NET_KW = ('layer_norm', 'num_layers', 'num_hidden', 'convs', 'fc_hidden', 'pad', 'nvec')


def _make_pair(network, ob_shape, ob_dtype, pd_kind, nact, value_network, seed, chunk, **kw):
    ops = _ops()
    np.random.seed(seed)
    om = OracleModel(network=network, ob_shape=ob_shape, ob_dtype=ob_dtype, pd_kind=pd_kind, nact=nact,
                     value_network=value_network, ent_coef=kw.pop('ent_coef', 0.01), vf_coef=0.5,
                     max_grad_norm=kw.pop('max_grad_norm', 0.5), **kw)
    dm = ops.DeviceModel(network=network, ob_shape=ob_shape, ob_dtype=ob_dtype, pd_kind=pd_kind, nact=nact,
                         value_copy=(value_network == 'copy'), chunk=chunk,
                         num_layers=kw.get('num_layers', 2), num_hidden=kw.get('num_hidden', 64),
                         layer_norm=kw.get('layer_norm', False), convs=kw.get('convs'), fc_hidden=kw.get('fc_hidden', 512),
                         pad=kw.get('pad', 'VALID'), nvec=kw.get('nvec'))
    # layout must equal the reference's variable order / shapes
    assert [t['name'] for t in dm.tensors] == om.names
    for t, (nm, shp, sc) in zip(dm.tensors, om.specs):
        assert tuple(t['shape']) == tuple(shp), (nm, t['shape'], shp)
        assert (t['init_scale'] is None) == (sc is None or sc == 'ones')
        assert (t['init_const'] == 1.0) == (sc == 'ones')
    flat = om.flat_params()
    assert flat.size == dm.P
    return om, dm, dev(flat.astype(np.float32))


def _perturb(om, seed, scale=0.05):
    """biases/logstd are zero at init; perturb everything so every gradient path is exercised"""
    rng = np.random.RandomState(seed)
    with torch.no_grad():
        for k in om.names:
            om.p[k] += torch.tensor(scale * rng.randn(*om.p[k].shape), dtype=om.p[k].dtype)


CONFIGS = {
    'cartpole_mlp_shared': dict(network='mlp', ob_shape=(4,), ob_dtype=np.float32, pd_kind='categorical', nact=2,
                                value_network=None, kind='cartpole', T=16, N=8, B=64, chunk=64),
    'mujoco_mlp_copy': dict(network='mlp', ob_shape=(376,), ob_dtype=np.float32, pd_kind='gaussian', nact=17,
                            value_network='copy', kind='mujoco', T=16, N=24, B=192, chunk=80),
    'atari_cnn': dict(network='cnn', ob_shape=(84, 84, 4), ob_dtype=np.uint8, pd_kind='categorical', nact=6,
                      value_network=None, kind='atari', T=4, N=8, B=24, chunk=16),
    'mlp_matching_fc': dict(network='mlp', ob_shape=(11,), ob_dtype=np.float32, pd_kind='gaussian', nact=64,
                            value_network=None, kind=None, T=8, N=8, B=32, chunk=32),
    # mlp(layer_norm=True) (common/models.py:97-98), separate value network, three layers of 48 units
    'mlp_layer_norm': dict(network='mlp', ob_shape=(11,), ob_dtype=np.float32, pd_kind='gaussian', nact=3,
                           value_network='copy', kind=None, T=8, N=8, B=48, chunk=32, layer_norm=True, num_layers=3, num_hidden=48),
    # nature_cnn on images that are not Atari frame stacks (common/models.py:19 casts and scales any image): RGB uint8,
    # single-channel float, 4-channel float -- first layer on the generic tiled engine, /255 in its loader
    'cnn_rgb_u8': dict(network='cnn', ob_shape=(44, 48, 3), ob_dtype=np.uint8, pd_kind='categorical', nact=4,
                       value_network=None, kind='image', T=3, N=6, B=12, chunk=8),
    'cnn_gray_f32': dict(network='cnn', ob_shape=(36, 40, 1), ob_dtype=np.float32, pd_kind='categorical', nact=3,
                         value_network=None, kind='image', T=3, N=6, B=12, chunk=16),
    'cnn_f32_4ch': dict(network='cnn', ob_shape=(40, 40, 4), ob_dtype=np.float32, pd_kind='gaussian', nact=2,
                        value_network=None, kind='image', T=2, N=8, B=16, chunk=16),
    # cnn_small (common/models.py:117-129): conv 8 x 8x8 / 4, conv 16 x 4x4 / 2, fc 128 -- on Atari frame stacks and with a
    # separate value network
    'cnn_small_atari': dict(network='cnn', ob_shape=(84, 84, 4), ob_dtype=np.uint8, pd_kind='categorical', nact=6,
                            value_network=None, kind='atari', T=4, N=8, B=24, chunk=16, convs=((8, 8, 4), (16, 4, 2)), fc_hidden=128),
    'cnn_small_copy': dict(network='cnn', ob_shape=(52, 44, 3), ob_dtype=np.uint8, pd_kind='categorical', nact=3,
                           value_network='copy', kind='image', T=3, N=6, B=12, chunk=16, convs=((8, 8, 4), (16, 4, 2)), fc_hidden=128),
    # cnn(pad='SAME') (a **conv_kwargs entry of nature_cnn, a2c/utils.py:37): zero padding, the smaller half in front
    # MultiDiscrete / MultiBinary action spaces (common/distributions.py:206-225, 253-276, 285-288)
    'mlp_multidiscrete': dict(network='mlp', ob_shape=(9,), ob_dtype=np.float32, pd_kind='multicategorical', nact=9, nvec=(2, 3, 4),
                              value_network='copy', kind=None, T=8, N=8, B=48, chunk=32),
    'cnn_multidiscrete': dict(network='cnn', ob_shape=(84, 84, 4), ob_dtype=np.uint8, pd_kind='multicategorical', nact=7, nvec=(1, 2, 4),
                              value_network=None, kind='atari', T=3, N=6, B=12, chunk=16),
    'mlp_multibinary': dict(network='mlp', ob_shape=(7,), ob_dtype=np.float32, pd_kind='bernoulli', nact=5,
                            value_network=None, kind=None, T=8, N=8, B=40, chunk=64),
    'cnn_same_pad': dict(network='cnn', ob_shape=(30, 34, 4), ob_dtype=np.uint8, pd_kind='categorical', nact=4,
                         value_network=None, kind='image', T=3, N=6, B=12, chunk=16, pad='SAME'),
}


def _rollout(cfg, om, seed):
    """Seeded synthetic rollout whose actions / values / neglogpacs come from the (oracle) policy
    itself with teacher-forced noise, like a real Runner.run -> ratios near 1, realistic loss scale."""
    rng = np.random.RandomState(seed)
    T, N = cfg['T'], cfg['N']
    if cfg['kind'] == 'image':
        ro = O.synthetic_rollout('cartpole', T, N, seed)
        px = rng.randint(0, 256, (T, N) + cfg['ob_shape'])
        ro['obs'] = px.astype(np.uint8) if cfg['ob_dtype'] == np.uint8 else (px + rng.rand(*px.shape)).astype(np.float32)
    elif cfg['kind']:
        ro = O.synthetic_rollout(cfg['kind'], T, N, seed)
    else:
        ro = O.synthetic_rollout('cartpole', T, N, seed)
        ro['obs'] = rng.randn(*((T, N) + cfg['ob_shape'])).astype(np.float32)
    acts, vals, nlps = [], [], []
    for t in range(T):
        nz = (rng.rand(N, cfg['nact']) if cfg['pd_kind'] != 'gaussian' else rng.randn(N, cfg['nact']))
        a, v, _, nlp = om.step(ro['obs'][t], nz.astype(np.float32))
        acts.append(a); vals.append(v); nlps.append(nlp)
    ro['actions'], ro['values'], ro['neglogpacs'] = np.stack(acts), np.stack(vals), np.stack(nlps)
    ro['last_values'] = om.value(ro['obs'][0])
    ro['rewards'] = (ro['rewards'] * 0.3).astype(np.float32)
    return ro


@pytest.mark.parametrize('name', list(CONFIGS))
def test_model_act_grad_train_vs_oracle(name):
    cfg = dict(CONFIGS[name])
    ops = _ops()
    kind, T, N, B, chunk = cfg.pop('kind'), cfg.pop('T'), cfg.pop('N'), cfg.pop('B'), cfg.pop('chunk')
    full = dict(CONFIGS[name])
    om, dm, _ = _make_pair(seed=0, chunk=chunk, **cfg)
    _perturb(om, 1)
    ro = _rollout(full, om, 2)
    _perturb(om, 3, 0.01)          # "a few optimizer steps later": ratios != 1, some samples clipped
    params = dev(om.flat_params().astype(np.float32))
    om64 = OracleModel(network=cfg['network'], ob_shape=cfg['ob_shape'], ob_dtype=cfg['ob_dtype'],
                       pd_kind=cfg['pd_kind'], nact=cfg['nact'], value_network=cfg['value_network'],
                       ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, dtype=torch.float64,
                       params=om.params_numpy(), **{k: cfg[k] for k in NET_KW if k in cfg})
    returns, _ = O.gae(ro['rewards'], ro['values'], ro['dones'], ro['last_values'], ro['last_dones'], 0.99, 0.95)

    # ---- act side (teacher-forced noise) ----
    rng = np.random.RandomState(5)
    obs0 = ro['obs'][0]
    noise = (rng.rand(N, cfg['nact']) if cfg['pd_kind'] != 'gaussian' else rng.randn(N, cfg['nact'])).astype(np.float32)
    a_o, v_o, _, nlp_o = om.step(obs0, noise)
    a_d, v_d, nlp_d, pd_d = dm.act(params, dev(obs0), dev(noise), want_pdparam=True)
    with torch.no_grad():
        pd_o, _ = om.forward(obs0)
    np.testing.assert_allclose(pd_d.cpu().numpy(), pd_o.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(v_d.cpu().numpy(), v_o, rtol=1e-4, atol=2e-5)
    if cfg['pd_kind'] != 'gaussian':
        assert a_d.dtype == torch.int32 and tuple(a_d.shape) == tuple(a_o.shape)
        np.testing.assert_array_equal(a_d.cpu().numpy().astype(np.int64), a_o.astype(np.int64))
    else:
        np.testing.assert_allclose(a_d.cpu().numpy(), a_o, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(nlp_d.cpu().numpy(), nlp_o, rtol=1e-4, atol=2e-5)
    v_only = dm.act(params, dev(obs0), None, want_actions=False)[1]
    np.testing.assert_array_equal(v_only.cpu().numpy(), v_d.cpu().numpy())

    # ---- learner: gradient of the loss on a gathered minibatch ----
    np.random.seed(7)
    idx = next(iter(O.minibatch_indices(T * N, B, 1)))
    f = {k: O.sf01(ro[k]) for k in ('obs', 'actions', 'values', 'neglogpacs')}
    fret = O.sf01(returns)
    mb = dict(obs=f['obs'][idx], returns=fret[idx], actions=f['actions'][idx], values=f['values'][idx],
              neglogpacs=f['neglogpacs'][idx])
    cliprange = 0.2
    stats_o, g_o = om.compute_grads(cliprange, mb['obs'], mb['returns'], mb['actions'], mb['values'], mb['neglogpacs'])
    stats_64, g_64 = om64.compute_grads(cliprange, mb['obs'], mb['returns'], mb['actions'], mb['values'],
                                        mb['neglogpacs'])
    acts = ro['actions'].astype(np.int32) if cfg['pd_kind'] != 'gaussian' else ro['actions']
    d_obs, d_act, d_ret = dev(ro['obs']), dev(acts), dev(returns)
    d_val, d_nlp, d_idx = dev(ro['values']), dev(ro['neglogpacs']), dev(idx)
    grads = torch.empty(dm.P, dtype=torch.float32, device='cuda')
    stats = torch.empty(5, dtype=torch.float32, device='cuda')
    dm.grad(params, d_obs, d_act, d_ret, d_val, d_nlp, d_idx, B, T, N, cliprange, 0.01, 0.5, grads, stats)
    s_d = stats.cpu().numpy()
    g_d = grads.cpu().numpy().astype(np.float64)
    # north-star bar: fp32 losses within 1e-5 of the reference CPU path
    np.testing.assert_allclose(s_d, np.array(stats_o), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(s_d, np.array(stats_64), rtol=1e-5, atol=1e-5)
    g64 = g_64.numpy()
    scale = np.abs(g64).max()
    err_d = np.abs(g_d - g64).max() / scale
    err_o = np.abs(g_o.numpy().astype(np.float64) - g64).max() / scale
    assert err_d < 5e-5, (err_d, err_o)          # as close to fp64 as the fp32 CPU restatement is (same class)
    # per-tensor check so a wrong small tensor cannot hide behind a big one
    for t in dm.tensors:
        sl = slice(t['offset'], t['offset'] + t['size'])
        ref = g64[sl]
        tol = 5e-5 * max(np.abs(ref).max(), 1e-3 * scale)
        assert np.abs(g_d[sl] - ref).max() <= tol, (t['name'], np.abs(g_d[sl] - ref).max(), tol)

    # same minibatch passed directly (idx == NULL path)
    grads2 = torch.empty_like(grads)
    stats2 = torch.empty_like(stats)
    mb_act = mb['actions'].astype(np.int32) if cfg['pd_kind'] != 'gaussian' else mb['actions']
    dm.grad(params, dev(mb['obs']), dev(mb_act), dev(mb['returns']), dev(mb['values']), dev(mb['neglogpacs']), None,
            B, 1, 1, cliprange, 0.01, 0.5, grads2, stats2)
    np.testing.assert_array_equal(grads2.cpu().numpy(), grads.cpu().numpy())
    np.testing.assert_array_equal(stats2.cpu().numpy(), s_d)

    # ---- optimizer: 3 full train steps (clip + TF-Adam) ----
    m = torch.zeros_like(params)
    v = torch.zeros_like(params)
    scratch = torch.empty(8192, dtype=torch.uint8, device='cuda')
    b1p, b2p = np.float32(0.9), np.float32(0.999)
    lr = np.float32(3e-4)
    for it in range(3):
        om.train(float(lr), cliprange, mb['obs'], mb['returns'], None, mb['actions'], mb['values'], mb['neglogpacs'])
        dm.grad(params, d_obs, d_act, d_ret, d_val, d_nlp, d_idx, B, T, N, cliprange, 0.01, 0.5, grads, stats)
        alpha = lr * np.sqrt(np.float32(1) - b2p) / (np.float32(1) - b1p)
        gn = torch.empty(1, dtype=torch.float32, device='cuda')
        ops.adam_clip_step(params, grads, m, v, alpha, 0.9, 0.999, 1e-5, 0.5, 1.0, scratch, gn)
        b1p, b2p = np.float32(b1p * np.float32(0.9)), np.float32(b2p * np.float32(0.999))
        assert abs(float(gn.cpu()) - om.last_gnorm) <= 1e-4 * om.last_gnorm
    np.testing.assert_allclose(params.cpu().numpy(), om.flat_params(), rtol=0, atol=3e-6)


def test_adam_clip_matches_oracle_many_steps():
    ops = _ops()
    rng = np.random.RandomState(0)
    P = 100003
    om = OracleModel(network='mlp', ob_shape=(4,), ob_dtype=np.float32, pd_kind='categorical', nact=2)
    # hijack the oracle's optimizer on a single fake tensor
    om.names = ['x']
    om.p = {'x': torch.tensor(rng.randn(P).astype(np.float32))}
    om.m = {'x': torch.zeros(P)}
    om.v = {'x': torch.zeros(P)}
    params = dev(om.p['x'].numpy().copy())
    m, v = torch.zeros_like(params), torch.zeros_like(params)
    scratch = torch.empty(8192, dtype=torch.uint8, device='cuda')
    b1p, b2p = np.float32(0.9), np.float32(0.999)
    for it in range(20):
        g = (rng.randn(P) * (10.0 if it % 2 else 0.001)).astype(np.float32)   # clipped and unclipped steps
        om.total_weight, om.rank_weight = 1.0, 1.0
        om.apply_flat_grad(2.5e-4, torch.tensor(g))
        gd = dev(g)
        alpha = np.float32(2.5e-4) * np.sqrt(np.float32(1) - b2p) / (np.float32(1) - b1p)
        ops.adam_clip_step(params, gd, m, v, alpha, 0.9, 0.999, 1e-5, 0.5, 1.0, scratch)
        b1p, b2p = np.float32(b1p * np.float32(0.9)), np.float32(b2p * np.float32(0.999))
        np.testing.assert_allclose(gd.cpu().numpy(), om.last_grads.numpy(), rtol=2e-6, atol=1e-12)
    np.testing.assert_allclose(params.cpu().numpy(), om.p['x'].numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(m.cpu().numpy(), om.m['x'].numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(v.cpu().numpy(), om.v['x'].numpy(), rtol=1e-5, atol=1e-12)


def test_engine_options_agree():
    """The fast engines against their plain counterparts on the same minibatch:
       * first conv layer on the bf16 pipe (exact u8 x 3-way bf16 split) vs the fp32 MFMA path,
       * conv2/conv3/fc1 forward and the conv2/conv3/fc1 data gradients on the bf16 pipe (both operands split into 3 exact
         bf16 planes, six products per multiply -- eight in -DMRL_PRODUCTS8 builds) vs the fp32 MFMA paths (batch >= 1024 so that the tiled split engine
         takes the fc layer),
       * conv data gradients on the position-major tiled engine vs the LDS-resident fp32-MFMA engine,
       * fused whole-step MLP kernel vs layer-wise launches.
    Gradients agree to fp32 round-off on EVERY entry (the split products are exact; only the summation order and the
    folded 1/255 scale differ).  The NatureCNN cases run on minibatches screened by the fp64 oracle for ReLU margins
    (tests/test_gpu_large_batch.py): a pre-activation within round-off of the kink switches on in one engine and off in
    the other -- the forward value hardly moves, but a whole element of the gradient flowing back does -- which is a
    property of ReLU, not of an engine."""
    from baselines_amd import _lib as L
    from baselines_amd import ops
    from tests.test_gpu_large_batch import _problem

    def grads(network, ob_shape, ob_dtype, pd_kind, nact, value_copy, B, opts, screened=None):
        for o, v in opts.items():
            L.set_option(o, v)
        dm = ops.DeviceModel(network=network, ob_shape=ob_shape, ob_dtype=ob_dtype, pd_kind=pd_kind, nact=nact,
                             value_copy=value_copy, chunk=B)
        r = np.random.RandomState(1)
        if screened is not None:
            om, _, mb = screened
            params = dev(om.flat_params().astype(np.float32))
            obs, act = dev(mb['obs']), dev(mb['actions'].astype(np.int32))
            ret, val_, nlp = dev(mb['returns']), dev(mb['values']), dev(mb['neglogpacs'])
        else:
            params = torch.from_numpy((r.randn(dm.P) * 0.05).astype(np.float32)).cuda()
            if ob_dtype == np.uint8:
                obs = torch.from_numpy(r.randint(0, 256, (B,) + ob_shape).astype(np.uint8)).cuda()
            else:
                obs = torch.from_numpy(r.randn(B, *ob_shape).astype(np.float32)).cuda()
            if pd_kind == 'categorical':
                act = torch.from_numpy(r.randint(0, nact, B).astype(np.int32)).cuda()
            else:
                act = torch.from_numpy(r.randn(B, nact).astype(np.float32)).cuda()
            ret, val_, nlp = (torch.from_numpy(r.randn(B).astype(np.float32)).cuda() for _ in range(3))
            nlp = nlp.abs() + 1.0
        g = torch.empty(dm.P, dtype=torch.float32, device='cuda')
        st = torch.empty(5, dtype=torch.float32, device='cuda')
        dm.grad(params, obs, act, ret, val_, nlp, None, B, 1, 1, 0.2, 0.01, 0.5, g, st)
        return g.cpu().numpy(), st.cpu().numpy()

    names = ['u8_bf16x3', 'f32_bf16x6', 'mlp_fused', 'dgrad_x6', 'relu_bits', 'c1_lds', 'wgrad_x8', 'c1_wgrad2', 'tr_epilogue',
             'wgrad_tr', 'x6_pg', 'mlp_slice', 'mlp_waves', 'x6_dither', 'x6_frag', 'conv_x6c', 'wgrad_pipe', 'heads_wave', 'x6_ktm', 'wgrad_xcd']
    # builds with -DMRL_X6_EXPERIMENTS also carry the measured-and-dropped variants (plane tensors, separate load phase)
    experiments = True
    try:
        L.get_option('act_planes')
        names += ['act_planes', 'x6_il']
    except Exception:
        experiments = False
    defaults = {o: L.get_option(o) for o in names}
    assert defaults['tr_epilogue'] == 1, 'transposed-accumulator epilogues are the default'
    assert defaults['f32_bf16x6'] == 2 and L.get_option('f32_products') in (6, 8), 'split engines are the default'
    assert defaults['wgrad_tr'] == 1, 'conv2 / conv3 / fc1 weight gradients: transpose-read kernels by default'
    assert defaults['c1_lds'] == 4, 'first conv layer forward: software-pipelined image-resident kernel by default'
    assert defaults['x6_dither'] == 3, 'tiled split engines: sign alternation of the staged rows and conflict-free staging order by default'
    assert defaults['wgrad_pipe'] == 1, 'conv2 / conv3 weight gradients: next image split between the MFMAs of the current one by default'
    assert defaults['conv_x6c'] == 1, 'conv2 / conv3 forward at minibatch sizes: class-resident kernel by default'
    assert defaults['x6_frag'] == 1, 'tiled split engines: operand split on the fragment path, between the MFMAs, by default'
    assert defaults['dgrad_x6'] == 2, 'conv data gradients: position-major kernel with exact wait counts around its stores by default'
    assert defaults['heads_wave'] == 2, 'loss / head gradients of the NatureCNN head shape: two samples per wave-step by default'
    if experiments:
        assert defaults['act_planes'] == 76 and defaults['x6_il'] == 1
    try:
        cnn = ('cnn', (84, 84, 4), np.uint8, 'categorical', 6, False)
        # ---- small batches (screened samples for the NatureCNN cases, plain random data for the MLPs)
        om_s, _, mb_s = _problem(168, 23)
        small = [(cnn, 160, 'u8_bf16x3', 1), (cnn, 161, 'c1_lds', 1), (cnn, 163, 'c1_wgrad2', 1), (cnn, 165, 'tr_epilogue', 1),
                 (cnn, 168, 'x6_pg', 8), (cnn, 164, 'x6_dither', 3),
                 (('mlp', (376,), np.float32, 'gaussian', 17, True), 200, 'mlp_fused', 1),
                 (('mlp', (376,), np.float32, 'gaussian', 17, True), 203, 'mlp_slice', 1),       # (tile, net) workgroups vs one per tile
                 (('mlp', (12,), np.float32, 'categorical', 5, True), 97, 'mlp_slice', 1),
                 (('mlp', (4,), np.float32, 'categorical', 2, False), 96, 'mlp_fused', 1)]
        if experiments:
            small += [(cnn, 32, 'act_planes', 79), (cnn, 166, 'act_planes', 28), (cnn, 167, 'x6_il', 1)]
        for cfg, B, opt, on in small:
            scr_s = (om_s, None, {k: v[:B] for k, v in mb_s.items()}) if cfg is cnn else None
            g1, s1 = grads(*cfg, B, dict(defaults, **{opt: on}), scr_s)
            g0, s0 = grads(*cfg, B, dict(defaults, **{opt: 0 if opt != 'x6_pg' else 1}), scr_s)
            scale = np.abs(g0).max()
            assert np.abs(g1 - g0).max() <= 2e-6 * scale + 1e-9, (opt, np.abs(g1 - g0).max(), scale)
            np.testing.assert_allclose(s1, s0, rtol=1e-5, atol=1e-6)
        # the two-samples-per-wave-step loss kernel evaluates the same formulas on the same sums in the same orders as the one-sample
        # kernel: gradient and statistics BIT-identical, for odd batch sizes and waves whose last step holds one sample as well
        for B in (161, 165, 168, 33, 97):
            scr_s = (om_s, None, {k: v[:B] for k, v in mb_s.items()})
            g2, s2 = grads(*cnn, B, dict(defaults, heads_wave=2), scr_s)
            g1, s1 = grads(*cnn, B, dict(defaults, heads_wave=1), scr_s)
            np.testing.assert_array_equal(g2, g1, err_msg='heads_wave B=%d' % B)
            np.testing.assert_array_equal(s2, s1, err_msg='heads_wave B=%d' % B)
            g0, s0 = grads(*cnn, B, dict(defaults, heads_wave=0), scr_s)             # generic tile kernel
            assert np.abs(g2 - g0).max() <= 2e-6 * np.abs(g0).max() + 1e-9
            np.testing.assert_allclose(s2, s0, rtol=1e-5, atol=1e-6)
        # ---- B = 1152 (tiled split engines everywhere they apply), screened minibatch: every entry
        B = 1152
        scr = _problem(B, 21)
        # reference: all fp32 x fp32 sites on the fp32 MFMA pipe
        ref_opts = dict(defaults, f32_bf16x6=0, dgrad_x6=0, relu_bits=0, c1_lds=0, wgrad_x8=0, c1_wgrad2=0, tr_epilogue=0, wgrad_tr=0, x6_pg=1)
        if experiments:
            ref_opts.update(act_planes=0, x6_il=0)
        g0, s0 = grads(*cnn, B, ref_opts, scr)
        scale = np.abs(g0).max()
        cases = [('split engines (default)', dict(defaults), 3e-6),
                 ('split engines, LDS-resident fp32 data gradients', dict(defaults, dgrad_x6=0), 3e-6),
                 ('split engines, data gradients with the generic position-major kernel (round 5)', dict(defaults, dgrad_x6=1), 3e-6),
                 ('split engines, fc weight planes in [plane][n][k] order (round 5)', dict(defaults, x6_ktm=0), 3e-6),
                 ('split engines, fc weight gradient workgroups in launch order (round 5)', dict(defaults, wgrad_xcd=0), 3e-6),
                 ('split engines, act\' from the fp32 activations instead of the ReLU bit masks', dict(defaults, relu_bits=0), 3e-6),
                 ('split engines, row-major accumulators and epilogues (no transposed epilogues)', dict(defaults, tr_epilogue=0), 3e-6),
                 ('split engines, one row panel at a time through the column tiles', dict(defaults, x6_pg=1), 3e-6),
                 ('split engines, every row staged as is (no sign alternation)', dict(defaults, x6_dither=0), 3e-6),
                 ('split engines, sign alternation without the conflict-free staging order', dict(defaults, x6_dither=1), 3e-6),
                 ('split engines, conflict-free staging order without the sign alternation', dict(defaults, x6_dither=2), 3e-6),
                 ('split engines, no sign alternation, row-major accumulators', dict(defaults, x6_dither=0, tr_epilogue=0), 3e-6),
                 ('split engines, conv2 / conv3 forward on the tiled engine instead of the class-resident kernel', dict(defaults, conv_x6c=0), 3e-6),
                 ('split engines, operand split in the staging pass (planes in LDS) instead of on the fragment path', dict(defaults, x6_frag=0, conv_x6c=0), 3e-6),
                 ('split engines, split in the staging pass, no sign alternation', dict(defaults, x6_frag=0, x6_dither=2, conv_x6c=0), 3e-6),
                 ('split engines, split on the fragment path, conv k steps in natural order', dict(defaults, x6_frag=5, conv_x6c=0), 3e-6),
                 ('split engines, split on the fragment path, natural k order, no sign alternation', dict(defaults, x6_frag=5, x6_dither=2, conv_x6c=0), 3e-6),
                 ('split engines, split on the fragment path, two register sets of operand loads', dict(defaults, x6_frag=2, conv_x6c=0), 3e-6),
                 ('split engines, conv weight gradients with a split phase of its own (no software pipeline)', dict(defaults, wgrad_pipe=0), 3e-6),
                 ('split engines, loss / head gradients one sample per wave-step', dict(defaults, heads_wave=1), 3e-6),
                 ('split engines, class-resident conv forward without the sign alternation', dict(defaults, x6_dither=2), 3e-6),
                 ('split engines, tiled conv forward in class-major k order without the sign alternation', dict(defaults, x6_dither=2, conv_x6c=0), 3e-6),
                 ('split engines, first conv layer on the gather engine instead of the image-resident one', dict(defaults, c1_lds=0), 3e-6),
                 ('split engines, first conv layer forward: lock-step phases instead of the software pipeline',
                  dict(defaults, c1_lds=2), 3e-6),
                 ('split engines, first conv layer forward: software pipeline with row-major accumulators', dict(defaults, c1_lds=3), 3e-6),
                 ('split engines, first conv layer forward: uint8 image in LDS, converted per fragment', dict(defaults, c1_lds=1), 3e-6),
                 ('split engines, conv1 weight gradient: whole-image workgroups', dict(defaults, c1_wgrad2=1), 3e-6),
                 ('split engines, conv1 weight gradient: half-image units, operand loads one unit ahead', dict(defaults, c1_wgrad2=2), 3e-6),
                 ('split engines, weight gradients of conv2 / conv3 / fc1 on the fp32 MFMA engines',
                  dict(defaults, wgrad_x8=0, wgrad_tr=0), 3e-6),
                 ('split engines, conv2 / conv3 weight gradients on the fp32-MFMA image-resident engine, fc1 on the transposed-staging tiles',
                  dict(defaults, wgrad_tr=0), 3e-6),
                 ('split engines, weight gradients of conv2 / conv3 / fc1 on the transposed-staging tiles',
                  dict(defaults, wgrad_x8=2, wgrad_tr=0), 3e-6)]
        if experiments:
            cases += [('loads in a phase of their own instead of between the MFMAs', dict(defaults, x6_il=0), 3e-6),
                      ('plane tensors of the forward activations only', dict(defaults, act_planes=1), 3e-6),
                      ('plane tensors of the pre-activation gradients only', dict(defaults, act_planes=2), 3e-6),
                      ('plane tensors of both', dict(defaults, act_planes=3), 3e-6),
                      ('transposed-accumulator epilogues incl. fc1 forward, planes of the last dz only', dict(defaults, act_planes=108), 3e-6)]
        by_name = {}
        for name, opts, tol in cases:
            g1, s1 = grads(*cnn, B, opts, scr)
            by_name[name] = (g1, s1)
            d = np.abs(g1 - g0)
            assert d.max() <= tol * scale + 1e-9, (name, d.max(), scale, int(d.argmax()))
            np.testing.assert_allclose(s1, s0, rtol=1e-5, atol=1e-6)
        # the split-at-the-fragment kernel (gemmx6r.hip.h) forms the same products as the staged-planes kernel and, with the conv
        # layers' k steps in natural order (x6_frag = 5), adds them in the same order: gradient and statistics are then BIT-identical,
        # with and without the sign alternation (the default's class-major k order changes the order of the sums only)
        for a, b in (('split engines, split on the fragment path, conv k steps in natural order',
                      'split engines, operand split in the staging pass (planes in LDS) instead of on the fragment path'),
                     ('split engines, split on the fragment path, natural k order, no sign alternation',
                      'split engines, split in the staging pass, no sign alternation'),
                     # the class-resident kernel walks k in the class-major order of the tiled engine's default: the same sums (its tiles
                     # are whole images, so WHICH rows the sign alternation negates differs -- compared without it)
                     ('split engines, class-resident conv forward without the sign alternation',
                      'split engines, tiled conv forward in class-major k order without the sign alternation'),
                     # the pipelined weight gradients split the same values and multiply in the same order
                     ('split engines (default)', 'split engines, conv weight gradients with a split phase of its own (no software pipeline)'),
                     ('split engines (default)', 'split engines, loss / head gradients one sample per wave-step'),
                     # dgrad_x6p_kernel (round 6: exact wait counts around the epilogue stores) is the same arithmetic in the same order
                     ('split engines (default)', 'split engines, data gradients with the generic position-major kernel (round 5)'),
                     ('split engines (default)', 'split engines, fc weight planes in [plane][n][k] order (round 5)'),
                     ('split engines (default)', 'split engines, fc weight gradient workgroups in launch order (round 5)')):
            np.testing.assert_array_equal(by_name[a][0], by_name[b][0], err_msg=b)
            np.testing.assert_array_equal(by_name[a][1], by_name[b][1], err_msg=b)
        # ... also where conv3's 256-image tiles divide the batch (1152 % 256 != 0 sends conv3 to the generic kernel above), on
        # several tiles per workgroup, and on a batch neither tile size divides (both layers on the generic kernel)
        for B2 in (1280, 5120, 200):
            ga, sa = grads(*cnn, B2, dict(defaults), None)
            gb, sb = grads(*cnn, B2, dict(defaults, dgrad_x6=1), None)
            np.testing.assert_array_equal(ga, gb, err_msg='dgrad_x6 = 2 vs 1 at B = %d' % B2)
            np.testing.assert_array_equal(sa, sb, err_msg='dgrad_x6 = 2 vs 1 at B = %d' % B2)
    finally:
        for o, v in defaults.items():
            L.set_option(o, v)


@pytest.mark.parametrize('waves', [8, 4])
def test_precomputed_advantage_statistics_feed_the_fused_mlp_step_bit_identically(waves):
    """mrl_advstat_minibatches + mrl_model_set_advstat (the epoch graphs of Model.train_epoch): the statistics of all
    minibatches of an epoch in one launch are the (mean, std) of model.py:136-139 and bit-identical to what the fused step's
    workgroups form for themselves -- the gradient of a step fed from them equals the gradient of a plain step, bit for bit"""
    from baselines_amd import _lib as L
    from baselines_amd import ops
    T, N, M = 8, 96, 4
    B = T * N // M
    r = np.random.RandomState(7)
    old = L.get_option('mlp_waves')
    L.set_option('mlp_waves', waves)
    try:
        dm = ops.DeviceModel(network='mlp', ob_shape=(376,), ob_dtype=np.float32, pd_kind='gaussian', nact=17, value_copy=True,
                             chunk=B)
        params = dev((r.randn(dm.P) * 0.05).astype(np.float32))
        obs, act = dev(r.randn(T * N, 376).astype(np.float32)), dev(r.randn(T * N, 17).astype(np.float32))
        ret, val_ = r.randn(T * N).astype(np.float32), r.randn(T * N).astype(np.float32)
        nlp = dev((np.abs(r.randn(T * N)) + 1.0).astype(np.float32))
        perm = r.permutation(T * N).astype(np.int64).reshape(M, B)                 # env-major indices i = e * T + t
        idx = dev(perm)
        adv = torch.zeros((M, 2), dtype=torch.float32, device='cuda')
        d_ret, d_val = dev(ret), dev(val_)
        L.check(dm.lib.mrl_advstat_minibatches(L.ptr(d_ret), L.ptr(d_val), L.ptr(idx), M, B, T, N, L.ptr(adv),
                                               L.stream_ptr()), 'mrl_advstat_minibatches')
        rows = (perm % T) * N + perm // T                                          # storage rows of the time-major rollout
        x = (ret[rows] - val_[rows]).astype(np.float64)
        np.testing.assert_allclose(adv.cpu().numpy()[:, 0], x.mean(axis=1), rtol=0, atol=1e-7)
        np.testing.assert_allclose(adv.cpu().numpy()[:, 1], x.std(axis=1), rtol=2e-7, atol=0)
        for k in range(M):
            g0 = torch.empty(dm.P, dtype=torch.float32, device='cuda')
            g1 = torch.empty_like(g0)
            s0, s1 = torch.empty(5, dtype=torch.float32, device='cuda'), torch.empty(5, dtype=torch.float32, device='cuda')
            dm.grad(params, obs, act, d_ret, d_val, nlp, idx[k], B, T, N, 0.2, 0.01, 0.5, g0, s0)
            L.check(dm.lib.mrl_model_set_advstat(dm.handle, L.ptr(adv[k])), 'mrl_model_set_advstat')
            try:
                dm.grad(params, obs, act, d_ret, d_val, nlp, idx[k], B, T, N, 0.2, 0.01, 0.5, g1, s1)
            finally:
                L.check(dm.lib.mrl_model_set_advstat(dm.handle, None), 'mrl_model_set_advstat')
            assert torch.equal(g0, g1) and torch.equal(s0, s1) and float(g0.abs().max()) > 0
        # wrong statistics DO change the result (the pointer is what the step reads)
        bad = torch.tensor([5.0, 0.1], dtype=torch.float32, device='cuda')
        L.check(dm.lib.mrl_model_set_advstat(dm.handle, L.ptr(bad)), 'mrl_model_set_advstat')
        try:
            dm.grad(params, obs, act, d_ret, d_val, nlp, idx[0], B, T, N, 0.2, 0.01, 0.5, g1, s1)
        finally:
            L.check(dm.lib.mrl_model_set_advstat(dm.handle, None), 'mrl_model_set_advstat')
        assert not torch.equal(g0, g1)
    finally:
        L.set_option('mlp_waves', old)


@pytest.mark.parametrize('B', [1, 257, 4109])
def test_conv1_forward_engines_are_bit_identical(B):
    """The image-resident first-layer kernels (c1fwd.hip.h: software pipeline with transposed / row-major accumulators,
    lock-step phases) form the same exact products: logits, values and the whole
    gradient (it goes through the ReLU bit masks the forward writes) are bit-identical between the row-major pipeline and the lock-step
    kernel, and within 1e-6 of them for the balanced transposed pipeline (see below) -- one image, a batch that
    leaves persistent workgroups with 1-2 images and one with 16 (two peeled phases + the steady-state loop)."""
    from baselines_amd import _lib as L
    from baselines_amd import ops
    old = L.get_option('c1_lds')
    outs = []
    try:
        for v in (4, 3, 2):
            L.set_option('c1_lds', v)
            dm = ops.DeviceModel(network='cnn', ob_shape=(84, 84, 4), ob_dtype=np.uint8, pd_kind='categorical', nact=6, chunk=B)
            r = np.random.RandomState(B)
            params = dev((r.randn(dm.P) * 0.05).astype(np.float32))
            obs = dev(r.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8))
            act = dev(r.randint(0, 6, B).astype(np.int32))
            ret, val_, nlp = (dev(r.randn(B).astype(np.float32)) for _ in range(3))
            nlp = nlp.abs() + 1.0
            _, vv, _, pd = dm.act(params, obs, None, want_actions=False, want_pdparam=True)
            g = torch.empty(dm.P, dtype=torch.float32, device='cuda')
            st = torch.empty(5, dtype=torch.float32, device='cuda')
            dm.grad(params, obs, act, ret, val_, nlp, None, B, 1, 1, 0.2, 0.01, 0.5, g, st)
            outs.append((vv.cpu(), pd.cpu(), g.cpu(), st.cpu()))
    finally:
        L.set_option('c1_lds', old)
    assert float(outs[0][2].abs().max()) > 0 and bool(torch.isfinite(outs[0][2]).all())
    # c1_lds = 3 and 2: bit-identical.  c1_lds = 4 (round 6): the same products, but the pixels 384 .. 399 of every image are multiplied
    # by v_mfma_f32_16x16x32_bf16 (one patch row = 32 k per instruction) instead of padded 32x32x16 tiles, which adds the same exact
    # products in groups of 32 instead of 16: last-bit differences in those 16 pixels, everything held to 1e-6 of its scale
    for a, b in zip(outs[1], outs[2]):
        assert torch.equal(a, b)
    for a, b in zip(outs[0], outs[1]):
        assert float((a.double() - b.double()).abs().max()) <= 1e-6 * max(float(b.double().abs().max()), 1e-30)


@pytest.mark.parametrize('cfg', [dict(ob=376, pd='gaussian', nact=17, copy=True, n=1024), dict(ob=376, pd='gaussian', nact=17, copy=True, n=37),
                                 dict(ob=11, pd='gaussian', nact=3, copy=False, n=200), dict(ob=4, pd='categorical', nact=2, copy=False, n=8),
                                 dict(ob=27, pd='categorical', nact=5, copy=True, n=1001)])
def test_fused_mlp_act_agrees_with_the_layer_wise_path(cfg):
    """Act side of the 2 x 64 tanh MLP (common/models.py:74-103 behind policies.py:77-96 step()): the one-launch forward of both
    layers of both nets (csrc/mlpact.hip.h, option mlp_act) forms fp32 products and fp32 sums like the layer-wise path; the order
    of the sums differs (the tiled GEMM splits K over partial slabs at act batch sizes), so values, neglogp and pdparam agree to
    2e-6 and categorical actions exactly; odd observation widths, shared and separate value nets, ragged last workgroups."""
    from baselines_amd import _lib as L
    from baselines_amd import ops
    r = np.random.RandomState(cfg['ob'] + cfg['n'])
    old = L.get_option('mlp_act')
    outs = []
    try:
        for v in (1, 0):
            L.set_option('mlp_act', v)
            dm = ops.DeviceModel(network='mlp', ob_shape=(cfg['ob'],), ob_dtype=np.float32, pd_kind=cfg['pd'], nact=cfg['nact'],
                                 value_copy=cfg['copy'], chunk=cfg['n'])
            if v == 1:
                params = dev((r.randn(dm.P) * 0.1).astype(np.float32))
                obs = dev(r.randn(cfg['n'], cfg['ob']).astype(np.float32))
                noise = dev((r.rand(cfg['n'], cfg['nact']) if cfg['pd'] == 'categorical' else r.randn(cfg['n'], cfg['nact'])).astype(np.float32))
            a, vv, nlp, pd = dm.act(params, obs, noise, want_pdparam=True)
            outs.append((a.cpu(), vv.cpu(), nlp.cpu(), pd.cpu()))
    finally:
        L.set_option('mlp_act', old)
    assert bool(torch.isfinite(outs[0][1]).all()) and float(outs[0][1].abs().max()) > 0
    for q, (x, y) in enumerate(zip(outs[0], outs[1])):
        assert x.dtype == y.dtype and x.shape == y.shape
        if q == 0 and cfg['pd'] == 'categorical':
            assert float((x != y).float().mean()) <= 0.002           # (an argmax over Gumbel-perturbed logits can flip on a 1e-7 tie)
        else:
            assert float((x.double() - y.double()).abs().max()) <= 2e-6 * max(1.0, float(y.double().abs().max()))


def test_act_heads_wave_kernel_agrees_with_the_tile_kernel():
    """heads_wave also selects the ACT side's wave-per-sample heads kernel (policies.py:77-96 on the NatureCNN latent): actions equal,
    values / neglogp / logits within 1e-6 of the tile kernel's (a different order of the 512-term dot products)."""
    from baselines_amd import _lib as L
    from baselines_amd import ops
    n = 333
    r = np.random.RandomState(5)
    old = L.get_option('heads_wave')
    outs = []
    try:
        for v in (2, 0):
            L.set_option('heads_wave', v)
            dm = ops.DeviceModel(network='cnn', ob_shape=(84, 84, 4), ob_dtype=np.uint8, pd_kind='categorical', nact=6, chunk=n)
            if v == 2:
                params = dev((r.randn(dm.P) * 0.02).astype(np.float32))
                obs = dev(r.randint(0, 256, (n, 84, 84, 4)).astype(np.uint8))
                noise = dev(r.rand(n, 6).astype(np.float32))
            a, vv, nlp, pd = dm.act(params, obs, noise, want_pdparam=True)
            outs.append((a.cpu(), vv.cpu(), nlp.cpu(), pd.cpu()))
    finally:
        L.set_option('heads_wave', old)
    assert float((outs[0][0] != outs[1][0]).float().mean()) <= 0.01
    for x, y in zip(outs[0][1:], outs[1][1:]):
        assert float((x.double() - y.double()).abs().max()) <= 1e-6 * max(1.0, float(y.double().abs().max()))
