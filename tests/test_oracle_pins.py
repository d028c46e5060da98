"""Pins of the TF half of the oracle (oracle/ppo2_torch.py) by the only statements of that mathematics the
reference itself holds outside TensorFlow:

  * the NumPy Adam of common/mpi_adam.py:38-41, which the reference's own `test_MpiAdam` (:64-99) proves equal to
    tf.train.AdamOptimizer over 10 steps within 1e-4 -- restated verbatim below and run against
    `OracleModel.apply_flat_grad` on the very problem of that test (a [3] and a [2,5] variable,
    loss = sum(a^2) + sum(sin(b)), stepsize 1e-2);
  * the statistical identities of common/distributions.py:320-348 `validate_probtype` on the fixed pdparams of
    `test_probtypes` (:303-311): entropy == -E[log p] and KL[p,q] == -H[p] - E_p[log q], N = 100,000, 3 sigma --
    run against the oracle's `_neglogp` / `_entropy` (the formulas its loss uses) and the reference's kl() formulas.

(The same identities are checked on the GPU kernels in tests/test_gpu_probtypes.py.)
"""
import math

import numpy as np
import torch

from oracle.ppo2_torch import OracleModel


# ---- common/mpi_adam.py:26-42, comm=None branch, verbatim arithmetic (float32 state, python-float hyper-parameters)
class _RefNumpyAdam(object):
    def __init__(self, size, beta1=0.9, beta2=0.999, epsilon=1e-08):
        self.beta1, self.beta2, self.epsilon = beta1, beta2, epsilon
        self.m = np.zeros(size, 'float32')
        self.v = np.zeros(size, 'float32')
        self.t = 0

    def update(self, theta, localg, stepsize):
        localg = localg.astype('float32')
        globalg = np.copy(localg)
        self.t += 1
        a = stepsize * np.sqrt(1 - self.beta2 ** self.t) / (1 - self.beta1 ** self.t)
        self.m = self.beta1 * self.m + (1 - self.beta1) * globalg
        self.v = self.beta2 * self.v + (1 - self.beta2) * (globalg * globalg)
        step = (- a) * self.m / (np.sqrt(self.v) + self.epsilon)
        return theta + step


def _tiny_oracle():
    """an OracleModel whose optimizer state is hijacked onto the two variables of test_MpiAdam"""
    om = OracleModel(network='mlp', ob_shape=(4,), ob_dtype=np.float32, pd_kind='categorical', nact=2, max_grad_norm=None)
    np.random.seed(0)
    a = np.random.randn(3).astype('float32')
    b = np.random.randn(2, 5).astype('float32')
    om.names = ['a', 'b']
    om.p = {'a': torch.tensor(a, requires_grad=True), 'b': torch.tensor(b, requires_grad=True)}
    om.m = {k: torch.zeros_like(v) for k, v in om.p.items()}
    om.v = {k: torch.zeros_like(v) for k, v in om.p.items()}
    om.eps = np.float32(1e-8)                       # tf.train.AdamOptimizer default of that test (ppo2 passes 1e-5)
    return om, a, b


def test_oracle_adam_equals_reference_numpy_adam_on_the_reference_test_problem():
    om, a, b = _tiny_oracle()
    stepsize = 1e-2
    ref = _RefNumpyAdam(a.size + b.size)
    theta = np.concatenate([a.reshape(-1), b.reshape(-1)])
    losses_ref, losses_om = [], []
    for i in range(10):
        # reference side: loss and flat gradient of sum(a^2) + sum(sin(b)) at theta
        ta, tb = theta[:3], theta[3:]
        losses_ref.append(float(np.sum(np.square(ta)) + np.sum(np.sin(tb))))
        g = np.concatenate([2 * ta, np.cos(tb)]).astype('float32')
        theta = ref.update(theta, g, stepsize).astype('float32')
        # oracle side: autograd gradient of the same loss, then its TF-1 ApplyAdam restatement
        for t in om.p.values():
            t.grad = None
        loss = (om.p['a'] ** 2).sum() + torch.sin(om.p['b']).sum()
        loss.backward()
        losses_om.append(float(loss.detach()))
        flat = torch.cat([om.p['a'].grad.reshape(-1), om.p['b'].grad.reshape(-1)])
        om.apply_flat_grad(stepsize, flat)
    # the reference's own bar (mpi_adam.py:99) ...
    np.testing.assert_allclose(np.array(losses_ref), np.array(losses_om), atol=1e-4)
    # ... and far tighter: same formula, float32 round-off only
    np.testing.assert_allclose(om.flat_params(), theta, rtol=0, atol=2e-6)
    np.testing.assert_allclose(torch.cat([om.m['a'].reshape(-1), om.m['b'].reshape(-1)]).numpy(), ref.m, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(torch.cat([om.v['a'].reshape(-1), om.v['b'].reshape(-1)]).numpy(), ref.v, rtol=5e-5, atol=1e-9)      # TF holds beta2 in float32: 1 - f32(0.999) is 1.3e-5 off 1e-3
    assert losses_ref[-1] < losses_ref[0]


def test_oracle_adam_with_ppo2_epsilon_differs_from_default_epsilon():
    """guards the pin above against vacuity: epsilon is where TF's Adam and torch.optim.Adam differ (not bias-corrected)"""
    om, a, b = _tiny_oracle()
    om2, _, _ = _tiny_oracle()
    om2.eps = np.float32(1e-2)
    g = torch.tensor(np.linspace(-1e-3, 1e-3, 13).astype('float32'))
    om.apply_flat_grad(1e-2, g.clone())
    om2.apply_flat_grad(1e-2, g.clone())
    assert np.abs(om.flat_params() - om2.flat_params()).max() > 1e-4


# ---- distributions.py:184-192, 242-244: the reference's kl() formulas (torch float64)
def _kl_categorical(l0, l1):
    a0 = l0 - l0.max(dim=-1, keepdim=True)[0]
    a1 = l1 - l1.max(dim=-1, keepdim=True)[0]
    ea0, ea1 = torch.exp(a0), torch.exp(a1)
    z0, z1 = ea0.sum(-1, keepdim=True), ea1.sum(-1, keepdim=True)
    p0 = ea0 / z0
    return (p0 * (a0 - torch.log(z0) - a1 + torch.log(z1))).sum(-1)


def _kl_gaussian(m0, ls0, m1, ls1):
    s0, s1 = torch.exp(ls0), torch.exp(ls1)
    return (ls1 - ls0 + (s0 ** 2 + (m0 - m1) ** 2) / (2.0 * s1 ** 2) - 0.5).sum(-1)


def _kl_bernoulli(l0, l1):
    """distributions.py:266-267: sum ce(logits = other, labels = ps) - sum ce(logits = self, labels = ps)"""
    ps = torch.sigmoid(l0)
    ce = lambda l, z: torch.clamp(l, min=0) - l * z + torch.log1p(torch.exp(-l.abs()))
    return ce(l1, ps).sum(-1) - ce(l0, ps).sum(-1)


def _kl_multicategorical(l0, l1, nvec):
    """distributions.py:217-218: add_n of the slices' Categorical kl"""
    tot, o = 0.0, 0
    for nv in nvec:
        tot = tot + _kl_categorical(l0[:, o:o + nv], l1[:, o:o + nv])
        o += nv
    return tot


PDPARAM_MULTICATEGORICAL, NVEC_MULTICATEGORICAL = np.array([-.2, .3, .5, .1, 1, -.1]), (1, 2, 3)     # distributions.py:311-312
PDPARAM_BERNOULLI = np.array([-.2, .3, .5])                               # distributions.py:316
PDPARAM_DIAG_GAUSS = np.array([-.2, .3, .4, -.5, .1, -.5, .1, 0.8])     # distributions.py:303
PDPARAM_CATEGORICAL = np.array([-.2, .3, .5])                             # distributions.py:307
N_SAMPLES = 100000


def _oracle_for(pd_kind, nact, nvec=None):
    return OracleModel(network='mlp', ob_shape=(4,), ob_dtype=np.float32, pd_kind=pd_kind, nact=nact, dtype=torch.float64, nvec=nvec)


def _validate(neglogp, entropy, kl, sample, pdparam, q):
    """distributions.py:320-348 with the oracle's functions in the place of the TF graph"""
    N = N_SAMPLES
    X = sample(pdparam, N)
    logliks = -neglogp(pdparam, X)
    entval_ll = -logliks.mean()
    entval_ll_stderr = logliks.std() / np.sqrt(N)
    entval = entropy(pdparam, N).mean()
    assert np.abs(entval - entval_ll) < 3 * entval_ll_stderr
    klval = kl(pdparam, q, N).mean()
    logliks = -neglogp(q, X)
    klval_ll = -entval - logliks.mean()
    klval_ll_stderr = logliks.std() / np.sqrt(N)
    assert np.abs(klval - klval_ll) < 3 * klval_ll_stderr


def test_validate_probtype_diag_gaussian_on_oracle():
    np.random.seed(0)
    nact = PDPARAM_DIAG_GAUSS.size // 2
    om = _oracle_for('gaussian', nact)
    rep = lambda v, n: torch.tensor(np.repeat(v[None, :], n, axis=0))

    def with_logstd(pdparam):
        om.p['ppo2_model/pi/logstd'] = torch.tensor(pdparam[None, nact:])
        return rep(pdparam[:nact], N_SAMPLES)

    def sample(pdparam, n):
        mean = with_logstd(pdparam)
        return mean + torch.exp(om.p['ppo2_model/pi/logstd']) * torch.tensor(np.random.randn(n, nact))

    q = PDPARAM_DIAG_GAUSS + np.random.randn(PDPARAM_DIAG_GAUSS.size) * 0.1
    _validate(neglogp=lambda pdp, X: om._neglogp(with_logstd(pdp), X).numpy(),
              entropy=lambda pdp, n: om._entropy(with_logstd(pdp)).numpy(),
              kl=lambda p, q_, n: _kl_gaussian(rep(p[:nact], n), rep(p[nact:], n), rep(q_[:nact], n), rep(q_[nact:], n)).numpy(),
              sample=sample, pdparam=PDPARAM_DIAG_GAUSS, q=q)


def test_validate_probtype_categorical_on_oracle():
    np.random.seed(0)
    nact = PDPARAM_CATEGORICAL.size
    om = _oracle_for('categorical', nact)
    rep = lambda v, n: torch.tensor(np.repeat(v[None, :], n, axis=0))

    def sample(pdparam, n):                 # distributions.py:199-201 (Gumbel-max), as in OracleModel.step
        u = torch.tensor(np.random.rand(n, nact))
        return torch.argmax(rep(pdparam, n) - torch.log(-torch.log(u)), dim=-1)

    q = PDPARAM_CATEGORICAL + np.random.randn(nact) * 0.1
    _validate(neglogp=lambda pdp, X: om._neglogp(rep(pdp, N_SAMPLES), X).numpy(),
              entropy=lambda pdp, n: om._entropy(rep(pdp, n)).numpy(),
              kl=lambda p, q_, n: _kl_categorical(rep(p, n), rep(q_, n)).numpy(),
              sample=sample, pdparam=PDPARAM_CATEGORICAL, q=q)
    # known answer: H = -sum p log p of softmax(pdparam)
    p = np.exp(PDPARAM_CATEGORICAL) / np.exp(PDPARAM_CATEGORICAL).sum()
    assert abs(float(om._entropy(rep(PDPARAM_CATEGORICAL, 1))[0]) - float(-(p * np.log(p)).sum())) < 1e-12
    assert abs(float(om._neglogp(rep(PDPARAM_CATEGORICAL, 1), torch.tensor([2]))[0]) + math.log(p[2])) < 1e-12


# ---- the network forward of the oracle against the DEFINING SUMS of the reference's layers -----------------------------
def _conv_defining_sum(x, w, b, stride):
    """a2c/utils.py:37-56 `conv`: tf.nn.conv2d(x, w, strides=[1, s, s, 1], padding='VALID', data_format='NHWC') + b with
    w[rf, rf, nin, nf] and b broadcast over [1, 1, 1, nf]; tf.nn.conv2d is defined (TF API documentation) as
        out[b, i, j, k] = sum_{di, dj, q} x[b, s*i + di, s*j + dj, q] * w[di, dj, q, k]
    -- restated with NumPy windows in float64, no library convolution involved."""
    rf = w.shape[0]
    win = np.lib.stride_tricks.sliding_window_view(x, (rf, rf), axis=(1, 2))[:, ::stride, ::stride]   # [B, OH, OW, C, rf, rf]
    return np.einsum('bijqde,deqk->bijk', win, w) + b.reshape(1, 1, 1, -1)


def test_oracle_network_forward_equals_the_defining_sums():
    """NatureCNN (common/models.py:15-26: /255, conv 8x8/4 -> 4x4/2 -> 3x3/1 with ReLU, conv_to_fc = NHWC flatten
    (a2c/utils.py:142-145), fc 512 with ReLU) and the heads (policies.py:43-64 via distributions.py:59-74: pi = latent @ w + b,
    vf = (latent @ w + b)[:, 0]) computed from their defining sums in float64 NumPy vs the oracle's torch restatement: pins the
    layout conventions (HWIO filters, NHWC flatten order, bias broadcast, scaling) that a wrong permute would silently break."""
    np.random.seed(3)
    om = OracleModel(network='cnn', ob_shape=(84, 84, 4), ob_dtype=np.uint8, pd_kind='categorical', nact=6,
                     dtype=torch.float64)
    rng = np.random.RandomState(4)
    with torch.no_grad():
        for k in om.names:                                   # biases are zero at init: make every term matter
            om.p[k] += torch.tensor(0.05 * rng.randn(*om.p[k].shape), dtype=torch.float64)
    P = {k: v.detach().numpy() for k, v in om.p.items()}
    obs = rng.randint(0, 256, (3, 84, 84, 4)).astype(np.uint8)
    h = obs.astype(np.float64) / 255.
    for name, s in (('c1', 4), ('c2', 2), ('c3', 1)):
        h = np.maximum(_conv_defining_sum(h, P['ppo2_model/pi/%s/w' % name], P['ppo2_model/pi/%s/b' % name], s), 0.)
    assert h.shape == (3, 7, 7, 64)
    lat = np.maximum(h.reshape(3, -1) @ P['ppo2_model/pi/fc1/w'] + P['ppo2_model/pi/fc1/b'], 0.)
    pi = lat @ P['ppo2_model/pi/w'] + P['ppo2_model/pi/b']
    vf = (lat @ P['ppo2_model/vf/w'] + P['ppo2_model/vf/b'])[:, 0]
    pi_o, vf_o = om.forward(obs)
    np.testing.assert_allclose(pi_o.detach().numpy(), pi, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(vf_o.detach().numpy(), vf, rtol=1e-10, atol=1e-12)
    # softmax cross-entropy / entropy of distributions.py:164-198 from their formulas
    a = np.array([0, 5, 2])
    a0 = pi - pi.max(axis=1, keepdims=True)
    z0 = np.exp(a0).sum(axis=1, keepdims=True)
    nlp = -(a0 - np.log(z0))[np.arange(3), a]
    ent = (np.exp(a0) / z0 * (np.log(z0) - a0)).sum(axis=1)
    np.testing.assert_allclose(om._neglogp(pi_o, torch.as_tensor(a)).detach().numpy(), nlp, rtol=1e-10)
    np.testing.assert_allclose(om._entropy(pi_o).detach().numpy(), ent, rtol=1e-10)


def test_oracle_loss_statistics_and_gradient_from_the_formulas_of_model_py():
    """ppo2/model.py:57-91 restated in float64 NumPy on the oracle's own network outputs (Gaussian policy: neglogp /
    entropy of distributions.py:238-246), and the gradient the oracle returns checked as a directional derivative of that
    loss by central differences -- the loss the gradients are taken of is the loss the statistics report."""
    np.random.seed(5)
    om = OracleModel(network='mlp', ob_shape=(11,), ob_dtype=np.float32, pd_kind='gaussian', nact=3, value_network='copy',
                     ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, dtype=torch.float64)
    rng = np.random.RandomState(6)
    with torch.no_grad():
        for k in om.names:
            om.p[k] += torch.tensor(0.1 * rng.randn(*om.p[k].shape), dtype=torch.float64)
    B, eps = 64, 0.2
    obs = rng.randn(B, 11).astype(np.float32)
    act = rng.randn(B, 3).astype(np.float32)
    ret, oldv = rng.randn(B), rng.randn(B) * 0.5
    oldnlp = rng.rand(B) * 2 + 2.0
    advs = om._normalised_advs(ret, oldv)
    np.testing.assert_allclose(advs, (ret - oldv - (ret - oldv).mean()) / ((ret - oldv).std() + 1e-8), rtol=1e-12)

    def loss_of():
        loss, stats = om.loss_and_stats(obs, ret, act, oldv, oldnlp, eps, advs)
        return loss, stats

    loss, stats = loss_of()
    pi, vf = (t.detach().numpy() for t in om.forward(obs))
    mean = pi                                                     # pdparam = [mean, mean * 0 + logstd] (distributions.py:102-106)
    logstd = np.zeros_like(mean) + om.p['ppo2_model/pi/logstd'].detach().numpy()
    nlp = 0.5 * (((act - mean) / np.exp(logstd)) ** 2).sum(-1) + 0.5 * np.log(2.0 * np.pi) * 3 + logstd.sum(-1)
    ent = (logstd + 0.5 * np.log(2.0 * np.pi * np.e)).sum(-1).mean()
    vclip = oldv + np.clip(vf - oldv, -eps, eps)
    vf_loss = 0.5 * np.maximum((vf - ret) ** 2, (vclip - ret) ** 2).mean()
    ratio = np.exp(oldnlp - nlp)
    pg_loss = np.maximum(-advs * ratio, -advs * np.clip(ratio, 1 - eps, 1 + eps)).mean()
    want = [pg_loss, vf_loss, ent, 0.5 * ((nlp - oldnlp) ** 2).mean(), (np.abs(ratio - 1) > eps).mean()]
    np.testing.assert_allclose([float(s) for s in stats], want, rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(float(loss), pg_loss - 0.01 * ent + 0.5 * vf_loss, rtol=1e-12)
    # directional derivative
    _, flat = om.compute_grads(eps, obs, ret, act, oldv, oldnlp, advs=advs)
    d = rng.randn(flat.numel())
    d /= np.linalg.norm(d)
    base = {k: om.p[k].detach().clone() for k in om.names}

    def shifted(h):
        off = 0
        with torch.no_grad():
            for k in om.names:
                n = base[k].numel()
                om.p[k].copy_(base[k] + h * torch.tensor(d[off:off + n].reshape(base[k].shape)))
                off += n
        return float(loss_of()[0])

    h = 1e-5
    fd = (shifted(h) - shifted(-h)) / (2 * h)
    shifted(0.0)
    assert abs(fd - float(flat.numpy() @ d)) <= 1e-7 * max(1.0, abs(fd)), (fd, float(flat.numpy() @ d))
    # tf.clip_by_global_norm: t * clip_norm / max(global_norm, clip_norm)
    gn = float(np.linalg.norm(flat.numpy()))
    np.testing.assert_allclose(om.average_and_clip(flat.clone()).numpy(), flat.numpy() * 0.5 / max(gn, 0.5), rtol=1e-12)


def test_oracle_lnlstm_equals_the_formulas_of_a2c_utils_in_numpy():
    """the layer-normalised cell of the oracle (oracle/ppo2_torch.py `_lstm` with layer_norm=True) against a float64 NumPy
    restatement written straight from a2c/utils.py:104-140 -- `_ln` = (x - mean) / sqrt(var + 1e-5) * g + b with the
    biased variance over axis 1, z = _ln(x@wx, gx, bx) + _ln(h@wh, gh, bh) + b, c = f c + i u, h = o tanh(_ln(c, gc, bc)),
    episode masks applied to (c, h) first -- forward values, final state, and the gradient of a random functional as a
    directional derivative (central differences in float64) against the oracle's autograd gradient"""
    import torch
    from oracle.ppo2_torch import OracleModel
    rng = np.random.RandomState(4)
    np.random.seed(4)
    nenv, T, nin, nh = 3, 5, 6, 8
    om = OracleModel(network='lstm', ob_shape=(nin,), ob_dtype=np.float32, pd_kind='categorical', nact=3, value_network=None,
                     nlstm=nh, layer_norm=True, dtype=torch.float64)
    with torch.no_grad():
        for k in om.names:
            om.p[k] += torch.tensor(0.3 * rng.randn(*om.p[k].shape), dtype=torch.float64)
    sc = 'ppo2_model/pi/lnlstm/'
    assert [n for n in om.names if 'lnlstm' in n] == [sc + s for s in ('wx', 'gx', 'bx', 'wh', 'gh', 'bh', 'b', 'gc', 'bc')]
    P = {k[len(sc):]: om.p[k].detach().numpy().copy() for k in om.names if k.startswith(sc)}
    feat = rng.randn(nenv * T, nin)                       # env-major rows like batch_to_seq expects
    M = (rng.rand(nenv * T) < 0.3).astype(np.float64)
    S0 = 0.5 * rng.randn(nenv, 2 * nh)

    def ln(x, g, b, e=1e-5):
        u = x.mean(axis=1, keepdims=True)
        s = ((x - u) ** 2).mean(axis=1, keepdims=True)
        return (x - u) / np.sqrt(s + e) * g + b

    def forward(P, feat):
        xs, ms = feat.reshape(nenv, T, nin), M.reshape(nenv, T)
        c, h = S0[:, :nh].copy(), S0[:, nh:].copy()
        out = np.zeros((nenv, T, nh))
        for t in range(T):
            m = ms[:, t:t + 1]
            c, h = c * (1 - m), h * (1 - m)
            z = ln(xs[:, t] @ P['wx'], P['gx'], P['bx']) + ln(h @ P['wh'], P['gh'], P['bh']) + P['b']
            i, f, o = (1.0 / (1.0 + np.exp(-z[:, k * nh:(k + 1) * nh])) for k in range(3))
            u = np.tanh(z[:, 3 * nh:])
            c = f * c + i * u
            h = o * np.tanh(ln(c, P['gc'], P['bc']))
            out[:, t] = h
        return out.reshape(nenv * T, nh), np.concatenate([c, h], axis=1)

    h_np, s_np = forward(P, feat)
    ft = torch.tensor(feat, dtype=torch.float64, requires_grad=True)
    h_t, s_t = om._lstm(ft, S0, M, nenv, 'ppo2_model/pi')
    np.testing.assert_allclose(h_t.detach().numpy(), h_np, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(s_t.detach().numpy(), s_np, rtol=1e-12, atol=1e-13)
    # gradient of L = <h, G> + <s, Gs> along a random direction in (parameters, features)
    G, Gs = rng.randn(*h_np.shape), rng.randn(*s_np.shape)
    L = (h_t * torch.tensor(G)).sum() + (s_t * torch.tensor(Gs)).sum()
    keys = list(P)
    grads = torch.autograd.grad(L, [om.p[sc + k] for k in keys] + [ft])
    D = {k: rng.randn(*P[k].shape) for k in keys}
    Df = rng.randn(*feat.shape)
    analytic = sum(float((g.numpy() * D[k]).sum()) for g, k in zip(grads[:-1], keys)) + float((grads[-1].numpy() * Df).sum())

    def Lnp(eps):
        Pp = {k: P[k] + eps * D[k] for k in keys}
        hh, ss = forward(Pp, feat + eps * Df)
        return float((hh * G).sum() + (ss * Gs).sum())

    eps = 1e-6
    numeric = (Lnp(eps) - Lnp(-eps)) / (2 * eps)
    assert abs(numeric - analytic) <= 1e-7 * max(1.0, abs(analytic)), (numeric, analytic)


def test_oracle_layer_norm_variants_equal_their_formulas_in_numpy():
    """tf.contrib.layers.layer_norm(center=True, scale=True) as the oracles restate it -- moments over the features of a row
    (biased variance), variance_epsilon 1e-12, gamma * xhat + beta -- in its two places: between every fc and its activation
    in `mlp(layer_norm=True)` (common/models.py:97-98) and between the hidden head layers and their ReLU in
    `build_q_func(layer_norm=True)` (deepq/models.py:24-41, dueling combination :42-44); float64 NumPy from the formulas"""
    import torch
    from oracle.ppo2_torch import OracleModel
    from oracle.dqn_torch import OracleQNet
    rng = np.random.RandomState(8)
    np.random.seed(8)

    def ln(x, g, b):
        u = x.mean(axis=1, keepdims=True)
        s = ((x - u) ** 2).mean(axis=1, keepdims=True)
        return (x - u) / np.sqrt(s + 1e-12) * g + b

    # ---- mlp(layer_norm=True): latent of the policy net
    om = OracleModel(network='mlp', ob_shape=(7,), ob_dtype=np.float32, pd_kind='categorical', nact=3, value_network=None,
                     num_layers=2, num_hidden=16, layer_norm=True, dtype=torch.float64)
    with torch.no_grad():
        for k in om.names:
            om.p[k] += torch.tensor(0.2 * rng.randn(*om.p[k].shape), dtype=torch.float64)
    P = {k: v.detach().numpy() for k, v in om.p.items()}
    x = rng.randn(9, 7)
    h = x
    for i in range(2):
        z = h @ P['ppo2_model/pi/mlp_fc%d/w' % i] + P['ppo2_model/pi/mlp_fc%d/b' % i]
        lnm = 'ppo2_model/pi/LayerNorm' + ('_%d' % i if i else '')
        h = np.tanh(ln(z, P[lnm + '/gamma'], P[lnm + '/beta']))
    np.testing.assert_allclose(om._net(torch.tensor(x), 'ppo2_model/pi').detach().numpy(), h, rtol=1e-12, atol=1e-13)

    # ---- Q heads with layer norm on mlp features, dueling
    tensors, off = [], 0
    def add(name, shape):
        nonlocal off
        n = int(np.prod(shape))
        tensors.append(dict(name=name, shape=tuple(shape), offset=off, size=n))
        off += n
    s = 'deepq/q_func'
    add(s + '/mlp_fc0/w', (5, 12)); add(s + '/mlp_fc0/b', (12,))
    for scope, nout in (('action_value', 4), ('state_value', 1)):
        add('%s/%s/fully_connected/weights' % (s, scope), (12, 10)); add('%s/%s/fully_connected/biases' % (s, scope), (10,))
        add('%s/%s/LayerNorm/beta' % (s, scope), (10,)); add('%s/%s/LayerNorm/gamma' % (s, scope), (10,))
        add('%s/%s/fully_connected_1/weights' % (s, scope), (10, nout)); add('%s/%s/fully_connected_1/biases' % (s, scope), (nout,))
    flat = 0.3 * rng.randn(off)
    oq = OracleQNet('mlp', tensors, flat, nact=4, hiddens=(10,), dueling=True, num_layers=1, activation='tanh',
                    dtype=torch.float64, layer_norm=True)
    W = {t['name']: flat[t['offset']:t['offset'] + t['size']].reshape(t['shape']) for t in tensors}
    obs = rng.randn(6, 5)
    feat = np.tanh(obs @ W[s + '/mlp_fc0/w'] + W[s + '/mlp_fc0/b'])

    def head(scope):
        pre = '%s/%s/' % (s, scope)
        o = feat @ W[pre + 'fully_connected/weights'] + W[pre + 'fully_connected/biases']
        o = np.maximum(ln(o, W[pre + 'LayerNorm/gamma'], W[pre + 'LayerNorm/beta']), 0.0)
        return o @ W[pre + 'fully_connected_1/weights'] + W[pre + 'fully_connected_1/biases']

    a, v = head('action_value'), head('state_value')
    np.testing.assert_allclose(oq.q_values(obs), v + (a - a.mean(axis=1, keepdims=True)), rtol=1e-12, atol=1e-13)


def test_validate_probtype_multicategorical_on_oracle():
    """distributions.py:310-313: MultiCategoricalPdType([1, 2, 3]) on the reference's pdparam"""
    np.random.seed(0)
    om = _oracle_for('multicategorical', PDPARAM_MULTICATEGORICAL.size, NVEC_MULTICATEGORICAL)
    rep = lambda v, n: torch.tensor(np.repeat(v[None, :], n, axis=0))

    def sample(pdparam, n):
        u = torch.tensor(np.random.rand(n, pdparam.size))
        g = rep(pdparam, n) - torch.log(-torch.log(u))
        return torch.stack([torch.argmax(x, dim=-1) for x in om._slices(g)], dim=-1)
    q = PDPARAM_MULTICATEGORICAL + np.random.randn(PDPARAM_MULTICATEGORICAL.size) * 0.1
    _validate(neglogp=lambda pdp, X: om._neglogp(rep(pdp, N_SAMPLES), X).numpy(),
              entropy=lambda pdp, n: om._entropy(rep(pdp, n)).numpy(),
              kl=lambda p, q_, n: _kl_multicategorical(rep(p, n), rep(q_, n), NVEC_MULTICATEGORICAL).numpy(),
              sample=sample, pdparam=PDPARAM_MULTICATEGORICAL, q=q)
    # known answer: entropies of softmax([.3, .5]) and softmax([.1, 1, -.1]); the one-class slice contributes 0
    ent = 0.0
    for l in (np.array([.3, .5]), np.array([.1, 1, -.1])):
        p = np.exp(l) / np.exp(l).sum()
        ent += float(-(p * np.log(p)).sum())
    assert abs(float(om._entropy(rep(PDPARAM_MULTICATEGORICAL, 1))[0]) - ent) < 1e-12


def test_validate_probtype_bernoulli_on_oracle():
    """distributions.py:315-317: BernoulliPdType(3) on the reference's pdparam"""
    np.random.seed(0)
    om = _oracle_for('bernoulli', PDPARAM_BERNOULLI.size)
    rep = lambda v, n: torch.tensor(np.repeat(v[None, :], n, axis=0))
    sample = lambda pdparam, n: (torch.tensor(np.random.rand(n, pdparam.size)) < torch.sigmoid(rep(pdparam, n))).double()
    q = PDPARAM_BERNOULLI + np.random.randn(PDPARAM_BERNOULLI.size) * 0.1
    _validate(neglogp=lambda pdp, X: om._neglogp(rep(pdp, N_SAMPLES), X).numpy(),
              entropy=lambda pdp, n: om._entropy(rep(pdp, n)).numpy(),
              kl=lambda p, q_, n: _kl_bernoulli(rep(p, n), rep(q_, n)).numpy(),
              sample=sample, pdparam=PDPARAM_BERNOULLI, q=q)
    p = 1.0 / (1.0 + np.exp(-PDPARAM_BERNOULLI))
    assert abs(float(om._entropy(rep(PDPARAM_BERNOULLI, 1))[0]) - float(-(p * np.log(p) + (1 - p) * np.log(1 - p)).sum())) < 1e-12
