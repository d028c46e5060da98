"""`validate_probtype` of the reference (common/distributions.py:320-348, fixed pdparams :303-311) run on the GPU
act kernel (`mrl_model_act`: Gumbel-max / reparameterised Gaussian sampling + neglogp, distributions.py:199-201,
238-248) and on the entropy the training kernel reports (`mrl_model_grad` stats[2], distributions.py:193-198, 245-246):

    entropy      == -E_p[log p]                 within 3 sigma over N = 100,000 samples
    KL[p, q]     == -H[p] - E_p[log q]          within 3 sigma  (kl() formulas of the reference restated in
                                                 tests/test_oracle_pins.py)

The policy is pinned to the test's pdparam by zeroing the pi-head weights and putting the pdparam into the head bias
(and `logstd`): logits / mean are then the same for every observation.  log q of the SAME samples is obtained by
teacher-forcing the sampler: noise chosen so that the q-policy reproduces exactly those actions.
"""
import numpy as np
import pytest
import torch

from tests.test_oracle_pins import (PDPARAM_CATEGORICAL, PDPARAM_DIAG_GAUSS, PDPARAM_MULTICATEGORICAL, NVEC_MULTICATEGORICAL,
                                    PDPARAM_BERNOULLI, N_SAMPLES, _kl_bernoulli, _kl_categorical, _kl_gaussian,
                                    _kl_multicategorical)

pytestmark = pytest.mark.gpu


def _pinned_model(pd_kind, nact, nvec=None):
    from baselines_amd import ops
    dm = ops.DeviceModel(network='mlp', ob_shape=(4,), ob_dtype=np.float32, pd_kind=pd_kind, nact=nact, nvec=nvec, chunk=N_SAMPLES)
    rng = np.random.RandomState(0)
    flat = (0.1 * rng.randn(dm.P)).astype(np.float32)
    off = {t['name']: (t['offset'], t['size']) for t in dm.tensors}
    o, n = off['ppo2_model/pi/w']
    flat[o:o + n] = 0.0
    return dm, flat, off


def _set(flat, off, name, values):
    o, n = off[name]
    flat[o:o + n] = np.asarray(values, np.float32).reshape(-1)


def _obs(n):
    return torch.from_numpy(np.random.RandomState(1).randn(n, 4).astype(np.float32)).cuda()


def _entropy_from_train_kernel(dm, params, obs, actions, nlp):
    n = obs.shape[0]
    z = torch.zeros(n, dtype=torch.float32, device='cuda')
    grads = torch.empty(dm.P, dtype=torch.float32, device='cuda')
    stats = torch.empty(5, dtype=torch.float32, device='cuda')
    dm.grad(params, obs, actions, z, z, nlp, None, n, 1, 1, 0.2, 0.0, 0.5, grads, stats)
    return float(stats[2].cpu())


def test_validate_probtype_categorical_on_gpu_kernels():
    np.random.seed(0)
    N, nact = N_SAMPLES, PDPARAM_CATEGORICAL.size
    dm, flat, off = _pinned_model('categorical', nact)
    q = PDPARAM_CATEGORICAL + np.random.randn(nact) * 0.1
    obs = _obs(N)
    gen = torch.Generator(device='cuda')
    gen.manual_seed(7)
    _set(flat, off, 'ppo2_model/pi/b', PDPARAM_CATEGORICAL)
    p_params = torch.from_numpy(flat.copy()).cuda()
    u = torch.rand((N, nact), generator=gen, device='cuda', dtype=torch.float32).clamp_(1e-7, 1 - 1e-7)
    X, _, nlp_p, pdp = dm.act(p_params, obs, u, want_pdparam=True)
    np.testing.assert_allclose(pdp.cpu().numpy(), np.repeat(PDPARAM_CATEGORICAL[None], N, 0), atol=1e-6)
    # empirical action frequencies are the softmax probabilities (sampler check, 5 sigma binomial)
    p = np.exp(PDPARAM_CATEGORICAL) / np.exp(PDPARAM_CATEGORICAL).sum()
    freq = np.bincount(X.cpu().numpy(), minlength=nact) / N
    assert np.all(np.abs(freq - p) < 5 * np.sqrt(p * (1 - p) / N)), (freq, p)
    logliks = -nlp_p.cpu().numpy().astype(np.float64)
    entval = _entropy_from_train_kernel(dm, p_params, obs, X, nlp_p)
    assert abs(entval - float(-(p * np.log(p)).sum())) < 2e-6                      # known answer
    assert abs(entval - (-logliks.mean())) < 3 * logliks.std() / np.sqrt(N)        # distributions.py:331-334
    # log q of the same samples: uniforms that make the Gumbel-max sampler of the q-policy pick X again
    _set(flat, off, 'ppo2_model/pi/b', q)
    q_params = torch.from_numpy(flat.copy()).cuda()
    forced = torch.full((N, nact), 1e-6, dtype=torch.float32, device='cuda')
    forced[torch.arange(N, device='cuda'), X.long()] = 1 - 1e-6
    Xq, _, nlp_q, _ = dm.act(q_params, obs, forced)
    assert torch.equal(Xq, X)
    logq = -nlp_q.cpu().numpy().astype(np.float64)
    rep = lambda v: torch.tensor(np.repeat(v[None, :], 4, axis=0))
    klval = float(_kl_categorical(rep(PDPARAM_CATEGORICAL), rep(q))[0])
    klval_ll = -entval - logq.mean()
    assert abs(klval - klval_ll) < 3 * logq.std() / np.sqrt(N)                     # distributions.py:337-347


def test_validate_probtype_diag_gaussian_on_gpu_kernels():
    np.random.seed(0)
    N, nact = N_SAMPLES, PDPARAM_DIAG_GAUSS.size // 2
    mean_p, logstd_p = PDPARAM_DIAG_GAUSS[:nact], PDPARAM_DIAG_GAUSS[nact:]
    dm, flat, off = _pinned_model('gaussian', nact)
    q = PDPARAM_DIAG_GAUSS + np.random.randn(PDPARAM_DIAG_GAUSS.size) * 0.1
    mean_q, logstd_q = q[:nact], q[nact:]
    obs = _obs(N)
    gen = torch.Generator(device='cuda')
    gen.manual_seed(8)
    _set(flat, off, 'ppo2_model/pi/b', mean_p)
    _set(flat, off, 'ppo2_model/pi/logstd', logstd_p)
    p_params = torch.from_numpy(flat.copy()).cuda()
    nz = torch.randn((N, nact), generator=gen, device='cuda', dtype=torch.float32)
    X, _, nlp_p, pdp = dm.act(p_params, obs, nz, want_pdparam=True)
    np.testing.assert_allclose(pdp.cpu().numpy(), np.repeat(mean_p[None], N, 0), atol=1e-6)
    x = X.cpu().numpy().astype(np.float64)
    assert np.all(np.abs(x.mean(0) - mean_p) < 5 * np.exp(logstd_p) / np.sqrt(N))
    assert np.all(np.abs(x.std(0) / np.exp(logstd_p) - 1) < 5 / np.sqrt(2 * N))
    logliks = -nlp_p.cpu().numpy().astype(np.float64)
    entval = _entropy_from_train_kernel(dm, p_params, obs, X, nlp_p)
    assert abs(entval - float((logstd_p + .5 * np.log(2.0 * np.pi * np.e)).sum())) < 2e-6   # known answer
    assert abs(entval - (-logliks.mean())) < 3 * logliks.std() / np.sqrt(N)
    # log q of the same samples: x = mean_q + std_q * noise  <=>  noise = (x - mean_q) / std_q
    _set(flat, off, 'ppo2_model/pi/b', mean_q)
    _set(flat, off, 'ppo2_model/pi/logstd', logstd_q)
    q_params = torch.from_numpy(flat.copy()).cuda()
    forced = ((X.double() - torch.tensor(mean_q, device='cuda')) / torch.tensor(np.exp(logstd_q), device='cuda')).float()
    Xq, _, nlp_q, _ = dm.act(q_params, obs, forced.contiguous())
    assert float((Xq - X).abs().max()) < 1e-5
    logq = -nlp_q.cpu().numpy().astype(np.float64)
    rep = lambda v: torch.tensor(np.repeat(v[None, :], 4, axis=0))
    klval = float(_kl_gaussian(rep(mean_p), rep(logstd_p), rep(mean_q), rep(logstd_q))[0])
    klval_ll = -entval - logq.mean()
    assert abs(klval - klval_ll) < 3 * logq.std() / np.sqrt(N)


def test_validate_probtype_multicategorical_on_gpu_kernels():
    """distributions.py:310-313: MultiCategoricalPdType([1, 2, 3]) -- one Gumbel-max per slice of the flat logits, neglogp and
    entropy summed over the slices"""
    np.random.seed(0)
    N, nact, nvec = N_SAMPLES, PDPARAM_MULTICATEGORICAL.size, NVEC_MULTICATEGORICAL
    dm, flat, off = _pinned_model('multicategorical', nact, nvec)
    q = PDPARAM_MULTICATEGORICAL + np.random.randn(nact) * 0.1
    obs = _obs(N)
    gen = torch.Generator(device='cuda')
    gen.manual_seed(9)
    _set(flat, off, 'ppo2_model/pi/b', PDPARAM_MULTICATEGORICAL)
    p_params = torch.from_numpy(flat.copy()).cuda()
    u = torch.rand((N, nact), generator=gen, device='cuda', dtype=torch.float32).clamp_(1e-7, 1 - 1e-7)
    X, _, nlp_p, pdp = dm.act(p_params, obs, u, want_pdparam=True)
    assert X.dtype == torch.int32 and tuple(X.shape) == (N, len(nvec))
    np.testing.assert_allclose(pdp.cpu().numpy(), np.repeat(PDPARAM_MULTICATEGORICAL[None], N, 0), atol=1e-6)
    x = X.cpu().numpy()
    o, ent = 0, 0.0
    for k, nv in enumerate(nvec):                                  # per-slice frequencies = that slice's softmax (5 sigma)
        l = PDPARAM_MULTICATEGORICAL[o:o + nv]
        p = np.exp(l) / np.exp(l).sum()
        freq = np.bincount(x[:, k], minlength=nv) / N
        assert np.all(np.abs(freq - p) <= 5 * np.sqrt(p * (1 - p) / N) + 1e-12), (k, freq, p)
        ent += float(-(p * np.log(p)).sum())
        o += nv
    logliks = -nlp_p.cpu().numpy().astype(np.float64)
    entval = _entropy_from_train_kernel(dm, p_params, obs, X, nlp_p)
    assert abs(entval - ent) < 3e-6                                                # known answer
    assert abs(entval - (-logliks.mean())) < 3 * logliks.std() / np.sqrt(N)        # distributions.py:331-334
    # log q of the same samples: uniforms that make every slice's Gumbel-max pick X again
    _set(flat, off, 'ppo2_model/pi/b', q)
    q_params = torch.from_numpy(flat.copy()).cuda()
    forced = torch.full((N, nact), 1e-6, dtype=torch.float32, device='cuda')
    o = 0
    for k, nv in enumerate(nvec):
        forced[torch.arange(N, device='cuda'), o + X[:, k].long()] = 1 - 1e-6
        o += nv
    Xq, _, nlp_q, _ = dm.act(q_params, obs, forced)
    assert torch.equal(Xq, X)
    logq = -nlp_q.cpu().numpy().astype(np.float64)
    rep = lambda v: torch.tensor(np.repeat(v[None, :], 4, axis=0))
    klval = float(_kl_multicategorical(rep(PDPARAM_MULTICATEGORICAL), rep(q), nvec)[0])
    klval_ll = -entval - logq.mean()
    assert abs(klval - klval_ll) < 3 * logq.std() / np.sqrt(N)                     # distributions.py:337-347


def test_validate_probtype_bernoulli_on_gpu_kernels():
    """distributions.py:315-317: BernoulliPdType(3) -- x = (u < sigmoid(logit)), sigmoid cross-entropies summed over the bits"""
    np.random.seed(0)
    N, nact = N_SAMPLES, PDPARAM_BERNOULLI.size
    dm, flat, off = _pinned_model('bernoulli', nact)
    q = PDPARAM_BERNOULLI + np.random.randn(nact) * 0.1
    obs = _obs(N)
    gen = torch.Generator(device='cuda')
    gen.manual_seed(10)
    _set(flat, off, 'ppo2_model/pi/b', PDPARAM_BERNOULLI)
    p_params = torch.from_numpy(flat.copy()).cuda()
    u = torch.rand((N, nact), generator=gen, device='cuda', dtype=torch.float32)
    X, _, nlp_p, pdp = dm.act(p_params, obs, u, want_pdparam=True)
    assert X.dtype == torch.int32 and tuple(X.shape) == (N, nact)
    p = 1.0 / (1.0 + np.exp(-PDPARAM_BERNOULLI))
    freq = X.cpu().numpy().mean(0)
    assert np.all(np.abs(freq - p) < 5 * np.sqrt(p * (1 - p) / N)), (freq, p)
    logliks = -nlp_p.cpu().numpy().astype(np.float64)
    entval = _entropy_from_train_kernel(dm, p_params, obs, X, nlp_p)
    assert abs(entval - float(-(p * np.log(p) + (1 - p) * np.log(1 - p)).sum())) < 3e-6
    assert abs(entval - (-logliks.mean())) < 3 * logliks.std() / np.sqrt(N)
    # log q of the same samples: u = 0 reproduces a 1, u = 1 a 0 whatever the probability
    _set(flat, off, 'ppo2_model/pi/b', q)
    q_params = torch.from_numpy(flat.copy()).cuda()
    forced = (1 - X).float().contiguous()
    Xq, _, nlp_q, _ = dm.act(q_params, obs, forced)
    assert torch.equal(Xq, X)
    logq = -nlp_q.cpu().numpy().astype(np.float64)
    rep = lambda v: torch.tensor(np.repeat(v[None, :], 4, axis=0))
    klval = float(_kl_bernoulli(rep(PDPARAM_BERNOULLI), rep(q))[0])
    klval_ll = -entval - logq.mean()
    assert abs(klval - klval_ll) < 3 * logq.std() / np.sqrt(N)
