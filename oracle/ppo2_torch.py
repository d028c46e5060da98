"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PyTorch-CPU restatement of the TensorFlow-1 graph the reference PPO2 builds
(tensorflow<2 is a third-party dependency that is absent from /root/reference
and from this image -> "parity unpinned" at the TF boundary).
Paths below are relative to /root/reference/baselines/.

  networks        common/models.py:15-26 (nature_cnn), :74-103 (mlp);
                  layers a2c/utils.py:37-63 (conv: NHWC/HWIO/VALID, fc), :142-145
  policy heads    common/policies.py:43-64, common/distributions.py:59-113, 351-355
  pd maths        common/distributions.py:153-204 (Categorical), :227-251 (DiagGaussian)
  loss            ppo2/model.py:57-91
  clip + Adam     ppo2/model.py:97-114 (tf.clip_by_global_norm, tf.train.AdamOptimizer eps=1e-5);
                  Adam update restated from the published TF-1 ApplyAdam kernel form
                  (m += (g-m)(1-b1); v += (g*g-v)(1-b2); var -= m*alpha/(sqrt(v)+eps),
                  alpha = lr*sqrt(1-b2^t)/(1-b1^t)), cross-pinned by common/mpi_adam.py:38-41.
  MPI averaging   common/mpi_adam_optimizer.py:21,39-40 (flat grad * w, sum, / sum w; THEN clip)
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .ppo2_numpy import ortho_init


NATURE_CONVS = ((32, 8, 4), (64, 4, 2), (64, 3, 1))        # common/models.py:21-23; cnn_small (:123-124): ((8, 8, 4), (16, 4, 2)), fc 128


def conv_out(n, k, s, pad):
    """tf.nn.conv2d output size: VALID floor((n - k) / s) + 1, SAME ceil(n / s)"""
    return (n + s - 1) // s if pad == 'SAME' else (n - k) // s + 1


def build_param_specs(network, ob_shape, pd_kind, nact, value_network=None,
                      num_layers=2, num_hidden=64, nlstm=128, layer_norm=False, convs=None, fc_hidden=512, pad='VALID'):
    """Ordered (name, shape, init_scale|None) list in TF variable-creation order
    (SURVEY.md App. A.6): policy net -> value net copy -> pi head -> [logstd] -> vf head."""
    specs = []

    def net(prefix):
        if network == 'cnn':
            h, w, c = ob_shape
            for i, (nf, rf, st) in enumerate(convs or NATURE_CONVS):
                specs.append((prefix + '/c%d/w' % (i + 1), (rf, rf, c, nf), math.sqrt(2)))
                specs.append((prefix + '/c%d/b' % (i + 1), (1, nf, 1, 1), None))
                h, w, c = conv_out(h, rf, st, pad), conv_out(w, rf, st, pad), nf
            specs.append((prefix + '/fc1/w', (h * w * c, fc_hidden), math.sqrt(2)))
            specs.append((prefix + '/fc1/b', (fc_hidden,), None))
            return fc_hidden
        elif network in ('lstm', 'cnn_lstm'):
            # common/models.py:132-210: [nature_cnn ->] a2c/utils.py:81-102 lstm(scope='lstm', init_scale=1.0):
            # wx [nin, 4nh], wh [nh, 4nh] orthogonal, b [4nh] zeros; created in that order after the conv stack
            if network == 'cnn_lstm':
                h, w, c = ob_shape
                for i, (nf, rf, st) in enumerate(convs or NATURE_CONVS):
                    specs.append((prefix + '/c%d/w' % (i + 1), (rf, rf, c, nf), math.sqrt(2)))
                    specs.append((prefix + '/c%d/b' % (i + 1), (1, nf, 1, 1), None))
                    h, w, c = conv_out(h, rf, st, pad), conv_out(w, rf, st, pad), nf
                specs.append((prefix + '/fc1/w', (h * w * c, fc_hidden), math.sqrt(2)))
                specs.append((prefix + '/fc1/b', (fc_hidden,), None))
                nin = fc_hidden
            else:
                nin = int(np.prod(ob_shape))
            if layer_norm:
                # a2c/utils.py:110-124 lnlstm(scope='lnlstm'): wx, gx (ones), bx, wh, gh (ones), bh, b, gc (ones), bc
                sc = prefix + '/lnlstm'
                specs.extend([(sc + '/wx', (nin, 4 * nlstm), 1.0), (sc + '/gx', (4 * nlstm,), 'ones'), (sc + '/bx', (4 * nlstm,), None),
                              (sc + '/wh', (nlstm, 4 * nlstm), 1.0), (sc + '/gh', (4 * nlstm,), 'ones'), (sc + '/bh', (4 * nlstm,), None),
                              (sc + '/b', (4 * nlstm,), None), (sc + '/gc', (nlstm,), 'ones'), (sc + '/bc', (nlstm,), None)])
                return nlstm
            specs.append((prefix + '/lstm/wx', (nin, 4 * nlstm), 1.0))
            specs.append((prefix + '/lstm/wh', (nlstm, 4 * nlstm), 1.0))
            specs.append((prefix + '/lstm/b', (4 * nlstm,), None))
            return nlstm
        elif network == 'mlp':
            nin = int(np.prod(ob_shape))
            for i in range(num_layers):
                specs.append((prefix + '/mlp_fc%d/w' % i, (nin, num_hidden), math.sqrt(2)))
                specs.append((prefix + '/mlp_fc%d/b' % i, (num_hidden,), None))
                if layer_norm:
                    # common/models.py:97-98: tf.contrib.layers.layer_norm(h, center=True, scale=True) -- variables
                    # LayerNorm[_i]/beta (zeros), LayerNorm[_i]/gamma (ones), created in that order in the enclosing scope
                    ln = prefix + ('/LayerNorm' if i == 0 else '/LayerNorm_%d' % i)
                    specs.append((ln + '/beta', (num_hidden,), None))
                    specs.append((ln + '/gamma', (num_hidden,), 'ones'))
                nin = num_hidden
            return nin
        raise ValueError('Unknown network type: {}'.format(network))

    nlat = net('ppo2_model/pi')
    nlat_v = nlat
    if value_network == 'copy':
        # policies.py:162-165: "recurrent architectures are not supported with value_network=copy yet"
        assert network not in ('lstm', 'cnn_lstm')
        nlat_v = net('ppo2_model/vf')
    # distributions.py:351-355 _matching_fc: no head when latent width == size
    has_pi_head = (nlat != nact)
    if has_pi_head:
        specs.append(('ppo2_model/pi/w', (nlat, nact), 0.01))
        specs.append(('ppo2_model/pi/b', (nact,), None))
    if pd_kind == 'gaussian':
        specs.append(('ppo2_model/pi/logstd', (1, nact), None))
    specs.append(('ppo2_model/vf/w', (nlat_v, 1), 1.0))
    specs.append(('ppo2_model/vf/b', (1,), None))
    return specs, has_pi_head


def init_params(specs):
    """Draw order == spec order; only `w` tensors consume np.random (a2c/utils.py:20-35)."""
    out = {}
    for name, shape, scale in specs:
        if scale is None:
            out[name] = np.zeros(shape, np.float32)
        elif scale == 'ones':
            out[name] = np.ones(shape, np.float32)
        else:
            out[name] = ortho_init(shape, scale)
    return out


class OracleModel(object):
    """Reference-shaped model object (ppo2/model.py:27-158) on torch-CPU.

    step/value take explicit noise (teacher forcing: TF's Philox stream cannot be
    reproduced) -- uniform [N,nA] for Categorical Gumbel-max
    (distributions.py:199-201), standard normal [N,nA] for DiagGaussian (:247-248).
    """
    loss_names = ['policy_loss', 'value_loss', 'policy_entropy', 'approxkl', 'clipfrac']

    def __init__(self, *, network, ob_shape, ob_dtype, pd_kind, nact, value_network=None,
                 ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, dtype=torch.float32,
                 params=None, num_layers=2, num_hidden=64, total_weight=1.0, rank_weight=1.0,
                 allreduce=None, nlstm=128, layer_norm=False, convs=None, fc_hidden=512, pad='VALID', nvec=None):
        self.nvec = tuple(int(v) for v in nvec) if nvec is not None else None      # MultiCategorical: slices of the flat logits
        self.network, self.ob_shape, self.ob_dtype = network, tuple(ob_shape), np.dtype(ob_dtype)
        self.pd_kind, self.nact, self.value_network = pd_kind, nact, value_network
        self.ent_coef, self.vf_coef, self.max_grad_norm = ent_coef, vf_coef, max_grad_norm
        self.dtype = dtype
        self.num_layers, self.num_hidden, self.nlstm = num_layers, num_hidden, nlstm
        self.layer_norm = layer_norm
        self.convs, self.fc_hidden, self.pad = tuple(convs) if convs else NATURE_CONVS, fc_hidden, pad
        self.recurrent = network in ('lstm', 'cnn_lstm')
        self.specs, self.has_pi_head = build_param_specs(network, ob_shape, pd_kind, nact, value_network,
                                                         num_layers, num_hidden, nlstm, layer_norm, convs, fc_hidden, pad)
        if params is None:
            params = init_params(self.specs)
        self.names = [s[0] for s in self.specs]
        self.p = {k: torch.tensor(np.asarray(params[k]), dtype=dtype, requires_grad=True) for k in self.names}
        self.m = {k: torch.zeros_like(self.p[k]) for k in self.names}
        self.v = {k: torch.zeros_like(self.p[k]) for k in self.names}
        # TF keeps beta powers as f32 variables multiplied after each apply
        npdt = np.float32 if dtype == torch.float32 else np.float64
        self.beta1, self.beta2, self.eps = npdt(0.9), npdt(0.999), npdt(1e-5)
        self.beta1_power, self.beta2_power = npdt(0.9), npdt(0.999)
        self.npdt = npdt
        self.rank_weight, self.total_weight, self.allreduce = rank_weight, total_weight, allreduce
        # common/models.py:171: initial_state = np.zeros([nenv, 2*nlstm]) of the act model (sized by the caller)
        self.initial_state = None
        self.last_grads = None

    # ---- networks ---------------------------------------------------------
    def _lstm(self, feat, S, M, nenv, prefix):
        """common/models.py:158-170 + a2c/utils.py:65-102: feat [nenv*nsteps, nin] in ENV-MAJOR order (batch_to_seq
        reshapes to [nenv, nsteps, -1]), S [nenv, 2nh] (c | h), M [nenv*nsteps] -> (h [nenv*nsteps, nh], snew)"""
        p = self.p
        nh = self.nlstm
        nsteps = feat.shape[0] // nenv
        xs = feat.reshape(nenv, nsteps, -1)
        ms = torch.as_tensor(np.asarray(M, dtype=np.float64)).to(self.dtype).reshape(nenv, nsteps)
        S = torch.as_tensor(np.asarray(S)).to(self.dtype)
        c, h = S[:, :nh], S[:, nh:]
        ln = self.layer_norm
        sc = prefix + ('/lnlstm' if ln else '/lstm')
        wx, wh, b = p[sc + '/wx'], p[sc + '/wh'], p[sc + '/b']

        def _ln(v, g, bb, e=1e-5):
            # a2c/utils.py:104-108: moments over axis 1 (biased variance), (v - u) / sqrt(s + e) * g + b
            u = v.mean(dim=1, keepdim=True)
            sv = ((v - u) ** 2).mean(dim=1, keepdim=True)
            return (v - u) / torch.sqrt(sv + e) * g + bb

        out = []
        for t in range(nsteps):
            m = ms[:, t:t + 1]
            c = c * (1 - m)
            h = h * (1 - m)
            if ln:                                                      # a2c/utils.py:130
                z = _ln(xs[:, t] @ wx, p[sc + '/gx'], p[sc + '/bx']) + _ln(h @ wh, p[sc + '/gh'], p[sc + '/bh']) + b
            else:
                z = xs[:, t] @ wx + h @ wh + b
            i, f, o, u = torch.sigmoid(z[:, :nh]), torch.sigmoid(z[:, nh:2 * nh]), torch.sigmoid(z[:, 2 * nh:3 * nh]), \
                torch.tanh(z[:, 3 * nh:])
            c = f * c + i * u
            h = o * torch.tanh(_ln(c, p[sc + '/gc'], p[sc + '/bc']) if ln else c)      # a2c/utils.py:137
            out.append(h)
        return torch.stack(out, dim=1).reshape(nenv * nsteps, nh), torch.cat([c, h], dim=1)

    def _net(self, x, prefix):
        p = self.p
        if self.network in ('cnn', 'cnn_lstm'):
            # models.py:19: tf.cast(float32) / 255.
            h = x.to(self.dtype) / 255.
            h = h.permute(0, 3, 1, 2)
            for i, (nf, rf, stride) in enumerate(self.convs):
                name = 'c%d' % (i + 1)
                w = p[prefix + '/%s/w' % name].permute(3, 2, 0, 1)  # HWIO -> OIHW
                b = p[prefix + '/%s/b' % name].reshape(-1)
                if self.pad == 'SAME':     # tf.nn.conv2d SAME: total = max((ceil(n/s) - 1) s + k - n, 0), the smaller half in front
                    def halves(n):
                        tot = max(((n + stride - 1) // stride - 1) * stride + rf - n, 0)
                        return tot // 2, tot - tot // 2
                    (pt, pb), (pl, pr) = halves(h.shape[2]), halves(h.shape[3])
                    h = F.pad(h, (pl, pr, pt, pb))
                h = F.relu(F.conv2d(h, w, b, stride=stride))
            h = h.permute(0, 2, 3, 1).reshape(h.shape[0], -1)  # conv_to_fc on NHWC
            return F.relu(h @ p[prefix + '/fc1/w'] + p[prefix + '/fc1/b'])
        elif self.network == 'lstm':
            return x.to(self.dtype).reshape(x.shape[0], -1)       # tf.layers.flatten (models.py:150)
        else:
            h = x.to(self.dtype).reshape(x.shape[0], -1)
            for i in range(self.num_layers):
                h = h @ p[prefix + '/mlp_fc%d/w' % i] + p[prefix + '/mlp_fc%d/b' % i]
                if self.layer_norm:
                    # tf.contrib.layers.layer_norm: tf.nn.moments over the features (biased variance), variance_epsilon 1e-12,
                    # tf.nn.batch_normalization(x, mean, var, offset=beta, scale=gamma, eps)
                    ln = prefix + ('/LayerNorm' if i == 0 else '/LayerNorm_%d' % i)
                    mean = h.mean(dim=1, keepdim=True)
                    var = ((h - mean) ** 2).mean(dim=1, keepdim=True)
                    h = (h - mean) * torch.rsqrt(var + 1e-12) * p[ln + '/gamma'] + p[ln + '/beta']
                h = torch.tanh(h)
            return h

    def forward(self, obs, S=None, M=None, nenv=None):
        """recurrent networks: obs rows in env-major order, S [nenv, 2nh], M [nenv*nsteps]; the new state is left in
        self.last_state (PolicyWithValue.state, policies.py:77-96)"""
        x = torch.as_tensor(np.asarray(obs))
        lat = self._net(x, 'ppo2_model/pi')
        if self.recurrent:
            lat, self.last_state = self._lstm(lat, S, M, nenv if nenv is not None else x.shape[0], 'ppo2_model/pi')
        vlat = self._net(x, 'ppo2_model/vf') if self.value_network == 'copy' else lat
        if self.has_pi_head:
            pi = lat @ self.p['ppo2_model/pi/w'] + self.p['ppo2_model/pi/b']
        else:
            pi = lat
        vf = (vlat @ self.p['ppo2_model/vf/w'] + self.p['ppo2_model/vf/b'])[:, 0]
        return pi, vf

    # ---- pd maths -----------------------------------------------------------
    def _slices(self, pi):
        """tf.split(flat, nvec, axis=-1) (distributions.py:209)"""
        out, o = [], 0
        for nv in self.nvec:
            out.append(pi[:, o:o + nv])
            o += nv
        return out

    def _neglogp(self, pi, a):
        if self.pd_kind == 'multicategorical':
            # distributions.py:215-216: add_n of the slices' Categorical neglogp
            a = a.long().reshape(pi.shape[0], -1)
            tot = 0.0
            for q, l in enumerate(self._slices(pi)):
                a0 = l - l.max(dim=-1, keepdim=True)[0]
                tot = tot + torch.log(torch.exp(a0).sum(-1)) - a0.gather(1, a[:, q:q + 1])[:, 0]
            return tot
        if self.pd_kind == 'bernoulli':
            # distributions.py:264-265: sum of sigmoid_cross_entropy_with_logits(logits, labels = x) =
            # max(l, 0) - l x + log(1 + exp(-|l|))
            x = a.to(pi.dtype).reshape(pi.shape)
            return (torch.clamp(pi, min=0) - pi * x + torch.log1p(torch.exp(-pi.abs()))).sum(-1)
        if self.pd_kind == 'categorical':
            # softmax_cross_entropy_with_logits_v2(logits, onehot)
            a0 = pi - pi.max(dim=-1, keepdim=True)[0]
            lse = torch.log(torch.exp(a0).sum(-1))
            return lse - a0.gather(1, a.long().reshape(-1, 1))[:, 0]
        logstd = self.p['ppo2_model/pi/logstd']
        logstd_b = pi * 0.0 + logstd
        std = torch.exp(logstd_b)
        return 0.5 * (((a - pi) / std) ** 2).sum(-1) + 0.5 * math.log(2.0 * math.pi) * float(self.nact) \
            + logstd_b.sum(-1)

    def _entropy(self, pi):
        if self.pd_kind == 'multicategorical':       # distributions.py:219-220
            tot = 0.0
            for l in self._slices(pi):
                a0 = l - l.max(dim=-1, keepdim=True)[0]
                ea0 = torch.exp(a0)
                z0 = ea0.sum(-1, keepdim=True)
                tot = tot + ((ea0 / z0) * (torch.log(z0) - a0)).sum(-1)
            return tot
        if self.pd_kind == 'bernoulli':              # distributions.py:268-269: labels = sigmoid(logits), differentiable
            ps = torch.sigmoid(pi)
            return (torch.clamp(pi, min=0) - pi * ps + torch.log1p(torch.exp(-pi.abs()))).sum(-1)
        if self.pd_kind == 'categorical':
            a0 = pi - pi.max(dim=-1, keepdim=True)[0]
            ea0 = torch.exp(a0)
            z0 = ea0.sum(-1, keepdim=True)
            p0 = ea0 / z0
            return (p0 * (torch.log(z0) - a0)).sum(-1)
        logstd_b = pi * 0.0 + self.p['ppo2_model/pi/logstd']
        return (logstd_b + .5 * math.log(2.0 * math.pi * math.e)).sum(-1)

    # ---- act side ---------------------------------------------------------
    def step(self, obs, noise, S=None, M=None, **_):
        with torch.no_grad():
            pi, vf = self.forward(obs, S, M)
            nz = torch.as_tensor(np.asarray(noise)).to(self.dtype)
            if self.pd_kind == 'categorical':
                a = torch.argmax(pi - torch.log(-torch.log(nz)), dim=-1)
            elif self.pd_kind == 'multicategorical':       # one Gumbel-max per slice, stacked (distributions.py:223-224)
                g = pi - torch.log(-torch.log(nz))
                a = torch.stack([torch.argmax(x, dim=-1) for x in self._slices(g)], dim=-1)
            elif self.pd_kind == 'bernoulli':              # tf.to_float(u < sigmoid(l)), distributions.py:271-273
                a = (nz < torch.sigmoid(pi)).to(self.dtype)
            else:
                a = pi + torch.exp(pi * 0.0 + self.p['ppo2_model/pi/logstd']) * nz
            nlp = self._neglogp(pi, a)
        a_np = (a.numpy().astype(np.int64) if self.pd_kind == 'categorical' else
                a.numpy().astype(np.int32) if self.pd_kind == 'multicategorical' else a.numpy().astype(np.float32))
        state = self.last_state.numpy().astype(np.float32) if self.recurrent else None
        return a_np, vf.numpy().astype(np.float32), state, nlp.numpy().astype(np.float32)

    def value(self, obs, S=None, M=None, **_):
        with torch.no_grad():
            return self.forward(obs, S, M)[1].numpy().astype(np.float32)

    # ---- learner ------------------------------------------------------------
    def loss_and_stats(self, obs, returns, actions, values, neglogpacs, cliprange, advs, states=None, masks=None):
        dt = self.dtype
        R = torch.as_tensor(np.asarray(returns)).to(dt)
        OLDV = torch.as_tensor(np.asarray(values)).to(dt)
        OLDNLP = torch.as_tensor(np.asarray(neglogpacs)).to(dt)
        ADV = torch.as_tensor(np.asarray(advs)).to(dt)
        A = torch.as_tensor(np.asarray(actions))
        if self.pd_kind in ('gaussian', 'bernoulli'):
            A = A.to(dt)
        if self.recurrent:      # model.py:153-155: S = states, M = masks; train model built with nsteps (model.py:40-43)
            pi, vpred = self.forward(obs, states, masks, nenv=np.asarray(states).shape[0])
        else:
            pi, vpred = self.forward(obs)
        neglogpac = self._neglogp(pi, A)
        entropy = self._entropy(pi).mean()
        vpredclipped = OLDV + torch.clamp(vpred - OLDV, -cliprange, cliprange)
        vf_losses1 = (vpred - R) ** 2
        vf_losses2 = (vpredclipped - R) ** 2
        vf_loss = .5 * torch.maximum(vf_losses1, vf_losses2).mean()
        ratio = torch.exp(OLDNLP - neglogpac)
        pg_losses = -ADV * ratio
        pg_losses2 = -ADV * torch.clamp(ratio, 1.0 - cliprange, 1.0 + cliprange)
        pg_loss = torch.maximum(pg_losses, pg_losses2).mean()
        approxkl = .5 * ((neglogpac - OLDNLP) ** 2).mean()
        clipfrac = (torch.abs(ratio - 1.0) > cliprange).to(dt).mean()
        loss = pg_loss - entropy * self.ent_coef + vf_loss * self.vf_coef
        return loss, [pg_loss, vf_loss, entropy, approxkl, clipfrac]

    def _normalised_advs(self, returns, values):
        returns = np.asarray(returns)
        values = np.asarray(values)
        if self.dtype == torch.float64:
            returns, values = returns.astype(np.float64), values.astype(np.float64)
        advs = returns - values
        return (advs - advs.mean()) / (advs.std() + 1e-8)

    def compute_grads(self, cliprange, obs, returns, actions, values, neglogpacs, advs=None, states=None, masks=None):
        """model.py:136-139 + graph gradients.  Returns (stats, flat unclipped grad ndarray).
        `advs` given: already normalised (the MicrobatchedModel slices, microbatched_model.py:40-56)."""
        returns = np.asarray(returns)
        values = np.asarray(values)
        if advs is None:
            advs = self._normalised_advs(returns, values)
        for t in self.p.values():
            t.grad = None
        loss, stats = self.loss_and_stats(obs, returns, actions, values, neglogpacs, cliprange, advs, states, masks)
        loss.backward()
        grads = []
        for k in self.names:
            g = self.p[k].grad
            grads.append(torch.zeros_like(self.p[k]) if g is None else g.detach().clone())
        flat = torch.cat([g.reshape(-1) for g in grads])
        return [float(s.detach()) for s in stats], flat

    def average_and_clip(self, flat):
        """`self.grads` of model.py:105-112: [MPI average] -> clip_by_global_norm.  flat: torch 1-D."""
        if self.allreduce is not None:
            # mpi_adam_optimizer.py:21,39-40
            flat = self.allreduce(flat * self.rank_weight) / self.total_weight
        if self.max_grad_norm is not None:
            # tf.clip_by_global_norm: scale = c * min(1/gn, 1/c)
            gn = torch.sqrt((flat * flat).sum())
            c = torch.tensor(self.max_grad_norm, dtype=self.dtype)
            scale = c * torch.minimum(1.0 / gn, 1.0 / c)
            flat = flat * scale
            self.last_gnorm = float(gn)
        return flat

    def apply_flat_grad(self, lr, flat, clipped=False):
        """[MPI average] -> clip_by_global_norm -> TF-1 Adam.  flat: torch 1-D."""
        if not clipped:
            flat = self.average_and_clip(flat)
        self.last_grads = flat.clone()
        lr = self.npdt(lr)
        one = self.npdt(1)
        alpha = lr * np.sqrt(one - self.beta2_power) / (one - self.beta1_power)
        off = 0
        with torch.no_grad():
            for k in self.names:
                n = self.p[k].numel()
                g = flat[off:off + n].reshape(self.p[k].shape)
                off += n
                self.m[k] += (g - self.m[k]) * float(one - self.beta1)
                self.v[k] += (g * g - self.v[k]) * float(one - self.beta2)
                self.p[k] -= (self.m[k] * float(alpha)) / (torch.sqrt(self.v[k]) + float(self.eps))
        self.beta1_power = self.npdt(self.beta1_power * self.beta1)
        self.beta2_power = self.npdt(self.beta2_power * self.beta2)

    def train(self, lr, cliprange, obs, returns, masks, actions, values, neglogpacs, states=None):
        stats, flat = self.compute_grads(cliprange, obs, returns, actions, values, neglogpacs, states=states, masks=masks)
        self.apply_flat_grad(lr, flat)
        return stats

    def train_micro(self, microbatch_size, lr, cliprange, obs, returns, masks, actions, values, neglogpacs, states=None):
        """MicrobatchedModel.train (microbatched_model.py:36-75): advantages normalised over the whole minibatch,
        per-slice post-clip gradients summed, divided by the number of microbatches, applied without a second
        clip; stats = np.mean over the microbatches."""
        assert states is None
        n = len(np.asarray(returns))
        assert n % microbatch_size == 0
        nmicro = n // microbatch_size
        advs = self._normalised_advs(returns, values)
        total, stats_all = None, []
        for k in range(nmicro):
            sl = slice(k * microbatch_size, (k + 1) * microbatch_size)
            stats, flat = self.compute_grads(cliprange, np.asarray(obs)[sl], np.asarray(returns)[sl], np.asarray(actions)[sl],
                                             np.asarray(values)[sl], np.asarray(neglogpacs)[sl], advs=advs[sl])
            flat = self.average_and_clip(flat)
            total = flat if total is None else total + flat
            stats_all.append(stats)
        self.apply_flat_grad(lr, total / nmicro, clipped=True)
        return np.mean(np.array(stats_all, dtype=np.float32), axis=0).tolist()

    # ---- helpers --------------------------------------------------------------
    def flat_params(self):
        return np.concatenate([self.p[k].detach().numpy().reshape(-1) for k in self.names])

    def params_numpy(self):
        return {k: self.p[k].detach().numpy().copy() for k in self.names}
