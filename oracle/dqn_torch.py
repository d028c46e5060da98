"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PyTorch-CPU restatement of the DQN graph the reference builds with TensorFlow 1 (third-party, absent here ->
"parity unpinned" at the TF boundary; the pieces the reference states in NumPy -- replay, segment trees, schedules --
are pinned in oracle/replay_numpy.py).  Paths relative to /root/reference/baselines/.

  q_func      deepq/models.py:5-45 (action_value / state_value heads, dueling combine),
              common/models.py:74-103 (mlp), :15-26 (nature_cnn), :222-249 (conv_only: contrib convolution2d,
              SAME padding -- total = max((ceil(in/s) - 1) * s + k - in, 0), the smaller half in front --, ReLU)
  TD loss     deepq/build_graph.py:380-413; Huber common/tf_util.py:39-45
  optimizer   deepq/build_graph.py:416-421 (tf.clip_by_norm per variable: t * c / max(||t||, c)),
              deepq/deepq.py:205 tf.train.AdamOptimizer(lr) -> epsilon 1e-8, TF-1 ApplyAdam form (as in ppo2_torch.py)

Parameters are taken by TF variable name from the layout the device model reports (QModel.tensors), so the same flat
buffer feeds both sides.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _same_pad(n, k, s):
    out = (n + s - 1) // s
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


class OracleQNet(object):
    def __init__(self, network, tensors, flat_params, nact, hiddens=(256,), dueling=True, convs=(), num_layers=2,
                 activation='tanh', dtype=torch.float32, lr=5e-4, gamma=1.0, clip=10.0, double_q=True, layer_norm=False,
                 body_layer_norm=False):
        self.network, self.nact, self.hiddens, self.dueling, self.convs = network, nact, tuple(hiddens), dueling, tuple(convs)
        self.num_layers, self.activation, self.dtype = num_layers, activation, dtype
        self.lr, self.gamma, self.clip, self.double_q = lr, gamma, clip, double_q
        self.tensors = tensors
        self.layer_norm = layer_norm
        self.body_layer_norm = body_layer_norm          # network = mlp(layer_norm=True), common/models.py:97-98
        self.names = [t['name'] for t in tensors]
        flat = np.asarray(flat_params)
        self.p = {t['name']: torch.tensor(flat[t['offset']:t['offset'] + t['size']].reshape(t['shape']), dtype=dtype,
                                          requires_grad=True) for t in tensors}
        self.target = {k: v.detach().clone() for k, v in self.p.items()}
        self.m = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.p.items()}
        npdt = np.float32 if dtype == torch.float32 else np.float64
        self.npdt = npdt
        self.beta1, self.beta2, self.eps = npdt(0.9), npdt(0.999), npdt(1e-8)
        self.beta1_power, self.beta2_power = npdt(0.9), npdt(0.999)

    # ---- forward ------------------------------------------------------------------------------------------------------
    def q(self, p, obs):
        s = 'deepq/q_func'
        x = torch.as_tensor(np.asarray(obs))
        if self.network == 'mlp':
            h = x.to(self.dtype).reshape(x.shape[0], -1)
            act = torch.tanh if self.activation == 'tanh' else F.relu
            for i in range(self.num_layers):
                h = h @ p[s + '/mlp_fc%d/w' % i] + p[s + '/mlp_fc%d/b' % i]
                if self.body_layer_norm:
                    # common/models.py:97-98: tf.contrib.layers.layer_norm(h, center=True, scale=True) in the q_func scope
                    ln = s + ('/LayerNorm_%d' % i if i else '/LayerNorm')
                    mu = h.mean(dim=1, keepdim=True)
                    var = ((h - mu) ** 2).mean(dim=1, keepdim=True)
                    h = (h - mu) * torch.rsqrt(var + 1e-12) * p[ln + '/gamma'] + p[ln + '/beta']
                h = act(h)
        elif self.network == 'cnn':
            h = (x.to(self.dtype) / 255.).permute(0, 3, 1, 2)
            for name, stride in (('c1', 4), ('c2', 2), ('c3', 1)):
                h = F.relu(F.conv2d(h, p[s + '/%s/w' % name].permute(3, 2, 0, 1), p[s + '/%s/b' % name].reshape(-1), stride=stride))
            h = h.permute(0, 2, 3, 1).reshape(h.shape[0], -1)
            h = F.relu(h @ p[s + '/fc1/w'] + p[s + '/fc1/b'])
        else:
            h = (x.to(self.dtype) / 255.).permute(0, 3, 1, 2)
            for i, (nf, k, st) in enumerate(self.convs):
                nm = s + '/convnet/' + ('Conv_%d' % i if i else 'Conv')
                pt, pb = _same_pad(h.shape[2], k, st)
                pl, pr = _same_pad(h.shape[3], k, st)
                h = F.pad(h, (pl, pr, pt, pb))
                h = F.relu(F.conv2d(h, p[nm + '/weights'].permute(3, 2, 0, 1), p[nm + '/biases'], stride=st))
            h = h.permute(0, 2, 3, 1).reshape(h.shape[0], -1)          # layers.flatten of the NHWC map
        def head(scope, nout):
            o = h
            for i in range(len(self.hiddens) + 1):
                nm = s + '/' + scope + ('/fully_connected_%d' % i if i else '/fully_connected')
                o = o @ p[nm + '/weights'] + p[nm + '/biases']
                if i < len(self.hiddens):
                    if self.layer_norm:
                        # deepq/models.py:27-29: layers.layer_norm(out, center=True, scale=True) -- moments over the features
                        # (biased variance), variance_epsilon 1e-12 (tf.contrib.layers.layer_norm -> tf.nn.batch_normalization)
                        ln = s + '/' + scope + ('/LayerNorm_%d' % i if i else '/LayerNorm')
                        mu = o.mean(dim=1, keepdim=True)
                        var = ((o - mu) ** 2).mean(dim=1, keepdim=True)
                        o = (o - mu) * torch.rsqrt(var + 1e-12) * p[ln + '/gamma'] + p[ln + '/beta']
                    o = F.relu(o)
            return o
        a = head('action_value', self.nact)
        if not self.dueling:
            return a
        v = head('state_value', 1)
        return v + (a - a.mean(dim=1, keepdim=True))

    def q_values(self, obs, target=False):
        with torch.no_grad():
            return self.q(self.target if target else self.p, obs).numpy()

    # ---- train ----------------------------------------------------------------------------------------------------------
    def td_and_grads(self, obs_t, act, rew, obs_tp1, done, w):
        dt = self.dtype
        act = torch.as_tensor(np.asarray(act)).long()
        rew, done, w = (torch.as_tensor(np.asarray(x)).to(dt) for x in (rew, done, w))
        for t in self.p.values():
            t.grad = None
        q_t = self.q(self.p, obs_t)
        with torch.no_grad():
            q_tp1 = self.q(self.target, obs_tp1)
            if self.double_q:
                best = self.q(self.p, obs_tp1).argmax(dim=1)
                q_tp1_best = q_tp1.gather(1, best[:, None])[:, 0]
            else:
                q_tp1_best = q_tp1.max(dim=1)[0]
            target = rew + self.gamma * (1.0 - done) * q_tp1_best
        q_sel = q_t.gather(1, act[:, None])[:, 0]
        td = q_sel - target
        huber = torch.where(td.abs() < 1.0, td * td * 0.5, td.abs() - 0.5)
        loss = (w * huber).mean()
        loss.backward()
        grads = {k: (self.p[k].grad.detach().clone() if self.p[k].grad is not None else torch.zeros_like(self.p[k]))
                 for k in self.names}
        return td.detach().numpy(), float(loss.detach()), grads

    def train(self, obs_t, act, rew, obs_tp1, done, w):
        td, loss, grads = self.td_and_grads(obs_t, act, rew, obs_tp1, done, w)
        self.apply_grads(grads)
        return td, loss

    def apply_grads(self, grads):
        """build_graph.py:416-421 per-variable clip_by_norm, then the optimizer's apply op"""
        one = self.npdt(1)
        alpha = self.npdt(self.lr) * np.sqrt(one - self.beta2_power) / (one - self.beta1_power)
        with torch.no_grad():
            for k in self.names:
                g = grads[k]
                if self.clip is not None and self.clip > 0:
                    g = g * self.clip / torch.clamp(torch.sqrt((g * g).sum()), min=self.clip)
                self.m[k] += (g - self.m[k]) * float(one - self.beta1)
                self.v[k] += (g * g - self.v[k]) * float(one - self.beta2)
                self.p[k] -= (self.m[k] * float(alpha)) / (torch.sqrt(self.v[k]) + float(self.eps))
        self.beta1_power = self.npdt(self.beta1_power * self.beta1)
        self.beta2_power = self.npdt(self.beta2_power * self.beta2)

    def update_target(self):
        self.target = {k: v.detach().clone() for k, v in self.p.items()}

    def flat_params(self):
        return np.concatenate([self.p[k].detach().numpy().reshape(-1) for k in self.names])
