"""`build_q_func` with the reference's signature (deepq/models.py:5-45).  Where the reference returns a TF graph
builder, this returns a *description* (`QFuncDesc`) that the device layout in libmrl (csrc/qnet.hip.h) understands:

    q = network(X) -> flatten -> action_value: [fc(h) -> [layer_norm] -> relu for h in hiddens] -> fc(num_actions)
        dueling:     state_value: [fc(h) -> relu ...] -> fc(1);   q = V + (A - mean_a A)

`network`: 'mlp' / 'cnn' (common/models.py, via our registry) or 'conv_only' (common/models.py:222-249:
convs=[(32, 8, 4), (64, 4, 2), (64, 3, 1)], tf.contrib convolution2d defaults = SAME padding + ReLU)."""
from ..common.models import NetworkDesc, get_network_builder, register


@register('conv_only')
def conv_only(convs=((32, 8, 4), (64, 4, 2), (64, 3, 1)), **conv_kwargs):
    if conv_kwargs:
        raise NotImplementedError('conv kwargs {} are outside the supported hot path'.format(sorted(conv_kwargs)))
    return NetworkDesc('conv_only', convs=tuple(tuple(int(v) for v in c) for c in convs))


class QFuncDesc(object):
    def __init__(self, network, hiddens, dueling, layer_norm=False):
        self.network, self.hiddens, self.dueling = network, tuple(int(h) for h in hiddens), bool(dueling)
        self.layer_norm = bool(layer_norm)

    def __repr__(self):
        return 'QFuncDesc(%r, hiddens=%r, dueling=%r, layer_norm=%r)' % (self.network, self.hiddens, self.dueling, self.layer_norm)


def build_q_func(network, hiddens=(256,), dueling=True, layer_norm=False, **network_kwargs):
    if isinstance(network, str):
        network = get_network_builder(network)(**network_kwargs)
    if not isinstance(network, NetworkDesc):
        raise NotImplementedError('custom TF network functions cannot run on the HIP path; register a NetworkDesc')
    if network.kind in ('lstm', 'cnn_lstm'):
        raise NotImplementedError('DQN is not compatible with recurrent policies yet')     # deepq/models.py:14-16
    if network.kind not in ('mlp', 'cnn', 'conv_only'):
        raise ValueError('Unknown network type: {}'.format(network.kind))
    return QFuncDesc(network, hiddens, dueling, layer_norm)
