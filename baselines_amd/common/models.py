"""Network registry with the reference's surface (common/models.py:7-13, 257-275): builders are
looked up by name and called with the user's **network_kwargs.  A builder here returns a
*description* that the HIP model layout (csrc/model.hip) understands, instead of a TF graph
function.  Supported on the hot path: 'mlp' (models.py:74-103), 'cnn' = NatureCNN
(models.py:15-26, incl. pad='SAME'), 'cnn_small' (models.py:117-129), 'lstm', 'cnn_lstm' and 'cnn_lnlstm' (models.py:132-218);
every other name raises like get_network_builder does (models.py:275)."""

mapping = {}


def register(name):
    def _thunk(func):
        mapping[name] = func
        return func
    return _thunk


class NetworkDesc(object):
    def __init__(self, kind, **kw):
        self.kind = kind
        self.kw = kw

    def __repr__(self):
        return 'NetworkDesc(%s, %s)' % (self.kind, self.kw)


def _activation_name(activation):
    if activation is None:
        return 'tanh'
    if isinstance(activation, str):
        name = activation
    else:
        name = getattr(activation, '__name__', str(activation))
    name = name.lower()
    if 'tanh' in name:
        return 'tanh'
    if 'relu' in name:
        return 'relu'
    raise ValueError('unsupported mlp activation for the HIP path: {}'.format(activation))


@register('mlp')
def mlp(num_layers=2, num_hidden=64, activation=None, layer_norm=False):
    """models.py:74-103; layer_norm=True: tf.contrib.layers.layer_norm(h, center=True, scale=True) between every fc and its
    activation (models.py:97-98)"""
    return NetworkDesc('mlp', num_layers=int(num_layers), num_hidden=int(num_hidden),
                       activation=_activation_name(activation), layer_norm=bool(layer_norm))


def _conv_kwargs(conv_kwargs):
    """**conv_kwargs of a2c/utils.py:37 conv() that the HIP path honours: pad='VALID' | 'SAME' (NHWC, [1, nf, 1, 1] biases are
    the only layout built)"""
    kw = dict(conv_kwargs)
    pad = kw.pop('pad', 'VALID')
    if pad not in ('VALID', 'SAME'):
        raise ValueError('pad must be VALID or SAME, got {!r}'.format(pad))
    if kw.pop('data_format', 'NHWC') != 'NHWC' or kw.pop('one_dim_bias', False):
        raise NotImplementedError('conv data_format / one_dim_bias other than the defaults are outside the supported hot path')
    if kw:
        raise NotImplementedError('conv kwargs {} are outside the supported hot path'.format(sorted(kw)))
    return {} if pad == 'VALID' else {'pad': pad}


@register('cnn')
def cnn(**conv_kwargs):
    """models.py:106-110: nature_cnn(X, **conv_kwargs)"""
    return NetworkDesc('cnn', **_conv_kwargs(conv_kwargs))


@register('cnn_small')
def cnn_small(**conv_kwargs):
    """models.py:117-129: conv 8 x (8 x 8) stride 4, conv 16 x (4 x 4) stride 2, fc 128 (all ReLU, sqrt(2) orthogonal init)"""
    return NetworkDesc('cnn', convs=((8, 8, 4), (16, 4, 2)), fc_hidden=128, **_conv_kwargs(conv_kwargs))


@register('lstm')
def lstm(nlstm=128, layer_norm=False):
    """common/models.py:132-176: flatten -> LSTM cell over the steps of a rollout, state managed outside the policy;
    layer_norm=True: the layer-normalised cell (a2c/utils.py:110-140 lnlstm, variables under scope 'lnlstm')"""
    return NetworkDesc('lstm', nlstm=int(nlstm), layer_norm=bool(layer_norm))


@register('cnn_lstm')
def cnn_lstm(nlstm=128, layer_norm=False, **conv_kwargs):
    """common/models.py:179-206: NatureCNN features -> LSTM cell (layer_norm=True: lnlstm)"""
    return NetworkDesc('cnn_lstm', nlstm=int(nlstm), layer_norm=bool(layer_norm), **_conv_kwargs(conv_kwargs))


@register('cnn_lnlstm')
def cnn_lnlstm(nlstm=128, **conv_kwargs):
    """common/models.py:216-218"""
    return cnn_lstm(nlstm, layer_norm=True, **conv_kwargs)


def get_network_builder(name):
    if callable(name):
        return name
    elif name in mapping:
        return mapping[name]
    else:
        raise ValueError('Unknown network type: {}'.format(name))
