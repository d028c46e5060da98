"""CPU: the C-ABI shared library loads without a GPU, exports every symbol include/mrl.h declares,
and its host-only entry points (layout object, error strings) behave.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

from baselines_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from baselines_amd.csrc.build import build
        build(verbose=False)
    return _lib.load()


def _declared():
    text = open(os.path.join(ROOT, 'include', 'mrl.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mrl_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported_and_bound(lib):
    names = _declared()
    assert len(names) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), 'libmrl.so does not export %s' % n
        assert n in _lib.SIGNATURES, 'no ctypes signature for %s' % n
    assert set(_lib.SIGNATURES) <= set(names), 'binding lists symbols the header does not declare'


def test_version_and_errors(lib):
    assert lib.mrl_version() == 1
    assert lib.mrl_strerror(0) == b'ok'
    assert b'invalid' in lib.mrl_strerror(-1)
    assert b'workspace' in lib.mrl_strerror(-2)
    assert b'unsupported' in lib.mrl_strerror(-3)


def _layout(lib, **kw):
    d = _lib.ModelDesc()
    d.network = kw['network']
    shape = kw['ob_shape']
    d.ob_ndim = len(shape)
    for i, s in enumerate(shape):
        d.ob_shape[i] = s
    d.ob_dtype = kw['ob_dtype']
    d.num_layers, d.num_hidden, d.activation = kw.get('num_layers', 2), kw.get('num_hidden', 64), _lib.ACT_TANH
    d.value_copy, d.pd_kind, d.nact = kw.get('value_copy', 0), kw['pd_kind'], kw['nact']
    h = ctypes.c_void_p()
    rc = lib.mrl_model_create(ctypes.byref(d), ctypes.byref(h))
    return rc, h


def _tensors(lib, h):
    out = []
    name = ctypes.create_string_buffer(128)
    for i in range(lib.mrl_model_num_tensors(h)):
        nd, shp, off, sc = ctypes.c_int(), (ctypes.c_int * 4)(), ctypes.c_long(), ctypes.c_double()
        assert lib.mrl_model_tensor_info(h, i, name, 128, ctypes.byref(nd), ctypes.byref(shp), ctypes.byref(off),
                                         ctypes.byref(sc)) == 0
        out.append((name.value.decode(), tuple(shp[k] for k in range(nd.value)), off.value, sc.value))
    return out


def test_layout_nature_cnn_matches_reference_inventory(lib):
    """variable names, shapes, creation order, P = 1,687,719 (nA = 6)."""
    rc, h = _layout(lib, network=_lib.NET_NATURE_CNN, ob_shape=(84, 84, 4), ob_dtype=_lib.OB_U8,
                    pd_kind=_lib.PD_CATEGORICAL, nact=6)
    assert rc == 0
    assert lib.mrl_model_num_params(h) == 1687719
    t = _tensors(lib, h)
    assert [x[0] for x in t] == ['ppo2_model/pi/c1/w', 'ppo2_model/pi/c1/b', 'ppo2_model/pi/c2/w', 'ppo2_model/pi/c2/b',
                                 'ppo2_model/pi/c3/w', 'ppo2_model/pi/c3/b', 'ppo2_model/pi/fc1/w', 'ppo2_model/pi/fc1/b',
                                 'ppo2_model/pi/w', 'ppo2_model/pi/b', 'ppo2_model/vf/w', 'ppo2_model/vf/b']
    shapes = dict((x[0], x[1]) for x in t)
    assert shapes['ppo2_model/pi/c1/w'] == (8, 8, 4, 32) and shapes['ppo2_model/pi/c1/b'] == (1, 32, 1, 1)
    assert shapes['ppo2_model/pi/c3/w'] == (3, 3, 64, 64) and shapes['ppo2_model/pi/fc1/w'] == (3136, 512)
    scales = dict((x[0], x[3]) for x in t)
    assert scales['ppo2_model/pi/c1/w'] == np.sqrt(2) and scales['ppo2_model/pi/w'] == 0.01 and scales['ppo2_model/vf/w'] == 1.0
    assert scales['ppo2_model/pi/c1/b'] < 0            # zeros
    offs = [x[2] for x in t]
    assert offs == sorted(offs) and offs[0] == 0       # flat buffer in creation order
    assert lib.mrl_model_workspace_bytes(h, 1024) > 1024 * 173000
    lib.mrl_model_destroy(h)


def test_layout_mlp_variants(lib):
    rc, h = _layout(lib, network=_lib.NET_MLP, ob_shape=(376,), ob_dtype=_lib.OB_F32, pd_kind=_lib.PD_DIAG_GAUSSIAN,
                    nact=17, value_copy=1)
    assert rc == 0 and lib.mrl_model_num_params(h) == 57763         # Humanoid-shaped, value_network='copy'
    names = [x[0] for x in _tensors(lib, h)]
    assert names.index('ppo2_model/vf/mlp_fc0/w') > names.index('ppo2_model/pi/mlp_fc1/b')
    assert 'ppo2_model/pi/logstd' in names
    lib.mrl_model_destroy(h)
    rc, h = _layout(lib, network=_lib.NET_MLP, ob_shape=(4,), ob_dtype=_lib.OB_F32, pd_kind=_lib.PD_CATEGORICAL, nact=2)
    assert rc == 0 and lib.mrl_model_num_params(h) == 4675          # CartPole, shared latent
    lib.mrl_model_destroy(h)
    # distributions.py:351-355 quirk: latent width == action dim -> no pi head
    rc, h = _layout(lib, network=_lib.NET_MLP, ob_shape=(8,), ob_dtype=_lib.OB_F32, pd_kind=_lib.PD_DIAG_GAUSSIAN,
                    nact=17, num_hidden=17)
    assert rc == 0
    assert 'ppo2_model/pi/w' not in [x[0] for x in _tensors(lib, h)]
    lib.mrl_model_destroy(h)


def test_layout_mlp_layer_norm_variable_names(lib):
    """mlp(layer_norm=True), common/models.py:97-98: tf.contrib.layers.layer_norm variables in creation order"""
    d = _lib.ModelDesc()
    d.network, d.ob_ndim, d.ob_dtype, d.num_layers, d.num_hidden = _lib.NET_MLP, 1, _lib.OB_F32, 2, 64
    d.ob_shape[0] = 8
    d.activation, d.value_copy, d.pd_kind, d.nact, d.layer_norm = _lib.ACT_TANH, 1, _lib.PD_DIAG_GAUSSIAN, 3, 1
    h = ctypes.c_void_p()
    assert lib.mrl_model_create(ctypes.byref(d), ctypes.byref(h)) == 0
    names = [x[0] for x in _tensors(lib, h)]
    assert names[:8] == ['ppo2_model/pi/mlp_fc0/w', 'ppo2_model/pi/mlp_fc0/b', 'ppo2_model/pi/LayerNorm/beta',
                         'ppo2_model/pi/LayerNorm/gamma', 'ppo2_model/pi/mlp_fc1/w', 'ppo2_model/pi/mlp_fc1/b',
                         'ppo2_model/pi/LayerNorm_1/beta', 'ppo2_model/pi/LayerNorm_1/gamma']
    assert 'ppo2_model/vf/LayerNorm_1/gamma' in names
    lib.mrl_model_destroy(h)


def test_layout_lnlstm_variable_names_match_the_oracle_inventory(lib):
    """lstm(layer_norm=True) / cnn_lnlstm (common/models.py:173-174, 216-218): a2c/utils.py:113-124 variables under 'lnlstm'"""
    from oracle.ppo2_torch import build_param_specs
    d = _lib.ModelDesc()
    d.network, d.ob_ndim, d.ob_dtype, d.nlstm, d.layer_norm = _lib.NET_LSTM, 1, _lib.OB_F32, 48, 1
    d.ob_shape[0] = 12
    d.pd_kind, d.nact = _lib.PD_CATEGORICAL, 4
    h = ctypes.c_void_p()
    assert lib.mrl_model_create(ctypes.byref(d), ctypes.byref(h)) == 0
    got = [(x[0], tuple(x[1])) for x in _tensors(lib, h)]
    specs, _ = build_param_specs('lstm', (12,), 'categorical', 4, nlstm=48, layer_norm=True)
    assert got == [(n, tuple(s)) for n, s, _ in specs]
    assert [n.rsplit('/', 1)[1] for n, _ in got[:9]] == ['wx', 'gx', 'bx', 'wh', 'gh', 'bh', 'b', 'gc', 'bc']
    assert lib.mrl_model_state_size(h) == 96
    lib.mrl_model_destroy(h)


def test_layout_qnet_layer_norm_heads_variable_names(lib):
    """build_q_func(layer_norm=True), deepq/models.py:24-41: LayerNorm variables between the hidden head layers, none
    after the output layer; gamma initialised to ones (init_kind 3), beta to zeros"""
    d = _lib.QNetDesc()
    d.network, d.ob_ndim, d.ob_dtype, d.num_layers, d.num_hidden, d.activation = _lib.NET_MLP, 1, _lib.OB_F32, 1, 16, _lib.ACT_TANH
    d.ob_shape[0] = 6
    d.nhidden, d.dueling, d.nact, d.layer_norm = 2, 1, 3, 1
    d.hiddens[0], d.hiddens[1] = 8, 4
    h = ctypes.c_void_p()
    assert lib.mrl_qnet_create(ctypes.byref(d), ctypes.byref(h)) == 0
    name = ctypes.create_string_buffer(160)
    names, kinds = [], {}
    for i in range(lib.mrl_qnet_num_tensors(h)):
        nd, shp, off, kind, sc = ctypes.c_int(), (ctypes.c_int * 4)(), ctypes.c_long(), ctypes.c_int(), ctypes.c_double()
        assert lib.mrl_qnet_tensor_info(h, i, name, 160, ctypes.byref(nd), ctypes.byref(shp), ctypes.byref(off), ctypes.byref(kind),
                                        ctypes.byref(sc)) == 0
        names.append(name.value.decode())
        kinds[names[-1]] = kind.value
    av = 'deepq/q_func/action_value/'
    assert names[2:12] == [av + 'fully_connected/weights', av + 'fully_connected/biases', av + 'LayerNorm/beta', av + 'LayerNorm/gamma',
                           av + 'fully_connected_1/weights', av + 'fully_connected_1/biases', av + 'LayerNorm_1/beta',
                           av + 'LayerNorm_1/gamma', av + 'fully_connected_2/weights', av + 'fully_connected_2/biases']
    assert names[12].startswith('deepq/q_func/state_value/') and names[-1] == 'deepq/q_func/state_value/fully_connected_2/biases'
    assert kinds[av + 'LayerNorm/gamma'] == 3 and kinds[av + 'LayerNorm/beta'] == 0
    lib.mrl_qnet_destroy(h)


def test_layout_qnet_mlp_body_layer_norm_variable_names(lib):
    """build_q_func(mlp(layer_norm=True)): the BODY's LayerNorm variables (common/models.py:97-98) exist, in the
    q_func scope, beta before gamma, gamma initialised to ones -- independent of the heads' layer_norm switch"""
    d = _lib.QNetDesc()
    d.network, d.ob_ndim, d.ob_dtype, d.num_layers, d.num_hidden, d.activation = _lib.NET_MLP, 1, _lib.OB_F32, 2, 16, _lib.ACT_TANH
    d.ob_shape[0] = 8
    d.nhidden, d.dueling, d.nact, d.layer_norm, d.body_layer_norm = 1, 0, 3, 0, 1
    d.hiddens[0] = 8
    h = ctypes.c_void_p()
    assert lib.mrl_qnet_create(ctypes.byref(d), ctypes.byref(h)) == 0
    name = ctypes.create_string_buffer(160)
    names, kinds = [], {}
    for i in range(lib.mrl_qnet_num_tensors(h)):
        nd, shp, off, kind, sc = ctypes.c_int(), (ctypes.c_int * 4)(), ctypes.c_long(), ctypes.c_int(), ctypes.c_double()
        assert lib.mrl_qnet_tensor_info(h, i, name, 160, ctypes.byref(nd), ctypes.byref(shp), ctypes.byref(off), ctypes.byref(kind),
                                        ctypes.byref(sc)) == 0
        names.append(name.value.decode())
        kinds[names[-1]] = kind.value
    s = 'deepq/q_func/'
    assert names == [s + 'mlp_fc0/w', s + 'mlp_fc0/b', s + 'LayerNorm/beta', s + 'LayerNorm/gamma', s + 'mlp_fc1/w', s + 'mlp_fc1/b',
                     s + 'LayerNorm_1/beta', s + 'LayerNorm_1/gamma', s + 'action_value/fully_connected/weights',
                     s + 'action_value/fully_connected/biases', s + 'action_value/fully_connected_1/weights',
                     s + 'action_value/fully_connected_1/biases']
    assert kinds[s + 'LayerNorm/gamma'] == 3 and kinds[s + 'LayerNorm_1/beta'] == 0 and kinds[s + 'mlp_fc1/w'] == 1
    lib.mrl_qnet_destroy(h)


def test_layout_cnn_small_variable_names_and_shapes(lib):
    """cnn_small (common/models.py:117-129): c1 [8, 8, C, 8], c2 [4, 4, 8, 16], fc1 [9 * 9 * 16, 128] on 84 x 84 x 4 frames"""
    d = _lib.ModelDesc()
    d.network, d.ob_ndim, d.ob_dtype, d.pd_kind, d.nact = _lib.NET_NATURE_CNN, 3, _lib.OB_U8, _lib.PD_CATEGORICAL, 6
    d.ob_shape[0], d.ob_shape[1], d.ob_shape[2] = 84, 84, 4
    d.nconv, d.fc_hidden = 2, 128
    for i, c in enumerate(((8, 8, 4), (16, 4, 2))):
        for k in range(3):
            d.convs[i][k] = c[k]
    h = ctypes.c_void_p()
    assert lib.mrl_model_create(ctypes.byref(d), ctypes.byref(h)) == 0
    name = ctypes.create_string_buffer(128)
    got = []
    for i in range(lib.mrl_model_num_tensors(h)):
        nd, shp, off, sc = ctypes.c_int(), (ctypes.c_int * 4)(), ctypes.c_long(), ctypes.c_double()
        assert lib.mrl_model_tensor_info(h, i, name, 128, ctypes.byref(nd), ctypes.byref(shp), ctypes.byref(off), ctypes.byref(sc)) == 0
        got.append((name.value.decode(), tuple(shp[k] for k in range(nd.value))))
    p = 'ppo2_model/pi/'
    assert got == [(p + 'c1/w', (8, 8, 4, 8)), (p + 'c1/b', (1, 8, 1, 1)), (p + 'c2/w', (4, 4, 8, 16)), (p + 'c2/b', (1, 16, 1, 1)),
                   (p + 'fc1/w', (1296, 128)), (p + 'fc1/b', (128,)), (p + 'w', (128, 6)), (p + 'b', (6,)),
                   ('ppo2_model/vf/w', (128, 1)), ('ppo2_model/vf/b', (1,))]
    lib.mrl_model_destroy(h)
    d.convs[0][0] = 6                              # filters % 4 != 0: outside what the engines load
    assert lib.mrl_model_create(ctypes.byref(d), ctypes.byref(h)) != 0


def test_layout_rejects_unsupported(lib):
    rc, h = _layout(lib, network=_lib.NET_NATURE_CNN, ob_shape=(84, 84, 3), ob_dtype=_lib.OB_U8,
                    pd_kind=_lib.PD_CATEGORICAL, nact=6)
    assert rc == 0             # any channel count / image dtype (common/models.py:19 casts whatever comes)
    lib.mrl_model_destroy(h)
    rc, _ = _layout(lib, network=_lib.NET_NATURE_CNN, ob_shape=(20, 20, 4), ob_dtype=_lib.OB_U8,
                    pd_kind=_lib.PD_CATEGORICAL, nact=6)
    assert rc == -3            # MRL_EUNSUP: image too small for the 8x8/4, 4x4/2, 3x3/1 VALID stack
    rc, _ = _layout(lib, network=_lib.NET_MLP, ob_shape=(4,), ob_dtype=_lib.OB_F32, pd_kind=7, nact=2)
    assert rc == -3


def test_product_has_no_cpu_path():
    """The package must fail loudly without a HIP device (no oracle / CPU fallback behind it)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError):
        _lib.require_gpu()
    src = ''
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'baselines_amd')):
        for f in files:
            if f.endswith('.py'):
                src += open(os.path.join(dirpath, f)).read()
    assert 'import oracle' not in src and 'from oracle' not in src


# ---- f1: checkpoint wire format (tf_util.py:345-372), host half of Model.load --------------------------------
def _reference_shaped_checkpoint(kind, rng):
    """What the reference's save_variables would write for this model -- built from SURVEY.md App. A.6 by hand,
    independently of mrl_model_tensor_info: GLOBAL_VARIABLES = model variables in creation order, then beta1_power,
    beta2_power, then the Adam slots (m, v) of each variable in turn."""
    if kind == 'nature_cnn':
        pv = [('ppo2_model/pi/c1/w', (8, 8, 4, 32)), ('ppo2_model/pi/c1/b', (1, 32, 1, 1)),
              ('ppo2_model/pi/c2/w', (4, 4, 32, 64)), ('ppo2_model/pi/c2/b', (1, 64, 1, 1)),
              ('ppo2_model/pi/c3/w', (3, 3, 64, 64)), ('ppo2_model/pi/c3/b', (1, 64, 1, 1)),
              ('ppo2_model/pi/fc1/w', (3136, 512)), ('ppo2_model/pi/fc1/b', (512,)),
              ('ppo2_model/pi/w', (512, 6)), ('ppo2_model/pi/b', (6,)), ('ppo2_model/vf/w', (512, 1)), ('ppo2_model/vf/b', (1,))]
    else:      # mlp 2x64, value_network='copy', Box(376,) observations, 17-dim DiagGaussian
        pv = []
        for net in ('pi', 'vf'):
            pv += [('ppo2_model/%s/mlp_fc0/w' % net, (376, 64)), ('ppo2_model/%s/mlp_fc0/b' % net, (64,)),
                   ('ppo2_model/%s/mlp_fc1/w' % net, (64, 64)), ('ppo2_model/%s/mlp_fc1/b' % net, (64,))]
        pv += [('ppo2_model/pi/w', (64, 17)), ('ppo2_model/pi/b', (17,)), ('ppo2_model/pi/logstd', (1, 17)),
               ('ppo2_model/vf/w', (64, 1)), ('ppo2_model/vf/b', (1,))]
    names = [n + ':0' for n, _ in pv] + ['beta1_power:0', 'beta2_power:0']
    arrays = [rng.randn(*s).astype(np.float32) for _, s in pv] + [np.float32(0.9 ** 7), np.float32(0.999 ** 7)]
    for n, s in pv:
        names += [n + '/Adam:0', n + '/Adam_1:0']
        arrays += [rng.randn(*s).astype(np.float32), rng.rand(*s).astype(np.float32)]
    return pv, names, arrays


@pytest.mark.parametrize('kind', ['nature_cnn', 'mlp_copy'])
def test_reference_shaped_checkpoints_load_by_name_and_as_legacy_list(lib, kind):
    from baselines_amd.ppo2.model import checkpoint_to_flat
    if kind == 'nature_cnn':
        rc, h = _layout(lib, network=_lib.NET_NATURE_CNN, ob_shape=(84, 84, 4), ob_dtype=_lib.OB_U8,
                        pd_kind=_lib.PD_CATEGORICAL, nact=6)
    else:
        rc, h = _layout(lib, network=_lib.NET_MLP, ob_shape=(376,), ob_dtype=_lib.OB_F32, value_copy=1,
                        pd_kind=_lib.PD_DIAG_GAUSSIAN, nact=17)
    assert rc == 0
    P = lib.mrl_model_num_params(h)
    tensors = [(n, s, o, int(np.prod(s))) for n, s, o, _ in _tensors(lib, h)]
    pv, names, arrays = _reference_shaped_checkpoint(kind, np.random.RandomState(0))
    n = len(pv)
    assert P == sum(int(np.prod(s)) for _, s in pv)
    # the flat buffers a loader has to end up with: variables concatenated in creation order
    want = {'params': np.concatenate([a.reshape(-1) for a in arrays[:n]]),
            'adam_m': np.concatenate([arrays[n + 2 + 2 * i].reshape(-1) for i in range(n)]),
            'adam_v': np.concatenate([arrays[n + 3 + 2 * i].reshape(-1) for i in range(n)])}

    def flat(out, key):
        f = np.full(P, np.nan, np.float32)
        for off, a in out[key]:
            f[off:off + a.size] = a
        return f

    for loaded in (dict(zip(names, arrays)), list(arrays)):          # tf_util.py:367-369 / :362-366
        out = checkpoint_to_flat(loaded, tensors)
        for key in want:
            np.testing.assert_array_equal(flat(out, key), want[key])
        assert out['beta_powers'] == (np.float32(0.9 ** 7), np.float32(0.999 ** 7))
    # parameters only: a dict without optimizer state leaves the slots alone; a list of the n model variables too
    for loaded in ({k: v for k, v in zip(names[:n], arrays[:n])}, list(arrays[:n])):
        out = checkpoint_to_flat(loaded, tensors)
        np.testing.assert_array_equal(flat(out, 'params'), want['params'])
        assert out['adam_m'] == [] and out['adam_v'] == [] and out['beta_powers'] is None
    with pytest.raises(ValueError, match='number of variables loaded mismatches'):
        checkpoint_to_flat(list(arrays[:n + 1]), tensors)
    with pytest.raises(KeyError):
        d = dict(zip(names, arrays))
        del d[names[3]]
        checkpoint_to_flat(d, tensors)
    lib.mrl_model_destroy(h)
