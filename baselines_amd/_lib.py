"""ctypes binding of libmrl.so (C ABI declared in include/mrl.h).

The product path has NO CPU fallback: if the library is missing or no HIP device is visible,
every op raises.  (The oracle under oracle/ is test infrastructure and is never imported here.)
"""
import contextlib
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MRL_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libmrl.so')      # MRL_LIB_PATH: A/B of two builds on one box

c_void_p, c_int, c_long, c_float, c_double, c_size_t, c_char_p = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_double, ctypes.c_size_t, ctypes.c_char_p)


class ModelDesc(ctypes.Structure):
    """mirror of mrl_model_desc (include/mrl.h)"""
    _fields_ = [('network', c_int), ('ob_ndim', c_int), ('ob_shape', c_int * 3), ('ob_dtype', c_int),
                ('num_layers', c_int), ('num_hidden', c_int), ('activation', c_int), ('value_copy', c_int),
                ('pd_kind', c_int), ('nact', c_int), ('nlstm', c_int), ('layer_norm', c_int),
                ('nconv', c_int), ('convs', (c_int * 3) * 4), ('fc_hidden', c_int), ('conv_pad', c_int),
                ('nsub', c_int), ('nvec', c_int * 16)]


NET_MLP, NET_NATURE_CNN, NET_LSTM, NET_CNN_LSTM, NET_CONV_ONLY = 0, 1, 2, 3, 4


class QNetDesc(ctypes.Structure):
    """mirror of mrl_qnet_desc (include/mrl.h)"""
    _fields_ = [('network', c_int), ('ob_ndim', c_int), ('ob_shape', c_int * 3), ('ob_dtype', c_int),
                ('num_layers', c_int), ('num_hidden', c_int), ('activation', c_int), ('nconv', c_int),
                ('convs', (c_int * 3) * 4), ('nhidden', c_int), ('hiddens', c_int * 4), ('dueling', c_int), ('nact', c_int),
                ('layer_norm', c_int), ('body_layer_norm', c_int)]

PD_CATEGORICAL, PD_DIAG_GAUSSIAN, PD_MULTICATEGORICAL, PD_BERNOULLI = 0, 1, 2, 3
OB_F32, OB_U8 = 0, 1
ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2

# name -> (restype, argtypes); the single source of truth for tests/test_abi.py as well
SIGNATURES = {
    'mrl_version': (c_int, []),
    'mrl_strerror': (c_char_p, [c_int]),
    'mrl_device_count': (c_int, []),
    'mrl_gae': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_void_p, c_void_p,
                        c_int, c_int, c_void_p]),
    'mrl_gather_rows': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'mrl_sf01': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'mrl_model_create': (c_int, [ctypes.POINTER(ModelDesc), ctypes.POINTER(c_void_p)]),
    'mrl_model_destroy': (None, [c_void_p]),
    'mrl_model_num_params': (c_long, [c_void_p]),
    'mrl_model_num_tensors': (c_int, [c_void_p]),
    'mrl_model_tensor_info': (c_int, [c_void_p, c_int, c_char_p, c_int, ctypes.POINTER(c_int),
                                      ctypes.POINTER(c_int * 4), ctypes.POINTER(c_long), ctypes.POINTER(c_double)]),
    'mrl_model_workspace_bytes': (c_size_t, [c_void_p, c_int]),
    'mrl_model_act': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_size_t, c_int, c_void_p]),
    'mrl_model_grad': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_int, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p,
                               c_size_t, c_int, c_void_p]),
    'mrl_model_state_size': (c_int, [c_void_p]),
    'mrl_model_act_rnn': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    'mrl_model_grad_rnn': (c_int, [c_void_p] * 9 + [c_int, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float,
                                                    c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    'mrl_model_grad_micro': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_int, c_void_p]),
    'mrl_clip_accumulate': (c_int, [c_void_p, c_void_p, c_long, c_float, c_float, c_int, c_void_p, c_void_p]),
    'mrl_adam_scratch_bytes': (c_size_t, [c_long]),
    'mrl_adam_clip_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float,
                                   c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    'mrl_adam_clip_step_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_float, c_float,
                                       c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    'mrl_qnet_create': (c_int, [ctypes.POINTER(QNetDesc), ctypes.POINTER(c_void_p)]),
    'mrl_qnet_destroy': (None, [c_void_p]),
    'mrl_qnet_num_params': (c_long, [c_void_p]),
    'mrl_qnet_num_tensors': (c_int, [c_void_p]),
    'mrl_qnet_tensor_info': (c_int, [c_void_p, c_int, c_char_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int * 4),
                                     ctypes.POINTER(c_long), ctypes.POINTER(c_int), ctypes.POINTER(c_double)]),
    'mrl_qnet_workspace_bytes': (c_size_t, [c_void_p, c_int]),
    'mrl_qnet_values': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    'mrl_qnet_act': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_size_t, c_int, c_void_p]),
    'mrl_qnet_td_grad': (c_int, [c_void_p] * 9 + [c_float, c_int, c_int] + [c_void_p] * 4 + [c_size_t, c_void_p]),
    'mrl_advstat_minibatches': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'mrl_model_set_advstat': (c_int, [c_void_p, c_void_p]),
    'mrl_qnet_policy_kl': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'mrl_qnet_adam_step': (c_int, [c_void_p] * 5 + [c_float, c_void_p] + [c_float] * 4 + [c_void_p, c_size_t, c_int, c_void_p]),
    'mrl_synth_env_obs': (c_int, [ctypes.c_uint32, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mrl_synth_env_step': (c_int, [ctypes.c_uint32, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    'mrl_replay_insert': (c_int, [c_void_p] * 5 + [c_long, c_long, c_int, c_int] + [c_void_p] * 5 + [c_void_p]),
    'mrl_replay_gather': (c_int, [c_void_p] * 6 + [c_int, c_int] + [c_void_p] * 5 + [c_void_p]),
    'mrl_segtree_init': (c_int, [c_void_p, c_void_p, c_long, c_void_p]),
    'mrl_segtree_set': (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_int, c_void_p]),
    'mrl_segtree_set_ring': (c_int, [c_void_p, c_void_p, c_long, c_long, c_long, c_int, c_double, c_void_p]),
    'mrl_segtree_set_ring_dev': (c_int, [c_void_p, c_void_p, c_long, c_long, c_long, c_int, c_void_p, c_double, c_void_p]),
    'mrl_per_update_from_td': (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_double, c_double, c_void_p,
                                       c_int, c_void_p]),
    'mrl_per_sample': (c_int, [c_void_p, c_void_p, c_long, c_long, c_int, c_void_p, c_double, c_void_p, c_void_p,
                               c_void_p, c_void_p]),
    'mrl_dqn_td_scratch_bytes': (c_size_t, [c_int]),
    'mrl_dqn_td': (c_int, [c_void_p] * 7 + [c_float, c_int, c_int] + [c_void_p] * 4 + [c_void_p]),
    'mrl_framestack_step': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_int, c_int, c_void_p]),
    'mrl_vecnorm_ob': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_double, c_double, c_double, c_void_p,
                               c_void_p, c_void_p]),
    'mrl_vecnorm_rew': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_double, c_double, c_double, c_double,
                                c_void_p, c_void_p, c_void_p]),
    'mrl_tune_set': (c_int, [c_char_p, c_int]),
    'mrl_set_option': (c_int, [c_char_p, c_int]),
    'mrl_get_option': (c_int, [c_char_p, ctypes.POINTER(c_int)]),
    'mrl_model_train_step': (c_int, [c_void_p] * 11 + [c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_void_p,
                                     c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_size_t,
                                     c_int, c_void_p]),
    'mrl_comm_unique_id': (c_int, [c_void_p]),
    'mrl_comm_create': (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    'mrl_comm_destroy': (None, [c_void_p]),
    'mrl_comm_size': (c_int, [c_void_p]),
    'mrl_comm_rank': (c_int, [c_void_p]),
    'mrl_comm_last_error': (c_char_p, []),
    'mrl_allreduce_grads': (c_int, [c_void_p, c_void_p, c_long, c_void_p]),
    'mrl_broadcast_state': (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    'mrl_model_attach_comm': (c_int, [c_void_p, c_void_p, c_float]),
    'mrl_prof_enable': (c_int, [c_int]),
    'mrl_prof_num_labels': (c_int, []),
    'mrl_prof_get': (c_int, [c_int, c_char_p, c_int, ctypes.POINTER(c_long), ctypes.POINTER(c_double),
                             ctypes.POINTER(c_double), ctypes.POINTER(c_double)]),
}

_lib = None


class MrlError(RuntimeError):
    pass


def load():
    """Load libmrl.so; raises if it has not been built (python -m baselines_amd.csrc.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MrlError('libmrl.so not found at %s -- build it with `python -m baselines_amd.csrc.build` '
                       '(there is no CPU fallback)' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if not hasattr(lib, name):
            continue   # optional groups (envs / replay) are bound by their own modules
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def bind(name, restype, argtypes):
    """Bind an additional symbol (used by optional modules)."""
    lib = load()
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    SIGNATURES.setdefault(name, (restype, argtypes))
    return fn


def check(rc, what='mrl call'):
    if rc != 0:
        msg = load().mrl_strerror(int(rc))
        text = msg.decode() if msg else '?'
        if int(rc) == -4:
            text += ': ' + load().mrl_comm_last_error().decode()
        raise MrlError('%s failed: rc=%d (%s)' % (what, rc, text))


_gpu_ok = None


def require_gpu():
    """Raise unless a HIP device is usable.  Called by every device object constructor."""
    global _gpu_ok
    if _gpu_ok:
        return
    import torch
    lib = load()
    if not torch.cuda.is_available() or lib.mrl_device_count() <= 0:
        raise MrlError('baselines_amd needs an AMD GPU (gfx950): no HIP device visible and there is no CPU path')
    _gpu_ok = True


def stream_ptr():
    """hipStream_t of torch's current stream (so our launches order with torch's allocator/copies)."""
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)"""
    if t is None:
        return c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), 'expected a contiguous device tensor'
    return c_void_p(t.data_ptr())


_PROF_ON = False


def prof_enable(on=True):
    global _PROF_ON
    _PROF_ON = bool(on)
    load().mrl_prof_enable(1 if on else 0)


def prof_enabled():
    """True while per-kernel HIP events are being recorded (they cannot be recorded inside a launch-graph capture)."""
    return _PROF_ON


def prof_report():
    """{label: dict(count, ms, flops, bytes)} accumulated since prof_enable(True)."""
    lib = load()
    out = {}
    name = ctypes.create_string_buffer(128)
    for i in range(lib.mrl_prof_num_labels()):
        cnt, ms, fl, by = c_long(), c_double(), c_double(), c_double()
        check(lib.mrl_prof_get(i, name, 128, ctypes.byref(cnt), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)))
        if cnt.value:
            out[name.value.decode()] = dict(count=cnt.value, ms=ms.value, flops=fl.value, bytes=by.value)
    return out


def tune_set(label, variant):
    """Override the GEMM tile variant of one launch site ("c1.fwd", ...); variant < 0 restores the default."""
    check(load().mrl_tune_set(label.encode(), int(variant)), 'mrl_tune_set')


def set_option(name, value):
    """Engine option (include/mrl.h: "u8_bf16x3", "mlp_fused", ...)."""
    check(load().mrl_set_option(name.encode(), int(value)), 'mrl_set_option')


@contextlib.contextmanager
def capture_graph(graph):
    """`torch.cuda.graph(graph, capture_error_mode='thread_local')` with the cyclic garbage collector held off.  A collection
    that happens to run while the stream is capturing destroys whatever unreachable objects it finds -- an old Model's pinned
    staging buffers, events, device tensors of a finished test -- and the runtime refuses those calls during capture (the
    process aborts).  `torch.cuda.graph` runs one explicit collection on entry, before the capture starts; the automatic
    collector stays off until the capture has ended."""
    import gc
    import torch
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):
            yield
    finally:
        if was_enabled:
            gc.enable()


def get_option(name):
    """Value in effect of an engine option (default, environment variable or set_option)."""
    v = c_int()
    check(load().mrl_get_option(name.encode(), ctypes.byref(v)), 'mrl_get_option')
    return v.value
