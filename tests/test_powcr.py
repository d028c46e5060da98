"""csrc/powcr.hip.h: the correctly rounded pow the device kernels use for prioritized-replay leaves and importance weights
(deepq/replay_buffer.py:169-191 computes them as `priority ** alpha` on Python floats).  Plain C++, so the host compiler can
check it here: against a 70-digit decimal evaluation (every result must be the nearest double) and against Python's own `**`
(glibc's pow, which misrounds about one input in a thousand: that is the only disagreement allowed)."""
import ctypes
import os
import subprocess
from decimal import Decimal, getcontext

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def powcr(tmp_path_factory):
    d = tmp_path_factory.mktemp('powcr')
    src = d / 'powcr_host.cpp'
    src.write_text('#include "powcr.hip.h"\n'
                   'extern "C" void pow_cr_many(const double* x, double a, double* out, long n) {\n'
                   '    for (long i = 0; i < n; ++i) out[i] = mrl::pow_cr(x[i], a);\n}\n')
    lib = d / 'libpowcr_host.so'
    subprocess.check_call(['g++', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-I' + os.path.join(ROOT, 'baselines_amd', 'csrc'),
                           str(src), '-o', str(lib)])
    L = ctypes.CDLL(str(lib))

    def f(x, a):
        x = np.ascontiguousarray(x, np.float64)
        out = np.empty_like(x)
        L.pow_cr_many(x.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(a), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(x.size))
        return out
    return f


def _inputs(n, seed):
    rng = np.random.RandomState(seed)
    return np.concatenate([np.abs(rng.randn(n)) * 3 + 1e-6,            # |td| + eps of a learner step
                           10.0 ** rng.uniform(-12, 6, n // 2),        # many decades
                           rng.uniform(0.5, 2.0, n // 2)])             # around 1, where log x cancels


@pytest.mark.parametrize('a', [0.6, 0.4, -0.4, 0.7, 1.0 / 3.0, -1.0])
def test_pow_cr_is_the_nearest_double(powcr, a):
    getcontext().prec = 70
    x = _inputs(1500, 3)
    got = powcr(x, a)
    ad = Decimal(a)
    for v, g in zip(x, got):
        exact = (Decimal(float(v)).ln() * ad).exp()
        assert float(exact) == g, (v, a, g, float(exact))


def test_pow_cr_equals_pythons_pow_except_where_glibc_misrounds(powcr):
    x = _inputs(100000, 4)
    for a in (0.6, 0.4, -0.4):
        got = powcr(x, a)
        ref = np.array([float(v) ** a for v in x])
        ulp = np.abs(got.view(np.int64) - ref.view(np.int64))
        assert ulp.max() <= 1
        assert (ulp != 0).mean() <= 3e-3, (a, (ulp != 0).mean())       # measured 0.7e-3 .. 1.0e-3: glibc's own misroundings


def test_pow_cr_special_values(powcr):
    assert powcr(np.array([1.0, 2.0, 0.0, 4.0, 5e-324, 1e308]), 0.5).tolist()[:4] == [1.0, 2.0 ** 0.5, 0.0, 2.0]
    assert powcr(np.array([3.0]), 0.0)[0] == 1.0 and powcr(np.array([3.0]), 1.0)[0] == 3.0
    sub = powcr(np.array([5e-324, 1e-310]), 0.6)
    assert np.all(sub == np.array([5e-324 ** 0.6, 1e-310 ** 0.6]))
