import numpy as np


def explained_variance(ypred, y):
    """1 - Var[y - ypred] / Var[y]; nan when Var[y] == 0  (reference: common/math_util.py:25-38)."""
    assert y.ndim == 1 and ypred.ndim == 1
    vary = np.var(y)
    return np.nan if vary == 0 else 1 - np.var(y - ypred) / vary


def safemean(xs):
    """reference: ppo2/ppo2.py:220-221"""
    return np.nan if len(xs) == 0 else np.mean(xs)
