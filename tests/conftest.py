import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests skip (instead of failing) on a box without a HIP device or without the built library"""
    reason = None
    if not os.path.exists(os.path.join(ROOT, 'baselines_amd', 'csrc', 'libmrl.so')):
        reason = 'libmrl.so is not built (python -m baselines_amd.csrc.build)'
    else:
        try:
            import torch
            if not torch.cuda.is_available():
                reason = 'no HIP device visible'
        except Exception as exc:           # pragma: no cover
            reason = 'torch unavailable: %s' % (exc,)
    if reason:
        skip = pytest.mark.skip(reason=reason)
        for item in items:
            if 'gpu' in item.keywords:
                item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
