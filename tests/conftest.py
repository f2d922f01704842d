import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: full-network CPU oracle runs (tens of seconds)')


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


@pytest.fixture(scope='session')
def hip():
    """The C-ABI library through the product loader; GPU tests fail loudly if it is missing."""
    import torch
    assert torch.cuda.is_available(), 'gpu-marked test needs a GPU'
    from segland_amd import _lib
    return _lib.lib()
