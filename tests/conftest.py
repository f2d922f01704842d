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


@pytest.fixture(autouse=True)
def _reset_debug_hooks(request):
    """The sl_debug_* hooks (include/segland_hip_debug.h) write process-wide dispatch state: whatever a GPU test set -- also one that failed between set and reset -- is
    put back before the next test runs."""
    yield
    if request.node.get_closest_marker('gpu') is not None:
        from segland_amd import _lib
        if _lib._lib is not None:
            _lib._lib.sl_debug_reset()


@pytest.fixture(autouse=True)
def _parity_log(request):
    """SEGLAND_PARITY_LOG=<file>: what the tests print (differing-pixel counts, cosines, relative errors: the numbers behind the tolerance gates) is appended to that file,
    one section per test -- `pytest -q` drops it otherwise.  profiles/r4_parity_log.txt is such a file from one MI355X box."""
    path = os.environ.get('SEGLAND_PARITY_LOG')
    capsys = request.getfixturevalue('capsys') if path else None      # without the variable nothing about the capture changes
    yield
    if path:
        out = capsys.readouterr().out
        if out.strip():
            with open(path, 'a') as f:
                f.write('## %s\n%s\n' % (request.node.nodeid, out.rstrip()))
