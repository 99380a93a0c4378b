import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture
def golden():
    return load_golden


@pytest.fixture
def cpu_double():
    """Install the NumPy test double as the process-wide context (CPU tests of the host layer)."""
    from krypy_amd import _hip
    from tests.support.numpy_context import NumpyContext

    ctx = NumpyContext()
    old = _hip._install_context_for_testing(ctx)
    yield ctx
    _hip._install_context_for_testing(old)


_real_ctx = []


@pytest.fixture
def hip():
    """The real HIP context (GPU tests).  Fails - never skips - when the library or GPU is absent."""
    from krypy_amd import _hip

    if not _real_ctx:
        if not os.path.exists(_hip.library_path()):
            import __graft_entry__
            __graft_entry__.build()
        _hip._install_context_for_testing(None)
        _real_ctx.append(_hip.get_context())
    _hip._install_context_for_testing(_real_ctx[0])
    return _real_ctx[0]
