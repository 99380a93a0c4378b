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


def _gpu_visible():
    try:
        import ctypes
        from krypy_amd import _hip
        lib = ctypes.CDLL(_hip.library_path())
        n = ctypes.c_int(0)
        return lib.kh_device_count(ctypes.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """A plain ``pytest tests`` on a machine without a GPU runs the CPU suite and SKIPS the gpu-marked tests.
    With an explicit ``-m`` expression nothing is skipped: ``-m gpu`` on a box without a working device or without
    the HIP library fails loudly (the `hip` fixture raises), it never passes on a fallback."""
    if config.getoption("-m") or _gpu_visible():
        return
    skip = pytest.mark.skip(reason="no MI355X visible (run with -m gpu on a GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _kernel_expectations_are_judged_last(request):
    """tests/support/kernel_expect.py: 'which kernel ran' expectations are reported after the test body (all numeric
    comparisons) has passed - as an error at teardown that says so."""
    from tests.support import kernel_expect
    kernel_expect.drain()
    yield
    failed = kernel_expect.drain()
    if failed and getattr(request.node, "_call_passed", False):      # (a failed body already speaks for itself)
        pytest.fail("KERNEL-PATH EXPECTATION (all numeric comparisons of this test passed): " + "; ".join(failed),
                    pytrace=False)


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    outcome = yield
    rep = outcome.get_result()
    if rep.when == "call":
        item._call_passed = rep.passed


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


def _cap_host_memory():
    """Fail-safe for the GPU box: a test that asks for absurd host memory (scipy.sparse.random at n = 3e5 wants
    n^2 integers, 671 GiB) must die with a MemoryError, not take the box down with it.  RLIMIT_DATA counts private
    writable mappings (NumPy's buffers), not the device / pinned mappings of the HIP runtime; applied after the HIP
    context exists.  KRYPY_AMD_TEST_RLIMIT_GB=0 switches it off (default 96 GB; the largest test holds ~30 GB)."""
    try:
        import resource
        gb = float(os.environ.get("KRYPY_AMD_TEST_RLIMIT_GB", "96"))
        if gb > 0:
            soft, hard = resource.getrlimit(resource.RLIMIT_DATA)
            want = int(gb * (1 << 30))
            if hard != resource.RLIM_INFINITY:
                want = min(want, hard)
            resource.setrlimit(resource.RLIMIT_DATA, (want, hard))
    except Exception:
        pass


@pytest.fixture
def hip():
    """The real HIP context (GPU tests).  Fails - never skips - when the library or GPU is absent."""
    from krypy_amd import _hip

    if not _real_ctx:
        if not os.path.exists(_hip.library_path()):
            import __graft_entry__
            __graft_entry__.build()
        _hip._install_context_for_testing(None)
        forced = os.environ.get("KRYPY_AMD_TEST_FORCE_MULTI", "") == "1"
        if forced:      # robustness runs: the WHOLE suite through the multi-rank code path (all-reduces of every
            os.environ["KRYPY_AMD_FORCE_MULTI"] = "1"      # inner product, no chain kernel) on a 1-rank communicator
        _real_ctx.append(_hip.get_context())
        if forced:
            _real_ctx[0].comm_init(0, 1, _real_ctx[0].comm_unique_id())
        _cap_host_memory()
    _hip._install_context_for_testing(_real_ctx[0])
    return _real_ctx[0]
