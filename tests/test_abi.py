"""CPU-side checks of the C ABI: the shared library builds for gfx950, loads, and exports every
symbol that include/krylov_hip.h declares (no compute calls - there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "krylov_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(kh_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from krypy_amd import _hip

    path = _hip.library_path()
    assert os.path.exists(path), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(path)
    declared = _header_symbols()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), "libkrylov_hip.so lacks %s declared in include/krylov_hip.h" % name
    # the ctypes binding declares exactly the header's entry points
    assert sorted(_hip.exported_symbols()) == declared
    lib.kh_version.restype = ctypes.c_int
    assert lib.kh_version() >= 100


def test_no_gpu_fails_loudly_instead_of_falling_back():
    """Without a visible GPU the product raises BackendError; it never computes on the CPU."""
    from krypy_amd import _hip

    lib = _hip.load_library()
    n = ctypes.c_int(0)
    lib.kh_device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is visible on this machine")
    old = _hip._install_context_for_testing(None)
    try:
        with pytest.raises(_hip.BackendError):
            _hip.get_context()
        import numpy as np
        import krypy_amd
        with pytest.raises(_hip.BackendError):
            krypy_amd.gmres(np.eye(4), np.ones(4))
    finally:
        _hip._install_context_for_testing(old)


def test_product_does_not_import_oracle_or_test_double():
    pkg = os.path.join(ROOT, "krypy_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn
            assert "numpy_context" not in src and "tests.support" not in src, fn
            assert "import torch" not in src, fn
