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


def _sources_reading_the_environment():
    files = []
    for base, _, names in os.walk(os.path.join(ROOT, "krypy_amd")):
        if "build" in base or "__pycache__" in base:
            continue
        files += [os.path.join(base, f) for f in names if f.endswith((".py", ".hip", ".h"))]
    files += [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for base, _, names in os.walk(os.path.join(ROOT, "tests")):
        if "__pycache__" not in base:
            files += [os.path.join(base, f) for f in names if f.endswith(".py")]
    return files


def test_every_environment_switch_is_in_the_registry_documented_and_its_off_path_tested():
    """krypy_amd/_switches.py is THE table of the package's KRYPY_AMD_* variables: (i) every variable any source of the
    package, bench.py or the tests reads through getenv / os.environ is a row of it, (ii) INTEGRATION.md section 4 is that
    table rendered, row for row, (iii) every `kernel-path` switch - one that selects another kernel or host loop for the
    same result - has its off-setting among the fallback suites of tools/r05_evidence.sh, so that no path exists that
    no GPU run exercises."""
    from krypy_amd import _switches

    table = set(_switches.names())
    assert len(table) == len(_switches.names()), "duplicate rows"
    read = set()
    for fn in _sources_reading_the_environment():
        if fn.endswith("_switches.py") or fn.endswith("test_abi.py"):
            continue
        src = open(fn).read()
        read |= set(re.findall(r"(?:getenv\(|environ(?:\.get|\.setdefault|\.pop)?[\(\[]\s*|setenv\(|delenv\(|environ,\s*)[\"'](KRYPY_AMD_[A-Z0-9_]+)", src))
        read |= set(re.findall(r"\b(KRYPY_AMD_[A-Z0-9_]+)=", src))            # env=dict(os.environ, KRYPY_AMD_X=...), shell-style
    missing = sorted(read - table)
    assert not missing, "read somewhere, not in krypy_amd/_switches.py: %s" % missing
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for line in _switches.markdown().splitlines():
        assert line in doc, "INTEGRATION.md section 4 is out of date (python -m krypy_amd._switches): %s" % line[:80]
    script = open(os.path.join(ROOT, "tools", "r05_evidence.sh")).read()
    fb = script[script.index("fallback)"):script.index("fuzz)")]
    for name, off in _switches.kernel_path_switches():
        assert "%s=%s" % (name, off) in fb, "the off-path %s=%s is not among the fallback suites of tools/r05_evidence.sh" % (name, off)
