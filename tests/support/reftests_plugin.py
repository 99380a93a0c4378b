"""pytest plugin for running the REFERENCE'S OWN test-suite (/root/reference/test, read where it lies, never copied) against
either the unmodified reference or krypy_amd's host layer on the NumPy test double - TEST INFRASTRUCTURE, build container only.

``REFTESTS_TARGET=reference``: ``import krypy`` is the reference (through oracle/refshim.py: the names NumPy 2 / SciPy 1.15
removed are re-injected first).  ``REFTESTS_TARGET=krypy_amd``: ``import krypy`` is krypy_amd with the double installed as the
process-wide context - the drop-in claim of INTEGRATION.md section 1 ("switch the import") put to the reference's own tests.
Used by tests/test_reference_suite.py; nothing of the reference travels to the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import refshim  # noqa: E402

_target = os.environ.get("REFTESTS_TARGET", "reference")
ref = refshim.load()        # (also re-injects numpy.float & co., which the reference's test files use themselves)
if _target == "krypy_amd":
    for name in [k for k in sys.modules if k == "krypy" or k.startswith("krypy.")]:
        del sys.modules[name]
    import krypy_amd
    from krypy_amd import _hip
    from tests.support.numpy_context import NumpyContext

    _hip._install_context_for_testing(NumpyContext())
    import krypy_amd.deflation
    import krypy_amd.linsys
    import krypy_amd.recycling
    import krypy_amd.utils

    sys.modules["krypy"] = krypy_amd
    for name in [k for k in list(sys.modules) if k.startswith("krypy_amd.")]:
        sys.modules["krypy." + name[len("krypy_amd."):]] = sys.modules[name]
elif _target != "reference":
    raise RuntimeError("REFTESTS_TARGET: reference or krypy_amd")
