"""Import the unmodified reference (``/root/reference/krypy``) on NumPy 2 / SciPy 1.15.

BUILD-CONTAINER ONLY: ``/root/reference`` does not exist on the GPU box, and
nothing in ``tests -m gpu``, ``smoke()`` or ``bench.py`` imports this module.
It is used by ``oracle/gen_golden.py`` (to emit the committed fixtures under
``tests/golden/``) and by ``tests/test_differential_vs_reference.py`` (skipped when the
reference tree is absent).

The reference uses names that NumPy 2 / SciPy 1.15 removed;
they are re-injected here *before* ``import krypy``; the reference tree itself
is never modified or copied.
"""
import os
import sys

REFERENCE_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "krypy"))


def load():
    """Return the reference ``krypy`` module (raises ImportError when absent)."""
    if not available():
        raise ImportError("reference tree %s not present" % REFERENCE_ROOT)
    import types

    import numpy
    import scipy.sparse
    from scipy.sparse._sputils import isintlike

    if "scipy.sparse.sputils" not in sys.modules or not hasattr(
            sys.modules["scipy.sparse.sputils"], "isintlike"):
        mod = types.ModuleType("scipy.sparse.sputils")
        mod.isintlike = isintlike
        sys.modules["scipy.sparse.sputils"] = mod
        scipy.sparse.sputils = mod

    def find_common_type(array_types, scalar_types):
        ts = [numpy.dtype(t) for t in list(array_types) + list(scalar_types) if t is not None]
        return numpy.result_type(*ts) if ts else numpy.dtype(float)

    if not hasattr(numpy, "find_common_type"):
        numpy.find_common_type = find_common_type
    for name, val in (("float", float), ("complex", complex), ("int", int),
                      ("Inf", numpy.inf), ("Infinity", numpy.inf)):
        if name not in numpy.__dict__:
            setattr(numpy, name, val)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import krypy

    return krypy
