"""krypy_amd - an MI355X-native Krylov solver core behind the KryPy API.

Drop-in for the hot path of ``krypy`` (``krypy.gmres(A, b)``, ``krypy.linsys``,
``krypy.deflation``, ``krypy.utils.Arnoldi``): same names and semantics, with the
per-iteration work (CSR SpMV / dense matvec, Gram-Schmidt against the growing basis,
vector recurrences of CG/MINRES/GMRES, the deflation projector) running as hand-written
HIP kernels for gfx950 through a ctypes C-ABI (``include/krylov_hip.h``).

There is no CPU fallback: without ``libkrylov_hip.so`` and an MI355X every solve raises
``krypy_amd.utils.BackendError``.
"""
from . import deflation, linsys, recycling, utils
from .__about__ import __version__
from ._convenience import cg, gmres, minres

__all__ = ["linsys", "deflation", "recycling", "utils", "cg", "minres", "gmres", "__version__"]
