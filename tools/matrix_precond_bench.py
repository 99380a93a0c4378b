#!/usr/bin/env python
"""GMRES / MINRES with a preconditioner (a first-order approximate inverse with the sparsity of A) given three ways:
as a sparse MATRIX (inside the fused step), as a composite device operator and as a host callable (the Gram-Schmidt
part in the fused step, M applied once per step outside it).  python tools/matrix_precond_bench.py [nx]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import bench  # noqa: E402
from krypy_amd import _hip, linsys, utils  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
A = bench.laplace2d(nx, nx)
N = A.shape[0]
D = sp.identity(N) * 0.25
M = (2 * D - D @ A @ D).tocsr()
b = np.random.default_rng(0).standard_normal(N)
ctx = _hip.get_context()
for label, Mop in (("matrix M inside the fused step", M),
                   ("same M as a composite operator (device, applied between the fused GS and a rescaling)",
                    utils.MatrixLinearOperator(M) * utils.MatrixLinearOperator(sp.identity(N).tocsr())),
                   ("same M as a host callable (download -> callback -> upload once per step)",
                    utils.LinearOperator((N, N), float, dot=M.dot, dot_adj=M.dot))):
    for cls, kw, m in ((linsys.Gmres, {}, 60), (linsys.Minres, dict(self_adjoint=True), 200)):
        ls = linsys.LinearSystem(A, b, M=Mop, **kw)
        best = 1e9
        for _ in range(3):
            ctx.sync()
            t0 = time.perf_counter()
            try:
                s = cls(ls, tol=1e-14, maxiter=m)
            except utils.ConvergenceError as e:
                s = e.solver
            ctx.sync()
            best = min(best, time.perf_counter() - t0)
        n_it = len(s.resnorms) - 1
        print("N = %d %-8s %-90s %7.0f it/s (%.0f us per iteration)" % (N, cls.__name__, label, n_it / best, best / n_it * 1e6))
