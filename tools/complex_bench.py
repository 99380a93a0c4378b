#!/usr/bin/env python
"""GMRES(100) on a complex (c128) shifted 2-D Laplacian, N = 5*10^6 (same bytes per vector as the real
bench): python tools/complex_bench.py [ortho ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import bench  # noqa: E402
from krypy_amd import _hip, linsys, utils  # noqa: E402

ctx = _hip.get_context()
NX, NY = (int(os.environ.get('ZNX', 2500)), int(os.environ.get('ZNY', 2000)))
A = bench.laplace2d(NX, NY).astype(complex)
N = A.shape[0]
A = (A + sp.diags(1j * np.linspace(0.1, 1.0, N))).tocsr()
rng = np.random.default_rng(0)
b = rng.standard_normal(N) + 1j * rng.standard_normal(N)
ls = linsys.LinearSystem(A, b)
if "minres" in sys.argv[1:]:
    # complex Hermitian MINRES (krypy/linsys.py:791-853 on c128 data): the Laplacian between two diagonals of phases, 150 steps
    sys.argv.remove("minres")
    ph = sp.diags(np.exp(1j * rng.uniform(0, 2 * np.pi, N)))
    Ah = (ph.conj() @ bench.laplace2d(NX, NY).astype(complex) @ ph).tocsr()
    Ah.sort_indices()
    lsh = linsys.LinearSystem(Ah, b, self_adjoint=True)
    for it in range(4):          # (the first solves pay for code-object loading and the block pool: the fourth is reported)
        ctx.sync()
        t0 = time.perf_counter()
        try:
            s = linsys.Minres(lsh, maxiter=150, tol=1e-30)
        except utils.ConvergenceError as e:
            s = e.solver
        ctx.sync()
        dt = time.perf_counter() - t0
    print(json.dumps({"config": "complex Hermitian MINRES, N=%d, 150 steps" % N, "iterations_per_s": (len(s.resnorms) - 1) / dt,
                      "relres": float(s.resnorms[-1])}))
    if not sys.argv[1:]:
        sys.exit(0)
for ortho in (sys.argv[1:] or ["mgs", "cgs"]):
    for it in range(2):
        ctx.sync()
        t0 = time.perf_counter()
        try:
            s = linsys.RestartedGmres(ls, maxiter=100, max_restarts=1, tol=1e-14, ortho=ortho)
        except utils.ConvergenceError as e:
            s = e.solver
        ctx.sync()
        dt = time.perf_counter() - t0
    n_it = len(s.resnorms) - 1
    print(json.dumps({"config": "complex GMRES(100), N=%d, ortho=%s" % (N, ortho), "iterations_per_s": n_it / dt,
                      "ms_per_cycle": dt / 2 * 1e3, "relres": float(s.resnorms[-1])}))
