#!/usr/bin/env python
"""Soak: many solves of every kind in one process; device memory must come back (block pool flushed at the end),
the rates must not drift.  python tools/soak.py [rounds=30]"""
import gc
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import bench  # noqa: E402
from krypy_amd import _hip, deflation, linsys, recycling, utils  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ctx = _hip.get_context()
warnings.simplefilter("ignore")


def free_gb():
    gc.collect()
    ctx._pool_flush()
    ctx.sync()
    return ctx.info()["mem_free"] / 1e9


def run(make):
    try:
        return make()
    except utils.ConvergenceError as e:
        return e.solver


A = bench.laplace2d(1000, 800)
N = A.shape[0]
rng = np.random.default_rng(0)
b = rng.standard_normal(N)
Ac = (A + sp.diags(1j * np.linspace(0.1, 1.0, N))).tocsr()
bc = b + 1j * rng.standard_normal(N)
d = np.asarray(A.diagonal())
M, Minv = sp.diags(1.0 / d).tocsr(), sp.diags(d).tocsr()
U = np.linalg.qr(rng.standard_normal((N, 4)))[0]
start = free_gb()
print("free at start: %.2f GB" % start)
t_first = t_last = None
for r in range(rounds):
    t0 = time.perf_counter()
    run(lambda: linsys.RestartedGmres(linsys.LinearSystem(A, b), maxiter=40, max_restarts=2, tol=1e-12))
    run(lambda: linsys.Gmres(linsys.LinearSystem(Ac, bc), maxiter=40, tol=1e-12, ortho="cgs"))
    run(lambda: linsys.Minres(linsys.LinearSystem(A, b, M=M, Minv=Minv, self_adjoint=True), maxiter=60, tol=1e-12))
    run(lambda: linsys.Cg(linsys.LinearSystem(A, b, self_adjoint=True, positive_definite=True), maxiter=60, tol=1e-12,
                          store_arnoldi=True))
    run(lambda: deflation.DeflatedGmres(linsys.LinearSystem(A, b), U=U, maxiter=30, tol=1e-12, store_arnoldi=True))
    rec = recycling.RecyclingMinres(vector_factory=recycling.factories.RitzFactorySimple(n_vectors=3, which="sm"))
    for _ in range(2):
        run(lambda: rec.solve(linsys.LinearSystem(A, b, self_adjoint=True), maxiter=40, tol=1e-12))
    del rec
    ctx.sync()
    dt = time.perf_counter() - t0
    if r == 1:
        t_first = dt
    t_last = dt
    if r % 5 == 0 or r == rounds - 1:
        print("round %3d: %.2f s, free %.2f GB (pool %.2f GB)" % (r, dt, ctx.info()["mem_free"] / 1e9, ctx._pool_bytes / 1e9),
              flush=True)
end = free_gb()
print("free at end: %.2f GB (start %.2f GB); round time %.2f s -> %.2f s" % (end, start, t_first, t_last))
assert start - end < 0.3, "device memory did not come back"
assert t_last < 1.5 * t_first, "rounds got slower"
print("soak ok")
