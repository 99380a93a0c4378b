#!/usr/bin/env python
"""GMRES(100) / MINRES at the sizes KryPy is mostly used at (N = 10^4 ... 10^6): iterations/s with the chain kernel's
workgroups on one XCD (default) and spread over the chip (KRYPY_AMD_CHAIN_ONEX=0 in a child process).
    python tools/small_bench.py [nx ...]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


_LS = {}


def linsys_minres(linsys, utils, A, b, sp):
    ls = _LS.get(id(A))
    if ls is None:          # (one LinearSystem: the operator is uploaded once)
        ls = _LS[id(A)] = linsys.LinearSystem(A, b, M=sp.diags(1.0 / A.diagonal()).tocsr(), self_adjoint=True)
    try:
        return linsys.Minres(ls, maxiter=150, tol=1e-30)
    except utils.ConvergenceError as e:
        return e.solver


def one(nx):
    import numpy as np
    import scipy.sparse as sp
    import bench
    from krypy_amd import _hip, linsys, utils

    ctx = _hip.get_context()
    A = bench.laplace2d(nx, nx)
    b = np.random.default_rng(0).standard_normal(A.shape[0])
    out = []
    for name, make in (("gmres mgs", lambda x0, n: linsys.RestartedGmres(linsys.LinearSystem(A, b), x0=x0, maxiter=100, max_restarts=n - 1, tol=1e-14, ortho="mgs")),
                       ("gmres cgs", lambda x0, n: linsys.RestartedGmres(linsys.LinearSystem(A, b), x0=x0, maxiter=100, max_restarts=n - 1, tol=1e-14, ortho="cgs")),
                       # (the solve converges within a few hundred iterations at these sizes: fixed runs of 150 steps, n of them)
                       ("minres jacobi", lambda x0, n: [linsys_minres(linsys, utils, A, b, sp) for _ in range(n)][-1])):
        def run(x0, n):
            try:
                return make(x0, n)
            except utils.ConvergenceError as e:
                return e.solver
        s = run(None, 2)
        ctx.sync()
        t0 = time.perf_counter()
        s2 = run(None, 10)
        ctx.sync()
        dt = time.perf_counter() - t0
        out.append("%s %.0f it/s" % (name, ((len(s2.resnorms) - 1) * (10 if name.startswith("minres") else 1)) / dt))
    c = ctx.counters()
    print("N = %7d, onex = %s (%d one-XCD launches, %d chain launches of which %d with the operator in the prologue, %d "
          "iterations through kh_gmres_cycle): %s" % (A.shape[0], os.environ.get("KRYPY_AMD_CHAIN_ONEX", "1"),
                                                      ctx.get("n_chain_onex"), c["chain"], c["chain_fused"],
                                                      ctx.get("n_cycle_steps"), ", ".join(out)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "one":
        one(int(sys.argv[2]))
    else:
        for nx in [int(a) for a in sys.argv[1:]] or [100, 316, 500, 1000]:
            for onex in ("1", "0"):
                subprocess.run([sys.executable, __file__, "one", str(nx)], env=dict(os.environ, KRYPY_AMD_CHAIN_ONEX=onex))
