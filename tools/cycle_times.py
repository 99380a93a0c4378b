"""Debug: per-cycle wall times of RestartedGmres on the bench problem (prints to stdout)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from krypy_amd import _hip, linsys, utils

ctx = _hip.get_context()
A = bench.laplace2d(4000, 2500)
b = np.random.default_rng(0).standard_normal(A.shape[0])
ls = linsys.LinearSystem(A, b)
x0 = None
for c in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    ctx.sync(); t0 = time.perf_counter()
    try:
        s = linsys.Gmres(ls, x0=x0, maxiter=100, tol=1e-8, ortho="mgs")
    except utils.ConvergenceError as e:
        s = e.solver
    ctx.sync(); t1 = time.perf_counter()
    x0 = s.__dict__["_xk_dev"]
    print("cycle %d: %.1f ms" % (c, (t1 - t0) * 1e3), flush=True)
