#!/usr/bin/env python
"""Host side of GMRES(100) at a SMALL size (default N = 316^2 ~ 1e5), where a step's device work is tens of
microseconds: cProfile of the Python between the C calls (python tools/host_profile_small.py [nx] [ortho])."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from krypy_amd import _hip, linsys, utils  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 316
ortho = sys.argv[2] if len(sys.argv) > 2 else "mgs"
A = bench.laplace2d(nx, nx)
b = np.random.default_rng(0).standard_normal(A.shape[0])
ls = linsys.LinearSystem(A, b)
ctx = _hip.get_context()


def run(n, x0=None):
    try:
        s = linsys.RestartedGmres(ls, x0=x0, maxiter=100, max_restarts=n - 1, tol=1e-12, ortho=ortho)
    except utils.ConvergenceError as e:
        s = e.solver
    return s


s = run(2)
ctx.sync()
t0 = time.perf_counter()
s2 = run(10, s.__dict__["_xk_dev"])
ctx.sync()
dt = time.perf_counter() - t0
n_it = len(s2.resnorms) - 1
print("N = %d, ortho = %s: %.0f iterations/s (%.1f us per iteration)" % (A.shape[0], ortho, n_it / dt, dt / n_it * 1e6))
pr = cProfile.Profile()
pr.enable()
run(10, s.__dict__["_xk_dev"])
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(14)
