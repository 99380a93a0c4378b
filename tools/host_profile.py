#!/usr/bin/env python
"""cProfile of the host side of a few GMRES(100) cycles at the bench size: where does Python spend
its time while the GPU works (python tools/host_profile.py [cycles])."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from krypy_amd import linsys, utils  # noqa: E402

ncyc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
A = bench.laplace2d(4000, 2500)
b = np.random.default_rng(0).standard_normal(A.shape[0])
ls = linsys.LinearSystem(A, b)


def run(n, x0=None):
    try:
        s = linsys.RestartedGmres(ls, x0=x0, maxiter=100, max_restarts=n - 1, tol=1e-8)
    except utils.ConvergenceError as e:
        s = e.solver
    return s


s = run(2)
pr = cProfile.Profile()
pr.enable()
run(ncyc, s.__dict__["_xk_dev"])
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(12)
st.print_callers("download|upload|kh_vec_free|alloc")
