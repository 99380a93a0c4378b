#!/usr/bin/env python
"""cProfile of the host side of MINRES + Jacobi (config 3) at N = 10^7: python tools/host_profile_minres.py"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import bench  # noqa: E402
from krypy_amd import linsys, utils  # noqa: E402

A = bench.laplace2d(4000, 2500)
N = A.shape[0]
b = np.random.default_rng(0).standard_normal(N)
d = A.diagonal()
ls = linsys.LinearSystem(A, b, M=sp.diags(1.0 / d).tocsr(), Minv=sp.diags(d).tocsr(), self_adjoint=True)


def run(steps):
    try:
        return linsys.Minres(ls, ortho="lanczos", tol=1e-12, maxiter=steps)
    except utils.ConvergenceError as e:
        return e.solver


run(50)
pr = cProfile.Profile()
pr.enable()
run(400)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
