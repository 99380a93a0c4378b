#!/usr/bin/env python
"""Where the host time of a SHORT solve goes: cProfile of 200 MINRES + Jacobi solves of 150 steps at N = 10^4 (the solves
tools/small_bench.py times), sorted by cumulative time: python tools/host_profile_setup.py [nx]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import bench  # noqa: E402
from krypy_amd import _hip, linsys, utils  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 100
if len(sys.argv) > 2:                      # pin the Lanczos window (66: the window of long vectors)
    utils.Arnoldi._WINDOW_COLS = int(sys.argv[2])
A = bench.laplace2d(nx, nx)
b = np.random.default_rng(0).standard_normal(A.shape[0])
ls = linsys.LinearSystem(A, b, M=sp.diags(1.0 / A.diagonal()).tocsr(), self_adjoint=True)
ctx = _hip.get_context()


def run():
    try:
        return linsys.Minres(ls, maxiter=150, tol=1e-30)
    except utils.ConvergenceError as e:
        return e.solver


for _ in range(20):
    run()
ctx.sync()
t0 = time.perf_counter()
for _ in range(100):
    run()
ctx.sync()
dt = (time.perf_counter() - t0) / 100
print("one 150-step solve: %.1f us = %.0f it/s" % (dt * 1e6, 150 / dt))
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    run()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
