#!/usr/bin/env python
"""cProfile of the host side of the config-5 shape (DeflatedGmres(100), 16 Ritz vectors, 3-D 7-pt
200^3) plus the kernel timeline: python tools/host_profile_cfg5.py"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import oracle.krylov_ref as ref  # noqa: E402
from krypy_amd import deflation, linsys, utils  # noqa: E402

A = ref.laplace3d(200)
N = A.shape[0]
b = np.random.default_rng(0).standard_normal(N)
ls = linsys.LinearSystem(A, b, self_adjoint=True)


def run(U=None, **kw):
    try:
        return deflation.DeflatedGmres(ls, U=U, tol=1e-12, maxiter=100, **kw)
    except utils.ConvergenceError as e:
        return e.solver


s0 = run(store_arnoldi=True)
ritz = deflation.Ritz(s0)
U = ritz._get_vectors_dev(np.argsort(np.abs(ritz.values))[:16])
run(U)
pr = cProfile.Profile()
pr.enable()
run(U)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
