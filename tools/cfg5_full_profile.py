#!/usr/bin/env python
"""Where a deflated solve of config 5 at its full size (N = 10^8, one MI355X) spends its time: cProfile of the host side of one
DeflatedGmres(100) solve with 16 Ritz vectors + the HIP-event time of its Arnoldi loop.   python tools/cfg5_full_profile.py [nz]"""
import cProfile
import gc
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from krypy_amd import _hip, deflation, linsys, utils  # noqa: E402

nz = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ortho = sys.argv[2] if len(sys.argv) > 2 else "mgs"
ctx = _hip.get_context()
t0 = time.perf_counter()
A = bench.laplace3d(500, 500, nz)
N = A.shape[0]
b = np.random.default_rng(0).standard_normal(N)
ls = linsys.LinearSystem(A, b, self_adjoint=True)
print("setup %.1f s, N = %d" % (time.perf_counter() - t0, N), flush=True)


def run(U=None, **kw):
    try:
        return deflation.DeflatedGmres(ls, U=U, tol=1e-12, maxiter=100, ortho=ortho, **kw)
    except utils.ConvergenceError as e:
        return e.solver


t0 = time.perf_counter()
s0 = run(store_arnoldi=True)
ctx.sync()
print("plain solve %.2f s" % (time.perf_counter() - t0), flush=True)
ritz = deflation.Ritz(s0)
U = ritz._get_vectors_dev(np.argsort(np.abs(ritz.values))[:16])
del s0, ritz
gc.collect()
for rep in range(3):
    t0 = time.perf_counter()
    s = run(U)
    ctx.sync()
    t1 = time.perf_counter()
    del s
    gc.collect()
    t2 = time.perf_counter()
    print("deflated solve %d: %.2f s + %.2f s to drop it; pool %.1f GB, free %.1f GB" % (rep, t1 - t0, t2 - t1, ctx._pool_bytes / 1e9,
                                                                                    ctx.info()["mem_free"] / 1e9), flush=True)
pr = cProfile.Profile()
pr.enable()
s = run(U)
ctx.sync()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
