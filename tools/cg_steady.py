"""Config 4's shape in the steady state: CG on a dense SPD matrix of order 32768 (a cheap one: the symmetric part of a normal
matrix + 300 I - no n^3 product on the host), solves to three tolerances; the slope of time against iterations is the cost of
an iteration, the intercept what a solve costs besides (the explicit residual of the last iteration - one more product, as the
reference forms it, linsys.py:345-390 - and the set-up).  For the GEMV's rows-per-wave shapes.

    python tools/cg_steady.py"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krypy_amd import _hip, linsys
n = 32768
rng = np.random.default_rng(0)
G = rng.standard_normal((n, n))
A = G + G.T
A *= 0.5
A[np.diag_indices(n)] += 300.0
del G
b = rng.standard_normal(n)
ctx = _hip.get_context()
ls = linsys.LinearSystem(A, b, self_adjoint=True, positive_definite=True)
for rep in range(3):
    ctx.sync(); t0 = time.perf_counter()
    s = linsys.Cg(ls, tol=1e-8, maxiter=200)
    ctx.sync(); dt = time.perf_counter() - t0
    print("solve %d: %d iterations, %.3f ms, %.1f it/s, %.1f us per iteration" % (rep, s.iter, dt * 1e3, s.iter / dt, dt / s.iter * 1e6))
for rows in (0, 1, 2):
  ctx.set("gemv_rows", rows)
  pts = []
  for tol in (1e-2, 1e-6, 1e-12):
    best = 1e9
    for rep in range(4):
        ctx.sync(); t0 = time.perf_counter()
        s = linsys.Cg(ls, tol=tol, maxiter=200)
        ctx.sync(); best = min(best, time.perf_counter() - t0)
    pts.append((s.iter, best))
  slope = (pts[-1][1] - pts[0][1]) / (pts[-1][0] - pts[0][0])
  print("dense CG n = 32768, GEMV rows per wave %s: %.1f us per iteration (%.0f it/s in the steady state), %.2f ms per solve besides (the explicit residual of the last iteration is one more product); solves of %s iterations: %s ms" % (rows or "4 (default)", slope * 1e6, 1.0 / slope, (pts[0][1] - pts[0][0] * slope) * 1e3, [p[0] for p in pts], ["%.3f" % (p[1] * 1e3) for p in pts]), flush=True)
