import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle.krylov_ref as ref
from krypy_amd import _hip, deflation, linsys, utils
ctx = _hip.get_context()
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
for i in range(5):
    ctx.sync(); t0 = time.perf_counter()
    s = run(U)
    ctx.sync(); dt = time.perf_counter() - t0
    inf = ctx.info()
    print("solve %d: %.1f ms, free %.1f GB, counters %s spmm %d" % (i, dt * 1e3, inf["mem_free"] / 1e9, ctx.counters(), ctx.get("n_spmm")))
    del s
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); run(U); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
