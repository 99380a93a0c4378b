import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sp
import bench
from krypy_amd import _hip, linsys, utils
ctx = _hip.get_context()
for nx in (100, 316, 1000):
    A = bench.laplace2d(nx, nx)
    b = np.random.default_rng(0).standard_normal(A.shape[0])
    ls = linsys.LinearSystem(A, b, M=sp.diags(1.0 / A.diagonal()).tocsr(), self_adjoint=True, positive_definite=True)
    out = []
    for cyc in ("1", "0"):
        os.environ["KRYPY_AMD_CG_CYCLE"] = cyc
        def run():
            try: return linsys.Cg(ls, tol=1e-30, maxiter=300)
            except utils.ConvergenceError as e: return e.solver
        run(); ctx.sync()
        t0 = time.perf_counter(); n = 0
        for _ in range(5):
            s = run(); n += len(s.resnorms) - 1
        ctx.sync()
        out.append("%s %.0f it/s" % ("cycle" if cyc == "1" else "per-step", n / (time.perf_counter() - t0)))
    print("CG + Jacobi N = %d: %s" % (A.shape[0], ", ".join(out)), flush=True)
