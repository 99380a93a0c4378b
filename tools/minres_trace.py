#!/usr/bin/env python
"""Is a MINRES + Jacobi iteration at small N bound by the device or by the host loop?  Kernel trace of 3 x 150 steps:
    rocprofv3 --kernel-trace -d DIR -o mt -- python tools/minres_trace.py run [nx]
    python tools/minres_trace.py report DIR
kernel time per iteration (k_lanczos_fused) against the span per iteration."""
import glob
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(nx=100):
    import numpy as np
    import scipy.sparse as sp
    import bench
    from krypy_amd import _hip, linsys, utils

    ctx = _hip.get_context()
    A = bench.laplace2d(nx, nx)
    b = np.random.default_rng(0).standard_normal(A.shape[0])
    ls = linsys.LinearSystem(A, b, M=sp.diags(1.0 / A.diagonal()).tocsr(), self_adjoint=True)
    for _ in range(3):
        try:
            linsys.Minres(ls, maxiter=150, tol=1e-30)
        except utils.ConvergenceError:
            pass
    ctx.sync()


def report(src):
    import numpy as np
    db = sorted(glob.glob(src + "/**/*.db", recursive=True))[-1]
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    lz = [(i, s, e) for i, (nm, s, e) in enumerate(rows) if "k_lanczos_fused" in nm][-140:]
    d = np.array([(e - s) / 1e3 for _, s, e in lz])
    gaps = np.array([(lz[i + 1][1] - lz[i][2]) / 1e3 for i in range(len(lz) - 1)])
    other = sorted({rows[j][0].split("(")[0][:50] for i in range(len(lz) - 1) for j in range(lz[i][0] + 1, lz[i + 1][0])})
    span = (lz[-1][2] - lz[0][1]) / 1e3 / len(lz)
    print("last %d Lanczos launches: kernel %.2f us (median), gap to the next launch %.2f us (median; mean %.2f), span per "
          "iteration %.2f us = %.0f it/s; kernels between two of them: %s" % (len(lz), np.median(d), np.median(gaps), gaps.mean(), span, 1e6 / span, other))


if __name__ == "__main__":
    if sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        run(*[int(a) for a in sys.argv[2:3]])
