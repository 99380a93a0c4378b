#!/usr/bin/env python
"""Per-link cost of the chain kernel on short vectors, from a kernel trace.

    rocprofv3 --kernel-trace -d DIR -o st -- python tools/small_trace.py run [nx]
    python tools/small_trace.py report DIR

GMRES(100), reference-order MGS: the k-th step of a cycle orthogonalises against k + 1 columns, so the duration of the
chain launches of one cycle is a line in k - its slope is the cost of a link (dot, grid-wide sum, update), its intercept
the prologue (operator) + norm + store; the gaps between launches are what the host / the launch path add."""
import glob
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(nx=316):
    import numpy as np
    import bench
    from krypy_amd import _hip, linsys, utils

    ctx = _hip.get_context()
    A = bench.laplace2d(nx, nx)
    b = np.random.default_rng(0).standard_normal(A.shape[0])
    try:
        linsys.RestartedGmres(linsys.LinearSystem(A, b), maxiter=100, max_restarts=3, tol=1e-14, ortho="mgs")
    except utils.ConvergenceError:
        pass
    ctx.sync()


def report(src):
    import numpy as np
    db = sorted(glob.glob(src + "/**/*.db", recursive=True))[-1]
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    ch = [(i, s, e) for i, (nm, s, e) in enumerate(rows) if "k_mgs_chain" in nm]
    names = sorted({rows[i][0].split("(")[0][:70] for i, _, _ in ch})
    print("chain kernels:", names)
    # the last complete cycle: 100 launches
    last = ch[-100:]
    d = np.array([(e - s) / 1e3 for _, s, e in last])
    gaps = np.array([(last[i + 1][1] - last[i][2]) / 1e3 for i in range(len(last) - 1)])
    between = [last[i + 1][0] - last[i][0] - 1 for i in range(len(last) - 1)]
    k = np.arange(len(d))
    sl, ic = np.polyfit(k[5:], d[5:], 1)
    print("launches %d, duration = %.2f us + %.3f us per link (fit over the last cycle); first %.1f, last %.1f us" % (len(d), ic, sl, d[0], d[-1]))
    print("gap between chain launches: median %.2f us, mean %.2f us, max %.1f us; other kernels between them: %s" % (np.median(gaps), gaps.mean(), gaps.max(), sorted(set(between))))
    span = (last[-1][2] - last[0][1]) / 1e3
    print("cycle span %.1f us = %.1f us per iteration (%.0f it/s inside the cycle); kernel time %.1f us, gaps %.1f us" % (span, span / len(d), 1e6 * len(d) / span, d.sum(), gaps.sum()))


if __name__ == "__main__":
    if sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        run(*[int(a) for a in sys.argv[2:3]])
