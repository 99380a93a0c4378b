#!/usr/bin/env python
"""Soak of the short-vector chain launches (one XCD, column ring, communication wave): GMRES(100) restart cycles at
N = 4096 ... 130,000 for `seconds` (default 150), every cycle's residual history compared with the first one of its
size (bit for bit: the launches are deterministic), the context's timeout-recovery counter must stay at 0.
    python tools/onex_soak.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(seconds=150.0):
    import numpy as np
    import bench
    from krypy_amd import _hip, linsys, utils

    ctx = _hip.get_context()
    cases = []
    for nx, ny in ((64, 64), (100, 100), (200, 200), (316, 316), (400, 325)):
        A = bench.laplace2d(nx, ny)
        b = np.random.default_rng(nx).standard_normal(A.shape[0])
        cases.append((A, b, linsys.LinearSystem(A, b)))
    first = {}
    t0 = time.time()
    cycles = 0
    while time.time() - t0 < seconds:
        for i, (A, b, ls) in enumerate(cases):
            try:
                s = linsys.RestartedGmres(ls, maxiter=100, max_restarts=3, tol=1e-14, ortho="mgs")
            except utils.ConvergenceError as e:
                s = e.solver
            r = np.array(s.resnorms)
            if i not in first:
                first[i] = r
            elif not np.array_equal(first[i], r):
                print("MISMATCH at N = %d after %d cycles: max rel. deviation %.2e" % (
                    A.shape[0], cycles, np.max(np.abs(r - first[i]) / first[i])))
                return 1
            cycles += 4
    c = ctx.counters()
    rec = ctx.get("n_chain_recovered")
    print("onex_soak: %d GMRES(100) cycles in %.0f s, %d chain launches (%d on one XCD, %d of the column-ring kernel), "
          "residual histories bit-identical from cycle to cycle, timeouts recovered: %d" % (
              cycles, time.time() - t0, c["chain"], ctx.get("n_chain_onex"), ctx.get("n_chain_small"), rec))
    return 0 if rec == 0 else 1


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 150.0))
