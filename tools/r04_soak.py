#!/usr/bin/env python
"""Soak of round 4's kernels: the same solve again and again - residual histories must be bit-identical from repeat to repeat
(the grid-wide sums add in an order that depends on the workgroup numbers alone), no timeout of a sum may be recovered.
  (a) DeflatedGmres(40), 16 deflation vectors, 3-D Laplacian 130^3: the one-launch projector (proj_reg.h) + the chain kernel
  (b) GMRES(100) on the 2-D Laplacian at N = 10^6 and 2.5 * 10^5: the blocked kernel (chain_blk.h), spread over the chip
  (c) GMRES(100) at 1.25 M rows through the multi-rank path of a 1-rank communicator: the one-reduction reference order
    python tools/r04_soak.py [seconds per part = 60]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from krypy_amd import _hip, deflation, linsys, utils  # noqa: E402


def repeat(name, ctx, solve, seconds, counters):
    first, n, launches0 = None, 0, {c: ctx.get(c) for c in counters}
    rec0 = ctx.get("n_chain_recovered")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        r = np.asarray(solve().resnorms)
        if first is None:
            first = r
        elif not np.array_equal(first, r):
            raise SystemExit("%s: repeat %d differs from the first solve (max rel %.2e)" % (name, n, np.max(np.abs(r - first) / first)))
        n += 1
    print("%s: %d solves in %.0f s, residual histories (%d entries) bit-identical, %s, timeouts recovered: %d" % (
        name, n, time.perf_counter() - t0, len(first), ", ".join("%s +%d" % (c, ctx.get(c) - launches0[c]) for c in counters),
        ctx.get("n_chain_recovered") - rec0), flush=True)
    assert ctx.get("n_chain_recovered") == rec0


def main(seconds):
    ctx = _hip.get_context()
    rng = np.random.default_rng(0)

    def caught(make):
        try:
            return make()
        except utils.ConvergenceError as e:
            return e.solver

    A = bench.laplace3d(130, 130, 130)
    b = rng.standard_normal(A.shape[0])
    U = np.linalg.qr(rng.standard_normal((A.shape[0], 16)))[0]
    ls = linsys.LinearSystem(A, b, self_adjoint=True)
    repeat("(a) DeflatedGmres(40), N = 2.2e6, d = 16", ctx, lambda: caught(lambda: deflation.DeflatedGmres(ls, U=U, tol=1e-14, maxiter=40)),
           seconds, ("n_proj_reg", "n_chain_recovered"))
    for nx in (1000, 500):
        A2 = bench.laplace2d(nx, nx)
        ls2 = linsys.LinearSystem(A2, rng.standard_normal(A2.shape[0]))
        repeat("(b) GMRES(100), N = %d" % A2.shape[0], ctx, lambda: caught(lambda: linsys.Gmres(ls2, tol=1e-14, maxiter=100)), seconds / 2,
               ("n_chain_blk",))
    os.environ["KRYPY_AMD_FORCE_MULTI"] = "1"
    try:
        c2 = _hip.Context(0)
        c2.comm_init(0, 1, c2.comm_unique_id())
    finally:
        del os.environ["KRYPY_AMD_FORCE_MULTI"]
    old = _hip._install_context_for_testing(c2)
    try:
        A3 = bench.laplace2d(4000, 313)
        ls3 = linsys.LinearSystem(A3, rng.standard_normal(A3.shape[0]))
        repeat("(c) GMRES(100) mgs, 1.25 M rows, multi-rank path", c2, lambda: caught(lambda: linsys.Gmres(ls3, tol=1e-14, maxiter=100)), seconds,
               ("n_lowsync", "n_allreduce"))
    finally:
        _hip._install_context_for_testing(old)
        c2.close()
    print("r04_soak ok")


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
