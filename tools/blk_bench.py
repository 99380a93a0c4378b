#!/usr/bin/env python
"""Reference-order GMRES(100) on short vectors with the blocked Gram-Schmidt kernel (chain_blk.h: one grid-wide sum
per four columns) switched on and off in one process: iterations/s, launches taken by the blocked kernel, the final
residual and the loss of orthogonality ||V^T V - I||_F of one cycle's basis, both ways.
    python tools/blk_bench.py [nx ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(sizes):
    import numpy as np
    import bench
    from krypy_amd import _hip, linsys, utils

    ctx = _hip.get_context()
    for nx in sizes:
        A = bench.laplace2d(nx, nx)
        N = A.shape[0]
        b = np.random.default_rng(0).standard_normal(N)
        ls = linsys.LinearSystem(A, b)
        def run(n):
            try:
                return linsys.RestartedGmres(ls, maxiter=100, max_restarts=n - 1, tol=1e-14, ortho="mgs")
            except utils.ConvergenceError as e:
                return e.solver

        rates = {0: [], 1: []}
        nblk = 0
        run(5)
        for rep in range(3):
            for blk in (1, 0):
                ctx.set("chain_blk", blk)
                run(2)
                ctx.sync()
                n0 = ctx.get("n_chain_blk")
                t0 = time.perf_counter()
                s = run(20)
                ctx.sync()
                dt = time.perf_counter() - t0
                rates[blk].append((len(s.resnorms) - 1) / dt)
                if blk:
                    nblk = ctx.get("n_chain_blk") - n0
        del s
        hist, orth, rel = {}, {}, {}
        for blk in (1, 0):
            ctx.set("chain_blk", blk)
            try:
                s1 = linsys.Gmres(ls, maxiter=100, tol=1e-14, ortho="mgs", store_arnoldi=True)
            except utils.ConvergenceError as e:
                s1 = e.solver
            Vb = s1.arnoldi._V
            G = ctx.gemm_tn(Vb, 0, 101, Vb, 0, 101)
            orth[blk] = float(np.linalg.norm(G - np.eye(101)))
            hist[blk] = np.asarray(s1.resnorms)
            rel[blk] = float(s1.resnorms[-1])
            del s1, Vb
        ctx.set("chain_blk", 1)
        d = np.max(np.abs(hist[1] - hist[0]) / hist[0])
        print("N = %8d: blocked %s it/s (%d of 2000 steps blocked, relres %.6e, orth %.2e) | per-column sums %s it/s "
              "(relres %.6e, orth %.2e) | max rel diff of one cycle's residual history %.2e"
              % (N, "/".join("%.0f" % r for r in rates[1]), nblk, rel[1], orth[1],
                 "/".join("%.0f" % r for r in rates[0]), rel[0], orth[0], d), flush=True)


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [100, 316, 500, 1000])
