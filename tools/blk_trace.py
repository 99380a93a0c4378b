#!/usr/bin/env python
"""A few GMRES(100) cycles in reference order on the 2-D Laplacian at N = nx^2 (default 10^6) - short enough for
    rocprofv3 --kernel-trace --stats -- python tools/blk_trace.py [nx] [cycles]
to give the average launch of the blocked Gram-Schmidt kernel (chain_blk.h; steps k = 7 .. 99) next to the per-column
kernel of the steps in front of it."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(nx, cycles):
    import numpy as np
    import bench
    from krypy_amd import _hip, linsys, utils

    ctx = _hip.get_context()
    A = bench.laplace2d(nx, nx)
    b = np.random.default_rng(0).standard_normal(A.shape[0])
    ls = linsys.LinearSystem(A, b)
    n0 = ctx.get("n_chain_blk")
    try:
        linsys.RestartedGmres(ls, maxiter=100, max_restarts=cycles - 1, tol=1e-14, ortho="mgs")
    except utils.ConvergenceError:
        pass
    ctx.sync()
    print("N = %d: %d cycles, %d launches of the blocked kernel, %d of them with workgroups without rows" % (
        A.shape[0], cycles, ctx.get("n_chain_blk") - n0, ctx.get("n_blk_rowless")))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 5)
