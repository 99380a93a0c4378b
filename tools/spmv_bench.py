import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from krypy_amd import _hip
ctx = _hip.get_context()
A = bench.laplace2d(4000, 2500)
n = A.shape[0]
x = np.random.default_rng(0).standard_normal(n)
want = A.dot(x)
nb = 12.0 * A.nnz + 4.0 * (n + 1) + 16.0 * n
for tile in (1024, 2048, 4096):
    ctx.tune(0, tile)
    Ad = ctx.csr(A)
    X, Y = ctx.upload(x), ctx.alloc(n, 1)
    for _ in range(3): ctx.apply(Ad, X, 0, Y, 0, 1)
    ctx.timer_start()
    for _ in range(50): ctx.apply(Ad, X, 0, Y, 0, 1)
    ms = ctx.timer_stop() / 50
    ok = np.array_equal(Y.download()[:, 0], want)
    print("tile %d: %.1f us  %.0f GB/s  bit-identical=%s" % (tile, ms * 1e3, nb / ms / 1e6, ok))
