#!/usr/bin/env python
"""SpMV timing on the bench matrix (and a 3-D 7-point one): python tools/spmv_bench.py
(KRYPY_AMD_SPMV_DIA=0 selects the CSR kernel, 1/2/4 the row pairs per lane of the banded one)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from krypy_amd import _hip  # noqa: E402
from oracle.krylov_ref import laplace3d  # noqa: E402

ctx = _hip.get_context()
import scipy.sparse as sp  # noqa: E402


def banded(n, offsets, fill=1.0, seed=0):
    rng = np.random.default_rng(seed)
    diags = []
    for o in offsets:
        d = rng.standard_normal(n - abs(o)) + 3.0
        if fill < 1.0:
            d[rng.random(d.size) > fill] = 0.0
        diags.append(d)
    A = sp.diags(diags, offsets, shape=(n, n)).tocsr()
    A.eliminate_zeros()
    A.sort_indices()
    return A


mats = [("lap2d 4000x2500", bench.laplace2d(4000, 2500)), ("lap3d 200^3", laplace3d(200).tocsr())]
if "--wide" in sys.argv:     # the generic (not unrolled) instantiation and partially filled diagonals
    n = 4_000_000
    o27 = [a * 160 * 160 + b * 160 + c for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)]
    mats = [("11 diagonals", banded(n, (-2000, -40, -3, -2, -1, 0, 1, 2, 3, 40, 2000))),
            ("27-point 160^3 pattern", banded(160 ** 3, sorted(o27))),
            ("9 diagonals 75% full", banded(n, (-2001, -2000, -1999, -1, 0, 1, 1999, 2000, 2001), 0.75)),
            ("27 diagonals 72% full", banded(160 ** 3, sorted(o27), 0.72))]
for name, A in mats:
    n = A.shape[0]
    x = np.random.default_rng(0).standard_normal(n)
    want = A.dot(x)
    nb = 12.0 * A.nnz + 4.0 * (n + 1) + 16.0 * n
    Ad = ctx.csr(A)
    X, Y = ctx.upload(x), ctx.alloc(n, 1)
    for _ in range(3):
        ctx.apply(Ad, X, 0, Y, 0, 1)
    ctx.timer_start()
    for _ in range(50):
        ctx.apply(Ad, X, 0, Y, 0, 1)
    ms = ctx.timer_stop() / 50
    ok = np.array_equal(Y.download()[:, 0], want)
    nd = Ad.diagonals
    moved = (8.0 * nd * n + 16.0 * n) if nd else nb
    print("dia=%s %s: diagonals=%d  %.1f us  %.0f GB/s (CSR bytes)  %.0f GB/s (bytes of the format used)  bit-identical=%s" % (
        os.environ.get("KRYPY_AMD_SPMV_DIA", "default"), name, nd, ms * 1e3, nb / ms / 1e6, moved / ms / 1e6, ok))
