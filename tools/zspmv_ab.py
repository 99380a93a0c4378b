#!/usr/bin/env python
"""Complex banded SpMV (zpath.h: k_zspmv_dia) against the complex CSR-stream kernel on the same operator, N = 5e6 complex rows."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import bench  # noqa: E402
from krypy_amd import _hip  # noqa: E402

ctx = _hip.get_context()
A = bench.laplace2d(2500, 2000).astype(complex)
N = A.shape[0]
A = (A + sp.diags(1j * np.linspace(0.1, 1.0, N))).tocsr()
rng = np.random.default_rng(0)
x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
Ad = ctx.csr(A, dtype=complex)
X, Y = ctx.upload(x.reshape(-1, 1)), ctx.alloc(N, 1, dtype=complex)
want = A.dot(x)
for dia in (1, 0, 1, 0):
    ctx.set("spmv_dia", dia)
    for _ in range(3):
        ctx.apply(Ad, X, 0, Y, 0, 1)
    z0 = ctx.get("n_zspmv_dia")
    ctx.timer_start()
    for _ in range(30):
        ctx.apply(Ad, X, 0, Y, 0, 1)
    ms = ctx.timer_stop() / 30
    ok = np.array_equal(Y.download()[:, 0], want)
    nb = 20.0 * A.nnz + 4.0 * (N + 1) + 32.0 * N
    nd = 16.0 * 5 * N + 32.0 * N
    print("spmv_dia=%d: %.1f us, banded launches %d, bit-identical to SciPy %s, %.0f GB/s on the CSR bytes, %.0f on the banded copy's" % (
        dia, ms * 1e3, ctx.get("n_zspmv_dia") - z0, ok, nb / ms / 1e6, nd / ms / 1e6), flush=True)
ctx.set("spmv_dia", 1)
