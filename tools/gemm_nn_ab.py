#!/usr/bin/env python
"""A / B of kh_gemm_nn with 16 output columns (the Ritz vectors [V_n, U] @ coeffs, deflation.py:840-847) at config 5's slab
length: k_panel_gemm_mfma (the block read once) against one k_multiaxpy pass over the block per output column.
python tools/gemm_nn_ab.py [n] [k]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from krypy_amd import _hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 116
ctx = _hip.get_context()
rng = np.random.default_rng(0)
X = ctx.alloc(n, k)
for j in range(k):
    X.upload(j, rng.standard_normal(n))
C = rng.standard_normal((k, 16))
Y = ctx.alloc(n, 16)
for sw in (1, 0, 1, 0):
    ctx.set("gram_mfma", sw)
    ctx.gemm_nn(X, 0, k, C, 1.0, 0.0, Y, 0)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.gemm_nn(X, 0, k, C, 1.0, 0.0, Y, 0)
    ctx.sync()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print("gram_mfma=%d: %.2f ms per product (N = %d, %d x 16): %.2f TB/s on the block read once + the output" %
          (sw, ms, n, k, 8.0 * n * (k + 16) / ms / 1e9))
ctx.set("gram_mfma", 1)
