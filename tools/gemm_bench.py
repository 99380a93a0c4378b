#!/usr/bin/env python
"""Dense A times a 16-column panel (k_gemm_dense_mfma) against 16 single-column GEMVs:
python tools/gemm_bench.py [n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krypy_amd import _hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ctx = _hip.get_context()
rng = np.random.default_rng(0)
A = rng.standard_normal((n, n))
X = rng.standard_normal((n, 16)) * np.arange(1, 17)       # asymmetric: column j scaled by j+1
Ad, Xd, Yd = ctx.dense(A), ctx.upload(X), ctx.alloc(n, 16)
ctx.apply(Ad, Xd, 0, Yd, 0, 16)
want = A.dot(X)
err = np.abs(Yd.download() - want).max() / np.abs(want).max()
for nc in (16, 5, 1):
    for _ in range(3):
        ctx.apply(Ad, Xd, 0, Yd, 0, nc)
    ctx.timer_start()
    for _ in range(10):
        ctx.apply(Ad, Xd, 0, Yd, 0, nc)
    ms = ctx.timer_stop() / 10
    print("n=%d ncols=%2d: %.3f ms  A-stream %.2f TB/s  %.1f TFLOP/s  relerr(16)=%.1e" % (
        n, nc, ms, 8.0 * n * n / ms / 1e9, 2.0 * n * n * nc / ms / 1e9, err))
