#!/usr/bin/env python
"""Dense GEMV (config 4's matvec) timing: python tools/gemv_bench.py [n]   (KRYPY_AMD_GEMV_ROWS=1|2|4|8)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krypy_amd import _hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ctx = _hip.get_context()
rng = np.random.default_rng(0)
A = rng.standard_normal((n, n))
x = rng.standard_normal((n, 1))
Ad, X, Y = ctx.dense(A), ctx.upload(x), ctx.alloc(n, 1)
ctx.apply(Ad, X, 0, Y, 0, 1)
err = np.abs(Y.download() - A.dot(x)).max()
reps = 50
for _ in range(5):
    ctx.apply(Ad, X, 0, Y, 0, 1)
ctx.timer_start()
for _ in range(reps):
    ctx.apply(Ad, X, 0, Y, 0, 1)
ms = ctx.timer_stop() / reps
print("rows=%s n=%d  %.4f ms  %.2f TB/s  frac=%.3f  maxerr=%.2e" % (
    os.environ.get("KRYPY_AMD_GEMV_ROWS", "default"), n, ms, 8.0 * n * n / ms / 1e9, 8.0 * n * n / ms / 1e9 / 8.0, err))
