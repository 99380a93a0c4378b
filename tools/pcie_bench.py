#!/usr/bin/env python
"""Host <-> device rate of the boundary's vector transfers (kh_vec_upload / kh_vec_download), the cost of every
callable operator / preconditioner / inner product: python tools/pcie_bench.py [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from krypy_amd import _hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = _hip.get_context()
x = np.random.default_rng(0).standard_normal((n, 1))
V = ctx.alloc(n, 2)
for name, fn in (("upload", lambda: V.upload(0, x)), ("download", lambda: V.download(0, 1))):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / 10
    print("%-9s %d doubles: %.2f ms = %.1f GB/s" % (name, n, dt * 1e3, 8.0 * n / dt / 1e9))
