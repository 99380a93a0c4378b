"""A / B of the dense GEMV's rows-per-wave shapes on one box: time per product at n = 32768 (config 4's operator: 8.6 GB
streamed per product) and smaller orders, outputs compared bit for bit with the one-row-per-wave kernel.

    python tools/gemv_ab.py [n ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krypy_amd import _hip  # noqa: E402


def main(argv):
    ctx = _hip.get_context()
    for n in [int(a) for a in argv] or [32768, 8192, 3001]:
        rng = np.random.default_rng(0)
        A = rng.standard_normal((n, n))
        x = rng.standard_normal(n)
        Ad = ctx.dense(A)
        X, Y = ctx.upload(x), ctx.alloc(n, 1)
        ref = None
        line = []
        for rows in (1, 2, 4, 0):
            ctx.set("gemv_rows", rows)
            ctx.apply(Ad, X, 0, Y, 0, 1)
            y = Y.download()[:, 0]
            if ref is None:
                ref = y
                assert np.allclose(y, A.dot(x), rtol=1e-12, atol=1e-9)
            same = bool(np.array_equal(y, ref))
            reps = max(20, int(2e10 / (8.0 * n * n)))
            best = 1e9
            for _ in range(3):
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(reps):
                    ctx.apply(Ad, X, 0, Y, 0, 1)
                ctx.sync()
                best = min(best, (time.perf_counter() - t0) / reps)
            line.append("rows %d: %.1f us, %.2f TB/s%s" % (rows, best * 1e6, 8.0 * n * n / best / 1e12, "" if same else " DIFFERENT BITS"))
        print("dense GEMV n = %d: " % n + " | ".join(line) + "   (rows 0 = the default choice)", flush=True)
        del Ad


if __name__ == "__main__":
    main(sys.argv[1:])
