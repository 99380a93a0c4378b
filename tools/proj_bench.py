#!/usr/bin/env python
"""The deflation projector (utils.Projection.apply_complement, two sweeps, d = 16) on long vectors: one launch with the
vector in registers (csrc/proj_reg.h) against the four-launch form, microseconds per application and TB/s on the
bytes each form moves (528 N / 560 N).
    python tools/proj_bench.py [N ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(sizes):
    import numpy as np
    from krypy_amd import _hip

    ctx = _hip.get_context()
    rng = np.random.default_rng(0)
    d = 16
    for n in sizes:
        Wd, Vd = ctx.alloc(n, d), ctx.alloc(n, d)
        col = rng.standard_normal(n) / np.sqrt(n)
        for j in range(d):
            Wd.upload(j, np.roll(col, j))
            Vd.upload(j, np.roll(col, -j))
        pj = ctx.proj_create(Wd, Vd, d, rng.standard_normal((d, d)) * 0.1, None, 2)
        A, Z = ctx.upload(rng.standard_normal(n)), ctx.alloc(n, 1)
        out = []
        for reg in (1, 0, 2):
            ctx.set("proj_reg", 1 if reg == 1 else 0)
            ctx.set("proj_panel", 0 if reg == 2 else 1)
            ctx.proj_apply_complement(pj, A, 0, Z, 0, want_ya=True)
            best = 1e30
            for _ in range(3):
                ctx.timer_start()
                for _ in range(20):
                    ctx.proj_apply_complement(pj, Z, 0, Z, 0)
                best = min(best, ctx.timer_stop() / 20)
            nbytes = (528.0 if reg == 1 else 560.0) * n
            label = {1: "one launch", 0: "four launches per sweep, panel kernels (the N-rank form)", 2: "four launches per sweep, chunked kernels"}[reg]
            out.append("%s %.1f us (%.2f TB/s on %d N bytes)" % (label, best * 1e3, nbytes / (best * 1e-3) / 1e12, 528 if reg == 1 else 560))
        ctx.set("proj_reg", 1)
        ctx.set("proj_panel", 1)
        print("N = %9d: %s" % (n, "; ".join(out)), flush=True)
        del Wd, Vd, A, Z, pj


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [3000000, 8000000, 10200000, 12500000])
