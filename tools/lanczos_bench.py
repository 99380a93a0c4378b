#!/usr/bin/env python
"""Where does a MINRES + Jacobi iteration's one launch (k_lanczos_fused, csrc/lanczos.h) spend its time?

    python tools/lanczos_bench.py [nx ny]             HIP-event averages per variant (30 launches each, one host
                                                      synchronisation per launch included)
    rocprofv3 --kernel-trace -d DIR -o lz -- python tools/lanczos_bench.py
    python tools/lanczos_bench.py report DIR          kernel-trace durations per variant (exact)

Variants switch phases of the kernel off (kh_ctx_set "chain_debug": 8 pass 1, 16 pass 2, 32 pass 3, 64 the MINRES job;
results are garbage then - measurement only) and compare with the general chain kernel + k_minres_update pair."""
import glob
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

VARIANTS = [("all phases + MINRES job", 0), ("without the MINRES job", 64), ("pass 1 + 2 only", 64 | 32),
            ("pass 1 only", 64 | 32 | 16), ("pass 2 only", 8 | 32 | 64), ("pass 3 only", 8 | 16 | 64),
            ("MINRES job only", 8 | 16 | 32), ("sums only", 8 | 16 | 32 | 64)]
REPS = 30


def run(nx=4000, ny=2500):
    import numpy as np
    import bench
    from krypy_amd import _hip

    ctx = _hip.get_context()
    A = bench.laplace2d(nx, ny)
    n = A.shape[0]
    rng = np.random.default_rng(0)
    dj = 1.0 / A.diagonal()
    Ad, Md = ctx.csr(A), ctx.diag(dj)
    V, P, W = ctx.alloc(n, 8), ctx.alloc(n, 8), ctx.alloc(n, 2)
    Wm, yk = ctx.alloc(n, 2), ctx.alloc(n, 1)
    for blk in (V, P):
        for c in range(5):
            blk.upload(c, rng.standard_normal((n, 1)) / np.sqrt(n))
    print("N = %d" % n)
    for fused in (1, 0):
        ctx.set("lanczos_fused", fused)
        for name, bits in (VARIANTS if fused else [("general chain kernel + k_minres_update", 0)]):
            ctx.set("chain_debug", bits)
            for rep in range(3 + REPS):
                if rep == 3:
                    ctx.sync()
                    ctx.timer_start()
                ctx.minres_update(V, 3, Wm, rep & 1, 0.1, 0.2, 1.3, 0.4, yk, 0, defer=True)
                ctx.arnoldi_step(Ad, Md, V, P, W, 0, 4, 4, 1, 0, 0.37)
            ms = ctx.timer_stop()
            ctx.minres_flush()
            print("%-46s %7.1f us per iteration (HIP events, host round trip included)" % (name, ms * 1e3 / REPS), flush=True)
    ctx.set("chain_debug", 0)
    ctx.set("lanczos_fused", 1)


def report(src):
    db = sorted(glob.glob(src + "/**/*.db", recursive=True))[-1]
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    lz = [(e - s) / 1e3 for nm, s, e in rows if "k_lanczos_fused" in nm]
    import statistics as st
    per = 3 + REPS
    print("| variant | kernel time, median of %d launches (us) |\n|---|---:|" % REPS)
    for i, (name, bits) in enumerate(VARIANTS):
        chunk = lz[i * per + 3:(i + 1) * per]
        if chunk:
            print("| %s | %.1f |" % (name, st.median(chunk)))
    old_chain = [(e - s) / 1e3 for nm, s, e in rows if "k_mgs_chain" in nm]
    old_upd = [(e - s) / 1e3 for nm, s, e in rows if "k_minres_update" in nm]
    if old_chain and old_upd:
        print("| general chain kernel (six phases) | %.1f |" % st.median(old_chain[3:]))
        print("| k_minres_update (launch of its own) | %.1f |" % st.median(old_upd[3:]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        run(*[int(a) for a in sys.argv[1:3]])
