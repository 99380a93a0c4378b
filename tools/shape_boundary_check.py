#!/usr/bin/env python
"""Robustness sweep (GPU): Arnoldi steps at vector lengths right at the edges of the register shapes of the chain /
panel kernels (n = 262144 c -2 .. +2 for c = 4 ... 56 rows per lane: the last length a shape takes, the first one of
the next), at workgroup-chunk edges inside a shape, at the padding threshold and at tiny sizes.  For every length the
register-resident kernels (a context with the chain on) must agree with the per-column / chunked kernels (a context
with KRYPY_AMD_MGS_CHAIN=0) and satisfy the Arnoldi relation.  python tools/shape_boundary_check.py [quick]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402


def contexts():
    from krypy_amd import _hip
    out = []
    for chain in ("1", "0"):
        old = os.environ.get("KRYPY_AMD_MGS_CHAIN")
        os.environ["KRYPY_AMD_MGS_CHAIN"] = chain
        try:
            out.append(_hip.Context(0))
        finally:
            if old is None:
                del os.environ["KRYPY_AMD_MGS_CHAIN"]
            else:
                os.environ["KRYPY_AMD_MGS_CHAIN"] = old
    return out


def matrix(n, cplx):
    """five diagonals (0, +-1, +-K): the banded SpMV and, from 16 rows per lane on, the operator-in-prologue path"""
    K = min(1000, max(2, n // 3))
    rng = np.random.default_rng(n % 1000)
    d = [rng.uniform(2.0, 3.0, n)] + [rng.uniform(-1.0, -0.5, n - o) for o in (1, 1, K, K)]
    A = sp.diags(d, [0, 1, -1, K, -K], shape=(n, n), format="csr") if n > K else sp.diags([d[0]], [0], format="csr")
    if cplx:
        A = (A + sp.diags(1j * np.linspace(0.1, 0.5, n))).tocsr()
    return A


def steps(ctx, A, b, m, gs, dt):
    n = A.shape[0]
    Ad = ctx.csr(A)
    V, W = ctx.alloc(n, m + 1, dtype=dt), ctx.alloc(n, 2, dtype=dt)
    V.upload(0, b / np.linalg.norm(b))
    H = np.zeros((m + 1, m), dtype=dt)
    for k in range(m):
        H[: k + 2, k] = ctx.arnoldi_step(Ad, None, V, None, W, 0, k, 0, 2 if k == 2 else 1, gs)
    return H, V.download()


def check(n, cplx, ctxs, m=4):
    m = max(1, min(m, n - 1))          # (a Krylov space cannot outgrow the vector)
    dt = complex if cplx else float
    A = matrix(n, cplx)
    rng = np.random.default_rng(n % 977)
    b = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0.0)
    worst = 0.0
    for gs in (0, 1):          # reference-order MGS (chain kernel), panel form (k_cgs_* in registers)
        (Hc, Vc), (Hl, Vl) = [steps(c, A, b, m, gs, dt) for c in ctxs]
        e1 = np.linalg.norm(Hc - Hl) / np.linalg.norm(Hl)
        e2 = np.linalg.norm(Vc - Vl)
        e3 = np.linalg.norm(A.dot(Vc[:, :m]) - Vc.dot(Hc)) / np.linalg.norm(Hc)
        worst = max(worst, e1, e2, e3)
        assert e1 < 1e-12 and e2 < 1e-10 and e3 < 1e-12, (n, cplx, gs, e1, e2, e3)
    return worst


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    sizes = [2, 3, 5, 63, 64, 65, 1023, 1024, 1025, 2047, 2049, 4097, 65535, 65536, 65537, 100003]
    edge = 262144
    for c in (4, 8, 16, 24, 32, 40, 48, 56):
        for dlt in ((-1, 0, 1) if quick else (-2, -1, 0, 1, 2)):
            sizes.append(edge * c + dlt)
    for g, c in ((7, 4), (100, 8), (131, 16), (200, 24)):        # chunk edges inside a shape: n2 = g c 512 -+ 1
        sizes += [2 * (g * c * 512) - 1, 2 * (g * c * 512) + 2]
    ctxs = contexts()
    t0 = time.time()
    for n in sizes:
        cases = [False] + ([True] if n <= edge * 20 and (not quick or n < 5_000_000) else [])
        for cplx in cases:
            nn = n if not cplx else max(2, n // 2)       # complex: same register shape at half the entries
            t1 = time.time()
            w = check(nn, cplx, ctxs)
            print("n = %9d %-7s ok (worst deviation %.1e, %.1f s)" % (nn, "complex" if cplx else "real", w, time.time() - t1),
                  flush=True)
    print("all %d lengths agree, %.0f s" % (len(sizes), time.time() - t0))
