#!/usr/bin/env python
"""Randomised differential test of the device library's vector / operator entry points against NumPy / SciPy:
random lengths (odd ones, lengths around the padding threshold and the tile sizes), random column windows of wider
blocks, real and complex data, every operator kind (CSR with empty rows / one long row, banded, dense, diagonal),
single vectors and blocks (the SpMM kernels).  Seeds are fixed: a failure names (seed, call).
    python tools/abi_fuzz.py [rounds=60] [max_n=300000]
Runs on whatever context is installed (the GPU by default; tests/support/numpy_context for a dry run)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402


def rnd(rng, cplx, *shape):
    x = rng.standard_normal(shape)
    return x + 1j * rng.standard_normal(shape) if cplx else x


def pick_n(rng, max_n):
    kind = rng.integers(0, 5)
    if kind == 0:
        return int(rng.integers(1, 70))
    if kind == 1:
        return int(rng.choice([255, 256, 257, 1023, 1025, 2047, 2048, 2049, 4096, 4097, 65535, 65536, 65537]))
    if kind == 2:
        return int(rng.integers(1000, 20000)) | 1          # odd
    return int(rng.integers(70, max_n))


def random_operator(rng, n, cplx):
    kind = ["csr", "banded", "dense", "diag", "csr_holes"][rng.integers(0, 5)]
    if kind == "dense":
        n = min(n, 700)
        return kind, rnd(rng, cplx, n, n), n
    if kind == "diag":
        d = rnd(rng, cplx, n)
        return kind, sp.diags(d).tocsr(), n
    if kind == "banded":
        offs = sorted({0} | {int(o) for o in rng.integers(-min(n - 1, 900), min(n - 1, 900) + 1, size=rng.integers(1, 7))})
        A = sp.diags([rnd(rng, cplx, n - abs(o)) for o in offs], offs, shape=(n, n), format="csr")
        return kind, A, n
    nnz = int(rng.uniform(2, 9) * n)
    # (NOT scipy.sparse.random: it samples from n^2 integers - 671 GiB at n = 3e5; no lil / dense detours either)
    rows, cols = rng.integers(0, n, nnz), rng.integers(0, n, nnz)
    vals = rnd(rng, cplx, nnz)
    if kind == "csr_holes" and n > 8:
        holes = rng.integers(0, n, size=max(1, n // 50))
        keep = ~np.isin(rows, holes)                               # empty rows
        rows, cols, vals = rows[keep], cols[keep], vals[keep]
        m = min(n, 3000)                                           # one long row (beyond the LDS tile)
        lr = int(rng.integers(0, n))
        rows = np.r_[rows, np.full(m, lr)]
        cols = np.r_[cols, rng.choice(n, size=m, replace=False)]
        vals = np.r_[vals, rnd(rng, cplx, m)]
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    A.sum_duplicates()
    A = sp.csr_matrix(A, dtype=complex if cplx else float)
    return kind, A.tocsr(), n


def one_round(ctx, seed, max_n):
    rng = np.random.default_rng(seed)
    cplx = bool(rng.integers(0, 2))
    n = pick_n(rng, max_n)
    ncol = int(rng.integers(2, 21))
    dt = complex if cplx else float
    X, W = rnd(rng, cplx, n, ncol), rnd(rng, cplx, n, 3)
    Xd, Wd = ctx.upload(X), ctx.upload(W)
    tol = dict(rtol=1e-12, atol=1e-12 * np.sqrt(n) * 8)
    checks = 0

    def expect(name, got, want, **kw):
        nonlocal checks
        checks += 1
        kw = kw or tol
        if not np.allclose(got, want, **kw):
            raise AssertionError("seed %d (n=%d, %s): %s deviates by %.3e" % (
                seed, n, "complex" if cplx else "real", name, float(np.max(np.abs(np.asarray(got) - np.asarray(want))))))

    j0 = int(rng.integers(0, ncol - 1))
    k = int(rng.integers(1, ncol - j0 + 1))
    wc = int(rng.integers(0, 3))
    expect("download window", Xd.download(j0, k), X[:, j0:j0 + k], rtol=0, atol=0)
    expect("dot_panel", ctx.dot_panel(Xd, j0, k, Wd, wc), X[:, j0:j0 + k].conj().T.dot(W[:, wc]))
    expect("nrm2", ctx.nrm2(Wd, wc), np.linalg.norm(W[:, wc]), rtol=1e-13, atol=0)
    nx2 = int(rng.integers(1, min(ncol, 6) + 1))
    expect("gemm_tn", ctx.gemm_tn(Xd, 0, nx2, Xd, j0, k), X[:, :nx2].conj().T.dot(X[:, j0:j0 + k]))
    h = rnd(rng, cplx, k)
    w = W[:, wc].copy()
    for j in range(k):
        w = w - h[j] * X[:, j0 + j]
    ctx.axpy_panel(Xd, j0, k, h, Wd, wc)
    W[:, wc] = w
    expect("axpy_panel", Wd.download()[:, wc], w, rtol=1e-13, atol=1e-13 * k)
    C = rnd(rng, cplx, k, 2)
    beta = float(rng.choice([0.0, 1.0, -0.5]))
    Y = rnd(rng, cplx, n, 3)
    Yd = ctx.upload(Y)
    ctx.gemm_nn(Xd, j0, k, C, 1.0, beta, Yd, 1)
    Y[:, 1:3] = beta * Y[:, 1:3] + X[:, j0:j0 + k].dot(C)
    expect("gemm_nn", Yd.download(), Y, rtol=1e-12, atol=1e-12 * k)
    a, b = (rnd(rng, cplx, 2) if cplx and rng.integers(0, 2) else rng.standard_normal(2))
    ctx.waxpby(Yd, 0, a, Xd, j0, b, Wd, (wc + 1) % 3)
    Y[:, 0] = a * X[:, j0] + b * W[:, (wc + 1) % 3]
    expect("waxpby", Yd.download()[:, 0], Y[:, 0], rtol=1e-13, atol=1e-13)
    s = float(rng.uniform(0.5, 2.0))
    ctx.vdiv(Yd, 2, Xd, j0, s)
    expect("vdiv", Yd.download()[:, 2], X[:, j0] / s, rtol=1e-15, atol=0)
    Yd.copy_from(0, Xd, j0, min(k, 3))
    expect("copy_from", Yd.download()[:, :min(k, 3)], X[:, j0:j0 + min(k, 3)], rtol=0, atol=0)
    # operators: one vector, and a block through the same entry (SpMM kernels for CSR / banded)
    kind, A, na = random_operator(rng, n, cplx)
    if kind == "dense":
        Ad = ctx.dense(A)
        Adot = A.dot
    elif kind == "diag":
        Ad = ctx.diag(np.asarray(A.diagonal()))
        Adot = A.dot
    else:
        Ad = ctx.csr(A)
        Adot = A.dot
    d = int(rng.choice([1, 2, 3, 5, 16, 17]))
    Z = rnd(rng, cplx, na, d + 1)
    Zd, Od = ctx.upload(Z), ctx.alloc(na, d + 2, dtype=dt)
    x0c = int(rng.integers(0, 2))
    ctx.apply(Ad, Zd, x0c, Od, 1, d)
    want = Adot(Z[:, x0c:x0c + d])
    got = Od.download()
    # (a diagonal operator is a Hadamard product on the device: real bits only)
    exact = (kind == "diag" and not cplx) or (kind in ("csr", "banded", "csr_holes") and A.getnnz(axis=1).max() <= 1024)
    if exact:       # row sums left to right like SciPy: bit-identical
        expect("apply %s x%d (bits)" % (kind, d), got[:, 1:1 + d], want, rtol=0, atol=0)
    else:
        expect("apply %s x%d" % (kind, d), got[:, 1:1 + d], want, rtol=1e-12, atol=1e-12 * np.sqrt(na) * 30)
    expect("apply leaves the other columns alone", got[:, [0, d + 1]], 0.0, rtol=0, atol=0)
    return checks


def step_round(ctx, dbl, seed, max_n):
    """The fused entry points (Arnoldi / Lanczos step with every option, residual, MINRES / CG updates, CG step,
    projector) on random data: the device library against the NumPy restatement of each entry's semantics
    (tests/support/numpy_context.py), blocks and returned numbers."""
    rng = np.random.default_rng(10_000 + seed)
    cplx = bool(rng.integers(0, 2))
    dt = complex if cplx else float
    n = min(pick_n(rng, max_n), 120_000)
    n = max(n, 12)
    m = int(rng.integers(2, 9))
    k = int(rng.integers(0, m))
    checks = 0

    def expect(name, got, want, rtol=1e-10):
        nonlocal checks
        checks += 1
        got, want = np.asarray(got), np.asarray(want)
        scale = max(1.0, float(np.max(np.abs(want)))) if want.size else 1.0
        if got.shape != want.shape or not np.allclose(got, want, rtol=rtol, atol=rtol * scale):
            raise AssertionError("seed %d (n=%d, %s, k=%d of %d): %s deviates by %.3e" % (
                seed, n, "complex" if cplx else "real", k, m, name,
                float(np.max(np.abs(got - want))) if got.shape == want.shape else float("nan")))

    offs = sorted({0, 1, -1} | {int(o) for o in rng.integers(-min(n - 1, 500), min(n - 1, 500) + 1, size=2)})
    A = sp.diags([rnd(rng, cplx, n - abs(o)) + (4.0 if o == 0 else 0.0) for o in offs], offs, shape=(n, n), format="csr")
    d = rng.uniform(0.5, 2.0, n)
    with_m = bool(rng.integers(0, 2))
    Q = np.linalg.qr(rnd(rng, cplx, n, m + 1))[0]
    P0 = Q / np.sqrt(d)[:, None] if with_m else None          # V = D P, P^H D P = I
    V0 = Q * np.sqrt(d)[:, None] if with_m else Q
    mode = ["mgs", "dmgs", "cgs", "cgs2", "lanczos"][rng.integers(0, 5)]
    gs = 1 if mode.startswith("cgs") else 0
    sweeps = 2 if mode in ("dmgs", "cgs2") else 1
    start = k if mode == "lanczos" else 0
    hk = (complex(*rng.standard_normal(2)) if cplx else float(rng.standard_normal())) if (mode == "lanczos" and k > 0) else 0.0
    outs = []
    for c in (ctx, dbl):
        Vd, Wd = c.alloc(n, m + 2, dtype=dt), c.alloc(n, 2, dtype=dt)
        Vd.upload(0, V0[:, : k + 1])
        Pd = None
        Md = None
        if with_m:
            Pd = c.alloc(n, m + 2, dtype=dt)
            Pd.upload(0, P0[:, : k + 1])
            Md = c.diag(d, dtype=dt)
        Ad = c.csr(A)
        hcol = c.arnoldi_step(Ad, Md, Vd, Pd, Wd, 0, k, start, sweeps, gs, hk)
        outs.append((np.array(hcol), Vd.download(0, k + 2), Pd.download(0, k + 2) if with_m else None))
    expect("arnoldi_step %s%s H column" % (mode, " + Jacobi" if with_m else ""), outs[0][0][start:], outs[1][0][start:])
    expect("arnoldi_step %s v_{k+1}" % mode, outs[0][1][:, k + 1], outs[1][1][:, k + 1])
    if with_m:
        expect("arnoldi_step %s p_{k+1}" % mode, outs[0][2][:, k + 1], outs[1][2][:, k + 1])
    # residual, MINRES / CG recurrences
    x, b_ = rnd(rng, cplx, n, 1), rnd(rng, cplx, n, 1)
    r0, r1, r2, y0 = ((complex(*rng.standard_normal(2)) if cplx else float(rng.standard_normal())) for _ in range(4))
    Vm, Wm, ym = rnd(rng, cplx, n, 3), rnd(rng, cplx, n, 2), rnd(rng, cplx, n, 1)
    pv, rv, zv, yv = (rnd(rng, cplx, n, 1) for _ in range(4))
    slot = int(rng.integers(0, 2))
    first = bool(rng.integers(0, 2))
    B = A + A.conj().T + sp.identity(n) * 12.0             # Hermitian, diagonally dominant: a CG operator
    res = []
    for c in (ctx, dbl):
        Ad = c.csr(A)
        Rd = c.alloc(n, 1, dtype=dt)
        nr = c.residual(Ad, c.upload(b_), 0, c.upload(x), 0, Rd, 0)
        Vd, Wd, Yd = c.upload(Vm), c.upload(Wm), c.upload(ym)
        c.minres_update(Vd, 1, Wd, slot, r0, r1, r2, y0, Yd, 0)
        Bd = c.csr(B.tocsr())
        D2 = c.diag(np.repeat(d, 2) if cplx else d) if with_m else None
        p_, r_, z_, y_, ap_ = c.upload(pv), c.upload(rv), c.upload(zv), c.upload(yv), c.alloc(n, 1, dtype=dt)
        st = c.cg_step(Bd, D2, p_, 0, ap_, 0, y_, 0, r_, 0, z_ if with_m else None, 0, first, 0.37, 1.9)
        p2, r2_, z2, y2, ap2 = c.upload(pv), c.upload(rv), c.upload(zv), c.upload(yv), c.upload(rnd(np.random.default_rng(seed), cplx, n, 1))
        rho = c.cg_update(0.7, p2, 0, ap2, 0, y2, 0, r2_, 0, D2, z2 if with_m else None, 0)
        res.append((nr, Rd.download(), Wd.download(), Yd.download(), st, p_.download(), r_.download(), y_.download(),
                    ap_.download(), z_.download() if with_m else 0.0, rho, r2_.download(), y2.download()))
    names = ("residual norm", "residual", "minres_update W", "minres_update yk", "cg_step scalars", "cg_step p", "cg_step r",
             "cg_step yk", "cg_step Ap", "cg_step z", "cg_update rho", "cg_update r", "cg_update yk")
    for nm, g_, w_ in zip(names, res[0], res[1]):
        if nm == "cg_step scalars":
            assert g_[3] == w_[3], ("cg_step sanity word", g_[3], w_[3])
            g_ = [g_[0], g_[1], complex(g_[2]).real, complex(g_[2]).imag / max(1.0, abs(complex(g_[2]).real))]
            w_ = [w_[0], w_[1], complex(w_[2]).real, complex(w_[2]).imag / max(1.0, abs(complex(w_[2]).real))]
        expect(nm, g_, w_)
    # projector (deflation): complement of a random oblique projection, <Y, a> on request
    dd = int(rng.integers(1, 6))
    Wp, Vp = np.linalg.qr(rnd(rng, cplx, n, dd))[0], np.linalg.qr(rnd(rng, cplx, n, dd))[0]
    T = np.linalg.inv(Wp.conj().T.dot(Vp))
    WRH = rnd(rng, cplx, dd, dd)
    a_ = rnd(rng, cplx, n, 1)
    pr = []
    for c in (ctx, dbl):
        pj = c.proj_create(c.upload(Wp), c.upload(Vp), dd, T, WRH, 2)
        Zd = c.alloc(n, 1, dtype=dt)
        ya = c.proj_apply_complement(pj, c.upload(a_), 0, Zd, 0, want_ya=True)
        pr.append((Zd.download(), ya))
    expect("projector complement", pr[0][0], pr[1][0])
    expect("projector <Y, a>", pr[0][1], pr[1][1])
    return checks


def cycle_round(ctx, dbl, seed):
    """kh_gmres_cycle (a run of GMRES iterations in one C call) against the same steps taken one at a time on the NumPy
    double with the Givens recurrences in NumPy (krypy/linsys.py:980-997), and the deferred MINRES update + flush against
    the immediate one (bit for bit)."""
    rng = np.random.default_rng(130_000 + seed)
    n = int(rng.integers(30, 40_000))
    m = int(rng.integers(3, 14))
    offs = sorted({0, 1, -1} | {int(o) for o in rng.integers(-min(n - 1, 300), min(n - 1, 300) + 1, size=2)})
    A = sp.diags([rng.standard_normal(n - abs(o)) + (5.0 if o == 0 else 0.0) for o in offs], offs, shape=(n, n), format="csr")
    with_m = bool(rng.integers(0, 2))
    d = rng.uniform(0.5, 2.0, n)
    v = rng.standard_normal(n)
    gs, sweeps = [(0, 1), (0, 2), (1, 1), (1, 2)][rng.integers(0, 4)]
    tol, bnorm = float(10.0 ** rng.uniform(-9, -1)), float(rng.uniform(0.5, 2.0))
    checks = 0
    if not hasattr(ctx, "gmres_cycle"):
        return 0
    # device: one call
    Vd, Wd = ctx.alloc(n, m + 1), ctx.alloc(n, 2)
    Pd = ctx.alloc(n, m + 1) if with_m else None
    Md = ctx.diag(d) if with_m else None
    nrm = np.sqrt(np.dot(v, d * v)) if with_m else np.linalg.norm(v)
    if with_m:
        Pd.upload(0, (v / nrm).reshape(-1, 1))
        Vd.upload(0, (d * v / nrm).reshape(-1, 1))
    else:
        Vd.upload(0, (v / nrm).reshape(-1, 1))
    H, R = np.zeros((m + 1, m)), np.zeros((m + 1, m))
    cs, y, resn = np.zeros(2 * m), np.zeros(m + 1), np.zeros(m)
    y[0] = nrm
    k_done, enq, h2, why = ctx.gmres_cycle(ctx.csr(A), Md, Vd, Pd, Wd, 0, m - 1, m - 1, sweeps, gs, 0, tol, bnorm, H, R, cs,
                                            y, 0.0, resn)
    for j in range(k_done, enq):            # speculative steps: fetched and dropped
        ctx.arnoldi_step_end(j % 4, j + 2)
    # double: the same steps one at a time
    V2, W2 = dbl.alloc(n, m + 1), dbl.alloc(n, 2)
    P2 = dbl.alloc(n, m + 1) if with_m else None
    M2 = dbl.diag(d) if with_m else None
    if with_m:
        P2.upload(0, (v / nrm).reshape(-1, 1))
        V2.upload(0, (d * v / nrm).reshape(-1, 1))
    else:
        V2.upload(0, (v / nrm).reshape(-1, 1))
    A2 = dbl.csr(A)
    H2, R2, y2, rot, res2 = np.zeros((m + 1, m)), np.zeros((m + 1, m)), np.zeros(m + 1), [], []
    y2[0] = nrm
    kk = 0
    stop = 0
    for k in range(m - 1):
        col = np.array(dbl.arnoldi_step(A2, M2, V2, P2, W2, 0, k, 0, sweeps, gs, 0.0), dtype=float)
        H2[: k + 2, k] = col
        for i, (c, s_) in enumerate(rot):
            col[i], col[i + 1] = c * col[i] + s_ * col[i + 1], -s_ * col[i] + c * col[i + 1]
        a_, b_ = col[k], col[k + 1]
        r_ = np.hypot(a_, b_) * (1.0 if (a_ if abs(a_) > abs(b_) else b_) >= 0 else -1.0)
        c, s_ = (1.0, 0.0) if r_ == 0 else (a_ / r_, b_ / r_)
        rot.append((c, s_))
        col[k], col[k + 1] = c * a_ + s_ * b_, -s_ * a_ + c * b_
        R2[: k + 2, k] = col
        y2[k], y2[k + 1] = c * y2[k] + s_ * y2[k + 1], -s_ * y2[k] + c * y2[k + 1]
        kk = k + 1
        res2.append(abs(y2[k + 1]))
        if not (abs(y2[k + 1]) / bnorm > tol):
            stop = 1
            break

    def expect(name, got, want, rtol=1e-10):
        nonlocal checks
        checks += 1
        got, want = np.asarray(got), np.asarray(want)
        scale = max(1.0, float(np.max(np.abs(want)))) if want.size else 1.0
        if got.shape != want.shape or not np.allclose(got, want, rtol=rtol, atol=rtol * scale):
            raise AssertionError("seed %d (n=%d, m=%d, gs=%d x %d%s): gmres_cycle %s deviates" % (
                seed, n, m, gs, sweeps, ", Jacobi" if with_m else "", name))

    expect("number of recorded steps / stop reason", [k_done, why], [kk, stop])
    expect("H", H[:, :kk], H2[:, :kk])
    expect("R", R[:, :kk], R2[:, :kk])
    expect("y", y[: kk + 1], y2[: kk + 1])
    expect("residual recurrence", resn[:kk], res2)
    expect("basis", Vd.download(0, kk + 1), V2.download(0, kk + 1))
    # deferred MINRES update + flush == the immediate update
    Vm, Wm, ym = rng.standard_normal((n, 2)), rng.standard_normal((n, 2)), rng.standard_normal((n, 1))
    co = [float(x) for x in rng.standard_normal(4)]
    co[2] += 3.0
    outs = []
    for defer in (True, False):
        Vx, Wx, Yx = ctx.upload(Vm), ctx.upload(Wm), ctx.upload(ym)
        ctx.minres_update(Vx, 1, Wx, 1, co[0], co[1], co[2], co[3], Yx, 0, defer=defer)
        ctx.minres_flush()
        outs.append((Wx.download(), Yx.download()))
    checks += 1
    if not (np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])):
        raise AssertionError("seed %d: deferred MINRES update differs from the immediate one" % seed)
    return checks


def minres_cycle_round(ctx, dbl, seed):
    """kh_minres_cycle (a run of MINRES iterations in one C call, krypy/linsys.py:791-853) against the NumPy restatement of
    its contract (tests/support/numpy_context.py: the per-step entries of the double + the rotations in NumPy): recorded
    steps and stop reason, the three entries per column of the Lanczos matrix, rotations / rotated right-hand side, the
    residual recurrence, and - after a flush - the direction vectors and the iterate; with and without Jacobi, in two
    calls (the second continues where the first stopped)."""
    if not hasattr(ctx, "minres_cycle"):
        return 0
    rng = np.random.default_rng(170_000 + seed)
    n = int(rng.integers(30, 60_000))
    m = int(rng.integers(4, 24))
    offs = sorted({1} | {int(o) for o in rng.integers(2, min(n - 1, 300) + 1, size=2)})
    diags = [rng.standard_normal(n) + 6.0] + [rng.standard_normal(n - o) for o in offs]
    A = sp.diags(diags + diags[1:], [0] + offs + [-o for o in offs], shape=(n, n), format="csr")
    with_m = bool(rng.integers(0, 2))
    d = rng.uniform(0.5, 2.0, n)
    v = rng.standard_normal(n)
    tol, bnorm = float(10.0 ** rng.uniform(-12, -2)), float(rng.uniform(0.5, 2.0))
    nrm = np.sqrt(np.dot(v, d * v)) if with_m else np.linalg.norm(v)
    split = int(rng.integers(1, m - 1))
    out = []
    for c in (ctx, dbl):
        V, W = c.alloc(n, m + 1), c.alloc(n, 2)
        P = c.alloc(n, m + 1) if with_m else None
        Md = c.diag(d) if with_m else None
        if with_m:
            P.upload(0, (v / nrm).reshape(-1, 1))
            V.upload(0, (d * v / nrm).reshape(-1, 1))
        else:
            V.upload(0, (v / nrm).reshape(-1, 1))
        Wm, YK = c.alloc(n, 2), c.alloc(n, 1)
        H, st, resn = np.zeros((m + 1, m)), np.zeros(8), np.zeros(m)
        st[5] = nrm
        Ad = c.csr(A)
        k, enq, h2, slot, why = c.minres_cycle(Ad, Md, V, P, W, 0, split, m - 1, 0, 0, tol, bnorm, H, Wm, 0, YK, 0, st, 0.0, resn)
        first = (k, why)
        if why == 0 and k == split:
            k, enq, h2, slot, why = c.minres_cycle(Ad, Md, V, P, W, k, m - 1, m - 1, 0, enq, tol, bnorm, H, Wm, slot, YK, 0, st,
                                                   h2, resn)
        for j in range(k, enq):            # speculative steps: fetched and dropped
            c.arnoldi_step_end(j % 4, j + 2)
        c.minres_flush()
        out.append(dict(first=first, k=k, why=why, slot=slot, h2=h2, H=H.copy(), st=st.copy(), resn=resn.copy(),
                        W=Wm.download(), yk=YK.download(), V=V.download(0, k + 1)))
    g, w = out
    checks = 0

    def expect(name, got, want, rtol=1e-9):
        nonlocal checks
        checks += 1
        got, want = np.asarray(got, dtype=float), np.asarray(want, dtype=float)
        scale = max(1.0, float(np.max(np.abs(want)))) if want.size else 1.0
        if got.shape != want.shape or not np.allclose(got, want, rtol=rtol, atol=rtol * scale):
            raise AssertionError("seed %d (n=%d, m=%d%s): minres_cycle %s deviates" % (seed, n, m, ", Jacobi" if with_m else "", name))

    expect("steps / stop reasons / W slot", [g["first"][0], g["first"][1], g["k"], g["why"], g["slot"]],
           [w["first"][0], w["first"][1], w["k"], w["why"], w["slot"]], rtol=0.0)
    kk = w["k"]
    expect("Lanczos matrix", g["H"][:, :kk], w["H"][:, :kk])
    expect("rotations, right-hand side", g["st"][:7], w["st"][:7])
    expect("residual recurrence", g["resn"][:kk], w["resn"][:kk])
    expect("running Frobenius norm", g["h2"], w["h2"])
    expect("basis", g["V"], w["V"])
    expect("direction vectors", g["W"], w["W"], rtol=1e-8)
    expect("iterate", g["yk"], w["yk"], rtol=1e-8)
    return checks


def shard_round(ctx, seed):
    """Block-row shards on one device: a random banded / scattered matrix cut into random uneven slabs, ghost
    entries written by hand (kh_mat_set_ghost) instead of the halo exchange - every slab must reproduce its rows of
    the global product bit for bit, for one vector and for a block, real and complex."""
    from krypy_amd import dist
    rng = np.random.default_rng(90_000 + seed)
    cplx = bool(rng.integers(0, 2))
    dt = complex if cplx else float
    n = int(rng.integers(400, 60_000))
    bw = int(rng.integers(1, max(2, min(n // 8, 400))))
    if rng.integers(0, 2):
        offs = sorted({0} | {int(o) for o in rng.integers(-bw, bw + 1, size=rng.integers(2, 7))})
        A = sp.diags([rnd(rng, cplx, n - abs(o)) for o in offs], offs, shape=(n, n), format="csr")
    else:
        nnz = 6 * n
        rows = rng.integers(0, n, nnz)
        cols = np.clip(rows + rng.integers(-bw, bw + 1, nnz), 0, n - 1)
        A = sp.coo_matrix((rnd(rng, cplx, nnz), (rows, cols)), shape=(n, n)).tocsr()
        A.sum_duplicates()
    A = sp.csr_matrix(A, dtype=dt)
    A.sort_indices()
    nslab = int(rng.integers(2, 7))
    inner = np.sort(rng.choice(np.arange(1, n // max(bw, 1)), size=nslab - 1, replace=False)) * max(bw, 1)
    cuts = [0] + [int(c) for c in inner if bw <= c <= n - bw] + [n]
    cuts = sorted(set(cuts))
    d = int(rng.choice([1, 3]))
    x = rnd(rng, cplx, n, d)
    want = A.dot(x)
    checks = 0
    for p in range(len(cuts) - 1):
        r0, r1 = cuts[p], cuts[p + 1]
        if r1 - r0 < bw:
            continue
        A_local, nrp, nrn = dist.localize_columns(A[r0:r1], r0, n)
        Ad = ctx.csr(A_local, n_cols=A_local.shape[1], dtype=dt)
        ctx.set_halo(Ad, 0, 0, nrp, nrn)
        X, Y = ctx.upload(x[r0:r1]), ctx.alloc(r1 - r0, d, dtype=dt)
        for c in range(d):        # (the ghost buffer holds one vector's halo: a block is applied column by column)
            gh = np.concatenate([x[r0 - nrp:r0, c], x[r1:r1 + nrn, c]])
            ctx.set_ghost(Ad, gh)
            if hasattr(ctx, "get_ghost") and not np.array_equal(ctx.get_ghost(Ad, nrp + nrn), gh):
                raise AssertionError("seed %d: kh_mat_get_ghost does not return what kh_mat_set_ghost wrote" % seed)
            ctx.apply(Ad, X, c, Y, c, 1)
        got = Y.download()
        checks += 1
        if not np.array_equal(got, want[r0:r1]):
            raise AssertionError("seed %d: slab %d of %s (n=%d, bandwidth %d, %s): %.3e" % (
                seed, p, cuts, n, bw, "complex" if cplx else "real", float(np.max(np.abs(got - want[r0:r1])))))
    return checks


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    max_n = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
    from krypy_amd import _hip
    ctx = _hip.get_context()
    try:        # a generator bug must end in a MemoryError, not take the box down (tests/conftest.py does the same)
        import resource
        resource.setrlimit(resource.RLIMIT_DATA, (48 << 30, resource.getrlimit(resource.RLIMIT_DATA)[1]))
    except Exception:
        pass
    total = 0
    from tests.support.numpy_context import NumpyContext
    dbl = NumpyContext()
    for seed in range(rounds):
        total += one_round(ctx, seed, max_n)
        total += step_round(ctx, dbl, seed, max_n)
        total += shard_round(ctx, seed)
        total += cycle_round(ctx, dbl, seed)
        total += minres_cycle_round(ctx, dbl, seed)
    print("abi_fuzz: %d rounds, %d comparisons, all within tolerance" % (rounds, total))
