"""GPU tests of the complex (c128) kernels of libkrylov_hip (zpath.h), through the C ABI:
edge sizes, mixed shapes, bit-level agreement of the complex CSR SpMV with SciPy, and the complex
Arnoldi step against NumPy.  The solver-level complex parity cases run from tests/test_gpu_parity.py
(tests/parity_cases_complex.py)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from tests.support.kernel_expect import expect_kernel

pytestmark = pytest.mark.gpu


def _crand(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 4097, 100003, 1 << 20])
def test_complex_vector_kernels_edge_sizes(hip, n):
    rng = np.random.default_rng(n)
    k = 19
    X, w = _crand(rng, n, k), _crand(rng, n, 1)
    Xd, Wd = hip.upload(X), hip.upload(w)
    assert Xd.dtype == np.complex128 and Xd.n == n
    assert np.array_equal(Xd.download(), X) and np.array_equal(Xd.download(3, 2), X[:, 3:5])
    # <X, w> = X^H w
    got = hip.dot_panel(Xd, 0, k, Wd, 0)
    want = X.conj().T.dot(w)[:, 0]
    assert np.allclose(got, want, rtol=1e-13, atol=1e-13 * np.sqrt(n) * 4)
    G = hip.gemm_tn(Xd, 2, 5, Xd, 1, 3)
    assert np.allclose(G, X[:, 2:7].conj().T.dot(X[:, 1:4]), rtol=1e-12, atol=1e-12 * n)
    # 2-norm of the (re, im) view
    assert abs(hip.nrm2(Wd, 0) - np.linalg.norm(w)) <= 1e-14 * np.linalg.norm(w) * 4
    # w -= sum h_j X_j, left to right (NumPy's SIMD complex multiply may fuse: compare to 1e-14)
    h = _crand(rng, k)
    ref = w[:, 0].copy()
    for j in range(k):
        ref = ref - h[j] * X[:, j]
    hip.axpy_panel(Xd, 0, k, h, Wd, 0)
    assert np.allclose(Wd.download()[:, 0], ref, rtol=1e-14, atol=1e-14 * k)
    # Y = beta Y + X C
    C = _crand(rng, 7, 2)
    Y0 = _crand(rng, n, 2)
    Yd = hip.upload(Y0)
    hip.gemm_nn(Xd, 4, 7, C, 1.0, 1.0, Yd, 0)
    assert np.allclose(Yd.download(), Y0 + X[:, 4:11].dot(C), rtol=1e-13, atol=1e-13)
    hip.gemm_nn(Xd, 4, 7, C, 2.0 - 1.0j, 0.0, Yd, 0)
    assert np.allclose(Yd.download(), X[:, 4:11].dot((2.0 - 1.0j) * C), rtol=1e-13, atol=1e-13)
    # z = alpha x + beta y, complex coefficients; real coefficients take the real kernel
    a, b = 0.3 - 1.2j, -0.7 + 0.1j
    Z = hip.alloc(n, 1, dtype=complex)
    hip.waxpby(Z, 0, a, Xd, 0, b, Xd, 1)
    assert np.allclose(Z.download()[:, 0], a * X[:, 0] + b * X[:, 1], rtol=1e-14, atol=1e-14)
    hip.waxpby(Z, 0, 2.0, Xd, 0, -1.0, Xd, 1)
    assert np.array_equal(Z.download()[:, 0], 2.0 * X[:, 0] - X[:, 1])
    hip.vdiv(Z, 0, Xd, 2, 3.0)
    assert np.allclose(Z.download()[:, 0], X[:, 2] / 3.0, rtol=1e-15, atol=0)   # (NumPy: Smith's division)
    hip.vdiv(Z, 0, Xd, 2, 1.0 + 2.0j)
    assert np.allclose(Z.download()[:, 0], X[:, 2] / (1.0 + 2.0j), rtol=1e-14)
    # widening a real block, element access, partial zero
    R = rng.standard_normal((n, 2))
    Rd = hip.upload(R)
    hip.promote(Rd, 1, Z, 0, 1)
    assert np.array_equal(Z.download()[:, 0], R[:, 1].astype(complex))
    Z.set(0, n - 1, [5.0 - 2.0j])
    assert Z.get(0, n - 1, 1)[0] == 5.0 - 2.0j
    Z.zero_range(0, 0, n)
    assert not Z.download().any()
    with pytest.raises(Exception):
        hip.dot_panel(Rd, 0, 1, Wd, 0)          # real and complex blocks never mix
    with pytest.raises(Exception):
        hip.axpy_panel(Rd, 0, 1, [1.0j], Rd, 1)  # complex coefficient for real blocks


@pytest.mark.parametrize("kind", ["laplace", "random", "empty_rows", "long_row", "rect"])
def test_complex_csr_spmv_bit_identical_to_scipy(hip, kind):
    rng = np.random.default_rng(3)
    if kind == "laplace":
        from oracle.inputs import complex_systems
        A = complex_systems(40)["nonh"]
    elif kind == "random":
        A = sp.random(3000, 3000, density=4e-3, random_state=3, format="csr")
    elif kind == "empty_rows":
        A = sp.random(5000, 5000, density=1e-3, random_state=4, format="csr")
        A = sp.vstack([A[:100], sp.csr_matrix((300, 5000)), A[400:]]).tocsr()
    elif kind == "long_row":
        A = sp.random(300, 50000, density=1e-3, random_state=5, format="lil")
        A[7, :] = rng.standard_normal(50000)
        A = A.tocsr()
    else:
        A = sp.random(1000, 3000, density=5e-3, random_state=6, format="csr")
    A = A.tocsr().astype(complex)
    if kind != "laplace":
        A.data = A.data + 1j * rng.standard_normal(A.nnz)
    A.sort_indices()
    x = _crand(rng, A.shape[1], 2)
    Ad = hip.csr(A)
    assert Ad.dtype == np.complex128
    X, Y = hip.upload(x), hip.alloc(A.shape[0], 2, dtype=complex)
    hip.apply(Ad, X, 0, Y, 0, 2)
    got, want = Y.download(), A.dot(x)
    if kind == "long_row":
        mask = np.ones(A.shape[0], bool)
        mask[7] = False
        assert np.array_equal(got[mask], want[mask])
        assert np.allclose(got[7], want[7], rtol=1e-12)
    else:
        assert np.array_equal(got, want)
    # a real matrix uploaded for complex vectors
    Ar = A.real.tocsr()
    Ard = hip.csr(Ar, dtype=complex)
    hip.apply(Ard, X, 0, Y, 0, 1)
    assert np.allclose(Y.download()[:, 0], Ar.dot(x[:, 0]), rtol=1e-13, atol=1e-13)
    with pytest.raises(Exception):
        hip.apply(hip.csr(Ar), X, 0, Y, 0, 1)    # real operator, complex blocks


def test_complex_dense_gemv_and_diag(hip):
    rng = np.random.default_rng(2)
    for n, m in ((1, 1), (37, 41), (512, 512), (1000, 999)):
        A, x = _crand(rng, n, m), _crand(rng, m, 1)
        Y = hip.alloc(n, 1, dtype=complex)
        hip.apply(hip.dense(A), hip.upload(x), 0, Y, 0, 1)
        assert np.allclose(Y.download(), A.dot(x), rtol=1e-13, atol=1e-12)
    d, x = _crand(rng, 777), _crand(rng, 777, 1)
    Y = hip.alloc(777, 1, dtype=complex)
    hip.apply(hip.diag(d), hip.upload(x), 0, Y, 0, 1)
    assert np.allclose(Y.download()[:, 0], d * x[:, 0], rtol=1e-15, atol=0)


@pytest.mark.parametrize("mode", ["mgs", "dmgs", "cgs", "lanczos"])
def test_complex_arnoldi_step_against_numpy(hip, mode):
    """kh_zarnoldi_step, step by step, against the same recurrence in NumPy (1e-12)."""
    from krypy_amd import _hip
    from oracle.inputs import complex_systems

    c = complex_systems(40)
    A = c["hind"] if mode == "lanczos" else c["nonh"]
    N, m = A.shape[0], 20
    rng = np.random.default_rng(1)
    v = _crand(rng, N)
    V = np.zeros((N, m + 1), dtype=complex, order="F")
    V[:, 0] = v / np.linalg.norm(v)
    Vd = hip.alloc(N, m + 1, dtype=complex)
    Vd.upload(0, V[:, [0]])
    Wd = hip.alloc(N, 2, dtype=complex)
    Ad = hip.csr(A)
    sweeps = 2 if mode == "dmgs" else 1
    gs = _hip.GS_CGS if mode == "cgs" else _hip.GS_MGS
    H = np.zeros((m + 1, m), dtype=complex)
    for k in range(m):
        start = k if mode == "lanczos" else 0
        hk = H[k, k - 1] if (mode == "lanczos" and k > 0) else 0.0
        hcol = hip.arnoldi_step(Ad, None, Vd, None, Wd, 0, k, start, sweeps, gs, hk)
        w = A.dot(V[:, k])
        if mode == "lanczos" and k > 0:
            w = w - hk * V[:, k - 1]
            H[k - 1, k] = hk
        for _ in range(sweeps):
            if mode == "cgs":
                h = V[:, : k + 1].conj().T.dot(w)
                H[: k + 1, k] += h
                for j in range(k + 1):
                    w = w - h[j] * V[:, j]
            else:
                for j in range(start, k + 1):
                    a = np.vdot(V[:, j], w)
                    H[j, k] += a
                    w = w - a * V[:, j]
        H[k + 1, k] = np.linalg.norm(w)
        V[:, k + 1] = w / H[k + 1, k]
        assert hcol.shape == (k + 2,) and hcol[k + 1].imag == 0.0
        assert np.allclose(hcol[start:], H[start: k + 2, k], rtol=1e-11, atol=1e-12), (mode, k)
    assert np.linalg.norm(Vd.download() - V) < 1e-10


@pytest.mark.parametrize("mode", ["mgs", "dmgs", "cgs", "lanczos"])
def test_complex_arnoldi_step_with_jacobi_against_numpy(hip, mode):
    """kh_zarnoldi_step_begin_md: the complex step with a diagonal preconditioner (V = M P, utils.py:1026-1045) -
    coefficients against V, updates with P, norm sqrt(Re <w, M w>) - step by step against NumPy."""
    from krypy_amd import _hip
    from oracle.inputs import complex_systems

    c = complex_systems(40)
    A = c["hind"] if mode == "lanczos" else c["nonh"]
    N, m = A.shape[0], 20
    rng = np.random.default_rng(4)
    d = 0.5 + rng.random(N)
    p0 = _crand(rng, N)
    nrm0 = np.sqrt(np.vdot(p0, d * p0).real)
    P = np.zeros((N, m + 1), dtype=complex, order="F")
    V = np.zeros((N, m + 1), dtype=complex, order="F")
    P[:, 0], V[:, 0] = p0 / nrm0, d * p0 / nrm0
    Vd, Pd = hip.alloc(N, m + 1, dtype=complex), hip.alloc(N, m + 1, dtype=complex)
    Vd.upload(0, V[:, [0]])
    Pd.upload(0, P[:, [0]])
    Wd = hip.alloc(N, 2, dtype=complex)
    Ad, Md = hip.csr(A), hip.diag(d, dtype=complex)
    sweeps = 2 if mode == "dmgs" else 1
    gs = _hip.GS_CGS if mode == "cgs" else _hip.GS_MGS
    H = np.zeros((m + 1, m), dtype=complex)
    for k in range(m):
        start = k if mode == "lanczos" else 0
        hk = H[k, k - 1] if (mode == "lanczos" and k > 0) else 0.0
        if k % 2:       # both entry styles: synchronous, and begin / end in a slot
            hcol = hip.arnoldi_step(Ad, Md, Vd, Pd, Wd, 0, k, start, sweeps, gs, hk)
        else:
            hip.arnoldi_step_begin(Ad, Md, Vd, Pd, Wd, 0, k, start, sweeps, gs, hk, k % 4)
            hcol = hip.arnoldi_step_end(k % 4, k + 2, cplx=True)
        w = A.dot(V[:, k])
        if mode == "lanczos" and k > 0:
            w = w - hk * P[:, k - 1]
            H[k - 1, k] = hk
        for _ in range(sweeps):
            if mode == "cgs":
                h = V[:, : k + 1].conj().T.dot(w)
                H[: k + 1, k] += h
                for j in range(k + 1):
                    w = w - h[j] * P[:, j]
            else:
                for j in range(start, k + 1):
                    a = np.vdot(V[:, j], w)
                    H[j, k] += a
                    w = w - a * P[:, j]
        H[k + 1, k] = np.sqrt(np.vdot(w, d * w).real)
        P[:, k + 1], V[:, k + 1] = w / H[k + 1, k], d * w / H[k + 1, k]
        assert hcol.shape == (k + 2,) and hcol[k + 1].imag == 0.0
        assert np.allclose(hcol[start:], H[start: k + 2, k], rtol=1e-11, atol=1e-12), (mode, k)
    assert np.linalg.norm(Vd.download() - V) < 1e-10 and np.linalg.norm(Pd.download() - P) < 1e-10
    # P^H V = P^H M P = I: the basis is orthonormal in the M inner product
    if mode != "lanczos":
        assert np.linalg.norm(Pd.download().conj().T.dot(Vd.download()) - np.eye(m + 1)) < 1e-10


@pytest.mark.parametrize("n", [1, 63, 4097, 300001])
def test_complex_minres_update_and_cg_step_kernels(hip, n):
    """kh_zminres_update and kh_zcg_step against the same formulas in NumPy (complex scalars; the CG step on the
    real views with a duplicated real Jacobi diagonal), edge sizes included."""
    rng = np.random.default_rng(n)
    # --- MINRES: z = (v - r0 W0 - r1 W1)/r2; W <- [W1, z]; yk += y0 z
    V, W, yk = _crand(rng, n, 3), _crand(rng, n, 2), _crand(rng, n, 1)
    r0, r1, r2, y0 = (complex(*rng.standard_normal(2)) for _ in range(4))
    for slot, rr2 in ((0, r2), (1, complex(r2.imag, 5 * r2.real)), (0, complex(2.5, 0.0))):    # |re| >= |im|, <, real
        Vd, Wd, Yd = hip.upload(V), hip.upload(W), hip.upload(yk)
        hip.minres_update(Vd, 1, Wd, slot, r0, r1, rr2, y0, Yd, 0)
        z = ((V[:, 1] - r0 * W[:, slot]) - r1 * W[:, 1 - slot]) / rr2
        got = Wd.download()
        assert np.allclose(got[:, slot], z, rtol=1e-14, atol=1e-14)
        assert np.array_equal(got[:, 1 - slot], W[:, 1 - slot])
        assert np.allclose(Yd.download()[:, 0], yk[:, 0] + y0 * z, rtol=1e-14, atol=1e-14)
    # --- CG: p = z + omega p; Ap = A p; alpha = Re(rho / <p, Ap>); yk += alpha p; r -= alpha Ap; z = D r; <r, z>
    # Hermitian, diagonally dominant, banded (NOT sp.random: its sampling allocates n^2 integers)
    offs = [o for o in (1, 7, 64) if o < n]
    B = sp.diags([_crand(rng, n - o) for o in offs], offs, shape=(n, n)) if offs else sp.csr_matrix((n, n), dtype=complex)
    A = (B + B.conj().T + sp.identity(n) * 12.0).tocsr()
    Ad = hip.csr(A)
    d = 0.5 + rng.random(n)
    Dd = hip.diag(np.repeat(d, 2))
    for jac in (False, True):
        for first in (True, False):
            p, r, zv, y = _crand(rng, n, 1), _crand(rng, n, 1), _crand(rng, n, 1), _crand(rng, n, 1)
            pd_, rd, zd, yd, apd = hip.upload(p), hip.upload(r), hip.upload(zv), hip.upload(y), hip.alloc(n, 1, dtype=complex)
            omega, rho = 0.37, 1.9
            den, rho_new, pap, flags = hip.cg_step(Ad, Dd if jac else None, pd_, 0, apd, 0, yd, 0, rd, 0, zd if jac else None, 0,
                                            first, omega, rho)
            pp = p[:, 0] if first else (zv[:, 0] if jac else r[:, 0]) + omega * p[:, 0]
            ap = A.dot(pp)
            want = np.vdot(pp, ap)
            assert abs(pap - want) <= 1e-13 * abs(want) * max(1.0, np.sqrt(n) / 30)
            assert flags == (0 if den > 0 else 2), flags        # (random p: <p, Ap> > 0 for this positive definite A)
            alpha = (rho / pap).real
            assert abs(rho / den - alpha) <= 4e-16 * abs(alpha)
            alpha = rho / den
            rn = r[:, 0] - alpha * ap
            zn = d * rn if jac else rn
            tol = dict(rtol=1e-13, atol=1e-13)
            assert np.allclose(pd_.download()[:, 0], pp, **tol) and np.allclose(apd.download()[:, 0], ap, **tol)
            assert np.allclose(yd.download()[:, 0], y[:, 0] + alpha * pp, **tol)
            assert np.allclose(rd.download()[:, 0], rn, **tol)
            if jac:
                assert np.allclose(zd.download()[:, 0], zn, **tol)
            assert abs(rho_new - np.vdot(rn, zn).real) <= 1e-12 * abs(np.vdot(rn, zn).real)


def test_complex_shard_with_ghost_columns_and_rccl_path(hip):
    """Complex block-row sharding on the device: (i) every slab of a 3-way split multiplied with its ghost entries
    written by hand (kh_mat_set_ghost) equals the global complex SpMV bit for bit; (ii) a complex solve and a
    complex DEFLATED solve (projector inside the complex step, kh_zproj_create) through the multi-rank code path
    on a 1-rank RCCL communicator in forced mode equal the plain single-GPU solves."""
    import os
    from krypy_amd import _hip, deflation, dist as kdist, linsys
    from oracle.inputs import complex_systems

    c = complex_systems(24)
    A, b = c["nonh"].tocsr(), c["b"]
    N = A.shape[0]
    rng = np.random.default_rng(9)
    x = _crand(rng, N)
    want = A.dot(x)
    cuts = kdist.slab_cuts(N, 3, align=24)
    for p in range(3):
        r0, r1 = cuts[p], cuts[p + 1]
        Al, nrp, nrn = kdist.localize_columns(A[r0:r1], r0, N)
        Ad = hip.csr(Al, n_cols=Al.shape[1])
        hip.set_halo(Ad, 0, 0, nrp, nrn)
        ghost = np.concatenate([x[r0 - nrp:r0], x[r1:r1 + nrn]])
        hip.set_ghost(Ad, ghost)
        X, Y = hip.upload(x[r0:r1]), hip.alloc(r1 - r0, 1, dtype=complex)
        hip.apply(Ad, X, 0, Y, 0, 1)
        assert np.array_equal(Y.download()[:, 0], want[r0:r1]), p
    U = np.linalg.qr(_crand(rng, N, 4))[0]
    ls0 = linsys.LinearSystem(A, b)
    want_s = (linsys.Gmres(ls0, tol=1e-10, maxiter=300), deflation.DeflatedGmres(ls0, U=U, tol=1e-9, maxiter=300))
    os.environ["KRYPY_AMD_FORCE_MULTI"] = "1"
    try:
        ctx = _hip.Context(0)
        ctx.comm_init(0, 1, ctx.comm_unique_id())
    finally:
        del os.environ["KRYPY_AMD_FORCE_MULTI"]
    old = _hip._install_context_for_testing(ctx)
    try:
        op = kdist.ShardedCSROperator(A, 0, N, ctx)
        ls = linsys.LinearSystem(op, b)
        got = (linsys.Gmres(ls, tol=1e-10, maxiter=300), deflation.DeflatedGmres(ls, U=U, tol=1e-9, maxiter=300))
        for g, w in zip(got, want_s):
            assert len(g.resnorms) == len(w.resnorms)
            assert np.allclose(g.resnorms[:40], w.resnorms[:40], rtol=1e-9)
            assert np.linalg.norm(g.xk - w.xk) < 1e-8 * np.linalg.norm(w.xk)
        # the panel modes (register-resident complex panel kernels, their (re, im) panel all-reduced) and the
        # complex CG step / MINRES with a Jacobi preconditioner through the same multi-rank code path
        before = ctx.counters()["cgs_register"]
        for ortho in ("cgs", "cgs2"):
            g = linsys.Gmres(ls, tol=1e-10, maxiter=300, ortho=ortho)
            assert len(g.resnorms) == len(want_s[0].resnorms), ortho
            assert np.linalg.norm(g.xk - want_s[0].xk) < 1e-8 * np.linalg.norm(want_s[0].xk), ortho
        expect_kernel(ctx.counters()["cgs_register"] > before, "ctx.counters()[\"cgs_register\"] > before")
        d = np.asarray(c["hpd"].diagonal()).real
        M, Minv = sp.diags(1.0 / d).tocsr(), sp.diags(d).tocsr()
        hpd = dict(self_adjoint=True, positive_definite=True)
        for Aname, cls, flags, mi in (("hpd", linsys.Cg, hpd, 300), ("hind", linsys.Minres, dict(self_adjoint=True), 600)):
            _hip._install_context_for_testing(old)
            w = cls(linsys.LinearSystem(c[Aname], b, M=M, Minv=Minv, **flags), tol=1e-10, maxiter=mi)
            _hip._install_context_for_testing(ctx)
            g = cls(linsys.LinearSystem(kdist.ShardedCSROperator(c[Aname], 0, N, ctx), b, M=M, Minv=Minv, **flags),
                    tol=1e-10, maxiter=mi)
            assert len(g.resnorms) == len(w.resnorms), Aname
            assert np.allclose(g.resnorms[:-1], w.resnorms[:-1], rtol=1e-8), Aname
            assert np.linalg.norm(g.xk - w.xk) < 1e-8 * np.linalg.norm(w.xk), Aname
    finally:
        _hip._install_context_for_testing(old)
        ctx.close()


@pytest.mark.parametrize("shape", [("lap2d", 1300, 1000), ("lap2d", 1700, 1500), ("lap2d", 2200, 1800), ("lap2d", 2500, 2000),
                                   ("lap3d", 120, 0)])
def test_complex_operator_fused_into_the_chain_prologue(hip, shape):
    """A banded COMPLEX operator (a shifted stencil matrix) + long vectors (round 4): the complex chain kernels compute
    w = A v_k in their prologue from a diagonal-major copy of (re, im) pairs instead of reading what k_zspmv_stream wrote -
    NumPy's product formula, sums from (0, 0) in storage order, empty slots skipped: H and the basis must come out bit for
    bit as with the SpMV launch (16 ... 40 rows per lane, 5 and 7 diagonals, single and double sweeps), and the Arnoldi
    relation must hold against SciPy's product."""
    from oracle import krylov_ref as ref

    kind, a, b_ = shape
    L = ref.laplace2d(a, b_) if kind == "lap2d" else ref.laplace3d(a).tocsr()
    n = L.shape[0]
    A = (L.astype(complex) + sp.diags(1j * np.linspace(0.1, 1.0, n))).tocsr()
    rng = np.random.default_rng(3)
    v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    m = 6
    out = []
    Ad = hip.csr(A, dtype=complex)
    for fused in (1, 0):
        hip.set("chain_spmv", fused)
        try:
            before = hip.counters()
            V, W = hip.alloc(n, m + 1, dtype=complex), hip.alloc(n, 2, dtype=complex)
            V.upload(0, (v / np.linalg.norm(v)).reshape(-1, 1))
            H = np.zeros((m + 1, m), dtype=complex)
            for k in range(m):
                hcol = hip.arnoldi_step(Ad, None, V, None, W, 0, k, 0, 2 if k == 4 else 1, 0, 0.0)
                H[: k + 2, k] = hcol[: k + 2]
            c = hip.counters()
            expect_kernel(c["chain"] - before["chain"] == m, "c[\"chain\"] - before[\"chain\"] == m: %r" % (c,))
            expect_kernel(c["chain_fused"] - before["chain_fused"] == (m if fused else 0), "c[\"chain_fused\"] - before[\"chain_fused\"] == (m if fused else 0): %r" % ((fused, c),))
            out.append((H, V.download()))
            del V, W
        finally:
            hip.set("chain_spmv", 1)
    (Hf, Vf), (Hs, Vs) = out
    assert np.array_equal(Hf, Hs) and np.array_equal(Vf, Vs)
    assert np.linalg.norm(A.dot(Vf[:, :m]) - Vf.dot(Hf)) < 1e-12 * np.linalg.norm(Hf)


@pytest.mark.parametrize("nx,ny", [(1300, 1000), (1700, 1500), (2200, 1800), (2500, 2000)])
def test_complex_lanczos_step_with_the_operator_in_the_prologue(hip, nx, ny):
    """A complex HERMITIAN banded operator (D^H L D with a diagonal of phases D): the Lanczos step of complex MINRES -
    operator, pre-subtraction of H[k,k-1] v_{k-1} (a real coefficient), one Gram-Schmidt link, norm, store - is ONE launch
    of the complex chain kernel with the operator in its prologue; H and the basis bit for bit as with the separate
    SpMV / axpy launches, H real-tridiagonal to rounding, the Lanczos relation against SciPy's product."""
    from oracle import krylov_ref as ref

    L = ref.laplace2d(nx, ny)
    n = L.shape[0]
    rng = np.random.default_rng(7)
    ph = sp.diags(np.exp(1j * rng.uniform(0, 2 * np.pi, n)))
    A = (ph.conj() @ L.astype(complex) @ ph).tocsr()
    A.sort_indices()
    v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    m = 6
    Ad = hip.csr(A, dtype=complex)
    out = []
    for fused in (1, 0):
        hip.set("chain_spmv", fused)
        try:
            before = hip.counters()
            zd0 = hip.get("n_zspmv_dia")
            V, W = hip.alloc(n, m + 1, dtype=complex), hip.alloc(n, 2, dtype=complex)
            V.upload(0, (v / np.linalg.norm(v)).reshape(-1, 1))
            H = np.zeros((m + 1, m), dtype=complex)
            for k in range(m):
                hk = H[k, k - 1] if k > 0 else 0.0
                hcol = hip.arnoldi_step(Ad, None, V, None, W, 0, k, k, 1, 0, hk)
                H[k: k + 2, k] = hcol[k: k + 2]
                if k > 0:
                    H[k - 1, k] = H[k, k - 1]
            c = hip.counters()
            expect_kernel(c["chain_fused"] - before["chain_fused"] == (m if fused else 0), "c[\"chain_fused\"] - before[\"chain_fused\"] == (m if fused else 0): %r" % ((fused, c),))
            # (round 6: the separate launches take the banded complex SpMV, zpath.h: k_zspmv_dia)
            expect_kernel(fused or hip.get("n_zspmv_dia") - zd0 == m, "banded complex SpMV launches: %r" % (hip.get("n_zspmv_dia") - zd0,))
            out.append((H, V.download()))
            del V, W
        finally:
            hip.set("chain_spmv", 1)
    (Hf, Vf), (Hs, Vs) = out
    assert np.array_equal(Hf, Hs) and np.array_equal(Vf, Vs)
    assert np.max(np.abs(Hf.imag)) < 1e-12
    assert np.linalg.norm(A.dot(Vf[:, :m]) - Vf.dot(Hf)) < 1e-10 * np.linalg.norm(Hf)
