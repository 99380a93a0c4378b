"""Full-size GPU runs of BASELINE.json configs 3, 4 and (single-GPU shape of) 5: size-independent
properties, since the CPU oracle cannot finish these sizes in seconds.  Marked gpu (and slow)."""
import time

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import krylov_ref as ref

pytestmark = pytest.mark.gpu


def test_config3_minres_jacobi_full_size(hip):
    """2-D 5-pt Laplacian N = 10^7, MINRES + Jacobi M, ortho='lanczos', 200 steps (V and P are both
    stored: maxiter must be bounded).  Checks: monotone residuals (MINRES minimises the
    M^-1-norm of the residual), Lanczos matrix tridiagonal + symmetric, true residual of the
    returned iterate equals the last recorded one."""
    from krypy_amd import linsys, utils

    A = ref.laplace2d(4000, 2500)
    N = A.shape[0]
    b = np.random.default_rng(0).standard_normal(N)
    d = A.diagonal()
    M, Minv = sp.diags(1.0 / d).tocsr(), sp.diags(d).tocsr()
    ls = linsys.LinearSystem(A, b, M=M, Minv=Minv, self_adjoint=True)
    t0 = time.perf_counter()
    try:
        sol = linsys.Minres(ls, ortho="lanczos", tol=1e-8, maxiter=200, store_arnoldi=False)
        raise AssertionError("tolerance cannot be reached in 200 steps at this N")
    except utils.ConvergenceError as e:
        sol = e.solver
    dt = time.perf_counter() - t0
    res = np.array(sol.resnorms)
    assert len(res) == 201 and np.all(np.diff(res[:-1]) <= 1e-14)
    H = sol.lanczos.H[:199, :199]
    assert np.allclose(H, H.T, atol=1e-10) and np.all(np.triu(H, 2) == 0)
    x = sol.xk[:, 0]
    r = b - A.dot(x)
    rn = np.sqrt(np.dot(r, r / d)) / np.sqrt(np.dot(b, b / d))
    assert abs(rn - res[-1]) < 1e-8 * res[-1]
    print("config 3: %.1f MINRES iterations/s" % (200 / dt))


def test_config4_dense_cg_full_size(hip):
    """Dense SPD n = 32768 (8.6 GB, streamed once per CG step through k_gemv_dense), CG to 1e-8.
    A = S S^T-free construction that is cheap on the host: symmetric random + diagonal shift (SPD by
    Gershgorin); checks the residual identity and CG's monotone A-norm error."""
    from krypy_amd import linsys

    n = 32768
    rng = np.random.default_rng(0)
    A = rng.standard_normal((n, n))
    A = (A + A.T) * (0.5 / np.sqrt(n))
    A[np.diag_indices(n)] += 3.0            # spectrum roughly in [1, 5]
    xs = rng.standard_normal(n)
    b = A.dot(xs)
    ls = linsys.LinearSystem(A, b, self_adjoint=True, positive_definite=True, exact_solution=xs)
    t0 = time.perf_counter()
    sol = linsys.Cg(ls, tol=1e-8, maxiter=200)
    dt = time.perf_counter() - t0
    assert sol.resnorms[-1] <= 1e-8 and sol.iter < 60
    x = sol.xk[:, 0]
    assert np.linalg.norm(b - A.dot(x)) <= 1.001e-8 * np.linalg.norm(b)
    assert np.linalg.norm(x - xs) < 1e-7 * np.linalg.norm(xs)
    print("config 4: %d CG iterations, %.1f iterations/s (incl. setup)" % (sol.iter, sol.iter / dt))


def test_config5_shape_deflated_gmres_single_gpu(hip):
    """3-D 7-pt Laplacian 200^3 (N = 8*10^6, the per-GPU share of config 5 is 1.25*10^7), plain
    GMRES(60) to harvest 16 smallest-magnitude Ritz vectors, then DeflatedGmres with them:
    deflation identities E = <U,AU>, C = <U, A V_n>, projected residual orthogonal to U, and the
    deflated solve reduces the residual further than the plain one in the same number of steps."""
    from krypy_amd import deflation, linsys, utils

    A = ref.laplace3d(200)
    N = A.shape[0]
    b = np.random.default_rng(0).standard_normal(N)
    ls = linsys.LinearSystem(A, b, self_adjoint=True)
    try:
        s0 = deflation.DeflatedGmres(ls, tol=1e-12, maxiter=60, store_arnoldi=True)
    except utils.ConvergenceError as e:
        s0 = e.solver
    ritz = deflation.Ritz(s0)
    idx = np.argsort(np.abs(ritz.values))[:16]
    Ud = ritz._get_vectors_dev(idx)                  # stays on the device
    try:
        s1 = deflation.DeflatedGmres(ls, U=Ud, tol=1e-12, maxiter=60, store_arnoldi=True)
    except utils.ConvergenceError as e:
        s1 = e.solver
    assert s1.resnorms[-1] < s0.resnorms[-1]
    ctx = ls._ctx
    U, AU = s1.projection._Ud, s1.projection._AUd
    E = ctx.gemm_tn(U, 0, 16, AU, 0, 16)
    assert np.linalg.norm(E - s1.E) < 1e-10 * np.linalg.norm(E)
    assert np.linalg.norm(ctx.gemm_tn(U, 0, 16, U, 0, 16) - np.eye(16)) < 1e-12
    n = s1.H.shape[1]
    T = ctx.alloc(N, 1)
    ctx.apply(ls.A._device_matrix(), s1.arnoldi._V, 3, T, 0, 1)
    c3 = ctx.gemm_tn(U, 0, 16, T, 0, 1)[:, 0]
    assert np.linalg.norm(c3 - s1.C[:, 3]) < 1e-9 * max(np.linalg.norm(c3), 1e-30)
    # the Krylov basis of the projected operator is orthogonal to U^* A (range of P = ker <U, .>)
    G = ctx.gemm_tn(U, 0, 16, s1.arnoldi._V, 0, n)
    assert np.linalg.norm(G) < 1e-9
