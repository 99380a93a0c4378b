"""Seeded synthetic inputs shared by the fixture generator and the parity tests.

Inputs are never stored in fixtures; both sides regenerate them from these
functions (``numpy.random.default_rng`` streams are stable across NumPy >= 1.17).
Shapes follow SURVEY.md section 8(d).
"""
import numpy as np
import scipy.sparse as sp

from .krylov_ref import laplace2d, laplace3d


def toy_system():
    """README toy (config 1): A = diag(1e-3, 2..100), b = ones(100)."""
    A = np.diag(np.concatenate([[1.0e-3], np.arange(2, 101)]).astype(float))
    return A, np.ones(100)


def _rhs(N, rhs):
    if rhs == "ones":
        return np.ones(N)
    if rhs.startswith("rng"):
        return np.random.default_rng(int(rhs[3:])).standard_normal(N)
    raise ValueError(rhs)


def lap2d_system(nx, ny=None, rhs="rng1"):
    A = laplace2d(nx, ny)
    return A, _rhs(A.shape[0], rhs)


def lap3d_system(nx, ny=None, nz=None, rhs="ones"):
    A = laplace3d(nx, ny, nz)
    return A, _rhs(A.shape[0], rhs)


def minres_jacobi_system(nx):
    """Config 3 shape: 2-D Laplacian, Jacobi M = diag(1/a_ii) and its inverse."""
    A, b = lap2d_system(nx, rhs="rng1")
    d = A.diagonal()
    return A, b, sp.diags(1.0 / d).tocsr(), sp.diags(d).tocsr()


def dense_spd_system(n):
    """Config 4 shape: G = rng(0) normal (n,n), A = G G^T / n + I, b = rng normal."""
    rng = np.random.default_rng(0)
    G = rng.standard_normal((n, n))
    A = G.dot(G.T) / n + np.eye(n)
    return A, rng.standard_normal(n)


def kernel_panel(N, k, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((N, k)), rng.standard_normal((N, 1))


def complex_systems(nx=24):
    """Complex (c128) systems on the 2-D Laplacian stencil: a Hermitian positive
    definite, a Hermitian indefinite and a non-Hermitian sparse matrix, a complex right-hand side,
    a complex initial guess and a complex deflation basis."""
    L = laplace2d(nx).tocsr()
    N = L.shape[0]
    rng = np.random.default_rng(11)
    K = sp.triu(L, 1).tocoo()
    K = sp.coo_matrix((0.3 * rng.standard_normal(K.nnz), (K.row, K.col)), shape=L.shape).tocsr()
    S = 1j * (K - K.T)                       # Hermitian: (iK - iK^T)^H = -iK^T + iK
    hpd = (L + S + 0.5 * sp.identity(N)).tocsr()
    # indefinite, well conditioned: diagonal +-4 (first / second half), half-weight off-diagonals:
    # Gershgorin puts the spectrum in [-6.1, -1.9] u [1.9, 6.1]
    sign = np.where(np.arange(N) < N // 2, 1.0, -1.0)
    hind = (sp.diags(4.0 * sign) + 0.5 * (L - 4.0 * sp.identity(N) + S)).tocsr()
    nonh = (L + sp.diags(1j * np.linspace(0.1, 1.0, N)) + 0.2 * sp.diags([np.ones(N - 1)], [1])).tocsr()
    b = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    x0 = 0.1 * (rng.standard_normal(N) + 1j * rng.standard_normal(N))
    U = rng.standard_normal((N, 4)) + 1j * rng.standard_normal((N, 4))
    return dict(L=L, hpd=hpd, hind=hind, nonh=nonh, b=b, x0=x0, U=U, N=N)


def complex_panel(N, k, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, k)) + 1j * rng.standard_normal((N, k))
    w = rng.standard_normal((N, 1)) + 1j * rng.standard_normal((N, 1))
    return X, w
