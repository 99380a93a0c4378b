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
