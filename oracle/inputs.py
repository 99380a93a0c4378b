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


def dense_spd_system_blocked(n, rows=4096):
    """dense_spd_system(n) with G G^T taken in row blocks: the same matrix to the rounding of the dgemm blocking.  (The one-shot
    product at n = 32768 brought the build container's OpenBLAS down twice - a segfault in a dgemm copy kernel; the full-size
    fixture of config 4 and everything compared with it build their matrix this way.)"""
    rng = np.random.default_rng(0)
    G = rng.standard_normal((n, n))
    A = np.empty((n, n))
    for i0 in range(0, n, rows):
        A[i0:i0 + rows] = G[i0:i0 + rows].dot(G.T)
    del G
    A /= n
    A[np.diag_indices(n)] += 1.0
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


# ---------------------------------------------------------------------------------------------
# the reference's solver test matrix (test/test_linsys.py:50-143, test/test_utils.py:14-58)
# ---------------------------------------------------------------------------------------------
def zoo_matrices():
    """The six 10 x 10 matrices of the reference's tests (test_utils.py:14-58) with their flags."""
    spd = np.linspace(1, 2, 10)
    spd[-1] = 1e-2
    hpd = np.array(np.linspace(1, 2, 10), dtype=complex)
    hpd[0], hpd[-1] = 5, 1e-1
    Ahpd = np.diag(hpd)
    Ahpd[-1, 0], Ahpd[0, -1] = 1e-1j, -1e-1j
    ind = np.linspace(1, 2, 10)
    ind[-1] = -1
    hind = np.array(np.linspace(1, 2, 10), dtype=complex)
    hind[-1] = 1e-3
    Ahind = np.diag(hind)
    Ahind[-1, 0], Ahind[0, -1] = 10j, -10j
    non = np.diag(np.arange(1, 11, dtype=float))
    non[-1, -1], non[0, -1] = -1e1, 1e1
    cnon = np.diag(np.arange(1, 11, dtype=complex))
    cnon[-1, -1], cnon[0, -1] = -1e1, 1.0e1j
    pd = dict(normal=True, self_adjoint=True, positive_definite=True)
    sa = dict(normal=True, self_adjoint=True)
    return [("spd", np.diag(spd), pd), ("hpd", Ahpd, pd), ("symm_indef", np.diag(ind), sa),
            ("herm_indef", Ahind, sa), ("nonsymm", non, {}), ("comp_nonsymm", cnon, {})]


def run_solver_matrix(linsys, ConvergenceError, visit, stride=1, offset=0, perturb=0.0):
    """Drive ``linsys`` (the reference's module or krypy_amd's) through the reference's solver test
    matrix - with the preconditioners really passed to LinearSystem, which the reference's own
    generator forgets (test_linsys.py:91-97 yields **ls_kwargs) - and call
    ``visit(idx, name, Solver, ls, params, sol, failed, A, B, M, Ml)`` for every solve.  Solve
    ``idx`` is carried out only if ``idx % stride == offset``; returns the total number of cases.
    ``perturb`` scales entry i of every right-hand side by ``1 + perturb*i`` (the fixture generator
    uses 1e-15 to find the solves whose outcome is rounding-chaotic in the reference itself)."""
    import itertools
    import warnings

    idx = 0
    xs = [np.ones((10, 1)), np.ones((10,)), (1 + 1j) * np.ones((10, 1)), np.zeros((10, 1))]
    for name, A0, flags in zoo_matrices():
        for B, x in itertools.product([None, np.diag(np.arange(1.0, 11.0))], xs):
            A = np.linalg.inv(B).dot(A0) if (B is not None and flags.get("self_adjoint")) else A0
            Ainv = np.linalg.inv(A)
            Ms, Mls, Mrs = [None], [None, Ainv], [None, Ainv]
            if flags.get("positive_definite"):
                Ms.append(Ainv)
            if np.linalg.norm(np.diag(np.diag(A)) - A) == 0 and B is None:
                Ms.append(np.diag(np.linspace(1, 10, 10)))
            for exact in (None, x):
                for M, Ml, Mr in itertools.product(Ms, Mls, Mrs):
                    kw = dict(flags)
                    kw.update(M=M, Ml=Ml, Mr=Mr)
                    if M is not None:
                        kw["Minv"] = np.linalg.inv(M)
                    ls = None
                    solvers = ["Gmres", "RestartedGmres"]
                    if flags.get("self_adjoint"):
                        solvers.append("Minres")
                    if flags.get("positive_definite"):
                        solvers.append("Cg")
                    b = A.dot(x)
                    if perturb:
                        b = b * (1.0 + perturb * np.arange(10).reshape(b.shape))
                    x0s = [None, np.zeros(b.shape), np.ones(b.shape)]
                    if exact is not None:
                        x0s.append(exact)
                    for sname, x0, tol in itertools.product(solvers, x0s, (1e-13, 1e-2)):
                        if idx % stride == offset:
                            if ls is None:
                                ls = linsys.LinearSystem(A, b, ip_B=B, exact_solution=exact, **kw)
                            Solver = getattr(linsys, sname)
                            params = dict(x0=x0, tol=tol, maxiter=15)
                            if sname == "RestartedGmres":
                                params.update(maxiter=7, max_restarts=20)
                            failed = False
                            with warnings.catch_warnings():
                                warnings.simplefilter("ignore")
                                try:
                                    sol = Solver(ls, **params)
                                except ConvergenceError as e:
                                    sol, failed = e.solver, True
                            visit(idx, name, Solver, ls, params, sol, failed, A, B, M, Ml)
                        idx += 1
    return idx


def run_deflation_matrix(linsys, deflation, ConvergenceError, visit):
    """The reference's deflated-solver test matrix (test/test_deflation.py:13-41): the six matrices
    x inner products x right-hand sides x exact solution x U in {None, e_1, e_1 + 1e-3} x
    Deflated{Gmres,Minres,Cg}, tol 1e-6, maxiter 15, store_arnoldi.  (Unpreconditioned, exactly as
    the reference's generator produces them.)  Calls ``visit(idx, name, Solver, ls, sol, failed, A, B)``;
    returns the number of cases."""
    import itertools
    import warnings

    idx = 0
    xs = [np.ones((10, 1)), np.ones((10,)), (1 + 1j) * np.ones((10, 1)), np.zeros((10, 1))]
    Us = [None, np.eye(10, 1), np.eye(10, 1) + 1e-3 * np.ones((10, 1))]
    for name, A0, flags in zoo_matrices():
        for B, x in itertools.product([None, np.diag(np.arange(1.0, 11.0))], xs):
            A = np.linalg.inv(B).dot(A0) if (B is not None and flags.get("self_adjoint")) else A0
            for exact in (None, x):
                ls = linsys.LinearSystem(A, A.dot(x), ip_B=B, exact_solution=exact, **flags)
                solvers = ["DeflatedGmres"]
                if flags.get("self_adjoint"):
                    solvers.append("DeflatedMinres")
                if flags.get("positive_definite"):
                    solvers.append("DeflatedCg")
                for U, sname in itertools.product(Us, solvers):
                    Solver = getattr(deflation, sname)
                    failed = False
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        try:
                            sol = Solver(ls, U=U, x0=None, tol=1e-6, maxiter=15, store_arnoldi=True)
                        except ConvergenceError as e:
                            sol, failed = e.solver, True
                    visit(idx, name, Solver, ls, sol, failed, A, B)
                    idx += 1
    return idx


# ---------------------------------------------------------------------------------------------
# edge cases of the solver API: what the reference does with degenerate input (zero right-hand side, exact initial
# guess, maxiter 0, tol 0, wrong shapes, a rank-deficient deflation basis ...).  Each scenario takes the package
# (`krypy` for the fixture, `krypy_amd` in the tests) and returns a solver; run_edge_scenarios reduces the outcome
# to (status, message, len(resnorms), resnorms[-1], ||xk||).
# ---------------------------------------------------------------------------------------------
def edge_scenarios():
    A = laplace2d(8).tocsr()
    N = A.shape[0]
    b = np.arange(1.0, N + 1.0)
    xex = np.linalg.solve(A.toarray(), b)
    sing = sp.diags(np.r_[0.0, np.ones(N - 1)]).tocsr()
    spd = dict(self_adjoint=True, positive_definite=True)
    return [
        ("zero rhs gmres", lambda m: m.linsys.Gmres(m.linsys.LinearSystem(A, np.zeros(N)))),
        ("zero rhs minres", lambda m: m.linsys.Minres(m.linsys.LinearSystem(A, np.zeros(N), self_adjoint=True))),
        ("zero rhs cg", lambda m: m.linsys.Cg(m.linsys.LinearSystem(A, np.zeros(N), **spd))),
        ("x0 exact", lambda m: m.linsys.Gmres(m.linsys.LinearSystem(A, b), x0=xex.reshape(-1, 1), tol=1e-8)),
        ("maxiter 0", lambda m: m.linsys.Gmres(m.linsys.LinearSystem(A, b), maxiter=0)),
        ("maxiter 0 cg", lambda m: m.linsys.Cg(m.linsys.LinearSystem(A, b, **spd), maxiter=0)),
        ("maxiter beyond N", lambda m: m.linsys.Gmres(m.linsys.LinearSystem(A, b), maxiter=5 * N, tol=1e-12)),
        ("tol 0", lambda m: m.linsys.Minres(m.linsys.LinearSystem(A, b, self_adjoint=True), maxiter=10, tol=0.0)),
        ("negative tol", lambda m: m.linsys.Gmres(m.linsys.LinearSystem(A, b), tol=-1.0, maxiter=5)),
        ("singular consistent", lambda m: m.linsys.Minres(
            m.linsys.LinearSystem(sing, np.r_[0.0, np.ones(N - 1)], self_adjoint=True), tol=1e-10)),
        ("b wrong length", lambda m: m.linsys.Gmres(m.linsys.LinearSystem(A, np.ones(N + 1)))),
        ("restart fails", lambda m: m.linsys.RestartedGmres(m.linsys.LinearSystem(A, b), maxiter=3, max_restarts=0,
                                                            tol=1e-12)),
        ("restart succeeds", lambda m: m.linsys.RestartedGmres(m.linsys.LinearSystem(A, b), maxiter=5, max_restarts=40,
                                                               tol=1e-8)),
        ("U rank deficient", lambda m: m.deflation.DeflatedGmres(m.linsys.LinearSystem(A, b), U=np.ones((N, 2)),
                                                                 tol=1e-8)),
        ("U spans the solution", lambda m: m.deflation.DeflatedMinres(
            m.linsys.LinearSystem(A, b, self_adjoint=True), U=xex.reshape(-1, 1), tol=1e-8)),
        ("U empty", lambda m: m.deflation.DeflatedCg(m.linsys.LinearSystem(A, b, **spd), U=np.zeros((N, 0)), tol=1e-8)),
        ("explicit residual", lambda m: m.linsys.Cg(m.linsys.LinearSystem(A, b, **spd), tol=1e-8,
                                                    explicit_residual=True)),
        ("exact_solution errnorms", lambda m: m.linsys.Gmres(
            m.linsys.LinearSystem(A, b, exact_solution=xex.reshape(-1, 1)), tol=1e-8)),
    ]


def run_edge_scenarios(pkg):
    import warnings
    out = []
    for name, fn in edge_scenarios():
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                sol = fn(pkg)
            extra = float(sol.errnorms[-1]) if getattr(sol, "errnorms", None) is not None and len(sol.errnorms) else -1.0
            out.append((name, "ok", "", len(sol.resnorms), float(sol.resnorms[-1]), float(np.linalg.norm(sol.xk)), extra))
        except Exception as e:      # noqa: BLE001  (the point is to record what is raised)
            sol = getattr(e, "solver", None)
            n = len(sol.resnorms) if sol is not None else -1
            out.append((name, type(e).__name__, str(e), n, float(sol.resnorms[-1]) if n > 0 else -1.0,
                        float(np.linalg.norm(sol.xk)) if sol is not None and getattr(sol, "xk", None) is not None else -1.0,
                        -1.0))
    return out
