"""CPU oracle: a NumPy restatement of KryPy's Arnoldi/Lanczos hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``krypy_amd/`` imports this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may use it, and there only as the checker / the reported CPU baseline.

Parity status: PINNED.  Every function below is checked in
``tests/test_oracle_golden.py`` against golden vectors that were produced by
importing the unmodified reference (``/root/reference/krypy``) in the build
container (``oracle/gen_golden.py``, fixtures under ``tests/golden/``), and
against the 18 known-answer scalars of the reference's own
``test/test_convenience_wrappers.py:10-12,37-39``.

Each function cites the reference lines whose arithmetic (operation order,
where sums are accumulated, which quantities are divided vs. multiplied) it
restates.  The code is written function-style on purpose so that it shares
no structure with the product's class-based host layer and can catch host
logic errors there.

All arithmetic is real fp64 with the Euclidean inner product (``ip_B=None``),
the only inner product the device kernels implement in this round.
"""
import math

import numpy as np
import scipy.linalg

__all__ = [
    "drotg",
    "apply_op",
    "residual",
    "arnoldi_step",
    "arnoldi",
    "gmres",
    "restarted_gmres",
    "minres",
    "cg",
    "mgs_qr",
    "Projection",
    "deflated_setup",
    "deflated_gmres",
    "ritz_vectors_smallest",
    "laplace2d",
    "laplace3d",
]


# --------------------------------------------------------------------------
# small dense helpers
# --------------------------------------------------------------------------
def drotg(a, b):
    """Reference BLAS ``drotg`` (what ``utils.Givens`` calls, utils.py:424).

    Returns ``(c, s, r)`` with ``[c s; -s c] [a; b] = [r; 0]`` and the BLAS
    sign convention (sign of ``r`` follows the larger-magnitude input).
    Written out from the netlib level-1 BLAS definition so that it does not
    depend on SciPy's wrapper (which the product uses).
    """
    a = float(a)
    b = float(b)
    roe = a if abs(a) > abs(b) else b
    scale = abs(a) + abs(b)
    if scale == 0.0:
        return 1.0, 0.0, 0.0
    r = scale * math.sqrt((a / scale) ** 2 + (b / scale) ** 2)
    r = math.copysign(1.0, roe) * r
    return a / r, b / r, r


def apply_op(op, x):
    """``op * x`` for None (identity), ndarray, scipy sparse, or callable.
    (utils.py:241-273 get_linearoperator + 1593-1594 MatrixLinearOperator._dot)"""
    if op is None:
        return x
    if callable(op) and not hasattr(op, "dot"):
        return op(x)
    return op.dot(x)


def _nrm(x, Mx=None):
    """utils.norm (utils.py:214-238): Euclidean 2-norm, or sqrt(<x,Mx>)."""
    if Mx is None:
        return float(np.sqrt(np.dot(x, x)))
    ip = float(np.dot(x, Mx))
    if ip < 0.0:
        raise ValueError("inner product <x,Mx> negative: M indefinite?")
    return math.sqrt(ip)


def residual(A, b, z, M=None, Ml=None):
    """LinearSystem.get_residual (linsys.py:130-161).

    Returns ``(MMlr, Mlr, ||MMlr||_{M^-1})`` for ``r = b - A z`` (``z=None``
    means the zero vector, i.e. the cached right-hand-side quantities).
    """
    r = b if z is None else b - apply_op(A, z)
    Mlr = apply_op(Ml, r)
    MMlr = apply_op(M, Mlr)
    nrm = _nrm(Mlr) if M is None else _nrm(Mlr, MMlr)
    return MMlr, Mlr, nrm


# --------------------------------------------------------------------------
# Arnoldi (mgs / dmgs / lanczos), utils.py:854-1081
# --------------------------------------------------------------------------
class _ArnoldiState(object):
    pass


def arnoldi_init(A, v, maxiter, ortho="mgs", M=None, Mv=None, Mv_norm=None):
    """utils.Arnoldi.__init__ (utils.py:855-952), mgs/dmgs/lanczos branches.

    ``V``/``P`` are stored column-contiguous (Fortran order): the values are
    identical to the reference's C-ordered arrays, only the strides differ.
    """
    if ortho not in ("mgs", "dmgs", "lanczos"):
        raise ValueError("oracle restates mgs, dmgs and lanczos only")
    N = v.shape[0]
    st = _ArnoldiState()
    st.A, st.M, st.ortho, st.maxiter = A, M, ortho, maxiter
    st.reorthos = 1 if ortho == "dmgs" else 0
    st.iter = 0
    st.invariant = False
    st.V = np.zeros((N, maxiter + 1), order="F")
    st.P = np.zeros((N, maxiter + 1), order="F") if M is not None else None
    st.H = np.zeros((maxiter + 1, maxiter))
    if M is not None:
        p = v
        v = apply_op(M, p) if Mv is None else Mv
        st.vnorm = _nrm(p, v) if Mv_norm is None else Mv_norm
        if st.vnorm > 0:
            st.P[:, 0] = p / st.vnorm
    else:
        st.vnorm = _nrm(v) if Mv_norm is None else Mv_norm
    if st.vnorm > 0:
        st.V[:, 0] = v / st.vnorm
    else:
        st.invariant = True
    return st


def arnoldi_step(st):
    """utils.Arnoldi.advance (utils.py:954-1048), non-Householder branch.

    Operation order kept: matvec; Lanczos pre-subtraction of the
    ``H[k,k-1]`` term (1000-1009); for each sweep, for each j: alpha=<v_j,Av>,
    ``H[j,k] += alpha``, ``Av -= alpha * (P_j | V_j)`` as multiply-then-
    subtract (1012-1029); norm (1030-1034); invariance test against the
    exact 2-norm of the Hessenberg so far (1035-1039); true division when
    storing the new basis vector (1041-1045).
    """
    if st.iter >= st.maxiter:
        raise ValueError("Maximum number of iterations reached.")
    if st.invariant:
        raise ValueError("Krylov subspace was found to be invariant")
    k = st.iter
    V, P, H = st.V, st.P, st.H
    B = P if st.M is not None else V
    Av = apply_op(st.A, V[:, k])
    start = 0
    if st.ortho == "lanczos":
        start = k
        if k > 0:
            H[k - 1, k] = H[k, k - 1]
            Av = Av - H[k, k - 1] * B[:, k - 1]
    for _ in range(st.reorthos + 1):
        for j in range(start, k + 1):
            alpha = float(np.dot(V[:, j], Av))
            H[j, k] += alpha
            Av = Av - alpha * B[:, j]
    if st.M is not None:
        MAv = apply_op(st.M, Av)
        H[k + 1, k] = _nrm(Av, MAv)
    else:
        H[k + 1, k] = _nrm(Av)
    if H[k + 1, k] / np.linalg.norm(H[: k + 2, : k + 1], 2) <= 1e-14:
        st.invariant = True
    else:
        if st.M is not None:
            P[:, k + 1] = Av / H[k + 1, k]
            V[:, k + 1] = MAv / H[k + 1, k]
        else:
            V[:, k + 1] = Av / H[k + 1, k]
    st.iter += 1


def arnoldi_get(st):
    """utils.Arnoldi.get (utils.py:1050-1061)."""
    k = st.iter
    if st.invariant:
        out = (st.V[:, :k], st.H[:k, :k])
        return out + ((st.P[:, :k],) if st.M is not None else ())
    out = (st.V[:, : k + 1], st.H[: k + 1, :k])
    return out + ((st.P[:, : k + 1],) if st.M is not None else ())


def arnoldi(A, v, maxiter, ortho="mgs", M=None):
    """utils.arnoldi (utils.py:1077-1081)."""
    st = arnoldi_init(A, v, maxiter, ortho=ortho, M=M)
    while st.iter < st.maxiter and not st.invariant:
        arnoldi_step(st)
    return arnoldi_get(st)


# --------------------------------------------------------------------------
# solvers, linsys.py
# --------------------------------------------------------------------------
class Result(dict):
    """Plain attribute dict holding what the reference exposes on a solver."""

    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _finalize_iteration(res, get_xk, yk, resnorm, A, b, M, Ml, bnorm, tol, maxiter,
                        explicit_residual):
    """_KrylovSolver._finalize_iteration (linsys.py:430-493).

    Returns ``rkn`` (explicit residual norm or None); sets ``res.failed`` when
    the reference would raise ConvergenceError in the last iteration.
    """
    res.xk = None
    rkn = None
    if explicit_residual or resnorm / bnorm <= tol or res.iter + 1 == maxiter:
        res.xk = get_xk(yk)
        _, _, rkn = residual(A, b, res.xk, M=M, Ml=Ml)
        res.resnorms.append(rkn / bnorm)
        if res.resnorms[-1] > tol and res.iter + 1 == maxiter:
            res.failed = True
    else:
        res.resnorms.append(resnorm / bnorm)
    return rkn


def gmres(A, b, x0=None, tol=1e-5, maxiter=None, M=None, Ml=None, Mr=None,
          ortho="mgs", explicit_residual=False, proj=None):
    """linsys.Gmres (linsys.py:912-1006) on LinearSystem(A,b,M,Ml,Mr).

    ``proj`` (an oracle ``Deflation``) turns this into DeflatedGmres
    (deflation.py:93-163,276): projected operator, projected initial residual
    and corrected iterates.
    Returns a Result with resnorms, xk, iter, V, H, R, y, converged, failed
    (failed == the reference raises ConvergenceError; xk is still set).
    """
    N = b.shape[0]
    maxiter = N if maxiter is None else maxiter
    res = Result(resnorms=[], iter=0, failed=False, xk=None)
    _, _, bnorm = residual(A, b, None, M=M, Ml=Ml)
    if proj is None:
        MMlr0, Mlr0, r0norm = residual(A, b, x0, M=M, Ml=Ml)
    else:
        MMlr0, Mlr0, r0norm = proj.initial_residual(x0)
    xstart = np.zeros(N) if x0 is None else x0
    res.x0 = xstart
    if bnorm == 0:
        res.xk = np.zeros(N)
        res.resnorms.append(0.0)
        res.converged = True
        return res
    res.resnorms.append(r0norm / bnorm)

    def MlAMr(x):
        y = apply_op(Ml, apply_op(A, apply_op(Mr, x)))
        return y if proj is None else proj.apply_projection(y)

    st = arnoldi_init(MlAMr, Mlr0, maxiter, ortho=ortho, M=M, Mv=MMlr0, Mv_norm=r0norm)
    R = np.zeros((maxiter + 1, maxiter))
    y = np.zeros(maxiter + 1)
    y[0] = r0norm
    G = []

    def get_xk(yk):
        # Gmres._get_xk (linsys.py:941-949)
        k = st.iter
        if yk is None or k == 0:
            xk = xstart
        else:
            yy = scipy.linalg.solve_triangular(R[:k, :k], yk)
            xk = xstart + apply_op(Mr, st.V[:, :k].dot(yy))
        return xk if proj is None else proj.correct(xk)

    while res.resnorms[-1] > tol and st.iter < st.maxiter and not st.invariant:
        k = res.iter = st.iter
        arnoldi_step(st)
        R[: k + 2, k] = st.H[: k + 2, k]
        for i in range(k):
            c, s = G[i]
            t0, t1 = R[i, k], R[i + 1, k]
            R[i, k], R[i + 1, k] = c * t0 + s * t1, -s * t0 + c * t1
        c, s, _ = drotg(R[k, k], R[k + 1, k])
        G.append((c, s))
        t0, t1 = R[k, k], R[k + 1, k]
        R[k, k], R[k + 1, k] = c * t0 + s * t1, -s * t0 + c * t1
        t0, t1 = y[k], y[k + 1]
        y[k], y[k + 1] = c * t0 + s * t1, -s * t0 + c * t1
        _finalize_iteration(res, get_xk, y[: k + 1], abs(y[k + 1]), A, b, M, Ml, bnorm,
                            tol, maxiter, explicit_residual)
        if res.failed:
            break
    if res.xk is None:
        res.xk = get_xk(y[: st.iter])
    res.V, res.H = arnoldi_get(st)[:2]
    res.P = arnoldi_get(st)[2] if M is not None else None
    res.R, res.y = R, y
    res.invariant = st.invariant
    res.converged = res.resnorms[-1] <= tol
    return res


def restarted_gmres(A, b, x0=None, tol=1e-5, maxiter=None, max_restarts=0, **kw):
    """linsys._RestartedSolver / RestartedGmres (linsys.py:1021-1081).

    Returns a Result with the spliced ``resnorms`` (the last entry of a cycle is
    replaced by the first of the next, 1061-1062), ``xk``, ``cycles`` (the
    per-cycle Results, for closed-loop parity checks) and ``failed``.
    """
    out = Result(resnorms=[np.inf], xk=None, cycles=[], failed=False)
    restart = 0
    while restart == 0 or (out.resnorms[-1] > tol and restart <= max_restarts):
        x_in = x0 if out.xk is None else out.xk
        sol = gmres(A, b, x0=x_in, tol=tol, maxiter=maxiter, **kw)
        out.cycles.append(sol)
        out.xk = sol.xk
        del out.resnorms[-1]
        out.resnorms += sol.resnorms
        restart += 1
    out.failed = out.resnorms[-1] > tol
    return out


def minres(A, b, x0=None, tol=1e-5, maxiter=None, M=None, Ml=None, Mr=None,
           ortho="lanczos", explicit_residual=False):
    """linsys.Minres (linsys.py:757-862).

    Lanczos via the Arnoldi restatement above, the 4x1 ``R`` QR update with the
    two remembered rotations (826-841) and the ``z``/``W``/``yk`` recurrences
    (844-847), in the reference's operation order.
    """
    N = b.shape[0]
    maxiter = N if maxiter is None else maxiter
    res = Result(resnorms=[], iter=0, failed=False, xk=None)
    _, _, bnorm = residual(A, b, None, M=M, Ml=Ml)
    MMlr0, Mlr0, r0norm = residual(A, b, x0, M=M, Ml=Ml)
    xstart = np.zeros(N) if x0 is None else x0
    if bnorm == 0:
        res.xk = np.zeros(N)
        res.resnorms.append(0.0)
        res.converged = True
        return res
    res.resnorms.append(r0norm / bnorm)

    def MlAMr(x):
        return apply_op(Ml, apply_op(A, apply_op(Mr, x)))

    st = arnoldi_init(MlAMr, Mlr0, maxiter, ortho=ortho, M=M, Mv=MMlr0, Mv_norm=r0norm)
    W0 = np.zeros(N)
    W1 = np.zeros(N)
    y = [r0norm, 0.0]
    G1 = G2 = None
    yk = np.zeros(N)

    def get_xk(yk_):
        return xstart if yk_ is None else xstart + apply_op(Mr, yk_)

    def rot(G, u, v):
        c, s = G
        return c * u + s * v, -s * u + c * v

    while res.resnorms[-1] > tol and st.iter < st.maxiter and not st.invariant:
        k = res.iter = st.iter
        arnoldi_step(st)
        H = st.H
        R = [0.0, 0.0, 0.0, 0.0]
        R[1] = H[k - 1, k]  # k == 0 reads H[-1, 0] like the reference (828): zero
        if G1 is not None:
            R[0], R[1] = rot(G1, R[0], R[1])
        R[2], R[3] = H[k, k], H[k + 1, k]
        if G2 is not None:
            R[1], R[2] = rot(G2, R[1], R[2])
        G1 = G2
        c, s, r = drotg(R[2], R[3])
        G2 = (c, s)
        R[2] = c * R[2] + s * R[3]  # Givens.r (utils.py:431)
        R[3] = 0.0
        y = list(rot(G2, y[0], y[1]))
        z = (st.V[:, k] - R[0] * W0 - R[1] * W1) / R[2]
        W0, W1 = W1, z
        yk = yk + y[0] * z
        y = [y[1], 0.0]
        _finalize_iteration(res, get_xk, yk, abs(y[0]), A, b, M, Ml, bnorm, tol, maxiter,
                            explicit_residual)
        if res.failed:
            break
    if res.xk is None:
        res.xk = get_xk(yk)
    got = arnoldi_get(st)
    res.V, res.H = got[0], got[1]
    res.P = got[2] if M is not None else None
    res.converged = res.resnorms[-1] <= tol
    return res


def cg(A, b, x0=None, tol=1e-5, maxiter=None, M=None, Ml=None, Mr=None,
       explicit_residual=False):
    """linsys.Cg._solve (linsys.py:593-689), without the store_arnoldi extras."""
    N = b.shape[0]
    maxiter = N if maxiter is None else maxiter
    res = Result(resnorms=[], iter=0, failed=False, xk=None)
    _, _, bnorm = residual(A, b, None, M=M, Ml=Ml)
    MMlr0, Mlr0, r0norm = residual(A, b, x0, M=M, Ml=Ml)
    xstart = np.zeros(N) if x0 is None else x0
    if bnorm == 0:
        res.xk = np.zeros(N)
        res.resnorms.append(0.0)
        res.converged = True
        return res
    res.resnorms.append(r0norm / bnorm)

    def get_xk(yk_):
        return xstart if yk_ is None else xstart + apply_op(Mr, yk_)

    yk = np.zeros(N)
    rhos = [r0norm ** 2]
    Mlrk = Mlr0.copy()
    MMlrk = MMlr0.copy()
    p = MMlrk.copy()
    while res.resnorms[-1] > tol and res.iter < maxiter:
        k = res.iter
        if k > 0:
            p = MMlrk + rhos[-1] / rhos[-2] * p
        Ap = apply_op(Ml, apply_op(A, apply_op(Mr, p)))
        alpha = rhos[-1] / float(np.dot(p, Ap))
        yk = yk + alpha * p
        Mlrk = Mlrk - alpha * Ap
        MMlrk = apply_op(M, Mlrk)
        nrm = _nrm(Mlrk) if M is None else _nrm(Mlrk, MMlrk)
        rhos.append(nrm ** 2)
        rkn = _finalize_iteration(res, get_xk, yk, nrm, A, b, M, Ml, bnorm, tol, maxiter,
                                  explicit_residual)
        if rkn is not None:
            rhos[-1] = rkn ** 2
        if res.failed:
            break
        res.iter += 1
    if res.xk is None:
        res.xk = get_xk(yk)
    res.rhos = rhos
    res.converged = res.resnorms[-1] <= tol
    return res


# --------------------------------------------------------------------------
# projections / deflation, utils.py:439-707, deflation.py:32-189
# --------------------------------------------------------------------------
def mgs_qr(X, reorthos=1):
    """utils.qr, modified Gram-Schmidt branch (utils.py:694-707).

    This is the branch the deflated solvers take: they pass an
    IdentityLinearOperator *instance* as ip_B, so ``ip_B is None`` is False
    (deflation.py:40, utils.py:692).
    """
    N, k = X.shape
    Q = np.array(X, dtype=float, order="F", copy=True)
    R = np.zeros((k, k))
    for i in range(k):
        for _ in range(reorthos + 1):
            for j in range(i):
                alpha = float(np.dot(Q[:, j], Q[:, i]))
                R[j, i] += alpha
                Q[:, i] -= alpha * Q[:, j]
        R[i, i] = _nrm(Q[:, i])
        if R[i, i] >= 1e-15:
            Q[:, i] /= R[i, i]
    return Q, R


class Projection(object):
    """utils.Projection with orthogonalize=True, Y given (utils.py:440-552,604-627)."""

    def __init__(self, X, Y, iterations=2, qr_reorthos=1):
        self.iterations = iterations
        self.k = X.shape[1]
        if self.k == 0:
            return
        self.V, self.VR = mgs_qr(X, reorthos=qr_reorthos)
        self.W, self.WR = mgs_qr(Y, reorthos=qr_reorthos)
        Mm = self.W.T.dot(self.V)
        self.Q, self.R = scipy.linalg.qr(Mm)

    def _apply(self, a, return_Ya=False):
        c = self.W.T.dot(a)
        Ya = self.WR.T.dot(c) if return_Ya else None
        c = scipy.linalg.solve_triangular(self.R, self.Q.T.dot(c))
        Pa = self.V.dot(c)
        return (Pa, Ya) if return_Ya else Pa

    def apply_complement(self, a, return_Ya=False):
        if self.k == 0:
            return (a.copy(), np.zeros(0)) if return_Ya else a.copy()
        if return_Ya:
            x, Ya = self._apply(a, True)
        else:
            x = self._apply(a)
        z = a - x
        for _ in range(self.iterations - 1):
            z = z - self._apply(z)
        return (z, Ya) if return_Ya else z


class Deflation(object):
    """deflation.ObliqueProjection + the _DeflationMixin hooks
    (deflation.py:32-76, 93-163), for M = identity or a given M with Minv=None
    only when M is None (the configs in scope)."""

    def __init__(self, A, b, U, M=None, Ml=None, Mr=None, qr_reorthos=0, iterations=2):
        self.A, self.b, self.M, self.Ml, self.Mr = A, b, M, Ml, Mr
        if M is not None:
            raise ValueError("oracle deflation restates the M=None case")
        # deflation.py:40 (MGS with reorthos=qr_reorthos, default 0)
        self.U, _ = mgs_qr(U, reorthos=qr_reorthos)
        d = self.U.shape[1]
        # deflation.py:47
        self.AU = np.zeros_like(self.U)
        for j in range(d):
            self.AU[:, j] = apply_op(Ml, apply_op(A, apply_op(Mr, self.U[:, j])))
        self.proj = Projection(self.AU, self.U, iterations=iterations)
        # deflation.py:104-111
        if d == 0:
            self.E = np.zeros((0, 0))
        else:
            E = self.proj.Q.dot(self.proj.R)
            self.E = self.proj.WR.T.dot(E.dot(self.proj.VR))
        self.C = np.zeros((d, 0))
        self.UMlr = None

    def apply_projection(self, Av):
        PAv, UAv = self.proj.apply_complement(Av, return_Ya=True)
        self.C = np.column_stack([self.C, UAv])
        return PAv

    def initial_residual(self, x0):
        r = self.b if x0 is None else self.b - apply_op(self.A, x0)
        Mlr = apply_op(self.Ml, r)
        PMlr, self.UMlr = self.proj.apply_complement(Mlr, return_Ya=True)
        return PMlr, PMlr, _nrm(PMlr)

    def correct(self, z):
        p = self.proj
        if p.k == 0:
            return z
        c = apply_op(self.Ml, self.b - apply_op(self.A, z))
        c = p.W.T.dot(c)
        c = scipy.linalg.solve_triangular(p.R, p.Q.T.dot(c))
        c = p.WR.dot(scipy.linalg.solve_triangular(p.VR, c))
        return z + p.W.dot(c)


def deflated_gmres(A, b, U, x0=None, tol=1e-5, maxiter=None, **kw):
    """deflation.DeflatedGmres (deflation.py:276) = gmres with a Deflation."""
    defl = Deflation(A, b, U, Ml=kw.get("Ml"), Mr=kw.get("Mr"))
    res = gmres(A, b, x0=x0, tol=tol, maxiter=maxiter, proj=defl, **kw)
    res.E, res.C, res.UMlr, res.U, res.AU = defl.E, defl.C, defl.UMlr, defl.U, defl.AU
    return res


def ritz_vectors_smallest(res, n_vectors, self_adjoint=False):
    """deflation.Ritz (mode 'ritz', deflation.py:738-847) + the 'sm' selection of
    recycling.factories.RitzFactorySimple (factories.py:167-194): the
    ``n_vectors`` Ritz vectors of smallest-magnitude Ritz value of the deflated
    Arnoldi relation held in ``res`` (from :func:`deflated_gmres` or
    :func:`gmres` with no deflation)."""
    V, H_ = res.V, res.H
    n_, n = H_.shape
    H = H_[:n, :n]
    U = res.get("U")
    d = 0 if U is None else U.shape[1]
    if d > 0:
        E, C, AU = res.E, res.C, res.AU
        # _DeflationMixin.B_ (deflation.py:165-189)
        if self_adjoint:
            B_ = C.T
            if n_ > n:
                B_ = np.vstack([B_, V[:, -1].dot(AU)])
        else:
            B_ = V.T.dot(AU)
        B = B_[:n, :]
        EinvC = np.linalg.solve(E, C)  # deflation.py:774
        M = np.block([[H + B.dot(EinvC), B], [C, E]])  # deflation.py:781
    else:
        M = H
    eig = scipy.linalg.eigh if self_adjoint else scipy.linalg.eig  # deflation.py:792
    vals, coeffs = eig(M)
    for i in range(n + d):  # deflation.py:812-813
        coeffs[:, i] /= np.linalg.norm(coeffs[:, i], 2)
    order = np.argsort(np.abs(vals))[:n_vectors]  # factories.py:176
    basis = V[:, :n] if d == 0 else np.column_stack([V[:, :n], U])
    return vals[order], basis.dot(coeffs[:, order])  # deflation.py:840-847


# --------------------------------------------------------------------------
# synthetic matrices of SURVEY.md section 8(d)
# --------------------------------------------------------------------------
def laplace2d(nx, ny=None):
    """5-point Laplacian ``kron(I_ny,T_nx)+kron(T_ny,I_nx)``, sorted int32 CSR."""
    import scipy.sparse as sp

    ny = nx if ny is None else ny

    def T(n):
        return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])

    A = sp.kron(sp.identity(ny), T(nx)) + sp.kron(T(ny), sp.identity(nx))
    A = sp.csr_matrix(A)
    A.sort_indices()
    return A


def laplace3d(nx, ny=None, nz=None):
    """7-point Laplacian on an nx*ny*nz grid (x fastest), sorted int32 CSR."""
    import scipy.sparse as sp

    ny = nx if ny is None else ny
    nz = nx if nz is None else nz

    def T(n):
        return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])

    Ix, Iy, Iz = sp.identity(nx), sp.identity(ny), sp.identity(nz)
    A = sp.kron(Iz, sp.kron(Iy, T(nx))) + sp.kron(Iz, sp.kron(T(ny), Ix)) + sp.kron(
        T(nz), sp.kron(Iy, Ix))
    A = sp.csr_matrix(A)
    A.sort_indices()
    return A
