"""Linear systems and Krylov solvers (CG, MINRES, GMRES, restarted GMRES) on the MI355X.

Host-side mirror of ``krypy/linsys.py``: identical class names, constructor signatures,
attributes and error behaviour; the solve still happens inside the constructor and a
solver object *is* the result.  All N-vectors stay in HBM between iterations; attributes
that the reference exposes as ndarrays (``xk``, ``x0``, ``MMlr0``, ``Mlr0``, ``V``, ``P``,
``LinearSystem.Mlb`` ...) are downloaded lazily on first access.  The Hessenberg matrix,
Givens rotations and the small triangular solves stay on the host like in the reference.

Reference lines are cited as ``linsys.py:<line>`` (= ``/root/reference/krypy/linsys.py``).
"""
import collections
import os
import warnings

import numpy
import scipy.linalg

from . import _hip, utils
from .utils import DVec

__all__ = ["LinearSystem", "TimedLinearSystem", "ConvertedTimedLinearSystem", "Cg", "Minres", "Gmres",
           "RestartedGmres"]


class _LazyHost(object):
    """Descriptor: attribute stored as a device vector, presented as an ``(N,1)`` ndarray."""

    def __init__(self, name):
        self.dev = "_" + name + "_dev"
        self.host = "_" + name + "_host"

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        h = obj.__dict__.get(self.host)
        if h is None:
            d = obj.__dict__.get(self.dev)
            if d is None:
                return None
            h = obj.__dict__[self.host] = d.download()
        return h

    def __set__(self, obj, value):
        if isinstance(value, DVec):
            obj.__dict__[self.dev] = value
            obj.__dict__[self.host] = None
        else:
            obj.__dict__[self.host] = value
            obj.__dict__[self.dev] = None


def _pyscalar(x):
    """numpy scalar -> Python float, or complex when it has an imaginary part"""
    x = complex(x)
    return x.real if x.imag == 0.0 else x


def _is_set(obj, name):
    """True if a _LazyHost attribute holds a value (device or host) - without downloading it."""
    return obj.__dict__.get("_" + name + "_dev") is not None or obj.__dict__.get("_" + name + "_host") is not None


def _dev_of(obj, name, ctx):
    """Device view of a _LazyHost attribute (uploads a host-assigned value once, in the owner's
    block dtype ``_bdt`` when it has one)."""
    d = obj.__dict__.get("_" + name + "_dev")
    if d is None:
        h = obj.__dict__.get("_" + name + "_host")
        if h is None:
            return None
        d = obj.__dict__["_" + name + "_dev"] = DVec.from_host(h, ctx, dtype=obj.__dict__.get("_bdt"))
    return d


class LinearSystem(object):
    Mlb = _LazyHost("Mlb")
    MMlb = _LazyHost("MMlb")

    def __init__(self, A, b, M=None, Minv=None, Ml=None, Mr=None, ip_B=None, normal=None,
                 self_adjoint=False, positive_definite=False, exact_solution=None):
        r"""Representation of a (preconditioned) linear system (linsys.py:12-128).

        .. math:: M M_l A M_r y = M M_l b \quad\text{with}\quad x = M_r y

        Same parameters as the reference.  ``A``, ``M``, ``Minv``, ``Ml``, ``Mr`` given as
        ndarray / SciPy sparse are uploaded once and applied by HIP kernels (CSR SpMV, dense
        GEMV, diagonal scaling); ``LinearOperator`` callables work through a host round trip.
        ``b`` is uploaded once and stays resident.
        """
        self.N = N = len(b)
        shape = (N, N)
        self.A = utils.get_linearoperator(shape, A)
        self.M = utils.get_linearoperator(shape, M)
        self.Minv = utils.get_linearoperator(shape, Minv)
        self.Ml = utils.get_linearoperator(shape, Ml)
        self.Mr = utils.get_linearoperator(shape, Mr)
        self.MlAMr = self.Ml * self.A * self.Mr
        try:
            self.ip_B = utils.get_linearoperator(shape, ip_B)
        except TypeError:
            self.ip_B = ip_B

        self.flat_vecs, (self.b, self.exact_solution) = utils.shape_vecs(b, exact_solution)

        self.self_adjoint = self_adjoint
        if self_adjoint:
            if normal is not None and not normal:
                warnings.warn("Setting normal=True because self_adjoint=True is provided.")
            normal = True
        if normal is None:
            normal = False
        self.normal = normal
        self.positive_definite = positive_definite
        if self_adjoint and not normal:
            raise utils.ArgumentError("self-adjointness implies normality")

        self.dtype = utils.find_common_dtype(self.A, self.b, self.M, self.Ml, self.Mr, self.ip_B)
        self._bdt = utils._bdt(self.dtype)     # float64 or complex128 device blocks

        # device-resident right hand side and its preconditioned forms (linsys.py:119-122)
        self._ctx = _hip.get_context()
        self._b_dev = DVec.from_host(self.b, self._ctx, dtype=self._bdt)
        self._plain = all(isinstance(op, utils.IdentityLinearOperator) for op in (self.M, self.Ml))
        self.Mlb = self.Ml * self._b_dev
        self.MMlb = self.M * _dev_of(self, "Mlb", self._ctx)
        self.MMlb_norm = utils.norm(_dev_of(self, "Mlb", self._ctx),
                                    _dev_of(self, "MMlb", self._ctx), ip_B=self.ip_B)

    # ---- device side ----------------------------------------------------------------
    def _residual_dev(self, z, compute_norm=False):
        """``(MMlr, Mlr[, norm])`` as device vectors for a device ``z`` (None = zero vector)."""
        ctx = self._ctx
        if z is None:
            out = (_dev_of(self, "MMlb", ctx), _dev_of(self, "Mlb", ctx))
            return out + (self.MMlb_norm,) if compute_norm else out
        dt = utils._bdt(self._bdt, z.dtype)      # a complex guess makes a real system complex
        z = z.astype(dt)
        b = self._b_dev.astype(dt)
        Amat = self.A._device_matrix(ctx, dt)
        euclid = self.ip_B is None or isinstance(self.ip_B, utils.IdentityLinearOperator)
        if self._plain and euclid and Amat is not None and compute_norm:
            # fused r = b - A z and ||r|| (one pass over the matrix, no extra vector pass)
            r = DVec(ctx.alloc(self.N, 1, dtype=dt))
            nrm = ctx.residual(Amat, b.block, b.col, z.block, z.col, r.block, 0)
            return r, r, nrm
        Az = (self.A * z).astype(dt)
        r = DVec(ctx.alloc(self.N, 1, dtype=dt))
        ctx.waxpby(r.block, 0, 1.0, b.block, b.col, -1.0, Az.block, Az.col)
        Mlr = self.Ml * r
        MMlr = self.M * Mlr
        if compute_norm:
            return MMlr, Mlr, utils.norm(Mlr, MMlr, ip_B=self.ip_B)
        return MMlr, Mlr

    # ---- reference API ----------------------------------------------------------------
    def get_residual(self, z, compute_norm=False):
        r"""Residual :math:`r = M M_l (b - A z)` and optionally
        :math:`\|M M_l (b-Az)\|_{M^{-1}}` (linsys.py:130-161); ``z`` is an ``(N,1)`` array
        (or ``None`` for the cached right-hand-side quantities); returns ndarrays."""
        if z is None:
            if compute_norm:
                return self.MMlb, self.Mlb, self.MMlb_norm
            return self.MMlb, self.Mlb
        zd = z if isinstance(z, DVec) else DVec.from_host(z, self._ctx, dtype=self._bdt)
        res = self._residual_dev(zd, compute_norm)
        out = (res[0].download(), res[1].download())
        return out + (res[2],) if compute_norm else out

    def get_ip_Minv_B(self):
        """Inner product implicitly used with the preconditioner ``M`` (linsys.py:163-176)."""
        if not isinstance(self.M, utils.IdentityLinearOperator):
            if isinstance(self.Minv, utils.IdentityLinearOperator):
                raise utils.ArgumentError(
                    "Minv has to be provided for the evaluation of the inner "
                    "product that is implicitly defined by M.")
            if isinstance(self.ip_B, utils.LinearOperator):
                return self.Minv * self.ip_B
            else:
                return lambda x, y: self.ip_B(x, self.Minv * y)
        return self.ip_B

    def __repr__(self):
        ret = "LinearSystem {\n"
        for k in ["A", "b", "M", "Minv", "Ml", "Mr", "ip_B", "normal", "self_adjoint",
                  "positive_definite", "exact_solution"]:
            op = self.__dict__.get(k)
            if op is not None and not isinstance(op, utils.IdentityLinearOperator):
                ret += "  " + k + ": " + op.__repr__() + "\n"
        return ret + "}"


class TimedLinearSystem(LinearSystem):
    def __init__(self, A, b, M=None, Minv=None, Ml=None, Mr=None, ip_B=None, normal=None,
                 self_adjoint=False, positive_definite=False, exact_solution=None):
        """LinearSystem whose operators (and a callable inner product) are timed
        (linsys.py:204-252); ``timings`` is a :class:`~krypy_amd.utils.Timings`."""
        self.timings = utils.Timings()
        N = len(b)
        shape = (N, N)
        try:
            _ip_B = utils.get_linearoperator(shape, ip_B, timer=self.timings["ip_B"])
        except TypeError:

            def _ip_B(X, Y):
                (_, m) = X.shape
                (_, n) = Y.shape
                if m == 0 or n == 0:
                    return ip_B(X, Y)
                with self.timings["ip_B"]:
                    ret = ip_B(X, Y)
                self.timings["ip_B"][-1] /= m * n
                return ret

        super(TimedLinearSystem, self).__init__(
            A=utils.get_linearoperator(shape, A, self.timings["A"]), b=b,
            M=utils.get_linearoperator(shape, M, self.timings["M"]),
            Minv=utils.get_linearoperator(shape, Minv, self.timings["Minv"]),
            Ml=utils.get_linearoperator(shape, Ml, self.timings["Ml"]),
            Mr=utils.get_linearoperator(shape, Mr, self.timings["Mr"]),
            ip_B=_ip_B, normal=normal, self_adjoint=self_adjoint,
            positive_definite=positive_definite, exact_solution=exact_solution)


class ConvertedTimedLinearSystem(TimedLinearSystem):
    def __init__(self, linear_system):
        """Timed copy of an existing LinearSystem (linsys.py:255-274)."""
        kwargs = {k: linear_system.__dict__[k]
                  for k in ["A", "b", "M", "Minv", "Ml", "Mr", "ip_B", "normal", "self_adjoint",
                            "positive_definite", "exact_solution"]}
        super(ConvertedTimedLinearSystem, self).__init__(**kwargs)


class _KrylovSolver(object):
    """Prototype of a Krylov subspace method for linear systems (linsys.py:277-517)."""

    x0 = _LazyHost("x0")
    xk = _LazyHost("xk")
    MMlr0 = _LazyHost("MMlr0")
    Mlr0 = _LazyHost("Mlr0")

    def __init__(self, linear_system, x0=None, tol=1e-5, maxiter=None, explicit_residual=False,
                 store_arnoldi=False, dtype=None):
        r"""Init standard attributes and perform checks (linsys.py:280-405).

        :param linear_system: a :class:`LinearSystem`.
        :param x0: initial guess, ``(N,1)``/``(N,)`` array (a device :class:`~utils.DVec` is
          accepted too, which is how restarts avoid a PCIe round trip).  Defaults to zero.
        :param tol: tolerance for the relative residual norm
          :math:`\|M M_l (b-A x_k)\|_{M^{-1}} / \|M M_l b\|_{M^{-1}}`.
        :param maxiter: maximum number of iterations (default ``N``).
        :param explicit_residual: compute the residual explicitly in each iteration.
        :param store_arnoldi: expose ``V``, ``H`` (and ``P``) trimmed to the computed part.
        :param dtype: optional dtype for the Arnoldi/Lanczos basis (real fp64 on the device).

        Attributes after the solve: ``xk``, ``resnorms``, ``errnorms`` (if an exact solution was
        given), ``iter``, and ``V``/``H``/``P`` with ``store_arnoldi``.  Non-convergence raises
        :class:`~utils.ConvergenceError` carrying the solver.
        """
        if not isinstance(linear_system, LinearSystem):
            raise utils.ArgumentError("linear_system is not an instance of LinearSystem")
        self.linear_system = linear_system
        self._ctx = linear_system._ctx
        N = linear_system.N
        self.maxiter = N if maxiter is None else maxiter
        # device blocks of this solve: complex as soon as the system, x0 or `dtype` is
        self._bdt = utils._bdt(linear_system.dtype, getattr(x0, "dtype", None), dtype)
        if isinstance(x0, DVec):
            self.flat_vecs = False
            self.x0 = x0.astype(self._bdt)
        else:
            self.flat_vecs, (x0,) = utils.shape_vecs(x0)
            self.x0 = x0
        self.explicit_residual = explicit_residual
        self.store_arnoldi = store_arnoldi

        # the hook sees the guess as it is held - a device vector stays on the device (reading the
        # `x0` attribute would download it: 80 MB over PCIe per restart cycle at N = 10^7)
        x0_cur = self.__dict__.get("_x0_dev")
        if x0_cur is None:
            x0_cur = self.__dict__.get("_x0_host")
        x0_new = self._get_initial_guess(x0_cur)
        if x0_new is not x0_cur:
            self.x0 = x0_new
        x0d = _dev_of(self, "x0", self._ctx)
        self.MMlr0, self.Mlr0, self.MMlr0_norm = self._get_initial_residual(x0d)

        if x0d is None:
            self.x0 = DVec.zeros(N, self._ctx, dtype=self._bdt)
        self.tol = tol
        self.xk = None

        x0dtype = numpy.dtype(float) if x0 is None else getattr(x0, "dtype", numpy.dtype(float))
        self.dtype = utils._common_type([linear_system.dtype, x0dtype, dtype])
        self.MlAMr = linear_system.MlAMr
        self.iter = 0
        self.resnorms = []

        if self.linear_system.MMlb_norm == 0:
            z = DVec.zeros(N, self._ctx, dtype=self._bdt)
            self.xk = z
            self.x0 = z
            self.resnorms.append(0.0)
        else:
            self.resnorms.append(self.MMlr0_norm / self.linear_system.MMlb_norm)

        if self.linear_system.exact_solution is not None:
            self.errnorms = []
            self.errnorms.append(self._errnorm(self._get_xk(None)))

        self._solve()
        self._finalize()

    def _errnorm(self, xk):
        ctx = self._ctx
        ex = self.linear_system.__dict__.get("_exact_dev")
        if ex is None:
            ex = self.linear_system.__dict__["_exact_dev"] = DVec.from_host(
                self.linear_system.exact_solution, ctx)
        dt = utils._bdt(ex.dtype, xk.dtype)
        ex, xk = ex.astype(dt), xk.astype(dt)
        d = DVec(ctx.alloc(self.linear_system.N, 1, dtype=dt))
        ctx.waxpby(d.block, 0, 1.0, ex.block, ex.col, -1.0, xk.block, xk.col)
        return utils.norm(d, ip_B=self.linear_system.ip_B)

    def _get_initial_guess(self, x0):
        """Hook to preprocess the initial guess (linsys.py:407-413)."""
        return x0

    def _get_initial_residual(self, x0):
        """Residual and its norm for the (device) initial guess (linsys.py:415-421)."""
        return self.linear_system._residual_dev(x0, compute_norm=True)

    def _get_xk(self, yk):
        """``x0 + Mr*yk`` as a device vector (linsys.py:423-428)."""
        x0 = _dev_of(self, "x0", self._ctx)
        if yk is not None:
            Mryk = (self.linear_system.Mr * yk).astype(self._bdt)
            x0 = x0.astype(self._bdt)
            xk = DVec(self._ctx.alloc(self.linear_system.N, 1, dtype=self._bdt))
            self._ctx.waxpby(xk.block, 0, 1.0, x0.block, x0.col, 1.0, Mryk.block, Mryk.col)
            return xk
        return x0

    def _finalize_iteration(self, yk, resnorm):
        """Compute solution, error norm and residual norm if required (linsys.py:430-493).

        :return: the explicit residual norm or ``None``."""
        self.xk = None
        xk = None
        if self.linear_system.exact_solution is not None:
            xk = self._get_xk(yk)
            self.xk = xk
            self.errnorms.append(self._errnorm(xk))
        rkn = None
        bnorm = self.linear_system.MMlb_norm
        if self.explicit_residual or resnorm / bnorm <= self.tol or self.iter + 1 == self.maxiter:
            if xk is None:
                xk = self._get_xk(yk)
                self.xk = xk
            _, _, rkn = self.linear_system._residual_dev(xk, compute_norm=True)
            self.resnorms.append(rkn / bnorm)
            if self.resnorms[-1] > self.tol:
                if self.iter + 1 == self.maxiter:
                    self._finalize()
                    raise utils.ConvergenceError(
                        ("No convergence in last iteration "
                         f"(maxiter: {self.maxiter}, "
                         f"residual: {self.resnorms[-1]})."),
                        self)
                elif not self.explicit_residual and resnorm / bnorm <= self.tol:
                    warnings.warn(
                        "updated residual is below tolerance, explicit residual is NOT! "
                        f"(upd={resnorm} <= tol={self.tol} < exp={self.resnorms[-1]})")
        else:
            self.resnorms.append(resnorm / bnorm)
        return rkn

    def _finalize(self):
        pass

    @staticmethod
    def operations(nsteps):
        """Number of operations needed for nsteps of the solver (linsys.py:498-510)."""
        raise NotImplementedError("operations() has to be overridden by the derived solver class.")

    def _solve(self):
        raise NotImplementedError("_solve has to be overridden by the derived solver class.")

    def _repr(self, title, extra=()):
        def ends(v):
            return "[{}, ..., {}]".format(v[0], v[-1])
        lines = [title,
                 "    MMlr0 = " + ends(self.MMlr0),
                 "    MMlr0_norm = {}".format(self.MMlr0_norm),
                 "    MlAMr: {} x {} matrix".format(*self.MlAMr.shape),
                 "    Mlr0: " + ends(self.Mlr0)]
        lines += list(extra)
        lines += ["    flat_vecs: {}".format(self.flat_vecs),
                  "    store_arnoldi: {}".format(self.store_arnoldi)]
        if hasattr(self, "ortho"):
            lines.append("    ortho: {}".format(self.ortho))
        lines += ["    tol: {}".format(self.tol),
                  "    maxiter: {}".format(self.maxiter),
                  "    iter: {}".format(self.iter),
                  "    explicit residual: {}".format(self.explicit_residual),
                  "    resnorms: [{}, ..., {}]".format(self.resnorms[0], self.resnorms[-1]),
                  "    x0: " + ends(self.x0),
                  "    xk: " + ends(self.xk)]
        return "\n".join(lines)


class Cg(_KrylovSolver):
    r"""Preconditioned CG method (linsys.py:520-708).

    One device pass per recurrence: the operator application, ``<p, Ap>``, then a fused
    kernel for ``yk += alpha p; r -= alpha Ap; z = M r; <r, z>`` when ``M`` is the identity or
    a diagonal (Jacobi) operator, and the direction update ``p = z + beta p``.
    """

    def __init__(self, linear_system, **kwargs):
        if not linear_system.self_adjoint or not linear_system.positive_definite:
            warnings.warn("Cg applied to a non-self-adjoint or non-definite "
                          "linear system. Consider using Minres or Gmres.")
        super(Cg, self).__init__(linear_system, **kwargs)

    def __repr__(self):
        return self._repr("krypy CG object")

    def _solve(self):
        ls = self.linear_system
        ctx = self._ctx
        N = ls.N
        euclid = ls.ip_B is None or isinstance(ls.ip_B, utils.IdentityLinearOperator)
        M_id = isinstance(ls.M, utils.IdentityLinearOperator)
        bdt = self._bdt
        cplx = utils._is_c(bdt)
        # the CG recurrences have real coefficients: on complex data they run on the real views of length 2N, with
        # the (real) Jacobi scaling as a diagonal of that length
        Md = None if M_id else (ls.M._real_diag_image(ctx) if cplx else ls.M._device_matrix())
        fused = euclid and (M_id or (Md is not None and Md.kind == "diag"))

        yk = DVec(ctx.alloc(N, 1, dtype=bdt))
        self.rhos = rhos = [self.MMlr0_norm ** 2]
        # working copies (linsys.py:603-607)
        self._Mlrk = _dev_of(self, "Mlr0", ctx).astype(bdt).copy()
        self._MMlrk = self._Mlrk if M_id else _dev_of(self, "MMlr0", ctx).astype(bdt).copy()
        p = _dev_of(self, "MMlr0", ctx).astype(bdt).copy()
        Ap = DVec(ctx.alloc(N, 1, dtype=bdt))
        self.iter = 0

        if self.store_arnoldi:
            self._Vb = ctx.alloc(N, self.maxiter + 1, dtype=bdt)
            if self.MMlr0_norm > 0:
                ctx.vdiv(self._Vb, 0, self._MMlrk.block, self._MMlrk.col, float(self.MMlr0_norm))
            self._Pb = None
            if not M_id:
                self._Pb = ctx.alloc(N, self.maxiter + 1, dtype=bdt)
                if self.MMlr0_norm > 0:
                    ctx.vdiv(self._Pb, 0, self._Mlrk.block, self._Mlrk.col, float(self.MMlr0_norm))
            self.H = numpy.zeros((self.maxiter + 1, self.maxiter))
            alpha_old = 0

        # whole iteration in one device call (one host synchronisation) when the operator is a plain
        # device matrix as well: direction update, A p, <p, Ap>, the fused update, <r, z>
        # (KRYPY_AMD_CG_STEP=0: the step-by-step path - operator, inner product and updates as separate calls, the
        # step length formed on the host - for diagnosis)
        Amat = self.MlAMr._device_matrix(ctx, bdt) if fused else None
        one_call = (Amat is not None and hasattr(ctx, "cg_step")
                    and os.environ.get("KRYPY_AMD_CG_STEP", "1") != "0")
        # the last steps' scalars (k, rho, d, <p, Ap>, rho_new, sanity word): what a ConvergenceError's solver
        # carries for diagnosis when the fused step is in use (the step length never visits the host there)
        self.cg_trace = trace = collections.deque(maxlen=16)

        # runs of iterations in ONE C call each (kh_cg_cycle: the fused step with omega, rho and the convergence test
        # formed in C instead of in this interpreter); the call stops after the iteration that reaches the tolerance
        # (finalised below like any other: iterate, explicit residual), at non-finite scalars and before the last iteration
        cyc = None
        if (one_call and not cplx and not self.store_arnoldi and not self.explicit_residual
                and ls.exact_solution is None and hasattr(ctx, "cg_cycle") and self.maxiter >= 3
                and type(self)._finalize_iteration is _KrylovSolver._finalize_iteration
                and os.environ.get("KRYPY_AMD_CG_CYCLE", "1") != "0"):
            cyc = dict(rhos=numpy.zeros(self.maxiter + 1), trace=numpy.zeros(6 * self.maxiter))
        bnorm = ls.MMlb_norm

        while self.resnorms[-1] > self.tol and self.iter < self.maxiter:
            k = self.iter
            if cyc is not None and k + 1 < self.maxiter:
                cr, ct = cyc["rhos"], cyc["trace"]
                cr[k] = rhos[-1]
                if k > 0:
                    cr[k - 1] = rhos[-2]
                k_done, why = ctx.cg_cycle(
                    Amat, None if M_id else Md, p.block, p.col, Ap.block, Ap.col, yk.block, yk.col,
                    self._Mlrk.block, self._Mlrk.col, None if M_id else self._MMlrk.block,
                    0 if M_id else self._MMlrk.col, k, self.maxiter - 1, float(self.tol), float(bnorm), cr, ct)
                for i in range(k, k_done + (1 if why == _hip.CYCLE_CHECK else 0)):
                    trace.append((i, float(ct[6 * i]), float(ct[6 * i + 1]), float(ct[6 * i + 2]), float(ct[6 * i + 3]),
                                  int(ct[6 * i + 4])))
                if why == _hip.CYCLE_CHECK:
                    # the iterations k .. k_done-1 the call DID record belong to the solver's state first (what the
                    # per-step path leaves behind when step k_done fails: rhos, resnorms and iter agree with each other)
                    for i in range(k, k_done):
                        rhos.append(float(cr[i + 1]))
                        self.resnorms.append(numpy.float64(ct[6 * i + 5]) / bnorm)
                    self.iter = k_done
                    self.xk = self._get_xk(yk)
                    raise _hip.BackendError(
                        "Cg: the fused step %d returned non-finite scalars (flags %d); last steps "
                        "(k, rho, d, <p,Ap>, rho_new, flags): %s" % (k_done, int(ct[6 * k_done + 4]), list(trace)))
                if k_done > k:
                    last_plain = k_done if why != _hip.CYCLE_TOL else k_done - 1
                    for i in range(k, k_done):
                        rhos.append(float(cr[i + 1]))
                    for i in range(k, last_plain):          # plain appends (linsys.py:476-477)
                        self.resnorms.append(numpy.float64(ct[6 * i + 5]) / bnorm)
                    self.xk = None
                    if why == _hip.CYCLE_TOL:
                        self.iter = k_done - 1
                        rkn = self._finalize_iteration(yk, numpy.float64(ct[6 * (k_done - 1) + 5]))
                        if rkn is not None:
                            rhos[-1] = rkn ** 2
                    self.iter = k_done
                    continue
            if k > 0:
                # p = MMlrk + rhos[-1]/rhos[-2] * p   (linsys.py:627)
                omega = rhos[-1] / rhos[-2]
            if one_call:
                den, rho_new, pAp, flags = ctx.cg_step(
                    Amat, None if M_id else Md, p.block, p.col, Ap.block, Ap.col, yk.block, yk.col,
                    self._Mlrk.block, self._Mlrk.col, None if M_id else self._MMlrk.block,
                    0 if M_id else self._MMlrk.col, k == 0, float(omega) if k > 0 else 0.0,
                    float(rhos[-1]))
                trace.append((k, float(rhos[-1]), den, pAp, rho_new, flags))
                if flags & (_hip.CG_NONFINITE_PAP | _hip.CG_NONFINITE_RHO | _hip.CG_STEP_CLAMPED) and numpy.isfinite(rhos[-1]):
                    # finite data in, inf / nan out: nothing to iterate on (yk and r were left as they were)
                    self.xk = self._get_xk(yk)
                    raise _hip.BackendError(
                        "Cg: the fused step %d returned non-finite scalars (flags %d); last steps "
                        "(k, rho, d, <p,Ap>, rho_new, flags): %s" % (k, flags, list(trace)))
                # (a divisor <= 0 - an operator that is not positive definite - is recorded in the trace only: the
                # reference iterates on, linsys.py:640-648)
            else:
                if k > 0:
                    ctx.waxpby(p.block, p.col, 1.0, self._MMlrk.block, self._MMlrk.col, float(omega),
                               p.block, p.col)
                self.MlAMr._apply_dev(p.block, p.col, Ap.block, Ap.col, 1)
                pAp = utils._inner_dev(p.block, p.col, 1, Ap.block, Ap.col, 1, ls.ip_B)[0, 0]
            alpha = rhos[-1] / pAp
            if abs(numpy.imag(alpha)) > 1e-12:
                warnings.warn(
                    f"Iter {k}: abs(alpha.imag) = {abs(alpha.imag)} > 1e-12. "
                    "Is your operator self-adjoint in the provided inner product?")
            # (one_call: the step length the device took, real part of the same quotient)
            alpha = float(numpy.float64(rhos[-1]) / numpy.float64(den)) if one_call else float(numpy.real(alpha))

            if self.store_arnoldi:
                if k > 0:
                    self.H[k - 1, k] = self.H[k, k - 1]
                    self.H[k, k] = (1.0 + alpha * omega / alpha_old) / alpha
                else:
                    self.H[k, k] = 1.0 / alpha

            if one_call:
                MMlrk_norm = numpy.sqrt(abs(rho_new))
            elif fused:
                rho_new = ctx.cg_update(alpha, p.block, p.col, Ap.block, Ap.col, yk.block, yk.col,
                                        self._Mlrk.block, self._Mlrk.col, None if M_id else Md,
                                        None if M_id else self._MMlrk.block,
                                        0 if M_id else self._MMlrk.col)
                MMlrk_norm = numpy.sqrt(abs(rho_new))
            else:
                ctx.waxpby(yk.block, yk.col, 1.0, yk.block, yk.col, alpha, p.block, p.col)
                ctx.waxpby(self._Mlrk.block, self._Mlrk.col, 1.0, self._Mlrk.block,
                           self._Mlrk.col, -alpha, Ap.block, Ap.col)
                self._MMlrk = (ls.M * self._Mlrk).astype(bdt)
                MMlrk_norm = utils.norm(self._Mlrk, self._MMlrk, ip_B=ls.ip_B)
            rhos.append(MMlrk_norm ** 2)

            if self.store_arnoldi:
                sgn = float((-1) ** (k + 1))
                ctx.vdiv(self._Vb, k + 1, self._MMlrk.block, self._MMlrk.col, sgn * MMlrk_norm)
                if self._Pb is not None:
                    ctx.vdiv(self._Pb, k + 1, self._Mlrk.block, self._Mlrk.col, sgn * MMlrk_norm)
                self.H[k + 1, k] = numpy.sqrt(rhos[-1] / rhos[-2]) / alpha
                alpha_old = alpha

            rkn = self._finalize_iteration(yk, MMlrk_norm)
            if rkn is not None:
                rhos[-1] = rkn ** 2
            self.iter += 1

        if not _is_set(self, "xk"):      # (reading self.xk would download it)
            self.xk = self._get_xk(yk)

    # the reference keeps Mlrk / MMlrk as ndarray attributes
    @property
    def Mlrk(self):
        return self._Mlrk.download()

    @property
    def MMlrk(self):
        return self._MMlrk.download()

    def _finalize(self):
        super(Cg, self)._finalize()
        if self.store_arnoldi and "V" not in self.__dict__ and hasattr(self, "_Vb"):
            self.V = self._Vb.download(0, self.iter + 1)
            if self._Pb is not None:
                self.P = self._Pb.download()
            self.H = self.H[: self.iter + 1, : self.iter]

    @staticmethod
    def operations(nsteps):
        """Number of operations needed for nsteps of CG (linsys.py:698-708)."""
        return {"A": 1 + nsteps, "M": 2 + nsteps, "Ml": 2 + nsteps, "Mr": 1 + nsteps,
                "ip_B": 2 + 2 * nsteps, "axpy": 2 + 2 * nsteps}


class _ArnoldiBasisMixin(object):
    """``V`` / ``P`` of Arnoldi-based solvers: device blocks, downloaded on access."""

    @property
    def V(self):
        v = self.__dict__.get("_V_trim")
        if v is not None:
            return v
        return self._basis_source().V

    @property
    def P(self):
        p = self.__dict__.get("_P_trim")
        if p is not None:
            return p
        return self._basis_source().P


class Minres(_ArnoldiBasisMixin, _KrylovSolver):
    r"""Preconditioned MINRES method (linsys.py:711-875).

    Lanczos three-term recurrence on the device (:class:`~utils.Arnoldi` with
    ``ortho='lanczos'``: SpMV, two fused axpy/dot kernels, Jacobi scaling fused into the norm),
    the 4x1 ``R`` QR update with two remembered Givens rotations on the host, and one fused
    kernel for ``z = (v_k - R0 W0 - R1 W1)/R2; W <- [W1, z]; yk += y0 z``.

    ``lanczos``: the Lanczos relation (an :class:`~utils.Arnoldi`).
    """

    def __init__(self, linear_system, ortho="lanczos", **kwargs):
        if not linear_system.self_adjoint:
            warnings.warn("Minres applied to a non-self-adjoint "
                          "linear system. Consider using Gmres.")
        self.ortho = ortho
        super(Minres, self).__init__(linear_system, **kwargs)

    def __repr__(self):
        return self._repr("krypy MINRES object")

    def _basis_source(self):
        return self.lanczos

    def _solve(self):
        ls = self.linear_system
        ctx = self._ctx
        N = ls.N
        # (without store_arnoldi nobody reads the Lanczos basis afterwards: a sliding window of it)
        self.lanczos = utils.Arnoldi(
            self.MlAMr, _dev_of(self, "Mlr0", ctx), maxiter=self.maxiter, ortho=self.ortho,
            M=ls.M, Mv=_dev_of(self, "MMlr0", ctx), Mv_norm=self.MMlr0_norm, ip_B=ls.ip_B,
            _window=not self.store_arnoldi)

        bdt = self.lanczos.dtype
        W = ctx.alloc(N, 2, dtype=bdt)     # the two remembered direction vectors (linsys.py:807)
        slot = 0                # column of W that currently holds W0
        y = [self.MMlr0_norm, 0.0]
        G2 = None
        G1 = None
        yk = DVec(ctx.alloc(N, 1, dtype=bdt))

        def rot(G, u, v):      # [[c, s], [-conj(s), c]] (utils.py:430)
            return G[0] * u + G[1] * v, -G[1].conjugate() * u + G[0] * v

        # runs of iterations in ONE C call each (kh_minres_cycle: Lanczos steps with look-ahead on the device, the
        # rotations and the deferred recurrence updates in C instead of in this interpreter - at N <= 10^5 the loop
        # below is what shows between two 18 us launches); the call stops after the step that reaches the tolerance,
        # at a possible invariant subspace, when the basis window is full and before the last iteration - those take
        # the per-step path below
        cyc = self._cycle_state()
        try:      # (whatever ends the loop - an exception included - a deferred update must not outlive W and yk)
            while (self.resnorms[-1] > self.tol and self.lanczos.iter < self.lanczos.maxiter
                   and not self.lanczos.invariant):
                if cyc is not None and self.lanczos.iter + 1 < self.lanczos.maxiter:
                    st = cyc["st"]
                    st[0:2] = G1 if G1 is not None else (0.0, 0.0)
                    st[2:4] = G2 if G2 is not None else (0.0, 0.0)
                    st[4] = (G1 is not None) + (G2 is not None)
                    st[5:7] = y
                    got = self._run_cycle(cyc, W, slot, yk)
                    if got is not None:
                        slot = got
                        n = int(st[4])
                        G2 = (float(st[2]), float(st[3])) if n >= 1 else None
                        G1 = (float(st[0]), float(st[1])) if n >= 2 else None
                        y = [float(st[5]), float(st[6])]
                        continue
                k = self.iter = self.lanczos.iter
                self.lanczos.advance()
                H = self.lanczos.H
                # QR update of the Lanczos matrix (linsys.py:826-841), scalars only
                R0 = 0.0
                R1 = H[k - 1, k].item()   # k == 0 reads H[-1, 0] like the reference: zero
                if G1 is not None:
                    R0, R1 = rot(G1, R0, R1)
                R2, R3 = H[k, k].item(), H[k + 1, k].item()
                if G2 is not None:
                    R1, R2 = rot(G2, R1, R2)
                G1 = G2
                gc, gs, gr = utils.givens_scalars(R2, R3)
                G2 = (_pyscalar(gc), _pyscalar(gs))
                R2 = _pyscalar(gr)
                y = list(rot(G2, y[0], y[1]))
                # z = (V_k - R0*W0 - R1*W1)/R2 ; W = [W1, z] ; yk += y[0]*z   (linsys.py:844-846)
                # (deferred: the next Lanczos launch carries the update in the shadow of its last pass; whoever reads
                # yk - _get_xk - flushes first)
                ctx.minres_update(self.lanczos._V, k - self.lanczos._base, W, slot, R0, R1, R2, y[0],
                                  yk.block, yk.col, defer=True)
                slot = 1 - slot
                y = [y[1], 0.0]
                self._finalize_iteration(yk, numpy.abs(y[0]))
        finally:
            ctx.minres_flush()
        if not _is_set(self, "xk"):      # (reading self.xk would download it)
            self.xk = self._get_xk(yk)

    def _cycle_state(self):
        """Scratch of the C cycle, or None when this solve is not eligible: real data, Euclidean inner product, the
        fused Lanczos step on a plain device matrix (Jacobi M allowed), nothing but the residual recurrence to record
        per iteration."""
        ar = self.lanczos
        ctx = self._ctx
        if (os.environ.get("KRYPY_AMD_MINRES_CYCLE", "1") == "0" or not hasattr(ctx, "minres_cycle")
                or not (ar._fused and ar._lookahead == 1 and ar._Amat is not None and ar._proj is None)
                or ar._cplx or ar.H.dtype != numpy.float64 or ar.ortho != "lanczos" or ar._BV is not None
                or (ar._Md is not None and ar._Md.kind != "diag")
                or self.explicit_residual or self.linear_system.exact_solution is not None
                or type(self)._finalize_iteration is not _KrylovSolver._finalize_iteration      # (a subclass watches every step)
                or ar.maxiter < 3 or not ar.H.flags.c_contiguous):
            return None
        return dict(st=numpy.zeros(8), resn=numpy.zeros(ar.maxiter))

    def _run_cycle(self, cyc, W, slot, yk):
        """One kh_minres_cycle call from the current iteration on; returns the new W slot when at least one iteration
        was recorded (the caller re-tests its loop condition), None otherwise."""
        ar = self.lanczos
        ctx = self._ctx
        k0 = ar.iter
        m = ar.maxiter
        # steps up to k_last can be begun in the columns the basis (or its sliding window) has now; beyond that
        # Arnoldi.advance slides the window / grows the blocks and the next call takes over again
        k_last = min(m - 1, ar._base + ar._cols - 2)
        k_stop = min(m - 1, k_last)
        if k_stop <= k0 or ar._enq > k_last + 1 or (k0 > 0 and k0 - 1 < ar._base):
            return None
        for sl in range(4):
            ar._claim(sl)
        bnorm = self.linear_system.MMlb_norm
        try:
            k_done, enq, h2, slot, why = ctx.minres_cycle(
                ar._Amat, ar._Md, ar._V, ar._P, ar._W, k0, k_stop, k_last, ar._base, max(ar._enq, k0), float(self.tol),
                float(bnorm), ar.H, W, slot, yk.block, yk.col, cyc["st"], ar._h2, cyc["resn"])
        except BaseException:
            ar._enq = ar.iter       # (as Gmres._run_cycle: nothing of the steps begun in C is used, the slots go back)
            for sl in range(4):
                ar._release(sl)
            raise
        ar._enq, ar._h2 = enq, h2
        in_flight = {j % 4 for j in range(k_done, enq)}       # (their slots stay claimed until advance() / _settle())
        for sl in range(4):
            if sl not in in_flight:
                ar._release(sl)
        if k_done == k0:
            return None                                 # (an invariance check is due: the per-step path takes it)
        ar.iter = k_done
        last_plain = k_done if why != _hip.CYCLE_TOL else k_done - 1
        for i in range(k0, last_plain):                 # plain appends (linsys.py:476-477)
            self.resnorms.append(float(cyc["resn"][i]) / bnorm)
        self.iter = k_done - 1
        self.xk = None
        if why == _hip.CYCLE_TOL:
            self._finalize_iteration(yk, float(cyc["resn"][k_done - 1]))
        return slot

    def _get_xk(self, yk):
        self._ctx.minres_flush()         # a deferred recurrence update may still be waiting for its launch
        return super(Minres, self)._get_xk(yk)

    def _finalize(self):
        self._ctx.minres_flush()
        super(Minres, self)._finalize()
        if hasattr(self, "lanczos"):
            self.lanczos._settle()      # drop a speculative look-ahead step, if any (as Gmres does)
        if self.store_arnoldi and hasattr(self, "lanczos"):
            got = self.lanczos.get()
            self._V_trim, self.H = got[0], got[1]
            if not isinstance(self.linear_system.M, utils.IdentityLinearOperator):
                self._P_trim = got[2]

    @staticmethod
    def operations(nsteps):
        """Number of operations needed for nsteps of MINRES (linsys.py:864-874)."""
        return {"A": 1 + nsteps, "M": 2 + nsteps, "Ml": 2 + nsteps, "Mr": 1 + nsteps,
                "ip_B": 2 + 2 * nsteps, "axpy": 4 + 8 * nsteps}


class Gmres(_ArnoldiBasisMixin, _KrylovSolver):
    r"""Preconditioned GMRES method (linsys.py:877-1018).

    Per iteration one device call (:meth:`utils.Arnoldi.advance`: SpMV, the Gram-Schmidt
    chain against the growing basis, norm, normalise) and an O(k) host update of the Givens QR
    of the Hessenberg matrix.  ``x_k = x_0 + M_r V_k R_k^{-1} y`` is formed on the device
    only when needed (convergence test with the explicit residual, last iteration).
    """

    def __init__(self, linear_system, ortho="mgs", **kwargs):
        self.ortho = ortho
        super(Gmres, self).__init__(linear_system, **kwargs)

    def __repr__(self):
        return self._repr("krypy GMRES object", extra=(
            "    R: {} x {} matrix".format(*self.R.shape),
            "    V: {} x {} matrix".format(self.linear_system.N, self.maxiter + 1)))

    def _basis_source(self):
        return self.arnoldi

    def _get_xk(self, y):
        if y is None:
            return _dev_of(self, "x0", self._ctx)
        k = self.arnoldi.iter
        if k > 0:
            yy = scipy.linalg.solve_triangular(self.R[:k, :k], y)
            ctx = self._ctx
            yk = DVec(ctx.alloc(self.linear_system.N, 1, dtype=self.arnoldi.dtype))
            ctx.gemm_nn(self.arnoldi._V, 0, k, yy, 1.0, 0.0, yk.block, 0)
            return super(Gmres, self)._get_xk(yk)
        return _dev_of(self, "x0", self._ctx)

    def _solve(self):
        ls = self.linear_system
        ctx = self._ctx
        self.arnoldi = utils.Arnoldi(
            self.MlAMr, _dev_of(self, "Mlr0", ctx), maxiter=self.maxiter, ortho=self.ortho,
            M=ls.M, Mv=_dev_of(self, "MMlr0", ctx), Mv_norm=self.MMlr0_norm, ip_B=ls.ip_B)
        cs = []  # Givens rotations as (c, s) Python scalars
        rdt = utils._bdt(self.dtype, self.arnoldi.dtype)
        self.R = numpy.zeros([self.maxiter + 1, self.maxiter], dtype=rdt)
        y = numpy.zeros((self.maxiter + 1, 1), dtype=rdt)
        y[0] = self.MMlr0_norm
        R = self.R

        # The iterations whose bookkeeping is a single append - no explicit residual, no error norm, not the last one -
        # run in ONE C call each time round: Arnoldi steps with look-ahead on the device, Givens QR and the residual
        # recurrence in C instead of in this interpreter (kh_gmres_cycle; what the per-step loop below does, minus
        # 30-40 us of Python per step - at N <= 10^5 that is the larger part of an iteration).  The call stops after the
        # step that reaches the tolerance (finalised below like any other), at a step that may have found an
        # invariant subspace (decided by Arnoldi.advance below) and before the last iteration.
        cyc = self._cycle_state(cs, y)
        while (self.resnorms[-1] > self.tol and self.arnoldi.iter < self.arnoldi.maxiter
               and not self.arnoldi.invariant):
            if cyc is not None and self.arnoldi.iter + 1 < self.arnoldi.maxiter:
                done = self._run_cycle(cyc, cs, y)
                if done:
                    continue
            k = self.iter = self.arnoldi.iter
            self.arnoldi.advance()
            # new Hessenberg column through the previous rotations (linsys.py:980-991); plain
            # floats: k tiny numpy calls per step would cost more than the device step itself
            col = self.arnoldi.H[: k + 2, k].tolist()
            for i in range(k):
                c, s = cs[i]
                t0, t1 = col[i], col[i + 1]
                col[i] = c * t0 + s * t1
                col[i + 1] = -s.conjugate() * t0 + c * t1
            gc, gs, _ = utils.givens_scalars(col[k], col[k + 1])
            c, s = _pyscalar(gc), _pyscalar(gs)
            cs.append((c, s))
            t0, t1 = col[k], col[k + 1]
            col[k] = c * t0 + s * t1
            col[k + 1] = -s.conjugate() * t0 + c * t1
            R[: k + 2, k] = col
            t0, t1 = y[k, 0].item(), y[k + 1, 0].item()
            y[k, 0] = c * t0 + s * t1
            y[k + 1, 0] = -s.conjugate() * t0 + c * t1
            self._finalize_iteration(y[: k + 1], abs(y[k + 1, 0]))

        if not _is_set(self, "xk"):      # (reading self.xk would download it)
            self.xk = self._get_xk(y[: self.arnoldi.iter])

    def _cycle_state(self, cs, y):
        """Scratch of the C cycle, or None when this solve is not eligible: real data, Euclidean inner product, the
        fused step on a plain device matrix (Jacobi M allowed), and nothing but the residual recurrence to record
        per iteration."""
        ar = self.arnoldi
        ctx = self._ctx
        if (os.environ.get("KRYPY_AMD_GMRES_CYCLE", "1") == "0" or not hasattr(ctx, "gmres_cycle")
                or not (ar._fused and ar._lookahead == 1 and ar._Amat is not None and ar._proj is None)
                or ar._cplx or self.R.dtype != numpy.float64 or ar.H.dtype != numpy.float64 or ar._win
                or ar.ortho not in ("mgs", "dmgs", "cgs", "cgs2") or ar._BV is not None
                or self.explicit_residual or self.linear_system.exact_solution is not None
                or type(self)._finalize_iteration is not _KrylovSolver._finalize_iteration      # (a subclass watches every step)
                or ar._base != 0 or ar.maxiter < 3
                or not (ar.H.flags.c_contiguous and self.R.flags.c_contiguous)):
            return None
        m = ar.maxiter
        return dict(cs=numpy.zeros(2 * m), y=numpy.zeros(m + 1), resn=numpy.zeros(m))

    def _run_cycle(self, cyc, cs, y):
        """One kh_gmres_cycle call from the current iteration on; returns True when at least one iteration was
        recorded (the caller re-tests its loop condition)."""
        ar = self.arnoldi
        ctx = self._ctx
        k0 = ar.iter
        m = ar.maxiter
        # (a basis that grows on demand - the default maxiter = N - holds ar._cols columns now: steps up to cols - 2 can be
        # begun; beyond that Arnoldi.advance grows the blocks and the next call takes over again)
        k_last = min(m - 1, ar._cols - 2)
        k_stop = min(m - 1, k_last)
        if k_stop <= k0 or ar._enq > k_last + 1:
            return False
        for i in range(len(cs)):                       # rotations so far (the per-step path may have produced some)
            cyc["cs"][2 * i], cyc["cs"][2 * i + 1] = cs[i]
        cyc["y"][: k0 + 2] = y[: k0 + 2, 0]
        for sl in range(4):
            ar._claim(sl)
        bnorm = self.linear_system.MMlb_norm
        try:
            k_done, enq, h2, why = ctx.gmres_cycle(
                ar._Amat, ar._Md, ar._V, ar._P, ar._W, k0, k_stop, k_last, ar._sweeps, ar._gs_mode, max(ar._enq, k0),
                float(self.tol), float(bnorm), ar.H, self.R, cyc["cs"], cyc["y"], ar._h2, cyc["resn"])
        except BaseException:
            # a failed launch / recovery inside the C loop: steps may have been begun that this object knows nothing
            # of.  Nothing of them is used (the error propagates); the slots go back so that another basis of the
            # context does not wait for an owner that will never fetch them.
            ar._enq = ar.iter
            for sl in range(4):
                ar._release(sl)
            raise
        ar._enq, ar._h2 = enq, h2
        in_flight = {j % 4 for j in range(k_done, enq)}       # (their slots stay claimed until advance() / _settle())
        for sl in range(4):
            if sl not in in_flight:
                ar._release(sl)
        if k_done == k0:
            return False                                # (an invariance check is due: the per-step path takes it)
        for i in range(len(cs), k_done):
            cs.append((float(cyc["cs"][2 * i]), float(cyc["cs"][2 * i + 1])))
        y[: k_done + 1, 0] = cyc["y"][: k_done + 1]
        ar.iter = k_done
        # every recorded iteration but (possibly) the last is a plain append (linsys.py:476-477)
        last_plain = k_done if why != _hip.CYCLE_TOL else k_done - 1
        for i in range(k0, last_plain):
            self.resnorms.append(float(cyc["resn"][i]) / bnorm)
        self.iter = k_done - 1
        if why == _hip.CYCLE_TOL:
            self.xk = None
            self._finalize_iteration(y[: k_done], float(cyc["resn"][k_done - 1]))
        else:
            self.xk = None
        return True

    def _finalize(self):
        super(Gmres, self)._finalize()
        if hasattr(self, "arnoldi"):
            self.arnoldi._settle()      # drop a speculative look-ahead step, if any
        if self.store_arnoldi and hasattr(self, "arnoldi"):
            got = self.arnoldi.get()
            self._V_trim, self.H = got[0], got[1]
            if not isinstance(self.linear_system.M, utils.IdentityLinearOperator):
                self._P_trim = got[2]

    @staticmethod
    def operations(nsteps):
        """Number of operations needed for nsteps of GMRES (linsys.py:1008-1018)."""
        return {"A": 1 + nsteps, "M": 2 + nsteps, "Ml": 2 + nsteps, "Mr": 1 + nsteps,
                "ip_B": 2 + nsteps + nsteps * (nsteps + 1) / 2,
                "axpy": 4 + 2 * nsteps + nsteps * (nsteps + 1) / 2}


class _RestartedSolver(object):
    """Base class for restarted solvers (linsys.py:1021-1072).

    The last approximation is handed to the next cycle as a device vector (no host round
    trip); ``xk`` downloads on access like for the plain solvers.
    """

    xk = _LazyHost("xk")

    def __init__(self, Solver, linear_system, max_restarts=0, **kwargs):
        """:param max_restarts: maximum number of restarts; the maximum number of iterations
        is ``(max_restarts+1)*maxiter``."""
        self.xk = None
        kwargs = dict(kwargs)
        self.resnorms = [numpy.inf]
        if linear_system.exact_solution is not None:
            self.errnorms = [numpy.inf]
        tol = None
        restart = 0
        while restart == 0 or (self.resnorms[-1] > tol and restart <= max_restarts):
            try:
                xk_dev = self.__dict__.get("_xk_dev")
                if xk_dev is not None:
                    kwargs.update({"x0": xk_dev})
                sol = Solver(linear_system, **kwargs)
            except utils.ConvergenceError as e:
                sol = e.solver
            self.xk = _dev_of(sol, "xk", sol._ctx)
            tol = sol.tol
            del self.resnorms[-1]
            self.resnorms += sol.resnorms
            if linear_system.exact_solution is not None:
                del self.errnorms[-1]
                self.errnorms += sol.errnorms
            restart += 1
            # the finished cycle's solver - and with it its basis, 8 GB at N = 10^7 - goes BEFORE the next cycle
            # allocates: one block cycles through the pool instead of two being alive at once (and the second
            # cycle of a solve does not pay a fresh hipMalloc, 230 ms for 8 GB)
            sol = None
        if self.resnorms[-1] > tol:
            raise utils.ConvergenceError(f"No convergence after {max_restarts} restarts.", self)


class RestartedGmres(_RestartedSolver):
    """Restarted GMRES method (linsys.py:1075-1081): ``GMRES(m)`` is
    ``RestartedGmres(ls, maxiter=m, max_restarts=R)``."""

    def __init__(self, *args, **kwargs):
        super(RestartedGmres, self).__init__(Gmres, *args, **kwargs)
