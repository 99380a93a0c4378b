"""Deflated Krylov solvers: projector ``P = I - AU (U^* AU)^{-1} U^*`` on the MI355X.

Host-side mirror of the hot-path part of ``krypy/deflation.py`` (``ObliqueProjection``,
``_DeflationMixin``, ``DeflatedCg/Minres/Gmres``, ``Ritz``).  The deflation bases ``U``,
``AU`` and the orthonormalised projector bases ``V``, ``W`` are ``(N, d)`` device blocks; one
application of the projected operator is, per refinement sweep, a tall-skinny ``W^T a`` panel
product, a ``d x d`` matrix-vector product with the precomputed ``R^{-1} Q^H`` and a ``z -= V c``
panel update - all on the device (``kh_proj``), inside the fused Arnoldi step for
DeflatedGmres/DeflatedMinres, so a deflated iteration needs no host round trip either.  ``Arnoldifyer`` / ``bound_pseudo`` (pseudospectral convergence prediction on small dense
matrices) are outside the hot path and not provided.

Reference lines are cited as ``deflation.py:<line>`` (= ``/root/reference/krypy/deflation.py``).
"""
import weakref

import numpy
import scipy.linalg

from . import _hip, linsys, utils
from .utils import DVec

__all__ = ["DeflatedCg", "DeflatedMinres", "DeflatedGmres", "_DeflationMixin",
           "ObliqueProjection", "_Projection", "Ritz"]


class _Projection(utils.Projection):
    def __init__(self, linear_system, U, **kwargs):
        """Abstract base class of a projection for deflation (deflation.py:19-29)."""
        raise NotImplementedError("abstract base class cannot be instanciated")


class ObliqueProjection(_Projection):
    def __init__(self, linear_system, U, qr_reorthos=0, **kwargs):
        """Oblique projection for left deflation (deflation.py:32-56).

        ``U``: ``(N, d)`` host array or device block.  ``U`` is orthonormalised in the
        ``Minv`` inner product by modified Gram-Schmidt on the device, ``AU = Ml A Mr U`` is
        one multi-column operator application, and ``Projection(AU, U)`` builds the XQRY
        factors.
        """
        self.linear_system = linear_system
        ctx = linear_system._ctx
        udt = U.dtype if hasattr(U, "dtype") else numpy.dtype(float)
        self._bdt = bdt = utils._bdt(linear_system.dtype, udt)
        Ud = utils._upload_block(U, ctx, dtype=bdt)
        # orthogonalize U in the Minv-inner-product (deflation.py:40).  get_ip_Minv_B() is an
        # IdentityLinearOperator *instance* for the unpreconditioned system, so this is the
        # modified Gram-Schmidt branch of utils.qr, never scipy's Householder QR.
        self._Ud, _ = utils.qr(Ud, ip_B=linear_system.get_ip_Minv_B(), reorthos=qr_reorthos)
        d = self._Ud.ncols
        self._AUd = ctx.alloc(linear_system.N, d, dtype=bdt)
        if d > 0:
            linear_system.MlAMr._apply_dev(self._Ud, 0, self._AUd, 0, d)
        self._MAUd = None
        super(_Projection, self).__init__(self._AUd, self._Ud, ip_B=linear_system.ip_B, **kwargs)

    @property
    def U(self):
        """Orthonormalised basis of the deflation space (host copy)."""
        return self._Ud.download()

    @property
    def AU(self):
        r""":math:`M_lAM_rU` (host copy)."""
        return self._AUd.download()

    @property
    def MAU(self):
        r""":math:`MM_lAM_rU` (deflation.py:70-76), host copy."""
        return self._MAU_dev().download()

    def _MAU_dev(self):
        if self._MAUd is None:
            d = self._AUd.ncols
            self._MAUd = self._AUd.ctx.alloc(self._AUd.n, d, dtype=self._bdt)
            if d > 0:
                self.linear_system.M._apply_dev(self._AUd, 0, self._MAUd, 0, d)
        return self._MAUd

    def _correct_dev(self, z):
        """``z + W WR VR^{-1} R^{-1} Q^H W^H Ml (b - A z)`` for a device ``z`` (deflation.py:58-68)."""
        if self._k == 0:
            return z
        ls = self.linear_system
        ctx = z.ctx
        dt = utils._bdt(self._bdt, z.dtype)
        z = z.astype(dt)
        Az = (ls.A * z).astype(dt)
        b = ls._b_dev.astype(dt)
        r = DVec(ctx.alloc(ls.N, 1, dtype=dt))
        ctx.waxpby(r.block, 0, 1.0, b.block, b.col, -1.0, Az.block, Az.col)
        c = ls.Ml * r
        c = utils._inner_dev(self._Wd, 0, self._k, c.block, c.col, 1, self.ip_B)
        if self.Q is not None and self.R is not None:
            c = scipy.linalg.solve_triangular(self.R, self.Q.T.conj().dot(c))
        if self.WR is not self.VR:
            c = self.WR.dot(scipy.linalg.solve_triangular(self.VR, c))
        out = z.copy()
        Wd = utils._promote_block(self._Wd, 0, self._k, dt)[0]
        ctx.gemm_nn(Wd, 0, self._k, c, 1.0, 1.0, out.block, out.col)
        return out

    def correct(self, z):
        """Correct the approximate solution ``z`` (host ``(N,1)`` array in and out)."""
        return self._correct_dev(utils._as_dvec(z, self.linear_system._ctx)).download()


class _DeflationMixin(object):
    """Mixin that adds deflation to a solver from :mod:`krypy_amd.linsys` (deflation.py:79-189).

    :param linear_system: the :class:`~krypy_amd.linsys.LinearSystem`.
    :param U: basis of the deflation space, ``(N, k)`` host array or device block.
    """

    def __init__(self, linear_system, U=None, projection_kwargs=None, *args, **kwargs):
        if U is None:
            U = numpy.zeros((linear_system.N, 0))
        if projection_kwargs is None:
            projection_kwargs = {}
        d = U.ncols if hasattr(U, "ncols") else U.shape[1]
        udtype = U.dtype
        projection = ObliqueProjection(linear_system, U, **projection_kwargs)
        self.projection = projection
        # E = <U, Ml A Mr U> from the factors (deflation.py:104-111)
        if projection.Q is None and projection.R is None:
            E = numpy.eye(d)
        else:
            E = projection.Q.dot(projection.R)
        if projection.VR is not None and projection.WR is not None:
            E = projection.WR.T.conj().dot(E.dot(projection.VR))
        self.E = E
        self.C = numpy.zeros((d, 0))
        self._B_ = None
        super(_DeflationMixin, self).__init__(linear_system, dtype=udtype, *args, **kwargs)

    def _solve(self):
        N = self.linear_system.N
        # the projected operator refers to its solver weakly: self.MlAMr -> P -> self would be a cycle, and the
        # solver's basis (10 GB at N = 10^7) would wait for the garbage collector instead of going back to the pool
        me = weakref.ref(self)

        def solver():
            s = me()
            if s is None:
                raise utils.RuntimeError("the deflated solver behind this projected operator is gone")
            return s

        P = utils.LinearOperator((N, N), self.projection._bdt, lambda Av: solver()._apply_projection(Av))
        P._apply_dev = lambda X, xcol, Y, ycol, ncols=1: solver()._apply_projection_dev(X, xcol, Y, ycol, ncols)
        if type(self)._store_UAv is _DeflationMixin._store_UAv:
            # (I - P) A runs inside the fused Arnoldi step; DeflatedCg keeps the Python path: its
            # C recurrence needs self.iter / self.rhos at application time
            P._kh_proj = self.projection._device_projector()
            P._on_ya = weakref.WeakMethod(self._store_UAv)     # (no solver <-> operator reference cycle)
        self.MlAMr = P * self.linear_system.MlAMr
        super(_DeflationMixin, self)._solve()

    def _apply_projection_dev(self, X, xcol, Y, ycol, ncols=1):
        for c in range(ncols):
            PAv, UAv = self.projection._apply_complement_dvec(DVec(X, xcol + c), return_Ya=True)
            self._store_UAv(UAv)
            PAv = PAv.astype(Y.dtype)
            Y.copy_from(ycol + c, PAv.block, PAv.col, 1)

    def _store_UAv(self, UAv):
        self.C = numpy.column_stack([self.C, UAv])

    def _apply_projection(self, Av):
        """Host-array form of the projected operator (deflation.py:135-143)."""
        blk = _hip.get_context().upload(Av, dtype=self.projection._bdt)
        out = self.linear_system._ctx.alloc(Av.shape[0], Av.shape[1], dtype=blk.dtype)
        self._apply_projection_dev(blk, 0, out, 0, Av.shape[1])
        return out.download()

    def _get_initial_residual(self, x0):
        """Projected initial residual :math:`MPM_l(b-Ax_0)` (deflation.py:145-159)."""
        ls = self.linear_system
        ctx = ls._ctx
        if x0 is None:
            Mlr = linsys._dev_of(ls, "Mlb", ctx)
        else:
            dt = utils._bdt(self.projection._bdt, x0.dtype)
            Ax = (ls.A * x0.astype(dt)).astype(dt)
            b = ls._b_dev.astype(dt)
            r = DVec(ctx.alloc(ls.N, 1, dtype=dt))
            ctx.waxpby(r.block, 0, 1.0, b.block, b.col, -1.0, Ax.block, Ax.col)
            Mlr = ls.Ml * r
        PMlr, self.UMlr = self.projection._apply_complement_dvec(Mlr, return_Ya=True)
        MPMlr = ls.M * PMlr
        MPMlr_norm = utils.norm(PMlr, MPMlr, ip_B=ls.ip_B)
        return MPMlr, PMlr, MPMlr_norm

    def _get_xk(self, yk):
        xk = super(_DeflationMixin, self)._get_xk(yk)
        return self.projection._correct_dev(xk)

    @property
    def B_(self):
        r""":math:`\underline{B}=\langle V_{n+1},M_lAM_rU\rangle` (deflation.py:165-189)."""
        (n_, n) = self.H.shape
        ls = self.linear_system
        if self._B_ is None or self._B_.shape[1] < n_:
            AUd = self.projection._AUd
            d = AUd.ncols
            Vd = self._basis_block()
            if ls.self_adjoint:
                self._B_ = self.C.T.conj()
                if n_ > n:
                    last = utils._inner_dev(Vd, n_ - 1, 1, AUd, 0, d, ls.ip_B) if d else \
                        numpy.zeros((1, 0))
                    self._B_ = numpy.vstack([self._B_, last])
            else:
                self._B_ = utils._inner_dev(Vd, 0, n_, AUd, 0, d, ls.ip_B) if d else \
                    numpy.zeros((n_, 0))
        return self._B_

    def _basis_block(self):
        if hasattr(self, "_basis_source"):
            return self._basis_source()._V
        return self._Vb  # Cg with store_arnoldi

    def estimate_time(self, nsteps, ndefl, deflweight=1.0):
        """Predicted run time of ``nsteps`` iterations with ``ndefl`` deflation vectors, from the measured
        timings of a :class:`~krypy_amd.linsys.TimedLinearSystem` (deflation.py:191-233).

        The solver's own operation counts come from ``operations(nsteps)``; on top of them the deflation costs
        ``ndefl`` applications of ``A``, ``M``, ``Ml``, ``Mr`` (building ``A U``), the inner products of the QR of
        ``U`` (``d(d+1)/2``), of ``E = <U, A U>`` (``d^2``) and of the two projection sweeps per application
        (``2 d`` per ``Ml`` application), and the matching vector updates.  ``deflweight`` scales the deflation
        part.  Raises :class:`utils.RuntimeError` without a timed linear system, like the reference."""
        ops = self.operations(nsteps)
        d, napp = ndefl, ops["Ml"]
        tri = d * (d + 1) / 2
        deflation_ops = {"A": d, "M": d, "Ml": d, "Mr": d,
                         "ip_B": tri + d * d + 2 * d * napp,
                         "axpy": tri + d * d + (2 * d + 2) * napp}
        if not isinstance(self.linear_system, linsys.TimedLinearSystem):
            raise utils.RuntimeError("A `TimedLinearSystem` has to be used in order to obtain timings.")
        t = self.linear_system.timings
        return t.get_ops(ops) + deflweight * t.get_ops(deflation_ops)


class DeflatedCg(_DeflationMixin, linsys.Cg):
    """Deflated preconditioned CG method (deflation.py:236-263)."""

    def __init__(self, *args, **kwargs):
        self._UAps = []
        super(DeflatedCg, self).__init__(*args, **kwargs)

    def _store_UAv(self, UAp):
        r"""Next column of :math:`C=\langle U, M_lAM_rV_n\rangle` (deflation.py:247-263).

        CG applies the operator to its search direction :math:`p_k`, not to the Lanczos vector :math:`v_k`.  With
        :math:`r_k = p_k - \omega_{k-1} p_{k-1}`, :math:`\omega_j=\rho_{j+1}/\rho_j`, and
        :math:`v_k = (-1)^k r_k/\sqrt{\rho_k}`, the column is a combination of the last three
        :math:`\langle U, A p_j\rangle` and of the previous column: weights first, then one pass."""
        hist = self._UAps
        hist.append(UAp)
        k, rho = self.iter, self.rhos
        weights = [1.0]
        if k >= 1:
            weights.append(-(1 + rho[-1] / rho[-2]))
        if k >= 2:
            weights.append(rho[-2] / rho[-3])
        col = numpy.array(hist[-1], copy=True)
        for back, wgt in enumerate(weights[1:], start=2):
            col += wgt * hist[-back]
        col *= (-1.0 if k % 2 else 1.0) / numpy.sqrt(rho[-1])
        if k >= 1:
            col -= numpy.sqrt(rho[-2] / rho[-1]) * self.C[:, -1:]
        self.C = numpy.column_stack([self.C, col])


class DeflatedMinres(_DeflationMixin, linsys.Minres):
    """Deflated preconditioned MINRES method (deflation.py:266-273)."""


class DeflatedGmres(_DeflationMixin, linsys.Gmres):
    """Deflated preconditioned GMRES method (deflation.py:276-283)."""


class Ritz(object):
    def __init__(self, deflated_solver, mode="ritz"):
        """Ritz pairs from a deflated Krylov subspace method (deflation.py:737-838).

        The block matrices are small (``n + d`` rows) and assembled on the host from ``H``,
        ``B_``, ``C``, ``E``; only ``F = <AU, MAU>`` and the Ritz vectors touch N-vectors and
        run on the device.
        """
        self._deflated_solver = deflated_solver
        linear_system = deflated_solver.linear_system
        self.values = None
        self.coeffs = None
        H_ = deflated_solver.H
        (n_, n) = H_.shape
        H = H_[:n, :n]
        projection = deflated_solver.projection
        m = projection._Ud.ncols
        eye, zeros = numpy.eye, numpy.zeros
        if n + m == 0:
            self.values = numpy.zeros((0,))
            self.coeffs = numpy.zeros((0,))
            self.resnorms = numpy.zeros((0,))
            return
        if not isinstance(projection, ObliqueProjection):
            raise utils.ArgumentError(
                "Invalid projection used in deflated_solver. Valid are ObliqueProjection")
        E = deflated_solver.E
        C = deflated_solver.C
        EinvC = numpy.linalg.solve(E, C) if m > 0 else C
        B_ = deflated_solver.B_
        B = B_[:n, :]
        M = numpy.block([[H + B.dot(EinvC), B], [C, E]])
        if m > 0:
            F = utils._inner_dev(projection._AUd, 0, m, projection._MAU_dev(), 0, m,
                                 linear_system.ip_B)
        else:
            F = zeros((0, 0))
        S = numpy.block([[eye(n_), B_, zeros((n_, m))],
                         [B_.T.conj(), F, E],
                         [zeros((m, n_)), E.T.conj(), eye(m)]])
        eig = scipy.linalg.eigh if linear_system.self_adjoint else scipy.linalg.eig
        if mode == "ritz":
            self.values, self.coeffs = eig(M)
        elif mode == "harmonic":
            L = numpy.block([[H_, zeros((n_, m))], [EinvC, eye(m)]])
            K = numpy.block([[eye(n_), B_], [B_.T.conj(), F]])
            sigmas, self.coeffs = eig(M.T.conj(), L.T.conj().dot(K.dot(L)))
            self.values = numpy.zeros(m + n, dtype=sigmas.dtype)
            zero = numpy.abs(sigmas) < numpy.finfo(float).eps
            self.values[~zero] = 1.0 / sigmas[~zero]
            self.values[zero] = numpy.inf
        else:
            raise utils.ArgumentError(
                f"Invalid value  '{mode}' for 'mode'. " + "Valid are ritz and harmonic.")
        for i in range(n + m):
            self.coeffs[:, [i]] /= numpy.linalg.norm(self.coeffs[:, [i]], 2)
        self.resnorms = numpy.zeros(m + n)
        for i in range(n + m):
            mu = self.values[i]
            y = self.coeffs[:, [i]]
            G = numpy.block([[H_ - mu * eye(n_, n), zeros((n_, m))],
                             [EinvC, eye(m)],
                             [zeros((m, n)), -mu * eye(m)]])
            Gy = G.dot(y)
            resnorm2 = Gy.T.conj().dot(S.dot(Gy))
            self.resnorms[i] = numpy.sqrt(numpy.abs(resnorm2).item())

    def _get_vectors_dev(self, indices=None):
        """``[V_n, U] @ coeffs[:, indices]`` as a device block: two tall-skinny GEMMs
        (deflation.py:840-847)."""
        s = self._deflated_solver
        (n_, n) = s.H.shape
        coeffs = self.coeffs if indices is None else self.coeffs[:, indices]
        coeffs = numpy.asarray(coeffs)
        if coeffs.ndim == 1:
            coeffs = coeffs.reshape(-1, 1)
        Ud = s.projection._Ud
        Vd = s._basis_block()
        ctx = Vd.ctx
        dt = utils._bdt(Vd.dtype, Ud.dtype)
        if numpy.iscomplexobj(coeffs) and coeffs.size:
            if numpy.abs(coeffs.imag).max() > 1e-12 * max(numpy.abs(coeffs).max(), 1e-300):
                dt = numpy.dtype(numpy.complex128)      # complex Ritz vectors
            else:
                coeffs = coeffs.real
        Vd = utils._promote_block(Vd, 0, n, dt)[0] if n > 0 else Vd
        Ud = utils._promote_block(Ud, 0, Ud.ncols, dt)[0] if Ud.ncols > 0 else Ud
        out = ctx.alloc(Vd.n, coeffs.shape[1], dtype=dt)
        if coeffs.shape[1] == 0:
            return out
        ctx.gemm_nn(Vd, 0, n, coeffs[:n, :], 1.0, 0.0, out, 0)
        if Ud.ncols > 0:
            ctx.gemm_nn(Ud, 0, Ud.ncols, coeffs[n:, :], 1.0, 1.0, out, 0)
        return out

    def get_vectors(self, indices=None):
        """Compute Ritz vectors (host ``(N, len(indices))`` array)."""
        return self._get_vectors_dev(indices).download()

    def get_explicit_residual(self, indices=None):
        """Explicitly computes the Ritz residual (deflation.py:849-855)."""
        vecs = self._get_vectors_dev(indices)
        ls = self._deflated_solver.linear_system
        out = vecs.ctx.alloc(vecs.n, vecs.ncols, dtype=vecs.dtype)
        ls.MlAMr._apply_dev(vecs, 0, out, 0, vecs.ncols)
        vals = self.values if indices is None else self.values[indices]
        return out.download() - vecs.download() * numpy.asarray(vals)

    def get_explicit_resnorms(self, indices=None):
        """Explicitly computes the Ritz residual norms (deflation.py:857-869)."""
        res = self.get_explicit_residual(indices)
        ls = self._deflated_solver.linear_system
        Mres = ls.M * res
        resnorms = numpy.zeros(res.shape[1])
        for i in range(resnorms.shape[0]):
            resnorms[i] = utils.norm(res[:, [i]], Mres[:, [i]], ip_B=ls.ip_B)
        return resnorms


# ---- out of scope: perturbation bounds on pseudospectra of small dense matrices
def __getattr__(name):
    if name in ("Arnoldifyer", "bound_pseudo"):
        def _stub(*args, **kwargs):
            raise NotImplementedError(
                "krypy_amd.deflation.%s: the a-priori bound machinery (Arnoldifyer / bound_pseudo, optional pseudopy) "
                "is host-side analysis outside the accelerated Krylov path and is not provided (SURVEY.md section "
                "2)" % name)
        _stub.__name__ = name
        return _stub
    raise AttributeError("module 'krypy_amd.deflation' has no attribute %r" % name)
