"""Krylov kernel utilities: operator algebra, inner products, Givens, Arnoldi, projections.

Host-side mirror of ``krypy/utils.py`` for the hot path named in SURVEY.md section 8
(same names, argument meaning and error behaviour), with every N-vector operation
executed on the MI355X through :mod:`krypy_amd._hip`.  Small dense algebra
(Hessenberg, Givens, d x d factors) stays on the host exactly as in the reference.

Vectors that never need to leave the GPU are handled as :class:`DVec` (a column
of a :class:`~krypy_amd._hip.DeviceVectors` block); the public functions accept and
return NumPy arrays with the reference's shapes.

Reference lines are cited as ``utils.py:<line>`` (= ``/root/reference/krypy/utils.py``).
"""
import time
import warnings
import weakref
from collections import defaultdict

import numpy
import scipy.linalg
import scipy.sparse
from scipy.linalg import blas

from . import _hip
from ._hip import BackendError  # noqa: F401  (re-exported)

__all__ = [
    "ArgumentError", "AssumptionError", "ConvergenceError", "LinearOperatorError",
    "InnerProductError", "RuntimeError", "BackendError",
    "Arnoldi", "Givens", "House", "IdentityLinearOperator", "LinearOperator", "MatrixLinearOperator",
    "ZeroLinearOperator", "Projection", "arnoldi", "arnoldi_res", "find_common_dtype",
    "get_linearoperator", "inner", "ip_euclid", "norm", "norm_squared", "orthonormality", "qr",
    "shape_vec", "shape_vecs", "DVec", "Timer", "Timings", "TimedLinearOperator", "ritz",
    "hegedus", "angles",
]


# ----------------------------------------------------------------------------------------
# exceptions (utils.py:62-103)
# ----------------------------------------------------------------------------------------
class ArgumentError(Exception):
    """Raised when an argument is invalid (krypy's analogue of ValueError)."""


class AssumptionError(Exception):
    """Raised when valid arguments turn out to violate an assumption of the method."""


class ConvergenceError(Exception):
    """Raised when a method did not converge; ``solver`` holds the last approximation."""

    def __init__(self, msg, solver):
        super(ConvergenceError, self).__init__(msg)
        self.solver = solver


class LinearOperatorError(Exception):
    """Raised when a :class:`LinearOperator` cannot be applied."""


class InnerProductError(Exception):
    """Raised when the inner product is indefinite."""


class RuntimeError(Exception):  # noqa: A001  (same name as in the reference, utils.py:102)
    """Raised for errors that do not fit in any other exception."""


# ----------------------------------------------------------------------------------------
# dtype / shape helpers (utils.py:106-143)
# ----------------------------------------------------------------------------------------
def _common_type(dtypes):
    ts = [numpy.dtype(t) for t in dtypes if t is not None]
    return numpy.result_type(*ts) if ts else numpy.dtype(float)


def _is_sparse(A):
    # scipy.sparse.isspmatrix is False for the newer sparse *arrays*; accept both
    return scipy.sparse.issparse(A)


def find_common_dtype(*args):
    """Common dtype of ndarrays, sparse matrices and LinearOperators (others are ignored)."""
    dtypes = []
    for arg in args:
        if type(arg) is numpy.ndarray or _is_sparse(arg) or isinstance(arg, (LinearOperator, DVec)):
            if hasattr(arg, "dtype"):
                dtypes.append(arg.dtype)
            else:
                warnings.warn("object %s does not have a dtype." % arg.__repr__)
    return _common_type(dtypes)


def shape_vec(x):
    """Take a (n,) ndarray and return it as (n,1) ndarray."""
    return numpy.reshape(x, (x.shape[0], 1))


def shape_vecs(*args):
    """Reshape all ndarrays with ``shape==(n,)`` to ``shape==(n,1)``; others pass through."""
    ret_args = []
    flat_vecs = True
    for arg in args:
        if type(arg) is numpy.ndarray:
            if len(arg.shape) == 1:
                arg = shape_vec(arg)
            else:
                flat_vecs = False
        ret_args.append(arg)
    return flat_vecs, ret_args


def _require_real(dtype, what):
    if numpy.dtype(dtype).kind == "c":
        raise NotImplementedError(
            "%s is complex: this part of the MI355X path is real fp64 only; there is no CPU "
            "fallback." % what)


def _bdt(*dtypes):
    """Device block dtype for a mix of input dtypes: complex128 if any is complex, else float64."""
    for t in dtypes:
        if t is not None and numpy.dtype(t).kind == "c":
            return numpy.dtype(numpy.complex128)
    return numpy.dtype(numpy.float64)


def _is_c(dtype):
    return numpy.dtype(dtype).kind == "c"


def _promote_block(X, c0, nc, dtype):
    """``(block, col0)`` holding columns ``[c0, c0+nc)`` of ``X`` with block dtype ``dtype``
    (a real block is widened to complex on the device; a complex one is never narrowed)."""
    if not _is_c(dtype) or _is_c(X.dtype):
        return X, c0
    Z = X.ctx.alloc(X.n, nc, dtype=numpy.complex128)
    X.ctx.promote(X, c0, Z, 0, nc)
    return Z, 0


# ----------------------------------------------------------------------------------------
# device vector view
# ----------------------------------------------------------------------------------------
class DVec(object):
    """One N-vector resident in HBM: column ``col`` of a DeviceVectors ``block``."""

    __slots__ = ("block", "col")

    def __init__(self, block, col=0):
        self.block, self.col = block, col

    @property
    def dtype(self):
        return self.block.dtype

    @property
    def n(self):
        return self.block.n

    @property
    def shape(self):
        return (self.block.n, 1)

    @property
    def ctx(self):
        return self.block.ctx

    def download(self):
        """Host copy with the reference's ``(N, 1)`` shape."""
        return numpy.ascontiguousarray(self.block.download(self.col, 1))

    def copy(self):
        out = self.ctx.alloc(self.n, 1, dtype=self.dtype)
        out.copy_from(0, self.block, self.col, 1)
        return DVec(out, 0)

    def astype(self, dtype):
        """This vector with (at least) the given block dtype; real -> complex widens on the device."""
        b, c = _promote_block(self.block, self.col, 1, dtype)
        return self if b is self.block else DVec(b, c)

    @staticmethod
    def from_host(x, ctx=None, dtype=None):
        ctx = _hip.get_context() if ctx is None else ctx
        x = numpy.asarray(x)
        if x.ndim == 2 and x.shape[1] != 1:
            raise ArgumentError("expected a single column, got shape %s" % (x.shape,))
        return DVec(ctx.upload(x.reshape(-1, 1), dtype=dtype), 0)

    @staticmethod
    def zeros(n, ctx=None, dtype=None):
        ctx = _hip.get_context() if ctx is None else ctx
        return DVec(ctx.alloc(n, 1, dtype=_bdt(dtype)), 0)


def _as_dvec(x, ctx=None, dtype=None):
    if isinstance(x, DVec):
        return x if dtype is None else x.astype(dtype)
    return DVec.from_host(x, ctx, dtype=dtype)


def _upload_block(X, ctx=None, dtype=None):
    """(N,k) host array or DeviceVectors -> DeviceVectors (widened to ``dtype`` if given)."""
    if isinstance(X, _hip.DeviceVectors) or hasattr(X, "download") and hasattr(X, "ncols"):
        return X if dtype is None else _promote_block(X, 0, X.ncols, dtype)[0]
    ctx = _hip.get_context() if ctx is None else ctx
    return ctx.upload(numpy.asarray(X), dtype=dtype)


# ----------------------------------------------------------------------------------------
# inner products and norms (utils.py:146-238)
# ----------------------------------------------------------------------------------------
def ip_euclid(X, Y):
    """Euclidean inner product ``X^* Y`` with ``shape==(m,n)`` (utils.py:146-157), on device."""
    return inner(X, Y)


def _inner_dev(Xb, x0, nx, Yb, y0, ny, ip_B=None):
    """<X[:, x0:x0+nx], Y[:, y0:y0+ny]> for device blocks -> (nx, ny) host array."""
    ctx = Xb.ctx
    N = Xb.n
    B = None
    if not (ip_B is None or isinstance(ip_B, IdentityLinearOperator)):
        try:
            B = get_linearoperator((N, N), ip_B)
        except TypeError:
            # user callable (X, Y) -> (m, n): it gets host arrays, like in the reference (utils.py:189)
            return numpy.asarray(ip_B(Xb.download(x0, nx), Yb.download(y0, ny)))
    dt = _bdt(Xb.dtype, Yb.dtype, None if B is None else B.dtype)
    Xb, x0 = _promote_block(Xb, x0, nx, dt)
    Yb, y0 = _promote_block(Yb, y0, ny, dt)
    if B is None:
        return ctx.gemm_tn(Xb, x0, nx, Yb, y0, ny)
    # operator B is applied to the thinner side (utils.py:190-193); <B x, y> = <x, B y>
    if nx > ny:
        T = ctx.alloc(N, nx, dtype=dt)
        B._apply_dev(Xb, x0, T, 0, nx)
        return ctx.gemm_tn(T, 0, nx, Yb, y0, ny)
    T = ctx.alloc(N, ny, dtype=dt)
    B._apply_dev(Yb, y0, T, 0, ny)
    return ctx.gemm_tn(Xb, x0, nx, T, 0, ny)


def inner(X, Y, ip_B=None):
    """Euclidean and non-Euclidean inner product (utils.py:160-193).

    :param X: array with ``shape==(N,m)`` (or a device block / :class:`DVec`).
    :param Y: array with ``shape==(N,n)``.
    :param ip_B: ``None`` (Euclidean), a self-adjoint positive definite operator ``B``
        (array / sparse / LinearOperator: ``X^* B Y``) or a callable ``(X, Y) -> <X,Y>``.
    :return: ``(m,n)`` ndarray, always 2-dimensional.
    """
    Xb, x0, nx = _block_of(X)
    Yb, y0, ny = _block_of(Y, Xb.ctx)
    if Xb.n != Yb.n:
        raise ArgumentError("inner: X and Y have different lengths")
    if nx == 0 or ny == 0:
        return numpy.zeros((nx, ny))
    return _inner_dev(Xb, x0, nx, Yb, y0, ny, ip_B)


def _block_of(X, ctx=None):
    if isinstance(X, DVec):
        return X.block, X.col, 1
    if hasattr(X, "download") and hasattr(X, "ncols"):
        return X, 0, X.ncols
    X = numpy.asarray(X)
    if X.ndim != 2:
        raise ArgumentError("expected a 2-dimensional array")
    b = _upload_block(X, ctx)
    return b, 0, X.shape[1]


def norm_squared(x, Mx=None, inner_product=ip_euclid):
    """Squared norm w.r.t. a given scalar product (utils.py:196-211)."""
    assert len(x.shape) == 2
    rho = inner_product(x, x) if Mx is None else inner_product(x, Mx)
    if rho.shape == (1, 1):
        if abs(rho[0, 0].imag) > abs(rho[0, 0]) * 1e-10 or rho[0, 0].real < 0.0:
            raise InnerProductError("<x,Mx> = %g. Is the inner product indefinite?" % rho[0, 0])
    return numpy.linalg.norm(rho, 2)


def norm(x, y=None, ip_B=None):
    r"""Norm :math:`\sqrt{\langle x,y\rangle}` (utils.py:214-238); ``y=None`` means ``y=x``."""
    euclid = ip_B is None or isinstance(ip_B, IdentityLinearOperator)
    if y is None and euclid:
        xb, x0, nx = _block_of(x)
        if nx == 1:
            return xb.ctx.nrm2(xb, x0)
        return numpy.linalg.norm(xb.download(x0, nx), 2)  # matrix 2-norm of a small panel
    if y is None:
        y = x
    ip = inner(x, y, ip_B=ip_B)
    nrm_diag = numpy.linalg.norm(numpy.diag(ip), 2)
    nrm_diag_imag = numpy.linalg.norm(numpy.imag(numpy.diag(ip)), 2)
    if nrm_diag_imag > nrm_diag * 1e-10:
        raise InnerProductError(
            "inner product defined by ip_B not positive definite? "
            "||diag(ip).imag||/||diag(ip)||=%g" % (nrm_diag_imag / nrm_diag))
    return numpy.sqrt(numpy.linalg.norm(ip, 2))


def orthonormality(V, ip_B=None):
    """:math:`\\| I_n - \\langle V,V \\rangle \\|_2` (utils.py:297-305)."""
    G = inner(V, V, ip_B=ip_B)
    return numpy.linalg.norm(numpy.eye(G.shape[0]) - G, 2)


def arnoldi_res(A, V, H, ip_B=None):
    """Arnoldi residual ``||A V_{n-1} - V_n H||`` or ``||A V_n - V_n H_n||`` (utils.py:308-329)."""
    N = V.shape[0]
    invariant = H.shape[0] == H.shape[1]
    A = get_linearoperator((N, N), A)
    res = A * (V if invariant else V[:, :-1]) - numpy.dot(V, H)
    return norm(res, ip_B=ip_B)


def hegedus(A, b, x0, M=None, Ml=None, ip_B=None):
    r"""Rescale an initial guess (Hegedues trick, utils.py:812-851).

    Returns :math:`\gamma_{\min} x_0` with
    :math:`\gamma_{\min} = \langle z, M M_l b\rangle_{M^{-1}} / \|z\|_{M^{-1}}^2`, :math:`z = M M_l A x_0`,
    which minimises :math:`\|M M_l (b - A \gamma x_0)\|_{M^{-1}}` over :math:`\gamma`; the zero
    vector if :math:`\|z\|^2 \le 10^{-15}`.  Same arguments as the solvers take.  The three operator
    applications and the two inner products run on the device; the result is a host ``(N,1)`` array.
    """
    N = len(b)
    A = get_linearoperator((N, N), A)
    M = get_linearoperator((N, N), M)
    Ml = get_linearoperator((N, N), Ml)
    x0 = numpy.asarray(x0)
    dt = _bdt(A.dtype, M.dtype, Ml.dtype, x0.dtype, numpy.asarray(b).dtype)
    MlAx0 = Ml * (A * DVec.from_host(x0, dtype=dt))
    z = M * MlAx0
    znorm2 = inner(z, MlAx0, ip_B=ip_B)
    if znorm2 <= 1e-15:
        return numpy.zeros((N, 1))
    gamma = inner(z, Ml * DVec.from_host(shape_vec(numpy.asarray(b)) if numpy.ndim(b) == 1 else b,
                                         dtype=dt), ip_B=ip_B) / znorm2
    return gamma * x0


def angles(F, G, ip_B=None, compute_vectors=False):
    r"""Principal angles between the subspaces spanned by ``F`` (N,k) and ``G`` (N,l)
    (utils.py:710-809; Knyazev & Argentati 2002, algorithm 6.2: cosines for large angles, sines
    of the residual for small ones).

    :return: ``theta`` (``max(k,l)`` angles, ascending, in :math:`[0,\pi/2]`), and with
      ``compute_vectors`` also the principal vectors ``U`` (from F) and ``V`` (from G).

    Orthonormalisations and the Gram matrices go through :func:`qr` / :func:`inner` (device); the
    SVDs are of ``k x l`` matrices on the host.
    """
    swapped = F.shape[1] < G.shape[1]
    if swapped:                      # work with the wider block first
        F, G = G, F
    k, l = F.shape[1], G.shape[1]
    QF, _ = qr(F, ip_B=ip_B)
    QG, _ = qr(G, ip_B=ip_B)
    half_pi = numpy.pi / 2
    if l == 0:
        theta, U, V = numpy.full(k, half_pi), QF, QG
    else:
        Y, cosines, Zh = scipy.linalg.svd(inner(QF, QG, ip_B=ip_B))
        Vcos = QG.dot(Zh.T.conj())
        small = int(numpy.count_nonzero(cosines ** 2 >= 0.5))       # angles below 45 degrees
        theta = numpy.concatenate([numpy.arccos(cosines[small:]), numpy.full(k - l, half_pi)])
        if compute_vectors:
            Ucos = QF.dot(Y)
            U, V = Ucos[:, small:], Vcos[:, small:]
        if small > 0:
            # small angles from the sines: the part of the G-side vectors outside span(F)
            RG = Vcos[:, :small]
            _, Rs = qr(RG - QF.dot(inner(QF, RG, ip_B=ip_B)), ip_B=ip_B)
            _, sines, Zh2 = scipy.linalg.svd(Rs)
            theta = numpy.concatenate([numpy.arcsin(sines[::-1][:small]), theta])
            if compute_vectors:
                c = cosines[:small]
                rot = numpy.diag(1.0 / c).dot(Zh2.T.conj().dot(numpy.diag(c)))
                U = numpy.column_stack([Ucos[:, :small].dot(rot), U])
                V = numpy.column_stack([RG.dot(Zh2.T.conj()), V])
    if not compute_vectors:
        return theta
    return (theta, V, U) if swapped else (theta, U, V)


# ----------------------------------------------------------------------------------------
# Householder reflectors (utils.py:332-402)
# ----------------------------------------------------------------------------------------
def _house_scalars(gamma, sigma, n):
    """The scalar part of Algorithm 5.1.1 (Golub, Van Loan) as the reference arranges it
    (utils.py:349-377): returns (v0, xnorm, alpha, beta) for a real vector with first entry
    ``gamma`` and ``sigma = ||x[1:]||``."""
    if n == 1 or sigma == 0:
        xnorm = abs(gamma)
        return 1.0, xnorm, (1.0 if gamma == 0 else gamma / xnorm), 0
    xnorm = numpy.sqrt(abs(gamma) ** 2 + sigma ** 2)
    if gamma == 0:
        return -sigma, xnorm, 1.0, 2
    return gamma + gamma / abs(gamma) * xnorm, xnorm, -gamma / abs(gamma), 2


class House(object):
    def __init__(self, x):
        """Householder transformation ``H`` with ``H x = alpha ||x||_2 e_1``, ``|alpha| = 1``
        (utils.py:332-377), for a host ``(N,1)`` vector (real)."""
        if len(x.shape) != 2 or x.shape[1] != 1:
            raise ArgumentError("x is not a vector of dim (N,1)")
        v = numpy.array(x, dtype=_bdt(x.dtype))
        gamma = v[0].item()
        sigma = 0.0 if x.shape[0] == 1 else numpy.linalg.norm(v[1:], 2)
        v0, self.xnorm, self.alpha, self.beta = _house_scalars(gamma, sigma, x.shape[0])
        v[0] = v0
        self.v = v / numpy.sqrt(abs(v0) ** 2 + sigma ** 2)

    def apply(self, x):
        """Apply the transformation to a ``(N, m)`` array: ``x - beta v (v^* x)``."""
        if len(x.shape) != 2:
            raise ArgumentError("x is not a matrix of shape (N,*)")
        if self.beta == 0:
            return x
        return x - self.beta * self.v * numpy.dot(self.v.T.conj(), x)

    def matrix(self):
        """Dense matrix ``I - beta v v^*`` (testing only)."""
        n = self.v.shape[0]
        return numpy.eye(n, n) - self.beta * numpy.dot(self.v, self.v.T.conj())


class _DevHouse(object):
    """A reflector for the sub-vector ``x[j:]`` of a device column, stored zero-padded as column
    ``j`` of the reflector block, so that applying it is one dot and one axpy on whole columns."""

    def __init__(self, ctx, Hv, j, X, xcol):
        N = Hv.n
        self.ctx, self.Hv, self.j = ctx, Hv, j
        Hv.copy_from(j, X, xcol, 1)
        Hv.zero_range(j, 0, j)
        gamma = Hv.get(j, j, 1)[0].item()
        Hv.set(j, j, [0.0])
        sigma = 0.0 if N - j == 1 else ctx.nrm2(Hv, j)          # ||x[j+1:]||
        v0, self.xnorm, self.alpha, self.beta = _house_scalars(gamma, sigma, N - j)
        Hv.set(j, j, [v0])
        ctx.vdiv(Hv, j, Hv, j, float(numpy.sqrt(abs(v0) ** 2 + sigma ** 2)))

    def apply(self, X, xcol):
        """``x[j:] -= beta v (v^* x[j:])`` in place (rows < j are untouched: v is zero there)."""
        if self.beta == 0:
            return
        d = self.ctx.dot_panel(self.Hv, self.j, 1, X, xcol)[0]
        self.ctx.axpy_panel(self.Hv, self.j, 1, [self.beta * d], X, xcol)


# ----------------------------------------------------------------------------------------
# Givens rotation (utils.py:405-436) - host, O(1)
# ----------------------------------------------------------------------------------------
class Givens(object):
    def __init__(self, x):
        """Givens rotation ``G=[[c,s],[-conj(s),c]]`` with ``G x = [r, 0]^T`` (BLAS ``drotg``
        sign convention, exactly as the reference obtains it)."""
        if x.shape != (2, 1):
            raise ArgumentError("x is not a vector of shape (2,1)")
        a = x[0].item()
        b = x[1].item()
        if numpy.isreal(x).all():
            a = numpy.real(a)
            b = numpy.real(b)
            c, s = blas.drotg(a, b)
        else:
            c, s = blas.zrotg(a, b)
        self.c = c
        self.s = s
        self.r = c * a + s * b
        self.G = numpy.array([[c, s], [-numpy.conj(s), c]])

    def apply(self, x):
        """Apply the rotation to a ``(2, m)`` array."""
        return numpy.dot(self.G, x)


def givens_scalars(a, b):
    """``(c, s, r)`` of ``Givens(numpy.array([[a], [b]]))`` for two Python scalars, without the arrays: the same BLAS
    call on the same values and the same expression for ``r`` - the same bits - in 0.4 instead of 10 microseconds.  (The
    MINRES loop at short vectors is host-bound: an iteration is one 18 us launch.)"""
    if isinstance(a, complex) or isinstance(b, complex):
        a = complex(a)          # (what the (2, 1) array would have made of the pair)
        b = complex(b)
    if a.imag == 0 and b.imag == 0:
        a = a.real
        b = b.real
        c, s = blas.drotg(a, b)
    else:
        c, s = blas.zrotg(a, b)
    return c, s, c * a + s * b


# ----------------------------------------------------------------------------------------
# linear operators (utils.py:1365-1602)
# ----------------------------------------------------------------------------------------
def _isintlike(x):
    try:
        return bool(int(x) == x)
    except (TypeError, ValueError):
        return False


class LinearOperator(object):
    """Linear operator ``C^n -> C^m`` defined by callables acting on ``(n, k)`` arrays.

    Same constructor and algebra as the reference (utils.py:1365-1456).  In addition every
    operator can act on device-resident vectors (:meth:`_apply_dev`); operators built from
    user callables do so through a download -> callback -> upload round trip, which is
    correct but slow - matrices (ndarray / CSR / diagonal) run as HIP kernels.
    """

    def __init__(self, shape, dtype, dot=None, dot_adj=None):
        if len(shape) != 2 or not _isintlike(shape[0]) or not _isintlike(shape[1]):
            raise LinearOperatorError("shape must be (m,n) with m and n integer")
        self.shape = shape
        self.dtype = numpy.dtype(dtype)  # defaults to float64
        if dot is None and dot_adj is None:
            raise LinearOperatorError("dot or dot_adj have to be defined")
        # a subclass handing over its OWN methods keeps them as class attributes: a bound method stored on the
        # instance would be a reference cycle, and whatever the operator holds (device images, a solver's basis
        # behind a projected operator) would then live until the garbage collector runs instead of until the
        # last reference goes - with 10 GB blocks that is the difference between a pool hit and a fresh hipMalloc
        for name, fn in (("_dot", dot), ("_dot_adj", dot_adj)):
            own = getattr(fn, "__self__", None) is self and \
                getattr(type(self), name, None) is getattr(fn, "__func__", None)
            if not own:
                setattr(self, name, fn)
        self._tmp = {}

    # -- host API ---------------------------------------------------------------------
    def _host_apply(self, fn, X, rows_needed, what):
        """Shared body of :meth:`dot` / :meth:`dot_adj`: the reference's contract (utils.py:1381-1405) - shape
        check against the operator, ``LinearOperatorError`` for a missing callable, an empty block passes through."""
        X = numpy.asanyarray(X)
        if X.shape[0] != rows_needed:
            raise LinearOperatorError("dimension mismatch")
        if fn is None:
            raise LinearOperatorError(what + " undefined")
        return fn(X) if X.shape[1] else numpy.zeros(X.shape)

    def dot(self, X):
        return self._host_apply(self._dot, X, self.shape[1], "dot")

    def dot_adj(self, X):
        return self._host_apply(self._dot_adj, X, self.shape[0], "dot_adj")

    @property
    def adj(self):
        return _AdjointLinearOperator(self)

    # -- device API -------------------------------------------------------------------
    def _device_matrix(self, ctx=None, dtype=None):
        """The DeviceMatrix this operator *is* (plain matrices only), else None."""
        return None

    def _real_diag_image(self, ctx=None):
        """Real diagonal of length 2N acting on the real view of a complex block (real diagonal matrices only)."""
        return None

    def _scratch(self, ctx, n, ncols, key=0, dtype=None):
        dtype = _bdt(dtype)
        k = (id(ctx), n, ncols, key, dtype.kind)
        buf = self._tmp.get(k)
        if buf is None:
            buf = self._tmp[k] = ctx.alloc(n, ncols, dtype=dtype)
        return buf

    def _apply_dev(self, X, xcol, Y, ycol, ncols=1):
        """``Y[:, ycol:ycol+ncols] = self * X[:, xcol:xcol+ncols]`` for device blocks.

        Default: the operator is a user callable on host arrays (utils.py:1371-1390 contract),
        so the columns make a host round trip."""
        if self._dot is None:
            raise LinearOperatorError("dot undefined")
        Y.upload(ycol, self._dot(X.download(xcol, ncols)))

    def _mul_dvec(self, x):
        if x.n != self.shape[1]:
            raise LinearOperatorError("dimension mismatch")
        x = x.astype(_bdt(x.dtype, self.dtype))   # a complex operator widens a real operand
        y = x.ctx.alloc(self.shape[0], 1, dtype=x.dtype)
        self._apply_dev(x.block, x.col, y, 0, 1)
        return DVec(y, 0)

    # -- algebra (utils.py:1407-1447) ---------------------------------------------------
    def __mul__(self, X):
        try:
            if isinstance(X, IdentityLinearOperator):
                return self
            elif isinstance(self, IdentityLinearOperator):
                return X
            elif isinstance(X, LinearOperator):
                return _ProductLinearOperator(self, X)
            elif isinstance(X, DVec):
                return self._mul_dvec(X)
            elif numpy.isscalar(X):
                return _ScaledLinearOperator(self, X)
            else:
                return self.dot(X)
        except LinearOperatorError:
            return NotImplemented

    def __rmul__(self, X):
        try:
            return _ScaledLinearOperator(self, X)
        except LinearOperatorError:
            return NotImplemented

    def __pow__(self, X):
        try:
            return _PowerLinearOperator(self, X)
        except LinearOperatorError:
            return NotImplemented

    def __add__(self, X):
        try:
            return _SumLinearOperator(self, X)
        except LinearOperatorError:
            return NotImplemented

    def __neg__(self):
        try:
            return _ScaledLinearOperator(self, -1)
        except LinearOperatorError:
            return NotImplemented

    def __sub__(self, X):
        return self + (-X)

    def __repr__(self):
        m, n = self.shape
        return "<%dx%d %s with dtype=%s>" % (m, n, self.__class__.__name__, str(self.dtype))


def _get_dtype(operators, dtypes=None):
    dtypes = [] if dtypes is None else dtypes
    for obj in operators:
        if obj is not None and hasattr(obj, "dtype"):
            dtypes.append(obj.dtype)
    return _common_type(dtypes)


class _SumLinearOperator(LinearOperator):
    def __init__(self, A, B):
        if not isinstance(A, LinearOperator) or not isinstance(B, LinearOperator):
            raise LinearOperatorError("both operands have to be a LinearOperator")
        if A.shape != B.shape:
            raise LinearOperatorError("shape mismatch")
        self.args = (A, B)
        super(_SumLinearOperator, self).__init__(A.shape, _get_dtype([A, B]), self._dot,
                                                 self._dot_adj)

    def _dot(self, X):
        return self.args[0].dot(X) + self.args[1].dot(X)

    def _dot_adj(self, X):
        return self.args[0].dot_adj(X) + self.args[1].dot_adj(X)

    def _apply_dev(self, X, xcol, Y, ycol, ncols=1):
        ctx = X.ctx
        T = self._scratch(ctx, self.shape[0], 1, dtype=X.dtype)
        for c in range(ncols):
            self.args[0]._apply_dev(X, xcol + c, Y, ycol + c, 1)
            self.args[1]._apply_dev(X, xcol + c, T, 0, 1)
            ctx.waxpby(Y, ycol + c, 1.0, Y, ycol + c, 1.0, T, 0)


class _ProductLinearOperator(LinearOperator):
    def __init__(self, A, B):
        if not isinstance(A, LinearOperator) or not isinstance(B, LinearOperator):
            raise LinearOperatorError("both operands have to be a LinearOperator")
        if A.shape[1] != B.shape[0]:
            raise LinearOperatorError("shape mismatch")
        self.args = (A, B)
        super(_ProductLinearOperator, self).__init__(
            (A.shape[0], B.shape[1]), _get_dtype([A, B]), self._dot, self._dot_adj)

    def _dot(self, X):
        return self.args[0].dot(self.args[1].dot(X))

    def _dot_adj(self, X):
        return self.args[1].dot_adj(self.args[0].dot_adj(X))

    def _apply_dev(self, X, xcol, Y, ycol, ncols=1):
        T = self._scratch(X.ctx, self.args[1].shape[0], ncols, dtype=X.dtype)
        self.args[1]._apply_dev(X, xcol, T, 0, ncols)
        self.args[0]._apply_dev(T, 0, Y, ycol, ncols)


class _ScaledLinearOperator(LinearOperator):
    def __init__(self, A, alpha):
        if not isinstance(A, LinearOperator):
            raise LinearOperatorError("LinearOperator expected as A")
        if not numpy.isscalar(alpha):
            raise LinearOperatorError("scalar expected as alpha")
        self.args = (A, alpha)
        super(_ScaledLinearOperator, self).__init__(
            A.shape, _get_dtype([A], [numpy.asarray(alpha).dtype]), self._dot, self._dot_adj)

    def _dot(self, X):
        return self.args[1] * self.args[0].dot(X)

    def _dot_adj(self, X):
        return numpy.conj(self.args[1]) * self.args[0].dot_adj(X)

    def _apply_dev(self, X, xcol, Y, ycol, ncols=1):
        self.args[0]._apply_dev(X, xcol, Y, ycol, ncols)
        alpha = self.args[1]
        alpha = float(alpha.real) if numpy.imag(alpha) == 0 else complex(alpha)
        for c in range(ncols):
            X.ctx.waxpby(Y, ycol + c, alpha, Y, ycol + c, 0.0, Y, ycol + c)


class _PowerLinearOperator(LinearOperator):
    def __init__(self, A, p):
        if not isinstance(A, LinearOperator):
            raise LinearOperatorError("LinearOperator expected as A")
        if A.shape[0] != A.shape[1]:
            raise LinearOperatorError("square LinearOperator expected as A")
        if not _isintlike(p):
            raise LinearOperatorError("integer expected as p")
        self.args = (A, p)
        super(_PowerLinearOperator, self).__init__(A.shape, A.dtype, self._dot, self._dot_adj)

    def _power(self, fun, X):
        res = X.copy()
        for _ in range(self.args[1]):
            res = fun(res)
        return res

    def _dot(self, X):
        return self._power(self.args[0].dot, X)

    def _dot_adj(self, X):
        return self._power(self.args[0]._dot_adj, X)


class _AdjointLinearOperator(LinearOperator):
    def __init__(self, A):
        if not isinstance(A, LinearOperator):
            raise LinearOperatorError("LinearOperator expected as A")
        self.args = (A,)
        m, n = A.shape
        super(_AdjointLinearOperator, self).__init__((n, m), A.dtype, A._dot_adj, A._dot)


class IdentityLinearOperator(LinearOperator):
    def __init__(self, shape):
        super(IdentityLinearOperator, self).__init__(shape, numpy.dtype(None), self._dot,
                                                     self._dot_adj)

    def _dot(self, X):
        return X

    def _dot_adj(self, X):
        return X

    def _apply_dev(self, X, xcol, Y, ycol, ncols=1):
        Y.copy_from(ycol, X, xcol, ncols)

    def _mul_dvec(self, x):
        return x


class ZeroLinearOperator(LinearOperator):
    def __init__(self, shape):
        super(ZeroLinearOperator, self).__init__(shape, numpy.dtype(None), self._dot,
                                                 self._dot_adj)

    def _dot(self, X):
        return numpy.zeros(X.shape)

    def _dot_adj(self, X):
        return numpy.zeros(X.shape)

    def _apply_dev(self, X, xcol, Y, ycol, ncols=1):
        Y.zero(ycol, ncols)


def _diagonal_of(A):
    """The diagonal if the sparse matrix ``A`` is square and purely diagonal, else None."""
    if A.shape[0] != A.shape[1]:
        return None
    if A.format == "dia":
        if len(A.offsets) == 1 and A.offsets[0] == 0:
            return numpy.asarray(A.diagonal(), dtype=_bdt(A.dtype))
        return None
    C = A.tocoo()
    if C.nnz <= A.shape[0] and numpy.array_equal(C.row, C.col):
        return numpy.asarray(A.diagonal(), dtype=_bdt(A.dtype))
    return None


class MatrixLinearOperator(LinearOperator):
    """Operator given by an ndarray or a SciPy sparse matrix (utils.py:1585-1602).

    The matrix is uploaded once (CSR as SciPy stores it / dense row-major / a plain
    diagonal) and applied by the HIP kernels; ``dot`` on host arrays uploads the
    operand, applies the kernel and downloads the result.
    """

    def __init__(self, A):
        super(MatrixLinearOperator, self).__init__(A.shape, A.dtype, self._dot, self._dot_adj)
        self._A = A
        self._A_adj = None
        self._dmats = {}
        self._adj_op = None

    def _device_matrix(self, ctx=None, dtype=None):
        """Device image for blocks of ``dtype`` (default: the matrix' own): a real matrix is
        uploaded a second time as c128 when it meets complex vectors."""
        dt = _bdt(self.dtype, dtype)
        dm = self._dmats.get(dt.kind)
        if dm is None:
            ctx = _hip.get_context() if ctx is None else ctx
            A = self._A
            if _is_sparse(A):
                d = _diagonal_of(A)
                if d is not None:
                    dm = ctx.diag(d, dtype=dt)
                else:
                    dm = ctx.csr(scipy.sparse.csr_matrix(A), dtype=dt)
            else:
                dm = ctx.dense(numpy.asarray(A), dtype=dt)
            self._dmats[dt.kind] = dm
        return dm

    def _real_diag_image(self, ctx=None):
        """A diagonal matrix with real entries as a REAL device diagonal of length 2N, every entry twice: how a
        Jacobi scaling acts on the real view (re, im interleaved) of a complex block.  None for anything else."""
        if "d2" not in self._dmats:
            d = _diagonal_of(self._A) if _is_sparse(self._A) else None
            dm = None
            if d is not None and not numpy.any(numpy.imag(d)):
                ctx = _hip.get_context() if ctx is None else ctx
                dm = ctx.diag(numpy.repeat(numpy.real(d).astype(float), 2))
            self._dmats["d2"] = dm
        return self._dmats["d2"]

    @property
    def _dmat(self):
        return self._dmats.get(_bdt(self.dtype).kind)

    def _apply_dev(self, X, xcol, Y, ycol, ncols=1):
        if _is_c(self.dtype) and not _is_c(X.dtype):
            raise LinearOperatorError("complex matrix applied to a real device block")
        X.ctx.apply(self._device_matrix(X.ctx, X.dtype), X, xcol, Y, ycol, ncols)

    def _dot(self, X):
        X = numpy.asarray(X)
        ctx = _hip.get_context()
        dt = _bdt(self.dtype, X.dtype)
        Xd = ctx.upload(X, dtype=dt)
        Yd = ctx.alloc(self.shape[0], X.shape[1], dtype=dt)
        self._apply_dev(Xd, 0, Yd, 0, X.shape[1])
        return numpy.ascontiguousarray(Yd.download())

    def _dot_adj(self, X):
        if self._adj_op is None:
            self._A_adj = self._A.T.conj()
            self._adj_op = MatrixLinearOperator(
                scipy.sparse.csr_matrix(self._A_adj) if _is_sparse(self._A_adj)
                else numpy.ascontiguousarray(self._A_adj))
        return self._adj_op._dot(X)

    def __repr__(self):
        return self._A.__repr__()


class DeviceOperator(MatrixLinearOperator):
    """An operator whose matrix is ALREADY on the device (a ``DeviceMatrix`` formed there, e.g. by
    ``Context.dense_from_block``): there is no host image.  Everything a ``MatrixLinearOperator`` can do on
    device blocks it can do; ``symmetric=True`` (the default) declares ``A == A^H`` so that the adjoint is the
    operator itself (utils.py:1585-1602 builds the adjoint from the host matrix, which does not exist here)."""

    def __init__(self, dmat, symmetric=True):
        LinearOperator.__init__(self, dmat.shape, dmat.dtype, self._dot, self._dot_adj)
        self._A = None
        self._A_adj = None
        self._dmats = {numpy.dtype(dmat.dtype).kind: dmat}
        self._adj_op = None
        self._symmetric = bool(symmetric)

    def _device_matrix(self, ctx=None, dtype=None):
        dm = self._dmats.get(_bdt(self.dtype, dtype).kind)
        if dm is None:
            raise LinearOperatorError("a device-only %s operator cannot be applied to %s blocks" % (self.dtype, dtype))
        return dm

    def _real_diag_image(self, ctx=None):
        return None

    def _dot_adj(self, X):
        if not self._symmetric:
            raise LinearOperatorError("no adjoint of a device-only operator that was not declared symmetric")
        return self._dot(X)

    def __repr__(self):
        return "<DeviceOperator %dx%d %s on the device>" % (self.shape[0], self.shape[1], self.dtype)


class Timer(list):
    """Measure execution time of multiple code blocks with ``with`` (utils.py:1289-1318)."""

    def __init__(self):
        super(Timer, self).__init__()

    def __enter__(self):
        self.tstart = time.time()

    def __exit__(self, a, b, c):
        self.append(time.time() - self.tstart)


class Timings(defaultdict):
    """Manages several timers, ``tm['A']`` etc. (utils.py:1321-1362)."""

    def __init__(self):
        super(Timings, self).__init__(Timer)

    def get(self, key):
        """Return the (minimal) timing for ``key``; 0 if not present."""
        if key in self and len(self[key]) > 0:
            return min(self[key])
        return 0

    def get_ops(self, ops):
        """Time for a dictionary of operation names -> number of applications."""
        total = 0.0
        for op, count in ops.items():
            total += self.get(op) * count
        return total

    def __repr__(self):
        return "Timings(" + ", ".join([f"{key}: {self.get(key)}" for key in self]) + ")"


class TimedLinearOperator(LinearOperator):
    """Operator whose applications are timed per column (utils.py:1605-1636).

    Applications through the host API and through the generic device path are timed (the device
    path synchronises the stream around the call so that the number is a kernel time).  The fused
    Arnoldi step uses the wrapped operator's device matrix directly and is not timed: the
    samples taken during the set-up of a solve are what ``Timings.get`` (a minimum) reports.
    """

    def __init__(self, linear_operator, timer=None):
        self._linear_operator = linear_operator
        super(TimedLinearOperator, self).__init__(
            shape=linear_operator.shape, dtype=linear_operator.dtype,
            dot=linear_operator.dot, dot_adj=linear_operator.dot_adj)
        self._timer = Timer() if timer is None else timer

    def dot(self, X):
        k = X.shape[1]
        if k == 0:
            return self._linear_operator.dot(X)
        with self._timer:
            ret = self._linear_operator.dot(X)
        self._timer[-1] /= k
        return ret

    def dot_adj(self, X):
        k = X.shape[1]
        if k == 0:
            return self._linear_operator.dot(X)
        with self._timer:
            ret = self._linear_operator.dot_adj(X)
        self._timer[-1] /= k
        return ret

    def _device_matrix(self, ctx=None, dtype=None):
        return self._linear_operator._device_matrix(ctx, dtype)

    def _real_diag_image(self, ctx=None):
        return self._linear_operator._real_diag_image(ctx)

    def _apply_dev(self, X, xcol, Y, ycol, ncols=1):
        if ncols == 0:
            return
        X.ctx.sync()
        with self._timer:
            self._linear_operator._apply_dev(X, xcol, Y, ycol, ncols)
            X.ctx.sync()
        self._timer[-1] /= ncols


def get_linearoperator(shape, A, timer=None):
    """Enhances aslinearoperator if A is None (utils.py:241-273).

    Accepts a :class:`LinearOperator`, ``None`` (identity), ndarray, SciPy sparse matrix or
    array (superset of the reference, which rejects ``csr_array``), ``numpy.matrix`` and
    SciPy ``LinearOperator`` with a ``dtype``.
    """
    import scipy.sparse.linalg as scipylinalg

    if isinstance(A, LinearOperator):
        ret = A
    elif A is None:
        ret = IdentityLinearOperator(shape)
    elif isinstance(A, numpy.matrix):
        ret = MatrixLinearOperator(numpy.atleast_2d(numpy.asarray(A)))
    elif isinstance(A, numpy.ndarray) or _is_sparse(A):
        ret = MatrixLinearOperator(A)
    elif isinstance(A, scipylinalg.LinearOperator):
        if not hasattr(A, "dtype"):
            raise ArgumentError("scipy LinearOperator has no dtype.")
        ret = LinearOperator(A.shape, dot=A.matmat, dot_adj=A.rmatmat, dtype=A.dtype)
    else:
        raise TypeError("type not understood")
    if A is not None and not isinstance(A, IdentityLinearOperator) and timer is not None:
        ret = TimedLinearOperator(ret, timer)
    if shape != ret.shape:
        raise LinearOperatorError("shape mismatch")
    return ret


# ----------------------------------------------------------------------------------------
# Arnoldi / Lanczos (utils.py:854-1081)
# ----------------------------------------------------------------------------------------
_GS_OF_ORTHO = {  # ortho -> (gs_mode, sweeps)
    "mgs": (_hip.GS_MGS, 1),
    "dmgs": (_hip.GS_MGS, 2),
    "lanczos": (_hip.GS_MGS, 1),
    # extensions (not in the reference): panel classical Gram-Schmidt, one reduction per sweep
    "cgs": (_hip.GS_CGS, 1),
    "cgs2": (_hip.GS_CGS, 2),
}


class Arnoldi(object):
    def __init__(self, A, v, maxiter=None, ortho="mgs", M=None, Mv=None, Mv_norm=None, ip_B=None,
                 _window=False):
        """Arnoldi algorithm: ``A V_n = V_{n+1} H_n`` (utils.py:855-952), basis in HBM.

        :param A: linear operator (anything :func:`get_linearoperator` accepts).
        :param v: initial vector, ``(N,1)`` ndarray (or a :class:`DVec`).
        :param maxiter: maximal number of iterations (default ``N``).
        :param ortho: ``'mgs'`` (default), ``'dmgs'``, ``'lanczos'`` as in the reference
            (``'house'`` is a sequential reflector chain, out of scope of the device path), plus
            the extensions ``'cgs'`` / ``'cgs2'`` (panel Gram-Schmidt, one / two passes).
        :param M: self-adjoint positive definite preconditioner; then ``P`` with ``V = M P``
            is built too.
        :param ip_B: inner product, see :func:`inner`.

        ``V`` and ``P`` are ``(N, maxiter+1)`` column-major blocks on the device; the
        attributes ``V`` / ``P`` download them on access.  ``H`` lives on the host.
        """
        N = v.shape[0]
        self.A = get_linearoperator((N, N), A)
        self.maxiter = N if maxiter is None else maxiter
        self.ortho = ortho
        self.M = get_linearoperator((N, N), M)
        if isinstance(self.M, IdentityLinearOperator):
            self.M = None
        self.ip_B = ip_B
        self.dtype = bdt = _bdt(find_common_dtype(self.A, v, self.M))
        cplx = _is_c(bdt)
        self.iter = 0
        self.invariant = False
        if ortho == "house":
            if self.M is not None or not (ip_B is None or isinstance(ip_B, IdentityLinearOperator)):
                raise ArgumentError(
                    "Only euclidean inner product allowed with Householder orthogonalization")
        elif ortho not in _GS_OF_ORTHO:
            raise ArgumentError(
                f"Invalid value '{ortho}' for argument 'ortho'. "
                + "Valid are house, mgs, dmgs and lanczos.")
        self._gs_mode, self._sweeps = _GS_OF_ORTHO.get(ortho, (_hip.GS_MGS, 1))
        self.reorthos = self._sweeps - 1

        ctx = self._ctx = v.ctx if isinstance(v, DVec) else _hip.get_context()
        # The reference allocates (N, maxiter+1) up front and lives on lazily committed host pages
        # (maxiter defaults to N).  Device memory is committed at once, so the basis starts with what
        # a fixed share of HBM allows and doubles when the iteration gets there (_grow); a restarted
        # or short run - anything that fits - gets its whole basis up front and never grows.
        self._cols = self._initial_cols(ctx, N, self.maxiter + 1, 2 if self.M is not None else 1, bdt)
        # Lanczos needs v_{k-1}, v_k, v_{k+1} only.  A caller that does not want the basis afterwards
        # (Minres without store_arnoldi) asks for a sliding WINDOW of _WINDOW_COLS columns instead of
        # maxiter+1: logical column j lives at j - _base, and when the window is full its last two
        # columns move to the front.  Nothing to zero-fill, nothing that grows with maxiter.
        self._base = 0
        self._win = (bool(_window) and ortho == "lanczos" and self._WINDOW_COLS < self._cols
                     and self._window_ok(ctx, bdt))
        if self._win:
            self._cols = self._WINDOW_COLS
        # (zero=False: every basis column is written before it is read, and V / P / get() download the
        # computed columns only - a recycled 8 GB block is not zero-filled again)
        self._V = ctx.alloc(N, self._cols, dtype=bdt, zero=False)
        self._P = ctx.alloc(N, self._cols, dtype=bdt, zero=False) if self.M is not None else None
        self._W = ctx.alloc(N, 2, dtype=bdt)
        self._BV = None
        self.H = numpy.zeros((self.maxiter + 1, self.maxiter), dtype=self.dtype)
        self._h2 = 0.0   # running sum of squares of H (Frobenius), for the invariance pre-test
        # fused device path: Euclidean inner product, M a plain diagonal (or absent)
        self._euclid = ip_B is None or isinstance(ip_B, IdentityLinearOperator)
        self._Md = None
        if self.M is not None and not cplx:
            md = self.M._device_matrix()
            if md is not None and md.kind == "diag":
                self._Md = md
            elif (md is not None and md.kind in ("csr", "dense") and md.dtype == _hip._F64
                  and ortho in _GS_OF_ORTHO and not self._win
                  and not (ortho in ("mgs", "dmgs") and N >= self._MATRIX_M_EXTERNAL_FROM and self._euclid)):
                # (long recurrences on long vectors: the register-resident chain kernel, which the matrix form of Md
                # does not use, is worth more than the look-ahead - those go the `_Mext` way below; measured
                # GMRES(60), N = 9e6: 753 vs 608 it/s; N = 1e6: 3630 vs 3400; N = 9e4: 5800 vs 6600)
                # a preconditioner given as a matrix (a sparse approximate inverse, a dense SPD block): the step
                # applies it itself - coefficients against V = M P, updates with P, M w for the norm - one C call
                # per step instead of a host loop over the Gram-Schmidt links (the matrix form of `Md`, as for ip_B)
                self._Md = md
        elif self.M is not None and self.M._real_diag_image(ctx) is not None:
            # complex data, real Jacobi scaling: the complex step takes it as a c128 diagonal
            self._Md = self.M._device_matrix(ctx, numpy.dtype(complex))
        elif self.M is not None and ortho in _GS_OF_ORTHO:
            md = self.M._device_matrix(ctx, numpy.dtype(complex))
            if md is not None and md.kind in ("csr", "dense"):       # a matrix preconditioner on complex data
                self._Md = md
        self._fused = self._euclid and (self.M is None or self._Md is not None)
        # Non-Euclidean inner product <x, y> = x^T B y with B a real matrix on the device (utils.py:184-193) and no
        # preconditioner: the step kernel's preconditioned recurrence with the roles of its two blocks swapped -
        # a second block BV = B V travels with the basis, the coefficients are <B v_j, w>, the updates w -= alpha v_j,
        # the norm sqrt(<w, B w>) - one C call and one host synchronisation per step instead of one per coefficient.
        self._ipB = None
        if (not self._euclid and self.M is None and not cplx and ortho in _GS_OF_ORTHO):
            try:
                Bop = get_linearoperator((N, N), ip_B)
            except TypeError:
                Bop = None          # a user callable: the general loop hands it host arrays
            bm = Bop._device_matrix() if Bop is not None else None
            if bm is not None and bm.dtype == _hip._F64 and bm.kind in ("diag", "csr", "dense"):
                self._ipB = bm
        self._Amat = self.A._device_matrix(ctx, bdt) if self._fused else None
        # deflated solvers hand in  P * MlAMr  with P the complement of a device projector: the
        # projection then runs inside the fused step and <U, A v_k> comes back with the H column
        self._proj = None
        self._on_ya = None
        if self._fused and self._Amat is None and isinstance(self.A, _ProductLinearOperator):
            Pop, inner = self.A.args
            kp = getattr(Pop, "_kh_proj", None)
            im = inner._device_matrix(ctx, bdt) if kp is not None else None
            if (kp is not None and im is not None and bool(getattr(kp, "cplx", False)) == cplx
                    and (self._Md is None or self._Md.kind == "diag")):      # (a matrix Md takes no projector)
                self._Amat, self._proj, self._on_ya = im, kp, Pop._on_ya
        # Look-ahead: when the operator is a plain device matrix, step k+1 depends on device data
        # only, so it is enqueued BEFORE the host waits for step k's Hessenberg column; the GPU
        # never idles while the host does its O(k) work.  A speculative step past the end of the
        # iteration is discarded by _settle().  (A Lanczos step takes H[k,k-1] from the previous step's
        # device-side H column, so it can run ahead as well.)
        self._lookahead = 1 if self._Amat is not None else 0
        self._cplx = cplx
        assert not self._win or (self._fused and self._lookahead), "sliding window outside the look-ahead path"
        self._enq = 0          # number of steps enqueued on the device so far
        if ortho == "house":
            # Householder Arnoldi (utils.py:910-922, 970-994): reflectors live zero-padded in their
            # own (N, maxiter+2) block; every application is a device dot + axpy.  Sequential by
            # nature (SURVEY: not a hot path), kept for its orthogonality guarantee.
            self._fused, self._lookahead = False, 0
            self._Hv = ctx.alloc(N, min(self.maxiter + 1, N) + 1, dtype=bdt)

        # A preconditioner that is NOT a device matrix (a callable such as an incomplete-factorisation solve, a
        # composite operator): the Gram-Schmidt part of the step does not depend on M at all - coefficients against
        # V_j, updates with P_j - so it runs in the fused step with a UNIT diagonal in M's place, which leaves
        # p' = w / ||w||_2 in column k+1; M is then applied once to p' and the column pair is rescaled:
        #   s = sqrt(<p', M p'>),  P_{k+1} = p'/s,  V_{k+1} = M p'/s,  H[k+1,k] = ||w||_2 s
        # (= sqrt(<w, M w>), w / H[k+1,k], M w / H[k+1,k] of utils.py:1030-1045 up to rounding).  Three host
        # synchronisations per step instead of k+3; Euclidean inner product, real or complex data.
        self._Mext = None
        if self.M is not None and self._Md is None and self._euclid and ortho in _GS_OF_ORTHO and not self._win:
            self._Mext = ctx.diag(numpy.ones(N), dtype=bdt)

        v = _as_dvec(v, ctx, dtype=bdt)
        if ortho == "house":
            self.houses = [_DevHouse(ctx, self._Hv, 0, v.block, v.col)]
            self.vnorm = norm(v)
        elif self.M is not None:
            p = v
            v = (self.M * p) if Mv is None else _as_dvec(Mv, ctx, dtype=bdt)
            self.vnorm = norm(p, v, ip_B=ip_B) if Mv_norm is None else Mv_norm
            if self.vnorm > 0:
                ctx.vdiv(self._P, 0, p.block, p.col, float(self.vnorm))
        else:
            self.vnorm = norm(v, ip_B=ip_B) if Mv_norm is None else Mv_norm
        if self.vnorm > 0:
            ctx.vdiv(self._V, 0, v.block, v.col, float(self.vnorm))
        else:
            self.invariant = True
        if self._ipB is not None and ortho != "house":
            self._BV = ctx.alloc(N, self._cols, dtype=bdt, zero=False)
            if self.vnorm > 0:
                ctx.apply(self._ipB, self._V, 0, self._BV, 0, 1)
            else:                    # v = 0: column 0 of V was never written (zero=False) - nothing uninitialised is
                self._V.zero(0, 1)   # multiplied into BV or handed back to the block pool
                self._BV.zero(0, 1)

    _WINDOW_COLS = 66        # columns of the sliding Lanczos window (re-based every 64 steps)
    _MATRIX_M_EXTERNAL_FROM = 1_500_000   # vector length from which a matrix preconditioner is applied outside the step
    _BASIS_SHARE = 0.30      # share of device memory the first allocation of V (and P) may take
    _max_initial_cols = None  # tests: force a small first allocation to exercise _grow()

    def _window_ok(self, ctx, bdt):
        """The window lives in the look-ahead path only: Euclidean inner product, plain device
        matrix (possibly behind a device projector), M absent or diagonal, real data."""
        if _is_c(bdt) or not (self.ip_B is None or isinstance(self.ip_B, IdentityLinearOperator)):
            return False
        if self.M is not None:
            md = self.M._device_matrix()
            if md is None or md.kind != "diag":
                return False
        if self.A._device_matrix(ctx, bdt) is not None:
            return True
        if isinstance(self.A, _ProductLinearOperator):
            Pop, inner = self.A.args
            return getattr(Pop, "_kh_proj", None) is not None and inner._device_matrix() is not None
        return False

    @classmethod
    def _initial_cols(cls, ctx, N, want, nblocks, bdt):
        if cls._max_initial_cols is not None:
            return max(2, min(want, cls._max_initial_cols))
        try:
            total = ctx.info()["mem_total"]
        except Exception:
            total = 0
        if total <= 0:
            return want
        per_col = 8 * (2 if _is_c(bdt) else 1) * max(N, 1) * nblocks
        fit = int(cls._BASIS_SHARE * total // per_col)
        return want if want <= fit else max(16, fit)

    def _grow(self, need):
        """Make room for ``need`` basis columns: double the blocks (at most maxiter+1 columns) and copy
        what has been computed.  Steps in flight are discarded first (they are re-enqueued)."""
        self._settle()
        if self._win:
            # slide: the two columns the next step reads (k-1, k) move to the front
            k = self.iter
            src = max(k - 1, 0) - self._base
            if src > 0:
                for blk in (self._V, self._P):
                    if blk is not None:
                        blk.copy_from(0, blk, src, 2 if k >= 1 else 1)
                self._base = max(k - 1, 0)
            return
        cols = min(self.maxiter + 1, max(need, 2 * self._cols))
        done = self.iter + 1
        # a deferred MINRES recurrence update of the previous iteration still names a column of the old block
        self._ctx.minres_flush()
        for name in ("_V", "_P", "_BV"):
            old = getattr(self, name)
            if old is None:
                continue
            new = self._ctx.alloc(old.n, cols, dtype=old.dtype, zero=False)
            new.copy_from(0, old, 0, min(done, old.ncols))
            setattr(self, name, new)
        self._cols = cols

    def _padded(self, block):
        """Host copy with the reference's (N, maxiter+1) shape: the computed columns, zeros behind them
        (the device block is not zero-filled when it is recycled, and may be shorter than maxiter+1)."""
        done = min(self.iter + 1, block.ncols)
        if self.invariant:
            done = min(self.iter, block.ncols)      # the reference leaves the last column untouched
        full = numpy.zeros((block.n, self.maxiter + 1), dtype=block.dtype, order="F")
        if done > 0:
            full[:, :done] = block.download(0, done)
        return full

    # the reference exposes ndarrays; here they are downloaded on demand
    @property
    def V(self):
        if self._win:
            raise AttributeError("V: this Lanczos run keeps a sliding window, not the basis")
        self._settle()
        return self._padded(self._V)

    @property
    def P(self):
        if self._P is None:
            raise AttributeError("P")
        if self._win:
            raise AttributeError("P: this Lanczos run keeps a sliding window, not the basis")
        self._settle()
        return self._padded(self._P)

    def _claim(self, slot):
        """The H-column slots (device column, pinned copy, event) belong to the context, four of them,
        keyed by step number mod 4.  Before this basis uses one, any OTHER basis that still has an
        unfetched step parked there (two Arnoldi objects advanced alternately, a solver started while
        another one has a look-ahead step in flight) settles first: its speculative steps are waited
        for, discarded and re-enqueued at its next ``advance()`` - it never reads a foreign column."""
        owners = self._ctx.__dict__.setdefault("_slot_owner", [None] * 4)
        ref = owners[slot]
        other = ref() if ref is not None else None
        if other is not None and other is not self and other._enq > other.iter:
            other._settle()
        owners[slot] = weakref.ref(self)

    def _release(self, slot):
        owners = self._ctx.__dict__.get("_slot_owner")
        if owners is not None and owners[slot] is not None and owners[slot]() is self:
            owners[slot] = None

    def _begin(self):
        """Enqueue Arnoldi step ``self._enq`` on the device (no host synchronisation)."""
        k = self._enq
        self._claim(k % 4)
        start, h_km1 = 0, 0.0
        if self.ortho == "lanczos":
            start = k
            if k > 0:
                # the previous step has been begun but maybe not fetched yet: NaN tells the library
                # to read H[k,k-1] from that step's device-side H column
                h_km1 = float(numpy.real(self.H[k, k - 1])) if self.iter >= k else float("nan")
        self._ctx.arnoldi_step_begin(self._Amat, self._Md, self._V, self._P, self._W, 0, k - self._base,
                                     start - self._base if start else 0, self._sweeps, self._gs_mode,
                                     h_km1, k % 4, proj=self._proj)
        self._enq = k + 1

    def _settle(self):
        """Discard speculative steps: wait for them and clear the basis columns they wrote, so
        that the arrays look exactly like the reference's (untouched columns are zero)."""
        while self._enq > self.iter:
            k = self._enq - 1
            kp = k - self._base
            self._ctx.arnoldi_step_end(k % 4, kp + 2 + (self._proj.d if self._proj is not None else 0),
                                       cplx=self._cplx)
            self._release(k % 4)
            self._V.zero(kp + 1, 1)
            if self._P is not None:
                self._P.zero(kp + 1, 1)
            self._enq = k

    def advance(self):
        """Carry out one iteration of Arnoldi (utils.py:954-1048)."""
        if self.iter >= self.maxiter:
            raise ArgumentError("Maximum number of iterations reached.")
        if self.invariant:
            raise ArgumentError("Krylov subspace was found to be invariant in the previous iteration.")
        k = self.iter
        ctx = self._ctx
        H = self.H
        start = 0
        h_km1 = 0.0
        need = min(k + self._lookahead, self.maxiter - 1) + 2 - self._base
        if need > self._cols:
            self._grow(need)
        if self.ortho == "lanczos":
            start = k
            if k > 0:
                hv = H[k, k - 1]
                H[k - 1, k] = hv
                h_km1 = float(hv.real)
        if self._fused:
            if self._lookahead:
                last = min(k + self._lookahead, self.maxiter - 1)
                while self._enq <= last:
                    self._begin()
                pd = self._proj.d if self._proj is not None else 0
                kp = k - self._base
                hcol = ctx.arnoldi_step_end(k % 4, kp + 2 + pd, cplx=self._cplx)
                self._release(k % 4)
                if pd:
                    cb = self._on_ya()          # (weak: the deflated solver owns the projected operator, not the reverse)
                    if cb is None:
                        raise RuntimeError("Arnoldi on a deflated solver's projected operator outlived the solver: "
                                           "keep the solver alive while its Arnoldi object / MlAMr is in use")
                    cb(hcol[kp + 2:].reshape(-1, 1).copy())
                    hcol = hcol[: kp + 2]

            elif self._Amat is not None:
                self._claim(0)       # the one-call step runs through slot 0
                hcol = ctx.arnoldi_step(self._Amat, self._Md, self._V, self._P, self._W, 0, k, start,
                                        self._sweeps, self._gs_mode, h_km1)
                self._release(0)
            else:
                self.A._apply_dev(self._V, k, self._W, 0, 1)
                self._claim(0)
                hcol = ctx.arnoldi_step(None, self._Md, self._V, self._P, self._W, 0, k, start,
                                        self._sweeps, self._gs_mode, h_km1)
                self._release(0)
            if self.ortho == "lanczos" and hcol.dtype.kind == "c":
                hcol = hcol.real       # alpha = real(alpha), utils.py:1024-1027
            off = self._base          # (window: the column arrives in window coordinates)
            if start == k:            # (Lanczos: one entry - no slices; an iteration at short vectors is one 18 us launch)
                H[k, k] += hcol[k - off]
            else:
                H[start: k + 1, k] += hcol[start - off: k + 1 - off]
            hn = float(hcol[k + 1 - off].real)
        elif self.ortho == "house":
            hn = self._advance_house(k)
        elif self._BV is not None:
            # inner product matrix on the device: dots against BV, updates with V, norm sqrt(<w, B w>)
            self.A._apply_dev(self._V, k, self._W, 0, 1)
            self._claim(0)
            hcol = ctx.arnoldi_step(None, self._ipB, self._BV, self._V, self._W, 0, k, start, self._sweeps,
                                    self._gs_mode, h_km1)
            self._release(0)
            H[start: k + 1, k] += hcol[start: k + 1]
            hn = float(hcol[k + 1])
        elif self._Mext is not None:
            hn = self._advance_external_m(k, start, h_km1)
        else:
            hn = self._advance_general(k, start, h_km1)
        H[k + 1, k] = hn
        # invariance test  H[k+1,k] / ||H[:k+2,:k+1]||_2 <= 1e-14  (utils.py:1035-1039).
        # ||.||_2 <= ||.||_F, so the quotient by the Frobenius norm is a lower bound: only when
        # THAT is tiny is the exact 2-norm (an O(k^3) SVD) needed.  Same decisions, O(k) cost.
        col = H[: k + 2, k]
        self._h2 += float(numpy.vdot(col, col).real)
        fro = numpy.sqrt(self._h2)
        is_inv = False
        if not (fro > 0) or not (hn / fro > 1e-14):
            nrm2 = numpy.linalg.norm(H[: k + 2, : k + 1], 2)
            is_inv = not (hn / nrm2 > 1e-14) if nrm2 > 0 else True
        self.iter += 1
        if is_inv:
            self.invariant = True
            # the reference leaves column k+1 untouched (zeros); undo the stores of this step and
            # of any step enqueued ahead of it
            self._settle()
            self._V.zero(k + 1 - self._base, 1)
            if self._P is not None:
                self._P.zero(k + 1 - self._base, 1)
            if self._BV is not None:
                self._BV.zero(k + 1 - self._base, 1)      # (0 / 0: nothing non-finite goes back to the block pool)

    def _advance_house(self, k):
        """One Householder Arnoldi step (utils.py:970-994) on the device."""
        ctx, V, W, H, N = self._ctx, self._V, self._W, self.H, self._V.n
        self.A._apply_dev(V, k, W, 0, 1)
        for j in range(k + 1):
            hj = self.houses[j]
            hj.apply(W, 0)
            if hj.alpha != 1.0:                          # Av[j] *= conj(alpha_j)
                W.set(0, j, W.get(0, j, 1) * numpy.conj(hj.alpha))
        if k + 1 < N:
            house = _DevHouse(ctx, self._Hv, k + 1, W, 0)
            self.houses.append(house)
            house.apply(W, 0)
            col = W.get(0, 0, k + 2)
            col[k + 1] *= numpy.conj(house.alpha)
            H[: k + 2, k] = col
            hn = abs(H[k + 1, k])
        else:
            H[: k + 1, k] = W.get(0, 0, k + 1)
            hn = 0.0
        if k + 1 < N and hn > 0:
            # v_{k+1} = H_0 ... H_{k+1} e_{k+1} * alpha_{k+1}   (utils.py:990-994)
            V.zero(k + 1, 1)
            V.set(k + 1, k + 1, [1.0])
            for j in range(k + 1, -1, -1):
                self.houses[j].apply(V, k + 1)
            a = self.houses[-1].alpha
            if a != 1.0:
                a = float(numpy.real(a)) if numpy.imag(a) == 0 else complex(a)
                ctx.waxpby(V, k + 1, a, V, k + 1, 0.0, V, k + 1)
        return hn

    def _advance_external_m(self, k, start, h_km1):
        """One step with a preconditioner that is a callable / composite operator (see __init__)."""
        ctx = self._ctx
        V, P, W, H = self._V, self._P, self._W, self.H
        self.A._apply_dev(V, k, W, 0, 1)
        self._claim(0)
        hcol = ctx.arnoldi_step(None, self._Mext, V, P, W, 0, k, start, self._sweeps, self._gs_mode, h_km1)
        self._release(0)
        if self.ortho == "lanczos" and hcol.dtype.kind == "c":
            hcol = hcol.real                         # alpha = real(alpha), utils.py:1024-1027
        H[start: k + 1, k] += hcol[start: k + 1]
        h2 = float(numpy.real(hcol[k + 1]))          # ||w||_2
        if not (h2 > 0.0):
            return 0.0                               # w = 0: invariant (the caller clears column k+1)
        self.M._apply_dev(P, k + 1, W, 1, 1)         # M p'
        s2 = _inner_dev(P, k + 1, 1, W, 1, 1, None)[0, 0]
        # the reference's norm(Av, MAv) (utils.py:226-238): sqrt of the MODULUS - an indefinite M is not reported -
        # but an inner product with an imaginary part is
        if abs(numpy.imag(s2)) > abs(s2) * 1e-10:
            raise InnerProductError("inner product defined by ip_B not positive definite? "
                                    "||diag(ip).imag||/||diag(ip)||={}".format(abs(numpy.imag(s2)) / abs(s2)))
        s1 = float(numpy.sqrt(abs(s2)))
        if not (s1 > 0.0):
            return 0.0
        ctx.vdiv(P, k + 1, P, k + 1, s1)
        ctx.vdiv(V, k + 1, W, 1, s1)
        return h2 * s1

    def _advance_general(self, k, start, h_km1):
        """Arnoldi step for a non-Euclidean inner product or a general preconditioner ``M``:
        same loop as the reference (utils.py:1011-1045) with each vector operation on the device."""
        ctx = self._ctx
        V, P, W, H = self._V, self._P, self._W, self.H
        B = P if self.M is not None else V
        self.A._apply_dev(V, k, W, 0, 1)
        if start > 0:
            ctx.axpy_panel(B, k - 1, 1, [h_km1], W, 0)
        for _ in range(self._sweeps):
            for j in range(start, k + 1):
                alpha = _inner_dev(V, j, 1, W, 0, 1, self.ip_B)[0, 0]
                if self.ortho == "lanczos":
                    if abs(numpy.imag(alpha)) > 1e-10:
                        warnings.warn(
                            f"Iter {self.iter}: abs(alpha.imag) = {abs(alpha.imag)} > 1e-10. "
                            "Is your operator self-adjoint in the provided inner product?")
                    alpha = numpy.real(alpha)
                H[j, k] += alpha
                ctx.axpy_panel(B, j, 1, [alpha], W, 0)
        if self.M is not None:
            self.M._apply_dev(W, 0, W, 1, 1)
            ip = _inner_dev(W, 0, 1, W, 1, 1, self.ip_B)
        else:
            ip = _inner_dev(W, 0, 1, W, 0, 1, self.ip_B)
        hn = float(numpy.sqrt(numpy.linalg.norm(ip, 2)))
        if hn > 0:
            if self.M is not None:
                ctx.vdiv(P, k + 1, W, 0, hn)
                ctx.vdiv(V, k + 1, W, 1, hn)
            else:
                ctx.vdiv(V, k + 1, W, 0, hn)
        return hn

    def get(self):
        """``(V, H[, P])`` trimmed to the computed part (utils.py:1050-1061)."""
        k = self.iter
        if self._win:
            raise ArgumentError("this Lanczos run keeps a sliding window, not the basis (get() needs it)")
        self._settle()
        nv, hr = (k, k) if self.invariant else (k + 1, k + 1)
        V, H = self._V.download(0, nv), self.H[:hr, :k]
        if self.M is not None:
            return V, H, self._P.download(0, nv)
        return V, H

    def get_last(self):
        """Last Arnoldi vector and Hessenberg column (utils.py:1063-1074)."""
        k = self.iter
        if self.invariant:
            V, H = None, self.H[:k, [k - 1]]
            return (V, H, None) if self.M is not None else (V, H)
        V, H = self._V.download(k - self._base, 1), self.H[: k + 1, [k - 1]]
        if self.M is not None:
            return V, H, self._P.download(k - self._base, 1)
        return V, H


def arnoldi(*args, **kwargs):
    """Run Arnoldi to ``maxiter`` or invariance and return ``get()`` (utils.py:1077-1081)."""
    _arnoldi = Arnoldi(*args, **kwargs)
    while _arnoldi.iter < _arnoldi.maxiter and not _arnoldi.invariant:
        _arnoldi.advance()
    return _arnoldi.get()


def ritz(H, V=None, hermitian=False, type="ritz"):
    """Ritz, harmonic Ritz or improved harmonic Ritz pairs of an Arnoldi/Lanczos relation
    (utils.py:1171-1286).

    :param H: Hessenberg matrix ``(n+1, n)`` or ``(n, n)``.
    :param V: (optional) Arnoldi vectors, ``(N, n+1)`` host array or device block; then the Ritz
      vectors ``Z = V[:, :n] U`` are returned too (formed on the device).
    :param hermitian: use ``eigh`` (``H[:n, :]`` must be Hermitian).
    :param type: ``'ritz'`` | ``'harmonic'`` | ``'harmonic_improved'``.
    :return: ``theta, U, resnorm`` and ``Z`` if ``V`` is given.

    The eigenproblem is n x n host work; only ``V[:, :n] @ U`` touches N-vectors.
    """
    n = H.shape[1]
    nv = None if V is None else (V.ncols if hasattr(V, "ncols") else V.shape[1])
    if V is not None and nv != H.shape[0]:
        raise ArgumentError("shape mismatch with V and H")
    if not H.shape[0] in [n, n + 1]:
        raise ArgumentError("H not of shape (n+1,n) or (n,n)")
    Hn = H[:n, :]
    symmres = numpy.linalg.norm(Hn - Hn.T.conj())
    if hermitian and symmres >= 5e-14:
        warnings.warn(f"Hessenberg matrix is not symmetric: |H-H^*|={symmres}")
    eig = scipy.linalg.eigh if hermitian else scipy.linalg.eig

    def residuals(theta, U):
        res = []
        for i in range(n):
            resi = numpy.array(numpy.dot(H, U[:, i]), dtype=numpy.result_type(U.dtype, theta.dtype))
            resi[:n] -= theta[i] * U[:, i]
            res.append(numpy.linalg.norm(resi, 2))
        return numpy.array(res)

    if type == "ritz":
        theta, U = eig(Hn)
        beta = 0 if H.shape[0] == n else H[-1, -1]
        resnorm = numpy.abs(beta * U[-1, :])
    elif type in ("harmonic", "harmonic_improved"):
        theta, U = eig(Hn.T.conj(), numpy.dot(H.T.conj(), H))
        for i in range(n):
            U[:, i] /= numpy.linalg.norm(U[:, i], 2)
        if type == "harmonic":
            theta = 1 / theta
        else:
            theta = numpy.array([numpy.dot(U[:, i].T.conj(), numpy.dot(Hn, U[:, i]))
                                 for i in range(n)])
        resnorm = residuals(theta, U)
    else:
        raise ArgumentError(f"unknown Ritz type {type}")
    if V is None:
        return theta, U, resnorm
    Ur = U
    Vd = _upload_block(V)
    if numpy.iscomplexobj(U) and n > 0:
        if numpy.abs(U.imag).max() > 1e-12 * max(numpy.abs(U).max(), 1e-300):
            Vd = _promote_block(Vd, 0, n, numpy.complex128)[0]   # complex Ritz vectors
        else:
            Ur = U.real
    Z = Vd.ctx.alloc(Vd.n, n, dtype=Vd.dtype)
    if n > 0:
        Vd.ctx.gemm_nn(Vd, 0, n, Ur, 1.0, 0.0, Z, 0)
    return theta, U, resnorm, numpy.ascontiguousarray(Z.download())


# ----------------------------------------------------------------------------------------
# QR and projections (utils.py:439-707)
# ----------------------------------------------------------------------------------------
def _qr_mgs_fused(Q, reorthos):
    """Euclidean MGS-QR with one fused device call per column: column i is orthogonalised against
    columns 0..i-1 in the reference's order, normed and normalised in place by the same kernels
    that run an Arnoldi step (``kh_arnoldi_step`` with no operator).  Returns None if a column
    turns out (numerically) dependent: the caller then takes the step-by-step path, which honours
    the reference's ``R[i,i] >= 1e-15`` guard exactly."""
    ctx = Q.ctx
    k = Q.ncols
    R = numpy.zeros((k, k), dtype=Q.dtype)
    for i in range(k):
        if i == 0:
            r00 = ctx.nrm2(Q, 0)
            R[0, 0] = r00
            if not r00 >= 1e-15:
                return None
            ctx.vdiv(Q, 0, Q, 0, float(r00))
            continue
        hcol = ctx.arnoldi_step(None, None, Q, None, Q, i, i - 1, 0, reorthos + 1, _hip.GS_MGS, 0.0)
        R[: i + 1, i] = hcol[: i + 1]
        if not hcol[i].real >= 1e-15:
            return None
    return R


def _qr_mgs_dev(Q, ip_B, reorthos, source=None):
    """In-place modified Gram-Schmidt of the device block ``Q`` (utils.py:694-707).  ``source``: a device block that still
    holds the columns ``Q`` was copied from (utils.qr's input) - the fused path's way back when it meets a dependent column;
    without one a copy of ``Q`` is kept for that."""
    ctx = Q.ctx
    k = Q.ncols
    if (ip_B is None or isinstance(ip_B, IdentityLinearOperator)) and k > 1 and hasattr(ctx, "arnoldi_step"):
        keep = source
        if keep is None:
            keep = ctx.alloc(Q.n, k, dtype=Q.dtype, zero=False)
            keep.copy_from(0, Q, 0, k)
        R = _qr_mgs_fused(Q, reorthos)
        if R is not None:
            return R
        Q.copy_from(0, keep, 0, k)      # dependent column: redo with the guarded path
    R = numpy.zeros((k, k), dtype=Q.dtype)
    for i in range(k):
        for _ in range(reorthos + 1):
            for j in range(i):
                alpha = _inner_dev(Q, j, 1, Q, i, 1, ip_B)[0, 0]
                R[j, i] += alpha
                ctx.axpy_panel(Q, j, 1, [alpha], Q, i)
        R[i, i] = numpy.sqrt(numpy.linalg.norm(_inner_dev(Q, i, 1, Q, i, 1, ip_B), 2))
        if R[i, i].real >= 1e-15:
            ctx.vdiv(Q, i, Q, i, float(R[i, i].real))
    return R


def qr(X, ip_B=None, reorthos=1):
    """QR factorization with customizable inner product (utils.py:680-707).

    ``ip_B is None`` -> ``scipy.linalg.qr(X, mode='economic')`` on the host array, exactly the
    reference's call (a LAPACK call there as well, and not on the Krylov hot path).  Any other
    ``ip_B`` - including an ``IdentityLinearOperator`` *instance*, which is what the deflated
    solvers pass - runs ``reorthos+1`` sweeps of modified Gram-Schmidt on the device.

    ``X`` may be a host ``(N,k)`` array (returns host ``Q``) or a device block (returns a
    device block).
    """
    on_device = hasattr(X, "download") and hasattr(X, "ncols")
    if ip_B is None and not on_device and X.shape[1] > 0:
        return scipy.linalg.qr(X, mode="economic")
    if on_device:
        Q = X.ctx.alloc(X.n, X.ncols, dtype=X.dtype, zero=False)      # (every column is written by the copy, padding included)
        Q.copy_from(0, X, 0, X.ncols)
    else:
        Q = _hip.get_context().upload(numpy.asarray(X))
    R = _qr_mgs_dev(Q, ip_B, reorthos, source=X if on_device else None)
    if on_device:
        return Q, R
    return numpy.ascontiguousarray(Q.download()), R


class Projection(object):
    def __init__(self, X, Y=None, ip_B=None, orthogonalize=True, iterations=2):
        """Generic (oblique) projection :math:`P_{\\mathcal{X},\\mathcal{Y}^\\perp}`
        (utils.py:440-520) with device-resident bases.

        ``X`` / ``Y``: ``(N,k)`` host arrays or device blocks.  The XQRY representation of the
        reference is kept: ``V,VR = qr(X)``, ``W,WR = qr(Y)``, ``Q,R = qr(<W,V>)`` (the last
        one a k x k host factorisation).  Attributes ``V``/``W`` download on access.
        """
        self.ip_B = ip_B
        if iterations < 1:
            raise ArgumentError("iterations < 1 not allowed")
        self.orthogonalize = orthogonalize
        self.iterations = iterations
        Y = X if Y is None else Y
        xs = (X.n, X.ncols) if hasattr(X, "ncols") else X.shape
        ys = (Y.n, Y.ncols) if hasattr(Y, "ncols") else Y.shape
        if len(xs) != 2:
            raise ArgumentError("X does not have shape==(N,k)")
        if xs != ys:
            raise ArgumentError("X and Y have different shapes")
        self._N, self._k = xs
        self.VR = self.WR = self.Q = self.R = None
        self._Vd = self._Wd = None
        if self._k == 0:
            return
        same = Y is X
        Xd = _upload_block(X)
        if orthogonalize:
            if ip_B is None:
                # reference: Householder QR of the host array (utils.py:692-693)
                Qh, self.VR = scipy.linalg.qr(Xd.download(), mode="economic")
                self._Vd = Xd.ctx.upload(Qh)
            else:
                self._Vd, self.VR = qr(Xd, ip_B=ip_B)
        else:
            self._Vd = Xd
        if same and orthogonalize:
            self._Wd, self.WR = self._Vd, self.VR
        else:
            Yd = _upload_block(Y, Xd.ctx)
            if Yd.dtype != Xd.dtype:      # mixed real / complex bases: widen the real one
                dt = _bdt(Xd.dtype, Yd.dtype)
                Yd = _promote_block(Yd, 0, Yd.ncols, dt)[0]
                self._Vd = _promote_block(self._Vd, 0, self._Vd.ncols, dt)[0]
            if orthogonalize:
                if ip_B is None:
                    Qh, self.WR = scipy.linalg.qr(Yd.download(), mode="economic")
                    self._Wd = Yd.ctx.upload(Qh)
                else:
                    self._Wd, self.WR = qr(Yd, ip_B=ip_B)
            else:
                self._Wd = Yd
            M = _inner_dev(self._Wd, 0, self._k, self._Vd, 0, self._k, ip_B)
            self.Q, self.R = scipy.linalg.qr(M)

    @property
    def V(self):
        return numpy.zeros((self._N, 0)) if self._k == 0 else self._Vd.download()

    @property
    def W(self):
        return numpy.zeros((self._N, 0)) if self._k == 0 else self._Wd.download()

    # -- device primitives ------------------------------------------------------------
    def _coeffs(self, a, want_Ya):
        """c = R^{-1} Q^H <W, a>  and  Ya = WR^H <W, a>  (utils.py:540-548)."""
        c = _inner_dev(self._Wd, 0, self._k, a.block, a.col, 1, self.ip_B)
        Ya = None
        if want_Ya:
            Ya = c.copy()
            if self.WR is not None:
                Ya = self.WR.T.conj().dot(Ya)
        if self.Q is not None and self.R is not None:
            c = scipy.linalg.solve_triangular(self.R, self.Q.T.conj().dot(c))
        return c, Ya

    def _apply_dvec(self, a, return_Ya=False):
        """Single application ``V c`` for a :class:`DVec` (utils.py:522-552)."""
        a = a.astype(self._Vd.dtype)
        c, Ya = self._coeffs(a, return_Ya)
        out = a.ctx.alloc(self._N, 1, dtype=a.dtype)
        Vd = _promote_block(self._Vd, 0, self._k, a.dtype)[0]
        a.ctx.gemm_nn(Vd, 0, self._k, c, 1.0, 0.0, out, 0)
        return (DVec(out), Ya) if return_Ya else DVec(out)

    def _device_projector(self):
        """The ``kh_proj`` image of this projection (Euclidean inner product only), or None.

        ``T = R^{-1} Q^H`` is formed once on the host (a d x d triangular solve, d ~ 16) so that a
        whole application - W^T z, T c, z -= V c, twice - runs on the device without a host round
        trip; ``WR^H`` maps the first sweep's coefficients to ``<Y, a>``."""
        if self._k == 0 or not (self.ip_B is None or isinstance(self.ip_B, IdentityLinearOperator)):
            return None
        if _is_c(self._Vd.dtype) != _is_c(self._Wd.dtype):
            return None     # one real and one complex block: the sweep-by-sweep path promotes what it needs
        if _is_c(self._Vd.dtype) and self._k > 512:
            return None     # (kh_zproj_create takes at most 512 vectors)
        if "_kh_proj" not in self.__dict__:
            T = None
            if self.Q is not None and self.R is not None:
                T = scipy.linalg.solve_triangular(self.R, self.Q.T.conj())
            WRH = None if self.WR is None else self.WR.T.conj()
            self._kh_proj = self._Vd.ctx.proj_create(self._Wd, self._Vd, self._k, T, WRH,
                                                     self.iterations)
        return self._kh_proj

    def _apply_complement_dvec(self, a, return_Ya=False):
        """``z = a - P a`` with ``iterations-1`` refinement sweeps (utils.py:604-627).

        Each sweep is one tall-skinny ``W^T z`` panel product and one ``z -= V c`` panel
        update on the device; ``z`` is a fresh vector."""
        ctx = a.ctx
        if self._k == 0:
            z = a.copy()
            return (z, numpy.zeros((0, 1))) if return_Ya else z
        a = a.astype(self._Vd.dtype)
        proj = self._device_projector()
        if proj is not None and _is_c(a.dtype) == bool(getattr(proj, "cplx", False)):
            z = DVec(ctx.alloc(self._N, 1, dtype=a.dtype))
            Ya = ctx.proj_apply_complement(proj, a.block, a.col, z.block, z.col, want_ya=return_Ya)
            return (z, Ya.reshape(-1, 1)) if return_Ya else z
        z = a.copy()
        Vd = _promote_block(self._Vd, 0, self._k, a.dtype)[0]
        c, Ya = self._coeffs(a, return_Ya)
        ctx.axpy_panel(Vd, 0, self._k, c, z.block, z.col)
        for _ in range(self.iterations - 1):
            c, _unused = self._coeffs(z, False)
            ctx.axpy_panel(Vd, 0, self._k, c, z.block, z.col)
        return (z, Ya) if return_Ya else z

    # -- host API (arrays in, arrays out) ---------------------------------------------
    def _columns(self, a, fn):
        a = numpy.asarray(a)
        blk = _hip.get_context().upload(a)
        outs = [fn(DVec(blk, j)) for j in range(a.shape[1])]
        return outs

    def _apply(self, a, return_Ya=False):
        if self._k == 0:
            Pa = numpy.zeros(a.shape)
            return (Pa, numpy.zeros((0, a.shape[1]))) if return_Ya else Pa
        res = self._columns(a, lambda v: self._apply_dvec(v, return_Ya))
        if return_Ya:
            return (numpy.column_stack([r[0].download() for r in res]),
                    numpy.column_stack([r[1] for r in res]))
        return numpy.column_stack([r.download() for r in res])

    def apply(self, a, return_Ya=False):
        """Apply the projection, ``iterations`` sweeps (utils.py:566-591)."""
        if self._k == 0:
            Pa = numpy.zeros(a.shape)
            return (Pa, numpy.zeros((0, a.shape[1]))) if return_Ya else Pa
        ctx = _hip.get_context()

        def one(v):
            v = v.astype(_bdt(self._Vd.dtype, self._Wd.dtype))
            if return_Ya:
                x, Ya = self._apply_dvec(v, True)
            else:
                x, Ya = self._apply_dvec(v), None
            for _ in range(self.iterations - 1):
                z = v.copy()
                ctx.waxpby(z.block, z.col, 1.0, v.block, v.col, -1.0, x.block, x.col)
                w = self._apply_dvec(z)
                ctx.waxpby(x.block, x.col, 1.0, x.block, x.col, 1.0, w.block, w.col)
            return x, Ya
        res = self._columns(a, one)
        X = numpy.column_stack([r[0].download() for r in res])
        if return_Ya:
            return X, numpy.column_stack([r[1] for r in res])
        return X

    def apply_complement(self, a, return_Ya=False):
        """Apply the complementary projection ``a - P a`` (utils.py:604-627)."""
        if self._k == 0:
            return (a.copy(), numpy.zeros((0, a.shape[1]))) if return_Ya else a.copy()
        res = self._columns(a, lambda v: self._apply_complement_dvec(v, return_Ya))
        if return_Ya:
            return (numpy.column_stack([r[0].download() for r in res]),
                    numpy.column_stack([r[1] for r in res]))
        return numpy.column_stack([r.download() for r in res])

    # -- adjoint projection (utils.py:554-564, 593-603, 629-638) ------------------------------
    def _apply_adj_dvec(self, a):
        """Single application of the adjoint: ``W Q R^{-H} <V, a>``."""
        a = a.astype(_bdt(self._Vd.dtype, self._Wd.dtype))
        c = _inner_dev(self._Vd, 0, self._k, a.block, a.col, 1, self.ip_B)
        if self.Q is not None and self.R is not None:
            c = self.Q.dot(scipy.linalg.solve_triangular(self.R.T.conj(), c, lower=True))
        out = a.ctx.alloc(self._N, 1, dtype=a.dtype)
        Wd = _promote_block(self._Wd, 0, self._k, a.dtype)[0]
        a.ctx.gemm_nn(Wd, 0, self._k, c, 1.0, 0.0, out, 0)
        return DVec(out)

    def _axpy(self, z, alpha, x):
        """z += alpha * x for device vectors (real alpha)."""
        z.ctx.waxpby(z.block, z.col, 1.0, z.block, z.col, alpha, x.block, x.col)

    def apply_adj(self, a):
        """Apply the adjoint projection, ``iterations`` sweeps (utils.py:593-603)."""
        if self._k == 0:
            return numpy.zeros(a.shape)

        def one(v):
            v = v.astype(_bdt(self._Vd.dtype, self._Wd.dtype))
            x = self._apply_adj_dvec(v)
            for _ in range(self.iterations - 1):
                z = v.copy()
                self._axpy(z, -1.0, x)
                self._axpy(x, 1.0, self._apply_adj_dvec(z))
            return x
        return numpy.column_stack([r.download() for r in self._columns(a, one)])

    def apply_complement_adj(self, a):
        """Apply the adjoint of the complementary projection (utils.py:629-638)."""
        if self._k == 0:
            return a.copy()

        def one(v):
            v = v.astype(_bdt(self._Vd.dtype, self._Wd.dtype))
            z = v.copy()
            self._axpy(z, -1.0, self._apply_adj_dvec(v))
            for _ in range(self.iterations - 1):
                self._axpy(z, -1.0, self._apply_adj_dvec(z))
            return z
        return numpy.column_stack([r.download() for r in self._columns(a, one)])

    def _get_operator(self, fun, fun_adj):
        dt = numpy.dtype(float) if self._Vd is None else _bdt(self._Vd.dtype, self._Wd.dtype)
        return LinearOperator((self._N, self._N), dt, fun, fun_adj)

    def operator(self):
        """``LinearOperator`` corresponding to :meth:`apply` (utils.py:645-654)."""
        if self._k == 0:
            return ZeroLinearOperator((self._N, self._N))
        return self._get_operator(self.apply, self.apply_adj)

    def operator_complement(self):
        """``LinearOperator`` corresponding to :meth:`apply_complement` (utils.py:656-665)."""
        if self._k == 0:
            return IdentityLinearOperator((self._N, self._N))
        op = self._get_operator(self.apply_complement, self.apply_complement_adj)
        op._apply_dev = self._complement_apply_dev
        return op

    def _complement_apply_dev(self, X, xcol, Y, ycol, ncols=1):
        for c in range(ncols):
            z = self._apply_complement_dvec(DVec(X, xcol + c))
            Y.copy_from(ycol + c, z.block, z.col, 1)

    def matrix(self):
        """Dense matrix of the projection - testing only (utils.py:667-677)."""
        return self.apply(numpy.eye(self._N))


# ---- names of the reference that are outside the hot path (spectral bounds, intervals,
# pseudospectrum helpers - scalar analysis on small dense data) are not rebuilt here.  Code written against krypy
# gets a clear message instead of an AttributeError.
_OUT_OF_SCOPE = ("BoundCG", "BoundMinres", "Interval", "Intervals", "NormalizedRootsPolynomial", "arnoldi_projected",
                 "bound_perturbed_gmres", "gap", "get_residual_norms", "norm_MMlr", "strakos")


def __getattr__(name):
    if name in _OUT_OF_SCOPE:
        def _stub(*args, **kwargs):
            raise NotImplementedError(
                "krypy_amd.utils.%s: krypy's convergence-bound / spectral helpers are host-side scalar analysis "
                "outside the accelerated Krylov path and are not provided (SURVEY.md section 2); use krypy.utils.%s "
                "on host arrays" % (name, name))
        _stub.__name__ = name
        return _stub
    raise AttributeError("module 'krypy_amd.utils' has no attribute %r" % name)
