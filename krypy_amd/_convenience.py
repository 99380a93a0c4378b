"""``cg`` / ``minres`` / ``gmres`` convenience wrappers (mirror of ``krypy/_convenience.py``).

Same signatures, defaults and return convention: ``(x or None, solver)`` where ``x`` has the
shape of ``b`` and is ``None`` when ``solver.resnorms[-1] >= tol``; a ``ConvergenceError``
raised by the solver is *not* caught (as in the reference).  ``Mr`` is accepted and dropped,
like in the reference (``_convenience.py:48-56,112-120,177-185``).
"""
import numpy

from .deflation import DeflatedCg, DeflatedGmres, DeflatedMinres
from .linsys import Cg, Gmres, LinearSystem, Minres


def wrap_inner_product(inner):
    """Wrap a vector inner product ``inner(a, b) -> scalar`` (e.g. ``numpy.dot``) into the
    2-D-returning callable the solvers expect (``_convenience.py:10-16``)."""

    def _wrap(a, b):
        if a.shape[1] == 0:
            return numpy.array([[]])
        return numpy.array([[inner(a[:, 0], b[:, 0])]])

    return _wrap


def _prepare(A, b, inner_product, U, x0):
    assert len(A.shape) == 2
    assert A.shape[0] == A.shape[1]
    assert A.shape[1] == b.shape[0]
    if inner_product:
        inner_product = wrap_inner_product(inner_product)
    if U is not None:
        U = U.reshape(U.shape[0], -1)
    if x0 is not None:
        x0 = x0.reshape(x0.shape[0], -1)
    return inner_product, U, x0


def _result(out, b):
    return out.xk.reshape(b.shape) if out.resnorms[-1] < out.tol else None, out


def cg(A, b, M=None, Minv=None, Ml=None, Mr=None, inner_product=None, exact_solution=None,
       x0=None, U=None, tol=1e-5, maxiter=None, use_explicit_residual=False,
       store_arnoldi=False):
    inner_product, U, x0 = _prepare(A, b, inner_product, U, x0)
    linear_system = LinearSystem(A=A, b=b, M=M, Minv=Minv, Ml=Ml, ip_B=inner_product,
                                 self_adjoint=True, positive_definite=True,
                                 exact_solution=exact_solution)
    kw = dict(x0=x0, tol=tol, maxiter=maxiter, explicit_residual=use_explicit_residual,
              store_arnoldi=store_arnoldi)
    out = Cg(linear_system, **kw) if U is None else DeflatedCg(linear_system, U=U, **kw)
    return _result(out, b)


def minres(A, b, M=None, Minv=None, Ml=None, Mr=None, inner_product=None, exact_solution=None,
           ortho="mgs", x0=None, U=None, tol=1e-5, maxiter=None, use_explicit_residual=False,
           store_arnoldi=False):
    inner_product, U, x0 = _prepare(A, b, inner_product, U, x0)
    linear_system = LinearSystem(A=A, b=b, M=M, Minv=Minv, Ml=Ml, ip_B=inner_product,
                                 self_adjoint=True, exact_solution=exact_solution)
    kw = dict(ortho=ortho, x0=x0, tol=tol, maxiter=maxiter,
              explicit_residual=use_explicit_residual, store_arnoldi=store_arnoldi)
    out = Minres(linear_system, **kw) if U is None else DeflatedMinres(linear_system, U=U, **kw)
    return _result(out, b)


def gmres(A, b, M=None, Minv=None, Ml=None, Mr=None, inner_product=None, exact_solution=None,
          ortho="mgs", x0=None, U=None, tol=1e-5, maxiter=None, use_explicit_residual=False,
          store_arnoldi=False):
    inner_product, U, x0 = _prepare(A, b, inner_product, U, x0)
    linear_system = LinearSystem(A=A, b=b, M=M, Minv=Minv, Ml=Ml, ip_B=inner_product,
                                 exact_solution=exact_solution)
    kw = dict(ortho=ortho, x0=x0, tol=tol, maxiter=maxiter,
              explicit_residual=use_explicit_residual, store_arnoldi=store_arnoldi)
    out = Gmres(linear_system, **kw) if U is None else DeflatedGmres(linear_system, U=U, **kw)
    return _result(out, b)
