"""Recycling Krylov solvers: a sequence of systems solved with deflation vectors harvested from
the previous solve (mirror of ``krypy/recycling``).

Provided: ``RecyclingCg/Minres/Gmres`` and the factories ``RitzFactorySimple`` / ``UnionFactory``.
The greedy ``RitzFactory`` with its a-priori / approximate-Krylov evaluators (convergence-bound
cost models on small dense matrices) is outside the hot path and not provided.
"""
from . import factories
from .linsys import RecyclingCg, RecyclingGmres, RecyclingMinres

__all__ = ["RecyclingCg", "RecyclingMinres", "RecyclingGmres", "factories"]


def __getattr__(name):
    if name in ("evaluators", "generators"):
        raise NotImplementedError(
            "krypy_amd.recycling.%s: the greedy RitzFactory with its a-priori / approximate-Krylov evaluators and "
            "subset generators (convergence-bound cost models) is outside the accelerated path and not provided "
            "(SURVEY.md section 2 rows 19-21); RitzFactorySimple and UnionFactory are" % name)
    raise AttributeError("module 'krypy_amd.recycling' has no attribute %r" % name)
