"""Recycling Krylov solvers: a sequence of systems solved with deflation vectors harvested from
the previous solve (mirror of ``krypy/recycling``).

Provided: ``RecyclingCg/Minres/Gmres`` and the factories ``RitzFactorySimple`` / ``UnionFactory``.
The greedy ``RitzFactory`` with its a-priori / approximate-Krylov evaluators (convergence-bound
cost models on small dense matrices) is outside the hot path and not provided.
"""
from . import factories
from .linsys import RecyclingCg, RecyclingGmres, RecyclingMinres

__all__ = ["RecyclingCg", "RecyclingMinres", "RecyclingGmres", "factories"]
