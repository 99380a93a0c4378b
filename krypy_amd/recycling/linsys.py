"""Recycling solvers (mirror of ``krypy/recycling/linsys.py``)."""
import numpy

from .. import deflation, linsys, utils
from . import factories


class _RecyclingSolver(object):
    """Base class for recycling solvers (recycling/linsys.py:7-103)."""

    def __init__(self, DeflatedSolver, vector_factory=None):
        """:param DeflatedSolver: a deflated solver from :mod:`krypy_amd.deflation`.
        :param vector_factory: a factory from :mod:`krypy_amd.recycling.factories` that constructs
          the deflation vectors from the previous solve (``None``: no recycling).  The string
          shortcuts of the reference select its bound-based ``RitzFactory``, which is out of scope
          here and raises.

        After :meth:`solve`, the solver instance is available as ``last_solver``."""
        self._DeflatedSolver = DeflatedSolver
        self._vector_factory = vector_factory
        self.timings = utils.Timings()
        self.last_solver = None

    def solve(self, linear_system, vector_factory=None, *args, **kwargs):
        """Solve the given linear system with recycling; remaining arguments go to the
        ``DeflatedSolver``.  Returns the solver instance (approximate solution in ``xk``).
        A ``ConvergenceError`` propagates and leaves ``last_solver`` untouched, as in the reference."""
        if not isinstance(linear_system, linsys.TimedLinearSystem):
            linear_system = linsys.ConvertedTimedLinearSystem(linear_system)
        with self.timings["vector_factory"]:
            if vector_factory is None:
                vector_factory = self._vector_factory
            if isinstance(vector_factory, str):
                raise NotImplementedError(
                    "the string shortcuts select krypy's RitzFactory with convergence-bound "
                    "evaluators (out of scope, SURVEY.md section 2 rows 19-21); pass a "
                    "RitzFactorySimple instance")
            if self.last_solver is None or vector_factory is None:
                U = numpy.zeros((linear_system.N, 0))
            else:
                U = vector_factory.get(self.last_solver)
        with self.timings["solve"]:
            self.last_solver = self._DeflatedSolver(linear_system, U=U, store_arnoldi=True,
                                                    *args, **kwargs)
        return self.last_solver


class RecyclingCg(_RecyclingSolver):
    """Recycling preconditioned CG method."""

    def __init__(self, *args, **kwargs):
        super(RecyclingCg, self).__init__(deflation.DeflatedCg, *args, **kwargs)


class RecyclingMinres(_RecyclingSolver):
    """Recycling preconditioned MINRES method."""

    def __init__(self, *args, **kwargs):
        super(RecyclingMinres, self).__init__(deflation.DeflatedMinres, *args, **kwargs)


class RecyclingGmres(_RecyclingSolver):
    """Recycling preconditioned GMRES method."""

    def __init__(self, *args, **kwargs):
        super(RecyclingGmres, self).__init__(deflation.DeflatedGmres, *args, **kwargs)


__all__ = ["RecyclingCg", "RecyclingMinres", "RecyclingGmres", "factories"]
