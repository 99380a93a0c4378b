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
        timed = linear_system if isinstance(linear_system, linsys.TimedLinearSystem) \
            else linsys.ConvertedTimedLinearSystem(linear_system)
        with self.timings["vector_factory"]:
            factory = self._vector_factory if vector_factory is None else vector_factory
            if isinstance(factory, str):
                raise NotImplementedError(
                    "the string shortcuts select krypy's RitzFactory with convergence-bound "
                    "evaluators (out of scope, SURVEY.md section 2 rows 19-21); pass a "
                    "RitzFactorySimple instance")
            recycle = factory is not None and self.last_solver is not None
            # first solve of a sequence / no factory: an empty basis, i.e. the plain solver
            U = factory.get(self.last_solver) if recycle else numpy.zeros((timed.N, 0))
        with self.timings["solve"]:
            solver = self._DeflatedSolver(timed, *args, U=U, store_arnoldi=True, **kwargs)
        self.last_solver = solver
        return solver


def _recycling(name, Deflated):
    """``Recycling<Method>()``: a :class:`_RecyclingSolver` bound to one deflated solver class."""

    def __init__(self, *args, **kwargs):
        _RecyclingSolver.__init__(self, Deflated, *args, **kwargs)

    return type(name, (_RecyclingSolver,), {
        "__init__": __init__, "__module__": __name__,
        "__doc__": "Recycling preconditioned %s method." % name[len("Recycling"):].upper()})


RecyclingCg = _recycling("RecyclingCg", deflation.DeflatedCg)
RecyclingMinres = _recycling("RecyclingMinres", deflation.DeflatedMinres)
RecyclingGmres = _recycling("RecyclingGmres", deflation.DeflatedGmres)


__all__ = ["RecyclingCg", "RecyclingMinres", "RecyclingGmres", "factories"]
