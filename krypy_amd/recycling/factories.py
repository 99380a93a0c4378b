"""Deflation vector factories (mirror of ``krypy/recycling/factories.py:9-18,142-208``)."""
import numpy

from .. import deflation, utils


class _DeflationVectorFactory(object):
    """Abstract base class for selectors."""

    def get(self, solver):
        """Get deflation vectors, shape ``(N,k)`` (host array or device block)."""
        raise NotImplementedError("abstract base class cannot be instanciated")


class RitzFactorySimple(_DeflationVectorFactory):
    def __init__(self, mode="ritz", n_vectors=0, which="sm"):
        """Select a fixed number of (harmonic) Ritz vectors by a criterion on the Ritz values:
        ``lm``/``sm`` largest/smallest magnitude, ``lr``/``sr`` real part, ``li``/``si`` imaginary
        part, ``smallest_res`` smallest Ritz residual norm.  The vectors are formed on the device
        (two tall-skinny GEMMs) and stay there for the next solve."""
        self.mode = mode
        self.n_vectors = n_vectors
        self.which = which

    # criterion -> (sort key of the Ritz values, take from the large end?)
    _BY_VALUE = {"lm": (numpy.abs, True), "sm": (numpy.abs, False),
                 "lr": (numpy.real, True), "sr": (numpy.real, False),
                 "li": (numpy.imag, True), "si": (numpy.imag, False)}

    def get(self, solver):
        ritz = deflation.Ritz(solver, mode=self.mode)
        n = self.n_vectors
        if self.which == "smallest_res":
            order, from_top = numpy.argsort(ritz.resnorms), False
        elif self.which in self._BY_VALUE:
            key, from_top = self._BY_VALUE[self.which]
            order = numpy.argsort(key(ritz.values))
        else:
            raise utils.ArgumentError("Invalid value '%s' for 'which'. Valid are %s and smallest_res."
                                      % (self.which, ", ".join(sorted(self._BY_VALUE))))
        # (order[-0:] would be everything: n_vectors == 0 selects nothing only from the small end,
        # exactly like the reference's slices [-n:] and [:n])
        chosen = order[-n:] if from_top else order[:n]
        return ritz._get_vectors_dev(chosen)


class UnionFactory(_DeflationVectorFactory):
    def __init__(self, factories):
        """Combine a list of factories (the union of their vectors)."""
        self._factories = factories

    def get(self, solver):
        vectors = []
        for factory in self._factories:
            v = factory.get(solver)
            vectors.append(v.download() if hasattr(v, "download") else v)
        return numpy.asarray(numpy.block(vectors))


class RitzFactory(_DeflationVectorFactory):
    def __init__(self, *args, **kwargs):
        """The reference's greedy selection driven by convergence-bound evaluators (factories.py:20-139): host-side
        cost models on small dense matrices, out of scope."""
        raise NotImplementedError(
            "RitzFactory (greedy subset selection with RitzApriori / RitzApproxKrylov evaluators) is not provided; "
            "use RitzFactorySimple(n_vectors=..., which=...) or UnionFactory")
