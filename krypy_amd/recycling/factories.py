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

    def get(self, solver):
        ritz = deflation.Ritz(solver, mode=self.mode)
        values, which, n_vectors = ritz.values, self.which, self.n_vectors
        if which == "lm":
            indices = numpy.argsort(numpy.abs(values))[-n_vectors:]
        elif which == "sm":
            indices = numpy.argsort(numpy.abs(values))[:n_vectors]
        elif which == "lr":
            indices = numpy.argsort(numpy.real(values))[-n_vectors:]
        elif which == "sr":
            indices = numpy.argsort(numpy.real(values))[:n_vectors]
        elif which == "li":
            indices = numpy.argsort(numpy.imag(values))[-n_vectors:]
        elif which == "si":
            indices = numpy.argsort(numpy.imag(values))[:n_vectors]
        elif which == "smallest_res":
            indices = numpy.argsort(ritz.resnorms)[:n_vectors]
        else:
            raise utils.ArgumentError(
                f"Invalid value '{which}' for 'which'. "
                + "Valid are lm, sm, lr, sr, li, si and smallest_res.")
        return ritz._get_vectors_dev(indices)


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
