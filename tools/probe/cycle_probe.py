#!/usr/bin/env python
"""Does dropping the last reference to a solver hand its device blocks back at once (reference counting), or only when the
cycle collector runs?  python tools/probe/cycle_probe.py   (needs a GPU)"""
import gc
import os
import sys
import weakref

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import oracle.krylov_ref as ref  # noqa: E402
from krypy_amd import deflation, linsys, utils  # noqa: E402

A = ref.laplace2d(300, 200)
N = A.shape[0]
b = np.random.default_rng(0).standard_normal(N)
d = A.diagonal()


def build(kind):
    try:
        if kind == "minres_jacobi":
            ls = linsys.LinearSystem(A, b, M=sp.diags(1.0 / d).tocsr(), Minv=sp.diags(d).tocsr(), self_adjoint=True)
            return linsys.Minres(ls, ortho="lanczos", tol=1e-14, maxiter=30)
        ls = linsys.LinearSystem(A, b, self_adjoint=True, positive_definite=True)
        if kind == "cg":
            return linsys.Cg(ls, tol=1e-14, maxiter=30)
        if kind == "minres":
            return linsys.Minres(ls, tol=1e-14, maxiter=30)
        if kind == "gmres":
            return linsys.Gmres(ls, tol=1e-14, maxiter=30)
        if kind == "restarted":
            return linsys.RestartedGmres(ls, tol=1e-14, maxiter=10, max_restarts=2)
        if kind == "deflated":
            U = np.random.default_rng(1).standard_normal((N, 4))
            return deflation.DeflatedGmres(ls, U=U, tol=1e-14, maxiter=30)
    except utils.ConvergenceError as e:
        return e.solver


def chain(obj, depth=0, seen=None, limit=4):
    seen = seen if seen is not None else set()
    if depth > limit or id(obj) in seen:
        return
    seen.add(id(obj))
    for r in gc.get_referrers(obj):
        if r is seen or type(r).__name__ in ("frame", "list") and depth == 0:
            continue
        name = type(r).__name__
        extra = ""
        if isinstance(r, dict):
            owners = [type(o).__name__ for o in gc.get_referrers(r) if hasattr(o, "__dict__") and o.__dict__ is r]
            keys = [k for k, v in r.items() if v is obj]
            extra = " keys=%s owner=%s" % (keys[:4], owners[:2])
        print("   " * depth + "<- %s%s" % (name, extra))
        if name not in ("module", "frame"):
            chain(r, depth + 1, seen, limit)


gc.collect()
gc.disable()
for kind in ("cg", "minres", "minres_jacobi", "gmres", "restarted", "deflated"):
    s = build(kind)
    w = weakref.ref(s)
    del s
    alive = w() is not None
    print("%-14s freed by reference counting alone: %s" % (kind, not alive))
    if alive:
        chain(w(), limit=3)
        gc.collect()
gc.enable()
