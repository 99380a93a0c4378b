import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krypy_amd import utils
np.set_printoptions(linewidth=200, precision=4)
for n in (10, 130, 1100, 5000):
    A = np.diag(np.linspace(1, 2, n))
    v = np.ones((n, 1))
    ar = utils.Arnoldi(A, v, maxiter=5, ortho="mgs")
    for k in range(5):
        ar.advance()
    print("n", n, "finite", np.all(np.isfinite(ar.H)), "H diag", np.diag(ar.H)[:5], "sub", np.diag(ar.H, -1)[:5])
