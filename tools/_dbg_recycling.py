import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krypy_amd import _hip, linsys, recycling
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'recycling_toy.npz'))
N = 100
d = np.linspace(1, 2, N); d[:5] = [1e-8, 1e-4, 1e-2, 2e-2, 3e-2]
ls = linsys.LinearSystem(np.diag(d), np.ones((N, 1)), normal=True, self_adjoint=True, positive_definite=True)
fac = recycling.factories.RitzFactorySimple(n_vectors=3, which='smallest_res')
rs = recycling.RecyclingGmres()
sols = [rs.solve(ls, vector_factory=fac, maxiter=50, tol=1e-5, x0=None) for _ in range(3)]
for s, row in zip(sols, (60, 61, 62)):
    print(len(s.resnorms), int(g['iters'][row]), s.resnorms[-4:], float(g['last'][row]))
    U = s.projection.U
    print('  U cols', U.shape[1], 'rayleigh', [float((U[:, j] * d).dot(U[:, j]) / U[:, j].dot(U[:, j])) for j in range(U.shape[1])])
