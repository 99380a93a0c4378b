import itertools, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krypy_amd import utils
from tests import parity_cases_utils as pcu
vs = [np.ones((10, 1)), np.eye(10, 1), (1 + 1j) * np.ones((10, 1))]
bad = 0
for (name, A, flags), as_op, (iv, v), maxiter, ortho, M, ipi in itertools.product(
        pcu.zoo_matrices(), (False, True), list(enumerate(vs)), (1, 5, 9, 10), ("mgs", "dmgs", "cgs2", "lanczos"), (None, pcu._B), range(4)):
    if ortho == "lanczos" and not flags.get("self_adjoint"):
        continue
    if ortho == "lanczos" and (ipi > 0 or M is not None):
        A_use = np.linalg.inv(pcu._B).dot(A) if ipi > 0 else A
        if M is not None:
            continue
    else:
        A_use = A
    try:
        res = utils.arnoldi(utils.MatrixLinearOperator(A_use) if as_op else A_use, v, maxiter=maxiter, ortho=ortho, M=M,
                            ip_B=pcu._ip_Bs(True)[ipi])
        H = res[1]
        if not np.all(np.isfinite(H)):
            raise ValueError("non-finite H")
    except Exception as e:
        bad += 1
        if bad < 12:
            print("FAIL", name, as_op, iv, maxiter, ortho, M is not None, ipi, type(pcu._ip_Bs(True)[ipi]), repr(e)[:80])
print("bad", bad)
