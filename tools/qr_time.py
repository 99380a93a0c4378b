import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np
from krypy_amd import _hip, utils
ctx = _hip.get_context()
n = 8_000_000
X = ctx.alloc(n, 16)
rng = np.random.default_rng(0)
for j in range(16): X.upload(j, rng.standard_normal(n))
ipI = utils.IdentityLinearOperator((n, n))
for rep in range(3):
    ctx.sync(); t0 = time.perf_counter()
    Q, R = utils.qr(X, ip_B=ipI, reorthos=1)
    ctx.sync(); print("qr fused: %.1f ms" % ((time.perf_counter() - t0) * 1e3), np.diag(R)[:3])
G = ctx.gemm_tn(Q, 0, 16, Q, 0, 16)
print("orth err", np.linalg.norm(G - np.eye(16)))
