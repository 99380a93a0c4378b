import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from krypy_amd import _hip
ctx = _hip.get_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
V = ctx.alloc(n, 18); W = ctx.alloc(n, 2)
rng = np.random.default_rng(1)
for j in range(18): V.upload(j, rng.standard_normal(n))
W.upload(0, rng.standard_normal(n))
print("KRYPY_AMD_CHAIN_LDS =", os.environ.get("KRYPY_AMD_CHAIN_LDS", "1"))
for which, name in ((0, "link kernel"), (5, "chain full"), (6, "chain no grid-sum"), (7, "chain grid-sum only")):
    ctx.bench_kernel(which, V, W, 3)
    ms = ctx.bench_kernel(which, V, W, 20)
    per = ms if which == 0 else ms / 64
    print("%-22s %.3f ms/launch  %.2f us per column" % (name, ms, per * 1e3))
