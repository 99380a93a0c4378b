"""Where a link of the chain kernel spends its time: phase SUMS of the light diagnostic build
(`make -C krypy_amd/csrc prof` builds krypy_amd/lib/libkrylov_hip_trace2.so with -DKH_CHAIN_TRACE=2, see chain.h CH_STAMP mode 2).

Every wave adds up, in scalar registers, the 100 MHz clock between the phase boundaries of every link of one 64-link
launch; wave 0 and wave 7 of every workgroup write their sums at the end.  Unlike the per-link stamps of
tools/chain_trace.py this does not change the register allocation of the kernel (the launch takes the same time)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("KRYPY_AMD_LIB", os.path.join(ROOT, "krypy_amd", "lib", "libkrylov_hip_trace2.so"))
import numpy as np
from krypy_amd import _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = _hip.get_context()
V = ctx.alloc(n, 18)
W = ctx.alloc(n, 2)
rng = np.random.default_rng(1)
for j in range(18):
    V.upload(j, rng.standard_normal(n))
W.upload(0, rng.standard_normal(n))
lib = ctx._lib
lib.kh_chain_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                               ctypes.POINTER(ctypes.c_uint64), ctypes.c_int64, ctypes.POINTER(ctypes.c_int)]
lib.kh_chain_trace.restype = ctypes.c_int
cap = 512 * 64 * 16
buf = np.zeros(cap, dtype=np.uint64)
g = ctypes.c_int(0)
rc = lib.kh_chain_trace(ctx._h, V.handle, W.handle, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), cap,
                        ctypes.byref(g))
if rc != 0:
    raise SystemExit(lib.kh_last_error().decode())
G = g.value
links = 64
T = buf[:G * 16].reshape(G, 2, 8).astype(np.float64) * 0.01 / links     # microseconds per link
ms = ctx.bench_kernel(5, V, W, 20)
print("n = %d, G = %d workgroups; %.3f ms per 64-link launch = %.2f us per link (events, this build)"
      % (n, G, ms, ms * 1e3 / links))
names = {1: "dot phase (stream the column)", 2: "sum: waiting for the other waves of the workgroup",
         3: "sum: workgroup partial published", 4: "sum: leaders - sweep of the fabric granules",
         5: "sum: result (leaders: add + store; others: poll the L2)",
         6: "update: ring batch + batches read again", 7: "update: LDS-parked batches"}
lead = T[:, 0, 4] > 0
print("%d leader workgroups" % lead.sum())
for label, sel in (("the %d other workgroups" % (~lead).sum(), ~lead), ("the leaders", lead)):
    for w in (0, 1):
        print("%s, wave %d: mean (p10 .. p90), us per link" % (label, 0 if w == 0 else 7))
        tot = 0.0
        for i in (1, 2, 3, 4, 5, 6, 7):
            d = T[sel, w, i]
            tot += d.mean()
            print("  %-58s %6.2f  (%5.2f .. %5.2f)" % (names[i], d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
        print("  %-58s %6.2f" % ("sum of the phases", tot))
