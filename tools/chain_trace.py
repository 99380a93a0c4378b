"""Phase timeline of the chain kernel (diagnostic build `make -C krypy_amd/csrc trace`).

For every workgroup and link, wave 0 and wave 7 stamp the 100 MHz clock at:
  0 link start   1 dot loop done   2 all waves in the reduction   3 partial published
  4 own granule sweep done   5 reduction returned   6 update from ring/memory done   7 LDS part done
Prints medians over workgroups / links (the first 8 links are skipped) in microseconds."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("KRYPY_AMD_LIB", os.path.join(ROOT, "krypy_amd", "lib", "libkrylov_hip_trace.so"))
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
T = buf[:G * 64 * 16].reshape(G, 64, 2, 8).astype(np.int64) * 0.01     # microseconds
print("n = %d, G = %d workgroups; ms per launch (events): %.3f" % (n, G, ctx.bench_kernel(5, V, W, 20)))
S = T[:, 8:, :, :]                                   # steady state
t0 = S[..., 0]
names = ["dot loop (0->1)", "wait for the workgroup's waves (1->2)", "publish (2->3)", "granule sweep (3->4)",
         "sum + barriers (4->5)", "update: ring + memory batches (5->6)", "update: LDS batches (6->7)"]
for w in (0, 1):
    print("wave %d:" % (0 if w == 0 else 7))
    for i, nm in enumerate(names):
        d = S[:, :, w, i + 1] - S[:, :, w, i]
        print("  %-42s median %6.2f  p10 %6.2f  p90 %6.2f us" % (nm, np.median(d), np.percentile(d, 10),
                                                                 np.percentile(d, 90)))
    d = S[:, 1:, w, 0] - S[:, :-1, w, 7]
    print("  %-42s median %6.2f us" % ("link end -> next link start", np.median(d)))
    d = S[:, 1:, w, 0] - S[:, :-1, w, 0]
    print("  %-42s median %6.2f  p10 %6.2f  p90 %6.2f us" % ("whole link", np.median(d), np.percentile(d, 10),
                                                             np.percentile(d, 90)))
# skew: when does each workgroup reach the reduction, relative to the last one of the same link
arr = S[:, :, 0, 2]
last = arr.max(axis=0, keepdims=True)
lag = last - arr
print("arrival at the reduction, behind the LAST workgroup: median %.2f  p10 %.2f  p90 %.2f  max %.2f us"
      % (np.median(lag), np.percentile(lag, 10), np.percentile(lag, 90), lag.max()))
ret = S[:, :, 0, 5]
print("reduction returned, after the last arrival: median %.2f  p10 %.2f  p90 %.2f us"
      % (np.median(ret - last), np.percentile(ret - last, 10), np.percentile(ret - last, 90)))
dl = S[:, :, 0, 1] - S[:, :, 0, 0]
print("dot loop per workgroup (median over links): min %.2f  median %.2f  max %.2f us"
      % (np.median(dl, axis=1).min(), np.median(dl), np.median(dl, axis=1).max()))
np.save(os.path.join(ROOT, "gpurun_out", "chain_trace_n%d.npy" % n), T)
