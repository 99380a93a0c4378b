#!/usr/bin/env python
"""Does the halo exchange of a sharded SpMV overlap the interior rows?  (one MI355X, loopback exchange)

    python tools/halo_overlap.py run [nx ny nz_slab reps]     the workload: config 5's per-rank slab (500 x 500 x 50 of a
                                                              7-point operator periodic in z), `reps` sharded SpMVs, each
                                                              followed by a norm (all-reduce on the compute stream)
    rocprofv3 --kernel-trace -d DIR -o halo -- python tools/halo_overlap.py run
    python tools/halo_overlap.py report DIR [out.md]          per SpMV: RCCL kernel vs interior launch on the timeline

The exchange is the library's grouped ncclSend / ncclRecv (to the rank itself: `halo_loopback`) on the communication
stream; the interior row blocks run on the compute stream meanwhile and the boundary blocks wait for the exchange's
event.  `report` prints, per SpMV, when the RCCL kernel ran relative to the interior launch and how long the boundary
launch had to wait after the interior launch ended (the exposed part of the exchange)."""
import glob
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(nx=500, ny=500, nz=50, reps=40):
    import numpy as np
    import scipy.sparse as sp
    from krypy_amd import _hip, dist

    os.environ["KRYPY_AMD_FORCE_MULTI"] = "1"
    ctx = _hip.Context(0)
    ctx.comm_init(0, 1, ctx.comm_unique_id())
    ctx.set("halo_loopback", 1)
    # rows [n, 2 n) of the 7-point operator on nx x ny x (3 nz) points (z slowest), global column indices
    plane = nx * ny
    n = plane * nz
    l = np.arange(n, dtype=np.int64)
    ix, iy = l % nx, (l // nx) % ny
    rows, cols, vals = [], [], []
    for off, mask in ((-plane, None), (-nx, iy > 0), (-1, ix > 0), (0, None), (1, ix < nx - 1), (nx, iy < ny - 1), (plane, None)):
        r = l if mask is None else l[mask]
        rows.append(r)
        cols.append(n + r + off)
        vals.append(np.full(r.size, 6.0 if off == 0 else -1.0))
    full = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, 3 * n))
    full.sort_indices()
    del rows, cols, vals
    A_local, nrp, nrn = dist.localize_columns(full, n, 3 * n)
    assert nrp == plane and nrn == plane, (nrp, nrn)
    Ad = ctx.csr(A_local, n_cols=A_local.shape[1])
    ctx.set_halo(Ad, nrn, nrp, nrp, nrn)
    x = np.random.default_rng(0).standard_normal(n)
    X, Y = ctx.upload(x), ctx.alloc(n, 1)
    for split in (1, 0):
        ctx.set("spmv_split", split)
        for _ in range(3):
            ctx.apply(Ad, X, 0, Y, 0, 1)
            ctx.nrm2(Y, 0)
        ctx.timer_start()
        for _ in range(reps):
            ctx.apply(Ad, X, 0, Y, 0, 1)
            ctx.nrm2(Y, 0)
        ms = ctx.timer_stop()
        print("split=%d: %.1f us per sharded SpMV + norm (N = %d rows, halo 2 x %d doubles, banded kernel: %s)" % (
            split, ms * 1e3 / reps, n, plane, Ad.diagonals > 0))
    want = full.dot(np.tile(x, 3))
    assert np.array_equal(Y.download()[:, 0], want)
    print("product == periodic operator (bit for bit); exchanges issued: %d" % ctx.get("n_halo_exchange"))
    ctx.close()


def report(src, dst=None):
    db = sorted(glob.glob(src + "/**/*.db", recursive=True))[-1]
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    out = ["# Halo exchange vs interior SpMV on the timeline (`rocprofv3 --kernel-trace`, one MI355X, loopback exchange)", "",
           "Workload: `python tools/halo_overlap.py run` - config 5's per-rank slab (500 x 500 x 50, 12.5 M rows, halo = one "
           "plane of 250,000 doubles = 2 MB each way), periodic in z so that the one rank exchanges with itself through the "
           "library's grouped `ncclSend` / `ncclRecv` on the communication stream.", ""]

    def short(nm):
        return nm.split("(")[0].replace("void kh::", "").replace("kh::", "")

    spmv = [(s, e, short(nm)) for nm, s, e in rows if "k_spmv" in nm]
    rccl = [(s, e, short(nm)) for nm, s, e in rows if "nccl" in nm.lower() or "rccl" in nm.lower()]
    out.append("Kernels seen: %d SpMV launches, %d RCCL kernels (%s)." % (
        len(spmv), len(rccl), ", ".join(sorted({r[2][:60] for r in rccl})) or "none"))
    out.append("")
    # split phase: SpMV launches come in (interior, boundary) pairs with an RCCL kernel between / beside them
    recs = []
    i = 0
    while i + 1 < len(spmv):
        a, b = spmv[i], spmv[i + 1]
        gap = b[0] - a[1]
        mid = [r for r in rccl if r[1] > a[0] and r[0] < b[0] and (r[1] - r[0]) > 0]
        if (a[1] - a[0]) > 5 * (b[1] - b[0]) and mid:          # interior (long) followed by boundary (short)
            r = max(mid, key=lambda q: q[1] - q[0])
            recs.append(dict(interior=(a[1] - a[0]) / 1e3, boundary=(b[1] - b[0]) / 1e3, rccl=(r[1] - r[0]) / 1e3,
                             rccl_start=(r[0] - a[0]) / 1e3, rccl_end_after_interior=(r[1] - a[1]) / 1e3, gap=gap / 1e3))
            i += 2
        else:
            i += 1
    if recs:
        import statistics as st
        med = lambda k: st.median(x[k] for x in recs)      # noqa: E731
        out += ["## Split launches (interior rows on the compute stream while the exchange runs), %d SpMVs" % len(recs), "",
                "| quantity (median, us) | |", "|---|---:|",
                "| interior launch | %.1f |" % med("interior"),
                "| RCCL send/recv kernel (communication stream) | %.1f |" % med("rccl"),
                "| RCCL kernel starts after the interior launch starts | %.1f |" % med("rccl_start"),
                "| RCCL kernel ends after the interior launch ends (> 0 = exposed) | %.1f |" % med("rccl_end_after_interior"),
                "| gap interior end -> boundary start | %.1f |" % med("gap"),
                "| boundary launch | %.1f |" % med("boundary"), ""]
        out.append("RCCL kernel inside the interior launch's interval in %d of %d SpMVs." % (
            sum(1 for x in recs if x["rccl_end_after_interior"] <= 0), len(recs)))
        out.append("")
    serial = []          # (the norms' all-reduces are RCCL kernels too: an exchange is the one directly in front of an SpMV launch)
    for idx in range(len(rows) - 1):
        nm, s0, e0 = rows[idx]
        if ("nccl" in nm.lower() or "rccl" in nm.lower()) and "k_spmv" in rows[idx + 1][0] and rows[idx + 1][1] >= e0:
            serial.append((e0 - s0) / 1e3)
    if serial:
        import statistics as st
        out += ["## Serial order (`spmv_split = 0`): exchange, then one SpMV launch", "",
                "RCCL kernels that overlap no SpMV launch: %d, median %.1f us each - the cost the split hides." % (
                    len(serial), st.median(serial)), ""]
    text = "\n".join(out)
    print(text)
    if dst:
        open(dst, "w").write(text + "\n")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "report":
        report(*sys.argv[2:4])
    else:
        run(*[int(a) for a in sys.argv[2:6]])
