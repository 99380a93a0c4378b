"""Live roofline measurements for bench.py: HIP-event timing of the hot kernels on the
library's own stream (``kh_bench_kernel`` / ``kh_timer_*``), with the algorithmic byte counts of
SURVEY.md section 8(d).  Not part of the solver path."""
import numpy

# kernel ids of kh_bench_kernel
K_GS_LINK, K_MULTIDOT16, K_MULTIAXPY16, K_AXPY_NRM, K_SCALE_STORE, K_CHAIN, K_CGS = 0, 1, 2, 3, 4, 5, 8
CHAIN_LINKS = 64   # kh_bench_kernel runs the chain over 16 columns x 4 sweeps


def _gbs(nbytes, ms):
    return nbytes / (ms * 1e-3) / 1e9


def roofline(ctx, ls, ortho, peak_gbs, reps=60):
    """Returns (roofline dict of the dominant kernel, extra dict with the other kernels)."""
    n = ls.N
    ncol = 18
    V = ctx.alloc(n, ncol)
    W = ctx.alloc(n, 2)
    rng = numpy.random.default_rng(1)
    for j in range(ncol):
        V.upload(j, rng.standard_normal(n))
    W.upload(0, rng.standard_normal(n))
    kernels = {}

    def run(which, nbytes, name, moved):
        ctx.bench_kernel(which, V, W, 5)             # warm-up
        ms = ctx.bench_kernel(which, V, W, reps)
        kernels[name] = {"avg_ms": ms, "algorithmic_bytes": nbytes, "achieved_gbs": _gbs(nbytes, ms),
                         "moved_bytes_model": moved, "moved_gbs_model": _gbs(moved, ms)}
        return ms

    # algorithmic bytes per launch ("16 N (k+1)" per step = 16 N per basis column:
    # the column is read once for the projection and once for the update; w is accounted once per
    # step, not per column).  moved_bytes_model = what the launch really streams through L2.
    run(K_GS_LINK, 16.0 * n, "k_gs_link<A_PART,T_DOT>", 32.0 * n)
    run(K_MULTIDOT16, 8.0 * n * 16 + 8.0 * n, "k_multidot<16>", 8.0 * n * 17)
    run(K_MULTIAXPY16, 8.0 * n * 16 + 16.0 * n, "k_multiaxpy<16>", 8.0 * n * 18)
    run(K_AXPY_NRM, 8.0 * n + 16.0 * n, "k_gs_link<A_PART,T_NRM>", 24.0 * n)
    run(K_SCALE_STORE, 16.0 * n, "k_scale_store", 16.0 * n)
    chain = None
    try:
        # register-resident MGS chain: one launch = load w (8N) + 64 links x (v_j read for the dot +
        # b_j read for the update = 16N, the SURVEY 8d per-column figure) + store v_{k+1} (8N)
        nb = 16.0 * n * CHAIN_LINKS + 16.0 * n
        run(K_CHAIN, nb, "k_mgs_chain (64 links/launch)", 8.0 * n * CHAIN_LINKS * 1.5 + 16.0 * n)
        chain = kernels["k_mgs_chain (64 links/launch)"]
        chain["us_per_link"] = chain["avg_ms"] * 1e3 / CHAIN_LINKS
    except Exception as exc:   # not eligible (odd n, w larger than the register file, multi-GPU)
        kernels["k_mgs_chain"] = {"unavailable": repr(exc)}
    cgs = None
    try:
        # register-resident panel GS: 16 columns read once for the dots, once for the update, w read
        # twice and written once
        nb = 16.0 * n * 16 + 32.0 * n
        run(K_CGS, nb, "k_cgs_dots+k_cgs_update (16 columns)", nb)
        cgs = kernels["k_cgs_dots+k_cgs_update (16 columns)"]
    except Exception as exc:
        kernels["k_cgs_dots+k_cgs_update"] = {"unavailable": repr(exc)}
    extra = {"kernels": kernels}
    Amat = ls.A._device_matrix()
    if Amat is not None and Amat.kind == "csr":
        X, Y = ctx.alloc(n, 1), ctx.alloc(Amat.shape[0], 1)
        X.upload(0, rng.standard_normal(n))
        for _ in range(3):
            ctx.apply(Amat, X, 0, Y, 0, 1)
        ctx.timer_start()
        for _ in range(reps):
            ctx.apply(Amat, X, 0, Y, 0, 1)
        ms = ctx.timer_stop() / reps
        nb = 12.0 * Amat.nnz + 4.0 * (Amat.shape[0] + 1) + 16.0 * Amat.shape[0]
        nd = Amat.diagonals
        extra["spmv"] = {"kernel": "k_spmv_dia (%d diagonals)" % nd if nd else "k_spmv_stream", "avg_ms": ms,
                         "algorithmic_bytes": nb, "achieved_gbs": _gbs(nb, ms),
                         "frac_of_peak": _gbs(nb, ms) / peak_gbs}
        if nd and getattr(ctx, "counters", None) is not None and ctx.counters().get("chain_fused", 0) > 0:
            extra["spmv"]["in_solver"] = ("fused: the GMRES / Lanczos steps of this run computed A v_k in the prologue of "
                                          "the chain kernel (no SpMV launch, w never written to HBM); the figures here "
                                          "are the stand-alone kernel, as used for residuals and by the panel modes")
        if nd:
            # the banded copy holds 8 B per diagonal slot and no indices: what the kernel really moves
            moved = 8.0 * nd * Amat.shape[0] + 16.0 * Amat.shape[0]
            extra["spmv"].update({
                "moved_bytes_model": moved, "moved_gbs_model": _gbs(moved, ms),
                "note": "algorithmic bytes = SURVEY 8(d) CSR figure (12 nnz + 4 (N+1) + 16 N); the operator "
                        "is banded, so the library multiplies with its diagonal-major copy (8 B per slot, no "
                        "index stream): achieved_gbs is the CSR-equivalent rate, moved_gbs_model the rate on "
                        "the bytes actually streamed"})
    if ortho in ("cgs", "cgs2") and cgs is not None:
        nb, ms, name = cgs["algorithmic_bytes"], cgs["avg_ms"], "k_cgs_dots+k_cgs_update (16 columns per launch pair)"
    elif ortho in ("cgs", "cgs2"):
        # the two panel kernels alternate; report the pair as one unit of 16 columns
        d, a = kernels["k_multidot<16>"], kernels["k_multiaxpy<16>"]
        nb = d["algorithmic_bytes"] + a["algorithmic_bytes"]
        ms = d["avg_ms"] + a["avg_ms"]
        name = "k_multidot<16>+k_multiaxpy<16>"
    elif chain is not None:
        nb, ms, name = chain["algorithmic_bytes"], chain["avg_ms"], "k_mgs_chain_lds (64 links per launch)"
    else:
        d = kernels["k_gs_link<A_PART,T_DOT>"]
        nb, ms, name = d["algorithmic_bytes"], d["avg_ms"], "k_gs_link<A_PART,T_DOT>"
    ach = _gbs(nb, ms)
    roof = {"bound": "hbm", "kernel": name, "achieved": ach, "peak": peak_gbs, "unit": "GB/s",
            "frac": ach / peak_gbs, "traffic": None, "avg_launch_ms": ms,
            "algorithmic_bytes_per_launch": nb}
    if chain is not None and name.startswith("k_mgs_chain"):
        roof["note"] = ("algorithmic bytes = SURVEY 8(d): 16 N per basis column (the column is read for the "
                        "projection and again for the update).  The kernel serves half of that second read "
                        "from LDS / the register ring, so its HBM traffic (`traffic`, PMC) is below the "
                        "algorithmic bytes and `frac` can exceed 1; traffic / avg_launch_ms is the HBM rate.  "
                        "Measured on identical 64-link launches that load w (k_mgs_chain_lds<40,false,false,0>); the "
                        "solver's own launches are the same kernel with w = A v_k computed in the prologue "
                        "(<40,false,false,5>, k+1 links each).")
    return roof, extra
