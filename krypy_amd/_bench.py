"""Live roofline measurements for bench.py: HIP-event timing of the hot kernels on the
library's own stream (``kh_bench_kernel`` / ``kh_timer_*``).  Not part of the solver path.

Every rate in here is bytes / time with the bytes named:

  compulsory   the bytes a launch cannot avoid reading from / writing to HBM: every basis column
               ONCE (8 N), w once in and v_{k+1} once out - the column's second use (the update after
               the grid-wide reduction) is served on chip where the kernel manages to
  moved        HBM traffic of the launch: the rocprofv3 FETCH_SIZE / WRITE_SIZE figure of
               ``profiles/*_traffic.json`` when that file was collected from THIS source tree (stamp =
               sha256 of the kernel sources), otherwise the kernel's own worst-case model (every byte it
               requests from L2 and beyond)
  survey_8d    the SURVEY.md 8(d) accounting (16 N per column: the column charged twice).  A kernel that
               keeps half a column on chip beats that model, so it is a labelled side number only and
               never a roofline fraction.

``frac`` fields are always moved-or-compulsory bytes over the 8 TB/s spec peak and cannot exceed 1;
``frac_of_attainable`` divides by the copy rate measured in the same run.
"""
import hashlib
import os

import numpy

# kernel ids of kh_bench_kernel
K_GS_LINK, K_MULTIDOT16, K_MULTIAXPY16, K_AXPY_NRM, K_SCALE_STORE, K_CHAIN, K_CGS = 0, 1, 2, 3, 4, 5, 8
K_COPY, K_TRIAD, K_READ = 9, 10, 11
CHAIN_LINKS = 64   # kh_bench_kernel runs the chain over 16 columns x 4 sweeps

_SOURCES = ("chain.h", "kernels.h", "krylov_hip.hip", "kh_internal.h", "zpath.h", "comm.hip", "lanczos.h", "chain_blk.h",
            "chain_blk.hip", "proj_reg.h", "proj_reg.hip", "xr.hip", "xr_dev.h", "chain_blk2.h", "chain_blk2.hip", "chain_xr.hip", "chain_long.h", "cycles.hip",
            "bench_abi.hip", "krylov_steps.h")


def source_stamp():
    """sha256 (first 16 hex digits) of the kernel sources: ties a PMC traffic file to the code it measured."""
    h = hashlib.sha256()
    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for fn in _SOURCES:
        with open(os.path.join(base, fn), "rb") as f:
            h.update(f.read())
    # the compiler flags too (-ffp-contract, -O: a flag change is a different kernel)
    with open(os.path.join(base, "Makefile"), "r") as f:
        for line in f:
            if line.startswith("FLAGS") or line.startswith("  ") and "-I" in line:
                h.update(line.encode())
    return h.hexdigest()[:16]


def _gbs(nbytes, ms):
    return nbytes / (ms * 1e-3) / 1e9


def chain_reread_fraction(n, ncu, lds=True):
    """Share of a column the LDS chain kernel requests a second time from memory (the rest of the second
    use comes from LDS and the register ring): NG*PB/R2 of chain.h's ChainShapeLds."""
    n2 = (n + 1) // 2
    for r2 in (4, 8, 16, 24, 32, 40, 48, 56):
        g = -(-n2 // (r2 * 512))
        if g <= ncu and g <= 512:
            if r2 == 48 and lds:
                # chain_long.h: batches 0, 1 parked in LDS, batches 10, 11 still in the ring: 32 of 48 rows come back from memory
                return r2, 32.0 / 48.0
            if not lds or r2 > 40:
                return r2, 1.0
            pb = 5 if r2 == 40 else (2 if r2 == 4 else 4)
            nb = r2 // pb
            lb = 3 if nb >= 4 else 1
            return r2, max(nb - lb - 1, 0) * pb / float(r2)
    return 0, 1.0


def projector_probe(ctx, n, d, reps=10):
    """The deflation projector as a deflated Arnoldi step applies it (utils.py:604-627: d columns, two sweeps) on synthetic
    bases of this length: (HIP-event ms per application, algorithmic bytes, kernel name).  One launch with z in registers
    where the vector fits (proj_reg.h: 528 N for d = 16), else four launches per sweep (560 N)."""
    rng = numpy.random.default_rng(3)
    Wd, Vd = ctx.alloc(n, d), ctx.alloc(n, d)
    col = rng.standard_normal(n) / numpy.sqrt(n)
    for j in range(d):
        Wd.upload(j, numpy.roll(col, j))
        Vd.upload(j, numpy.roll(col, -j))
    pj = ctx.proj_create(Wd, Vd, d, rng.standard_normal((d, d)) * 0.1, None, 2)
    A, Z = ctx.upload(rng.standard_normal(n)), ctx.alloc(n, 1)
    c0 = ctx.get("n_proj_reg")
    ctx.proj_apply_complement(pj, A, 0, Z, 0, want_ya=True)
    ctx.timer_start()
    for _ in range(reps):
        ctx.proj_apply_complement(pj, Z, 0, Z, 0)
    ms = ctx.timer_stop() / reps
    one = ctx.get("n_proj_reg") - c0 >= reps
    nbytes = (32.0 * d + 16.0) * n if one else (32.0 * d + 48.0) * n
    return ms, nbytes, ("k_proj_reg: the projector's two sweeps in one launch, z in registers (d = %d)" % d) if one else \
        ("the projector as four launches per sweep (d = %d)" % d)


def roofline(ctx, ls, ortho, peak_gbs, reps=60, traffic_files=None, m=100, solver_steps=True):
    """Returns (roofline dict of the dominant kernel, extra dict with the other kernels).  ``m``: the restart length
    of the timed solver (the solver's own launches are timed over k = 0 .. m-1)."""
    n = ls.N
    ncol = 18
    V = ctx.alloc(n, ncol)
    W = ctx.alloc(n, 2)
    rng = numpy.random.default_rng(1)
    for j in range(ncol):
        V.upload(j, rng.standard_normal(n))
    W.upload(0, rng.standard_normal(n))
    kernels = {}

    def run(which, name, compulsory, moved_model, survey=None):
        ctx.bench_kernel(which, V, W, 5)             # warm-up
        ms = ctx.bench_kernel(which, V, W, reps)
        return run_ms(ms, name, compulsory, moved_model, survey)

    def run_ms(ms, name, compulsory, moved_model, survey=None):
        # compulsory: what must cross the HBM interface; request model: every byte the kernel asks the L2 for (an
        # UPPER bound on its HBM traffic - re-reads served by L2 / Infinity Cache are in it); PMC traffic, where a
        # stamped profile exists, is attached to the dominant kernel below
        kernels[name] = {"avg_ms": ms, "compulsory_bytes": compulsory, "compulsory_gbs": _gbs(compulsory, ms),
                         "frac_compulsory": _gbs(compulsory, ms) / peak_gbs,
                         "moved_bytes_model": moved_model, "moved_gbs_model": _gbs(moved_model, ms),
                         "frac_request_model": _gbs(moved_model, ms) / peak_gbs}
        if survey is not None:
            kernels[name]["survey_8d_bytes"] = survey
        return kernels[name]

    # ---- attainable ceiling of this box, this run (SURVEY 8d / BASELINE.md section 3) ----
    ld = V.ld
    ceiling = {}
    for which, name, nbytes in ((K_COPY, "copy, grid-stride (8 columns -> 8 columns)", 8.0 * ld * 16),
                                (12, "copy, 1 x 16 B per lane", 8.0 * ld * 16),
                                (13, "copy, 4 x 16 B per lane", 8.0 * ld * 16),
                                (14, "copy, 8 x 16 B per lane", 8.0 * ld * 16),
                                (15, "copy, 4 x 16 B per lane, non-temporal stores", 8.0 * ld * 16),
                                (K_TRIAD, "triad a = b + s c (4-column chunks)", 8.0 * ld * 12),
                                (K_READ, "read-only sum, grid-stride (16 columns)", 8.0 * ld * 16),
                                (16, "read-only, 4 x 16 B per lane", 8.0 * ld * 16),
                                (17, "read-only, 8 x 16 B per lane", 8.0 * ld * 16),
                                (18, "read-only, 16 x 16 B per lane", 8.0 * ld * 16)):
        try:
            ctx.bench_kernel(which, V, W, 3)
            ms = ctx.bench_kernel(which, V, W, 20)
            ceiling[name] = {"avg_ms": ms, "bytes": nbytes, "gbs": _gbs(nbytes, ms), "frac_of_peak": _gbs(nbytes, ms) / peak_gbs}
        except Exception as exc:
            ceiling[name] = {"unavailable": repr(exc)}
    rates = [v["gbs"] for v in ceiling.values() if "gbs" in v]
    attainable = max(rates) if rates else None

    run(K_GS_LINK, "k_gs_link<A_PART,T_DOT>", 32.0 * n, 32.0 * n, 16.0 * n)
    run(K_MULTIDOT16, "k_multidot<16>", 8.0 * n * 17, 8.0 * n * 17)
    run(K_MULTIAXPY16, "k_multiaxpy<16>", 8.0 * n * 18, 8.0 * n * 18)
    run(K_AXPY_NRM, "k_gs_link<A_PART,T_NRM>", 24.0 * n, 24.0 * n)
    run(K_SCALE_STORE, "k_scale_store", 16.0 * n, 16.0 * n)
    info = ctx.info() if hasattr(ctx, "info") else {"compute_units": 256}
    counters0 = ctx.counters() if hasattr(ctx, "counters") else {}
    chain = None
    chain_name = None
    try:
        # register-resident MGS chain, 64 links per launch: w in (8N) + 64 columns (8N each) + v_{k+1} out (8N)
        ms_probe = ctx.bench_kernel(K_CHAIN, V, W, 2)
        c1 = ctx.counters() if hasattr(ctx, "counters") else {}
        lds = c1.get("chain_lds", 0) > counters0.get("chain_lds", 0)
        r2, rr = chain_reread_fraction(n, int(info.get("compute_units", 256)), lds)
        chain_name = "%s<%d> (64 links per launch)" % (("k_mgs_chain_long" if r2 == 48 else "k_mgs_chain_lds") if lds else "k_mgs_chain", r2)
        comp = 8.0 * n * CHAIN_LINKS + 16.0 * n
        chain = run(K_CHAIN, chain_name, comp, 8.0 * n * CHAIN_LINKS * (1.0 + rr) + 16.0 * n,
                    16.0 * n * CHAIN_LINKS + 16.0 * n)
        chain["us_per_link"] = chain["avg_ms"] * 1e3 / CHAIN_LINKS
        chain["reread_fraction_model"] = rr
        del ms_probe
    except Exception as exc:   # not eligible (w larger than the register file, multi-GPU)
        kernels["k_mgs_chain"] = {"unavailable": repr(exc)}
    # ---- the solver's OWN launches: Arnoldi steps k = 0 .. m-1 as Gmres._solve enqueues them ----
    solver = None
    solver_name = None
    Amat0 = ls.A._device_matrix() if hasattr(ls.A, "_device_matrix") else None
    if (solver_steps and chain is not None and ortho in ("mgs",) and Amat0 is not None and Amat0.kind == "csr"
            and hasattr(ctx, "bench_arnoldi")):
        try:
            c0 = ctx.counters()
            Vm = ctx.alloc(n, m + 1)
            v0 = rng.standard_normal(n)
            Vm.upload(0, v0 / numpy.linalg.norm(v0))
            ctx.bench_arnoldi(Amat0, Vm, W, m, 0, 1)             # warm-up
            c1 = ctx.counters()
            ms = ctx.bench_arnoldi(Amat0, Vm, W, m, 0, 3)
            c2 = ctx.counters()
            nd = Amat0.diagonals
            fused = nd and (c2.get("chain_fused", 0) - c1.get("chain_fused", 0)) == 3 * m
            one_launch = (c2.get("chain", 0) - c1.get("chain", 0)) == 3 * m
            if one_launch:
                # every datum once: the operator (diagonal-major copy, or the CSR arrays + w written and read back when
                # the step is SpMV + chain), columns 0..k, v_{k+1} out - averaged over k = 0 .. m-1
                op = 8.0 * nd * n if fused else (12.0 * Amat0.nnz + 4.0 * (n + 1) + 8.0 * n + 16.0 * n)
                comp = op + 8.0 * n * (m + 1) / 2.0 + 8.0 * n
                r2 = chain_reread_fraction(n, int(info.get("compute_units", 256)), True)[0]
                solver_name = ("k_mgs_chain_lds<%d,false,false,%d>: the solver's Arnoldi steps k = 0..%d, one launch each "
                               "(operator in the prologue, k+1 Gram-Schmidt links)" % (r2, nd, m - 1)) if fused else \
                              ("SpMV + k_mgs_chain_lds<%d>: the solver's Arnoldi steps k = 0..%d" % (r2, m - 1))
                solver = run_ms(ms, solver_name, comp, comp + 8.0 * n * (m + 1) / 2.0 * rr,
                                (8.0 * nd * n if fused else op) + 16.0 * n * (m + 1) / 2.0 + 8.0 * n)
                solver["fused_operator"] = bool(fused)
                # (the PMC passes see the chain kernel's launches, i.e. the steps k >= 1 - step k = 0 is the one-column
                # Lanczos kernel: what their traffic is compared with is the compulsory bytes of THOSE steps)
                solver["compulsory_bytes_k_ge_1"] = op + 8.0 * n * (m * (m + 1) / 2.0 - 1.0) / (m - 1.0) + 8.0 * n if m > 1 else comp
                solver["links_per_launch_avg"] = (m + 1) / 2.0
                solver["us_per_link"] = ms * 1e3 / ((m + 1) / 2.0)
            del Vm
        except Exception as exc:
            kernels["solver_arnoldi_steps"] = {"unavailable": repr(exc)}
    cgs = None
    try:
        # register-resident panel GS: 16 columns read for the dots, again for the update (the basis does not fit
        # any cache), w read twice and written once: compulsory == moved
        nb = 16.0 * n * 16 + 32.0 * n
        cgs = run(K_CGS, "k_cgs_dots+k_cgs_update (16 columns)", nb, nb)
    except Exception as exc:
        kernels["k_cgs_dots+k_cgs_update"] = {"unavailable": repr(exc)}
    extra = {"kernels": kernels, "attainable": {"probes": ceiling, "best_gbs": attainable,
                                                "note": "streaming probes timed in this run on 1.2-1.4 GB of the same device "
                                                        "block (beyond the 256 MB Infinity Cache); MI355X_MICROARCH.md quotes "
                                                        "6.29 TB/s for a float4 copy"}}

    # ---- SpMV: both kernels of the library on the bytes each one moves ----
    Amat = ls.A._device_matrix()
    if Amat is not None and Amat.kind == "csr":
        X, Y = ctx.alloc(n, 1), ctx.alloc(Amat.shape[0], 1)
        X.upload(0, rng.standard_normal(n))

        def time_spmv():
            for _ in range(3):
                ctx.apply(Amat, X, 0, Y, 0, 1)
            ctx.timer_start()
            for _ in range(reps):
                ctx.apply(Amat, X, 0, Y, 0, 1)
            return ctx.timer_stop() / reps

        rows = Amat.shape[0]
        csr_bytes = 12.0 * Amat.nnz + 4.0 * (rows + 1) + 16.0 * rows
        nd = Amat.diagonals
        spmv = {}
        if nd:
            ms = time_spmv()
            moved = 8.0 * nd * rows + 16.0 * rows
            spmv["k_spmv_dia (%d diagonals)" % nd] = {
                "avg_ms": ms, "moved_bytes": moved, "achieved_gbs": _gbs(moved, ms), "frac": _gbs(moved, ms) / peak_gbs,
                "csr_equivalent_gbs_side_number": _gbs(csr_bytes, ms),
                "note": "diagonal-major copy: 8 B per slot, no index stream; bytes = 8 nd N + 16 N (PMC: equal)"}
        # (a block-row shard whose halo travels inside the banded kernel's launch has no CSR form of the exchange to time:
        # kh_apply refuses it - every rank alike, but the refusal used to cost the N > 1 line its whole roofline object)
        sharded_in_launch = bool(getattr(ls.A, "halo_in_launch", False))
        if hasattr(ctx, "set") and not sharded_in_launch:
            ctx.set("spmv_dia", 0)
            try:
                ms = time_spmv()
            finally:
                ctx.set("spmv_dia", 1)
            spmv["k_spmv_stream (CSR)"] = {
                "avg_ms": ms, "moved_bytes": csr_bytes, "achieved_gbs": _gbs(csr_bytes, ms),
                "frac": _gbs(csr_bytes, ms) / peak_gbs,
                "note": "the CSR kernel north_star names: 12 nnz + 4 (N+1) + 16 N bytes (PMC: 804 MB vs 800 MB)"}
        for v in spmv.values():
            if attainable:
                v["frac_of_attainable"] = v["achieved_gbs"] / attainable
        # what a kernel trace of this command reproduces: the per-dispatch average of a stamped profile of these sources
        # (tools/summarize_prof.py).  It sits a few per cent ABOVE avg_ms: a dispatch timed by itself includes its own ramp
        # and drain, which back-to-back launches between two HIP events overlap with their neighbours'
        stamp_ = source_stamp()
        for fn in sorted(traffic_files or [], reverse=True):
            try:
                import json
                tj = json.load(open(fn))
                if tj.get("source_stamp") != stamp_ or int(tj.get("n", n)) != n or "kernel_trace_avg_us" not in tj:
                    continue
                for name, v in spmv.items():
                    key = "k_spmv_dia" if name.startswith("k_spmv_dia") else "k_spmv_stream"
                    if key in tj["kernel_trace_avg_us"]:
                        us = tj["kernel_trace_avg_us"][key]["avg_us"]
                        v["kernel_trace_avg_ms"] = us / 1e3
                        v["frac_kernel_trace"] = _gbs(v["moved_bytes"], us / 1e3) / peak_gbs
                        v["kernel_trace_source"] = os.path.basename(fn)
                break
            except Exception:
                continue
        extra["spmv"] = spmv
        if nd and counters0.get("chain_fused", 0) > 0:
            extra["spmv_in_solver"] = ("fused: the GMRES / Lanczos steps of this run computed A v_k in the prologue of the "
                                       "chain kernel (no SpMV launch, w never written to HBM); the figures above are the "
                                       "stand-alone kernels, as used for residuals and by the panel modes")

    # ---- the roofline object of the dominant kernel ----
    if ortho in ("cgs", "cgs2") and cgs is not None:
        k, name = cgs, "k_cgs_dots+k_cgs_update (16 columns per launch pair)"
        traffic_keys = ("k_cgs_dots", "k_cgs_update")
    elif ortho in ("cgs", "cgs2"):
        d, a = kernels["k_multidot<16>"], kernels["k_multiaxpy<16>"]
        k = {"avg_ms": d["avg_ms"] + a["avg_ms"], "compulsory_bytes": d["compulsory_bytes"] + a["compulsory_bytes"],
             "moved_bytes_model": d["moved_bytes_model"] + a["moved_bytes_model"]}
        name, traffic_keys = "k_multidot<16>+k_multiaxpy<16>", ()
    elif solver is not None:
        k, name, traffic_keys = solver, solver_name, ("k_mgs_chain_solver",)
    elif chain is not None:
        k, name, traffic_keys = chain, chain_name, ("k_mgs_chain_micro",)
    else:
        k, name, traffic_keys = kernels["k_gs_link<A_PART,T_DOT>"], "k_gs_link<A_PART,T_DOT>", ()
    ms = k["avg_ms"]
    comp = k["compulsory_bytes"]
    per_launch = (m + 1) / 2.0 if (solver is not None and k is solver) else (CHAIN_LINKS if (chain is not None and k is chain) else
                                                                             (16.0 if traffic_keys == ("k_cgs_dots", "k_cgs_update") or name.startswith("k_multidot") else 1.0))
    # `bytes_per_launch` = the ALGORITHMIC (compulsory) bytes of a launch - every datum once - and `frac` the
    # fraction of the peak they are moved at.  `traffic` = what the memory fabric was asked for, from the PMC passes
    # of this same command, attached only when the profile carries the stamp of the kernel sources that have just
    # been timed; traffic_over_bytes > 1 is re-read traffic (the second use of a column that missed LDS / L2).
    stamp = source_stamp()
    traffic, traffic_file = None, None
    for fn in sorted(traffic_files or [], reverse=True):
        try:
            import json
            tj = json.load(open(fn))
            if tj.get("source_stamp") != stamp or not traffic_keys:
                continue
            if int(tj.get("n", n)) != n:
                continue
            if not all(key in tj for key in traffic_keys):
                continue
            traffic = sum(tj[key]["hbm_read_bytes_per_launch"] + tj[key]["hbm_write_bytes_per_launch"]
                          for key in traffic_keys)
            traffic_file = os.path.basename(fn)
            break
        except Exception:
            continue
    ach = _gbs(comp, ms)
    roof = {"bound": "hbm", "kernel": name, "achieved": ach, "peak": peak_gbs, "unit": "GB/s",
            "frac": ach / peak_gbs, "traffic": traffic, "avg_launch_ms": ms,
            "bytes_per_launch": comp,
            "bytes_source": "compulsory (algorithmic) bytes: every basis column once (8 N each), the operator's arrays once, "
                            "v_{k+1} out - SURVEY 8(d)'s per-unit figures with the column's second use NOT charged to HBM",
            "source_stamp": stamp,
            # Gram-Schmidt links (chain / link kernels) or columns (panel kernels) one timed launch (pair) serves
            "links_or_columns_per_launch": per_launch, "traffic_key": traffic_keys[0] if len(traffic_keys) == 1 else None}
    if traffic is not None:
        # like with like: the solver key's traffic is the average over the steps k >= 1, so are the bytes under it
        roof["traffic_over_bytes"] = traffic / (k.get("compulsory_bytes_k_ge_1", comp) if traffic_keys == ("k_mgs_chain_solver",) else comp)
        roof["traffic_gbs"] = _gbs(traffic, ms)
        roof["frac_traffic"] = _gbs(traffic, ms) / peak_gbs
        roof["traffic_source"] = ("rocprofv3 PMC, separate passes: 2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950 "
                                  "correction), " + traffic_file + "; these counters sit in front of the Infinity "
                                  "Cache, so this is memory-FABRIC traffic - an upper bound on what reached HBM")
        k["fabric_bytes_pmc"] = traffic
        k["frac_pmc"] = _gbs(traffic, ms) / peak_gbs
    else:
        roof["traffic_source"] = ("null: no profiles/*_traffic.json carries the stamp of these kernel sources "
                                  "(tools/profile.sh + tools/summarize_prof.py produce it)")
    if attainable:
        roof["attainable_gbs"] = attainable
        roof["frac_of_attainable"] = ach / attainable
    if "survey_8d_bytes" in k:
        roof["survey_8d_side_number"] = {
            "bytes_per_launch": k["survey_8d_bytes"], "gbs": _gbs(k["survey_8d_bytes"], ms),
            "note": "SURVEY 8(d) charges 16 N per basis column (read for the projection, read again for the "
                    "update); the kernel serves part of the second use from LDS / registers / L2, so this rate "
                    "is NOT an HBM rate and is not compared with the peak"}
    if solver is not None and k is solver:
        roof["note"] = ("avg_launch_ms = HIP-event time of 3 x %d Arnoldi steps (kh_bench_arnoldi: the launches Gmres._solve "
                        "enqueues, k+1 = 1..%d links each, %.1f on average) / %d; frac = bytes_per_launch / avg_launch_ms / "
                        "peak.  The 64-link micro-launch of the same kernel without the operator (FND = 0) is in "
                        "`kernels`." % (m, m, (m + 1) / 2.0, 3 * m))
    elif chain is not None and k is chain:
        roof["note"] = ("64-link launches that load w (FND = 0 instantiation); the solver's own launches are the same "
                        "kernel with w = A v_k computed in the prologue and k+1 links each.  frac = bytes_per_launch / "
                        "avg_launch_ms / peak with every column counted once.")
    return roof, extra
