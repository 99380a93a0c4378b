#!/usr/bin/env python
"""Secondary measurements (not the bench.py contract): BASELINE.json configs 3, 4 and the
single-GPU shape of config 5 at full size, iterations/s with inputs resident in HBM.
  python tools/bench_configs.py [3] [4] [5]      -> one JSON line per config

Every line carries the same three keys:
  bytes_per_iteration   the ALGORITHMIC bytes of one iteration, every datum once (the convention of bench.py's roofline.bytes_per_launch)
  frac                  bytes_per_iteration x iterations/s / 8 TB/s
  traffic_over_bytes    fabric traffic of the solver's kernels (PMC passes of tools/profile_config.sh, which fills it in) over those
                        bytes; null in a plain run
and `iterations_in_process`, the iterations all solves of the process made (what the PMC totals are divided by).
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from krypy_amd import _hip, deflation, linsys, utils  # noqa: E402


def config3(ctx, steps=400):
    A = bench.laplace2d(4000, 2500)
    N = A.shape[0]
    b = np.random.default_rng(0).standard_normal(N)
    d = A.diagonal()
    ls = linsys.LinearSystem(A, b, M=sp.diags(1.0 / d).tocsr(), Minv=sp.diags(d).tocsr(),
                             self_adjoint=True)
    out = {}
    for it in range(2):          # first pass = warm-up
        ctx.sync()
        t0 = time.perf_counter()
        try:
            s = linsys.Minres(ls, ortho="lanczos", tol=1e-12, maxiter=steps)
        except utils.ConvergenceError as e:
            s = e.solver
        ctx.sync()
        dt = time.perf_counter() - t0
        n_it = len(s.resnorms) - 1
        del s
    # bytes the iteration MOVES (what the kernels request; the banded copy of the operator, not CSR):
    #   Lanczos launch (lanczos.h)  5 diagonals 40 N + v_k 8 N + p_{k-1} 8 N | p_k 8 N + D 8 N | D again (21 of 40 rows
    #                               come back from memory) 4.2 N + two stores 16 N                     = 92.2 N
    #   MINRES recurrences          v_k, W0, W1, yk in, z and yk out                                   = 48 N
    # SURVEY 8(d)'s fused lower bound (CSR bytes + 12 vector passes) is kept beside it for reference.
    fused = ctx.get("n_lanczos_fused") > 0
    moved = (92.2 if fused else 104.0) * N + 48.0 * N
    nb = 12.0 * A.nnz + 4.0 * (N + 1) + 16.0 * N + 12 * 8.0 * N
    comp = (88.0 if fused else 104.0) * N + 48.0 * N          # the same without the 4.2 N of rows that come back from memory
    out.update(config="3: MINRES + Jacobi, 2-D Laplacian N=1e7, ortho=lanczos, %d steps" % steps,
               iterations_per_s=n_it / dt, ms_per_iteration=dt / n_it * 1e3,
               bytes_per_iteration=comp, frac=comp * n_it / dt / 8e12, traffic_over_bytes=None, iterations_in_process=2 * n_it,
               moved_gb_per_iteration=moved / 1e9, moved_gbs=moved * n_it / dt / 1e9,
               frac_moved_of_8TBs=moved * n_it / dt / 8e12,
               survey_8d_gbs=nb * n_it / dt / 1e9,
               lanczos_fused_launches=ctx.get("n_lanczos_fused"), minres_updates_carried=ctx.get("n_minres_rides"))
    return out


def config4(ctx, n=32768):
    from oracle.inputs import dense_spd_system      # G = rng(0) normal, A = G G^T / n + I
    A, b = dense_spd_system(n)
    ls = linsys.LinearSystem(A, b, self_adjoint=True, positive_definite=True)
    for it in range(2):
        ctx.sync()
        t0 = time.perf_counter()
        s = linsys.Cg(ls, tol=1e-8, maxiter=200)
        ctx.sync()
        dt = time.perf_counter() - t0
    n_it = s.iter
    nb = 8.0 * n * n + 10 * 8.0 * n
    return dict(config="4: dense SPD n=%d CG tol 1e-8" % n, iterations=n_it,
                iterations_per_s=n_it / dt, ms_per_iteration=dt / n_it * 1e3,
                bytes_per_iteration=nb, frac=nb * n_it / dt / 8e12, traffic_over_bytes=None, iterations_in_process=2 * n_it,
                algorithmic_gbs=nb * n_it / dt / 1e9,
                final_relres=float(s.resnorms[-1]))


def config5(ctx, nx=200, m=100, d=16, nz=None, ortho="mgs"):
    import oracle.krylov_ref as ref
    A = ref.laplace3d(nx) if nz is None else ref.laplace3d(nx, nx, nz)
    N = A.shape[0]
    b = np.random.default_rng(0).standard_normal(N)
    ls = linsys.LinearSystem(A, b, self_adjoint=True)
    try:
        s0 = deflation.DeflatedGmres(ls, tol=1e-12, maxiter=m, store_arnoldi=True, ortho=ortho)
    except utils.ConvergenceError as e:
        s0 = e.solver
    ritz = deflation.Ritz(s0)
    U = ritz._get_vectors_dev(np.argsort(np.abs(ritz.values))[:d])
    dts = []
    for it in range(6):          # first pass = warm-up; best of the rest (a solve is only ~0.2 s)
        ctx.sync()
        t0 = time.perf_counter()
        try:
            s1 = deflation.DeflatedGmres(ls, U=U, tol=1e-12, maxiter=m, ortho=ortho)
        except utils.ConvergenceError as e:
            s1 = e.solver
        ctx.sync()
        dts.append(time.perf_counter() - t0)
    dt = min(dts[1:])
    n_it = len(s1.resnorms) - 1
    # every datum once: the operator (diagonal-major copy, nd diagonals) and v_k, w out and in again around the projector
    # (operator launch -> one-launch projector -> chain: 32 N), the projector's two bases in each of its two sweeps (32 d N), the
    # Gram-Schmidt columns (8 N each, (m + 1) / 2 on average), v_{k+1} out
    Amat = ls.A._device_matrix()
    nd = Amat.diagonals or 7
    comp = 8.0 * nd * N + 8.0 * N + 32.0 * N + 32.0 * d * N + 8.0 * N * (m + 1) / 2.0 + 8.0 * N
    return dict(bytes_per_iteration=comp, frac=comp * n_it / dt / 8e12, traffic_over_bytes=None,
                iterations_in_process=m + 6 * n_it,config="5 (one-GPU shape): 3-D 7-pt %dx%dx%d (N=%d), DeflatedGmres(%d) with %d Ritz vectors, ortho=%s"
                       % (nx, nx, nx if nz is None else nz, N, m, d, ortho), iterations_per_s=n_it / dt, ms_per_iteration=dt / n_it * 1e3,
                solve_ms=[round(x * 1e3, 1) for x in dts],
                plain_relres=float(s0.resnorms[-1]), deflated_relres=float(s1.resnorms[-1]))


def general_csr(kind, n, seed=11):
    """Two operators that are NOT stencils (no diagonal structure: the CSR-stream kernel, a separate SpMV launch and w
    through HBM in every Arnoldi step - what `MatrixLinearOperator._dot`, utils.py:1593-1594, gets for any other matrix):
      band    about nine entries per row at random places within +-899 of the diagonal (the pattern of
              tests/test_gpu_halo_loopback._stencil("random")), plus 3 I
      ragged  3 ... 40 entries per row (uniform) at random places within +-20000 of the diagonal, plus 5 I
    The shifts put the origin just outside the disc of the spectrum (radius ~ sqrt(entries per row)): GMRES(100) makes progress
    and does not finish.  (Rounds 3 and 4 had 16 I / 64 I: the updated residual fell below any tolerance after 40 steps, and from
    there on EVERY iteration assembled the iterate and formed the explicit residual - what the reference does when the updated
    residual passes the tolerance, linsys.py:345-390 - so those lines timed 800 products V y per run, not the iteration:
    profiles/r05_csr_band_trace.md.)"""
    r = np.random.default_rng(seed)
    if kind == "band":
        rows = np.repeat(np.arange(n, dtype=np.int64), 8)
        cols = np.clip(rows + r.integers(-899, 900, rows.size), 0, n - 1)
        shift = 3.0
    else:
        lens = r.integers(3, 41, n)
        rows = np.repeat(np.arange(n, dtype=np.int64), lens)
        cols = np.clip(rows + r.integers(-20000, 20001, rows.size), 0, n - 1)
        shift = 5.0
    A = (sp.coo_matrix((r.standard_normal(rows.size), (rows, cols)), shape=(n, n)) + shift * sp.identity(n)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A


def config_csr(ctx, kind, n):
    A = general_csr(kind, n)
    N = A.shape[0]
    b = np.random.default_rng(0).standard_normal(N)
    Ad = ctx.csr(A)
    assert Ad.diagonals == 0, "meant to be a general CSR operator"
    X, Y = ctx.upload(b), ctx.alloc(N, 1)
    for _ in range(3):
        ctx.apply(Ad, X, 0, Y, 0, 1)
    ctx.timer_start()
    reps = 50
    for _ in range(reps):
        ctx.apply(Ad, X, 0, Y, 0, 1)
    ms = ctx.timer_stop() / reps
    same = bool(np.array_equal(Y.download()[:, 0], A.dot(b)))
    nbytes = 12.0 * A.nnz + 4.0 * (N + 1) + 16.0 * N
    ls = linsys.LinearSystem(A, b)

    def run(ncyc):
        try:
            return linsys.RestartedGmres(ls, maxiter=100, max_restarts=ncyc - 1, tol=1e-30)
        except utils.ConvergenceError as e:
            return e.solver
    run(1)
    ctx.sync()
    t0 = time.perf_counter()
    s = run(4)
    ctx.sync()
    dt = time.perf_counter() - t0
    n_it = len(s.resnorms) - 1
    lens = np.diff(A.indptr)
    return dict(config="general CSR '%s': N=%d, nnz=%d (%d ... %d per row, mean %.1f), GMRES(100) mgs" % (
                    kind, N, A.nnz, lens.min(), lens.max(), lens.mean()),
                bytes_per_iteration=nbytes + 16.0 * N + 8.0 * N * 101 / 2.0 + 8.0 * N, frac=(nbytes + 16.0 * N + 8.0 * N * 101 / 2.0 + 8.0 * N) * n_it / dt / 8e12,
                traffic_over_bytes=None, iterations_in_process=500,
                spmv_us=ms * 1e3, spmv_algorithmic_bytes=nbytes, spmv_gbs=nbytes / ms / 1e6, spmv_frac_of_8TBs=nbytes / ms / 1e6 / 8000.0,
                spmv_bit_identical_to_scipy=same, iterations_per_s=n_it / dt, ms_per_iteration=dt / n_it * 1e3,
                chain_launches=ctx.counters()["chain"], chain_fused=ctx.counters()["chain_fused"])


if __name__ == "__main__":
    ctx = _hip.get_context()
    which = sys.argv[1:] or ["3", "4", "5"]
    for w in which:
        fn = {"3": config3, "4": config4, "5": config5,
              # the per-GPU share of config 5 at its stated size: a 500 x 500 x 50 slab, 12.5 M rows (beyond the
              # register file: 48 rows per lane, eight of them in LDS), reference-order MGS and the panel form
              "5s": lambda c: config5(c, nx=500, nz=50), "5sc": lambda c: config5(c, nx=500, nz=50, ortho="cgs"),
              # operators that are not stencils: the CSR-stream SpMV kernel + chain kernel (N = 5e6 / 2e6)
              "band": lambda c: config_csr(c, "band", 5_000_000), "ragged": lambda c: config_csr(c, "ragged", 2_000_000)}[w]
        print(json.dumps(fn(ctx)), flush=True)
