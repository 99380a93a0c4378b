"""Full-size GPU runs of BASELINE.json configs 2, 3, 4 against the CPU oracle (a minute or two of host time each) and
of the single-GPU shapes of config 5 through size-independent properties.  Marked gpu (and slow)."""
import time

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import krylov_ref as ref

pytestmark = pytest.mark.gpu


def _report(name, **vals):
    """The measured deviations of a full-size comparison, one line per test: printed (pytest -s) and, when
    KRYPY_AMD_PARITY_LOG names a file, appended there (profiles/r04_fullsize_parity.log comes from such a run)."""
    import os
    line = "PARITY %s: %s" % (name, ", ".join("%s = %.3e" % (k, float(v)) if isinstance(v, (float, np.floating)) else
                                                "%s = %s" % (k, v) for k, v in vals.items()))
    print(line)
    path = os.environ.get("KRYPY_AMD_PARITY_LOG")
    if path:
        with open(path, "a") as fh:
            fh.write(line + "\n")


def test_config2_whole_cycle_against_the_oracle_at_full_size(hip):
    """BASELINE.json config 2 at its stated size, the instantiation bench.py times (GMRES(100) through
    linsys.Gmres: operator fused into the prologue of the 40-rows-per-lane chain kernel): ONE whole restart
    cycle against the CPU oracle on the same inputs, iterate for iterate - every one of the 101 residual
    norms, Hessenberg columns 0 / 25 / 50 / 99, the iterate's norm - at north_star's 1e-10 (reference:
    linsys.py:951-997).  About two minutes of host time (one BLAS thread: the threaded ddot of a 256-core host
    is four times slower on these vectors)."""
    from krypy_amd import linsys, utils

    A = ref.laplace2d(4000, 2500)
    N = A.shape[0]
    b = np.random.default_rng(0).standard_normal(N)
    ctx = hip
    before = ctx.counters()
    try:
        sol = linsys.Gmres(linsys.LinearSystem(A, b), maxiter=100, tol=1e-8, store_arnoldi=True)
        raise AssertionError("tolerance cannot be reached in one cycle at this N")
    except utils.ConvergenceError as e:
        sol = e.solver
    after = ctx.counters()
    assert after["chain_fused"] - before["chain_fused"] >= 100      # the bench's kernel did run
    H = np.array(sol.H)
    xn = float(np.linalg.norm(sol.xk))
    res = np.array(sol.resnorms)
    vlast = sol.arnoldi._V.download(100, 1)[:, 0]
    del sol
    try:
        from threadpoolctl import threadpool_limits
        lim = threadpool_limits(limits=1)
    except ImportError:
        lim = None
    t0 = time.perf_counter()
    want = ref.gmres(A, b, tol=1e-8, maxiter=100)
    print("oracle cycle: %.1f s" % (time.perf_counter() - t0))
    if lim is not None:
        lim.restore_original_limits()
    assert len(res) == len(want.resnorms) == 101
    wres = np.array(want.resnorms)
    _report("config 2, one GMRES(100) cycle at N = 1e7 vs the oracle (bar 1e-10)",
            resnorms_max_rel=np.max(np.abs(res - wres) / wres), H_rel_fro=np.linalg.norm(H - want.H) / np.linalg.norm(want.H),
            xk_norm_rel=abs(xn - np.linalg.norm(want.xk)) / np.linalg.norm(want.xk),
            v101_abs=np.linalg.norm(vlast - want.V[:, 100]))
    assert np.max(np.abs(res - wres) / wres) < 1e-10
    for k in (0, 25, 50, 99):
        assert np.linalg.norm(H[: k + 2, k] - want.H[: k + 2, k]) < 1e-10 * np.linalg.norm(want.H[: k + 2, k]), k
    assert np.linalg.norm(H - want.H) < 1e-10 * np.linalg.norm(want.H)
    assert abs(xn - np.linalg.norm(want.xk)) < 1e-10 * np.linalg.norm(want.xk)
    assert np.linalg.norm(vlast - want.V[:, 100]) < 1e-10


def test_config3_minres_jacobi_full_size(hip):
    """2-D 5-pt Laplacian N = 10^7, MINRES + Jacobi M, ortho='lanczos', 200 steps (V and P are both
    stored: maxiter must be bounded).  Checks: monotone residuals (MINRES minimises the
    M^-1-norm of the residual), Lanczos matrix tridiagonal + symmetric, true residual of the
    returned iterate equals the last recorded one."""
    from krypy_amd import linsys, utils

    A = ref.laplace2d(4000, 2500)
    N = A.shape[0]
    b = np.random.default_rng(0).standard_normal(N)
    d = A.diagonal()
    M, Minv = sp.diags(1.0 / d).tocsr(), sp.diags(d).tocsr()
    ls = linsys.LinearSystem(A, b, M=M, Minv=Minv, self_adjoint=True)
    t0 = time.perf_counter()
    try:
        sol = linsys.Minres(ls, ortho="lanczos", tol=1e-8, maxiter=200, store_arnoldi=False)
        raise AssertionError("tolerance cannot be reached in 200 steps at this N")
    except utils.ConvergenceError as e:
        sol = e.solver
    dt = time.perf_counter() - t0
    res = np.array(sol.resnorms)
    assert len(res) == 201 and np.all(np.diff(res[:-1]) <= 1e-14)
    H = sol.lanczos.H[:199, :199]
    assert np.allclose(H, H.T, atol=1e-10) and np.all(np.triu(H, 2) == 0)
    x = sol.xk[:, 0]
    r = b - A.dot(x)
    rn = np.sqrt(np.dot(r, r / d)) / np.sqrt(np.dot(b, b / d))
    assert abs(rn - res[-1]) < 1e-8 * res[-1]
    print("config 3: %.1f MINRES iterations/s" % (200 / dt))


def _one_blas_thread():
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=1)
    except ImportError:
        return None


def test_config3_against_the_oracle_at_full_size(hip):
    """BASELINE.json config 3 at its stated size against the CPU oracle, iterate for iterate: 60 steps of MINRES with the
    Jacobi preconditioner on the N = 10^7 Laplacian (reference: linsys.py:791-853).  What runs here and in no golden
    fixture: the one-column Lanczos chain launch with the operator in its prologue at 40 rows per lane
    (k_mgs_chain<40, ., ., 5>) and the fused MINRES update on 80 MB vectors.  Residual history, the tridiagonal
    Lanczos matrix and the iterate's norm at 1e-10 - or ten times what the oracle's own output moves under one rounding
    error per datum (un-reorthogonalised Lanczos; measured, tests/parity_cases.rounding_sensitivity)."""
    from krypy_amd import linsys, utils
    from tests.parity_cases import ptol, rounding_sensitivity

    A = ref.laplace2d(4000, 2500)
    N = A.shape[0]
    b = np.random.default_rng(0).standard_normal(N)
    d = A.diagonal()
    M, Minv = sp.diags(1.0 / d).tocsr(), sp.diags(d).tocsr()
    steps = 60
    try:
        sol = linsys.Minres(linsys.LinearSystem(A, b, M=M, Minv=Minv, self_adjoint=True), ortho="lanczos", tol=1e-8,
                            maxiter=steps, store_arnoldi=False)
        raise AssertionError("tolerance cannot be reached in 60 steps at this N")
    except utils.ConvergenceError as e:
        sol = e.solver
    res, H, xn = np.array(sol.resnorms), np.array(sol.lanczos.H[: steps + 1, :steps]), float(np.linalg.norm(sol.xk))
    del sol

    def run(A_, b_):
        o = ref.minres(A_, b_, tol=1e-8, maxiter=steps, M=sp.diags(1.0 / A_.diagonal()).tocsr())
        return dict(resnorms=np.array(o.resnorms), H=np.array(o.H), xnorm=np.array([np.linalg.norm(o.xk)]))

    lim = _one_blas_thread()
    t0 = time.perf_counter()
    want = run(A, b)
    sens = rounding_sensitivity(run, A, b, seeds=(11,), elementwise=("resnorms",))
    print("oracle: 3 x 60 MINRES steps in %.1f s; rounding sensitivity %r" % (time.perf_counter() - t0, sens))
    if lim is not None:
        lim.restore_original_limits()
    assert len(res) == len(want["resnorms"]) == steps + 1
    # (the last entry is the explicitly computed residual the failing solve ends with; the recurrence's are compared)
    dev = np.max(np.abs(res[:-1] - want["resnorms"][:-1]) / want["resnorms"][:-1])
    _report("config 3, 60 MINRES + Jacobi steps at N = 1e7 vs the oracle", resnorms_max_rel=dev, bar_resnorms=ptol(sens, "resnorms"),
            H_rel_fro=np.linalg.norm(H - want["H"]) / np.linalg.norm(want["H"]), bar_H=ptol(sens, "H"),
            xk_norm_rel=abs(xn - want["xnorm"][0]) / want["xnorm"][0], bar_xnorm=ptol(sens, "xnorm"))
    assert dev < ptol(sens, "resnorms"), (dev, sens)
    assert np.linalg.norm(H - want["H"]) < ptol(sens, "H") * np.linalg.norm(want["H"])
    assert abs(xn - want["xnorm"][0]) < ptol(sens, "xnorm") * want["xnorm"][0]
    assert max(ptol(sens, k) for k in sens) < 1e-7, sens          # (the bar stays a bar)


def test_config4_against_the_oracle_at_full_size(hip):
    """BASELINE.json config 4 at its stated size against the CPU oracle: the whole CG solve on the dense SPD matrix
    of order 32768 (8.6 GB streamed per step by k_gemv_dense; reference: linsys.py:593-689).  Same number of
    iterations, residual history and iterate at 1e-10 (the system is well conditioned: kappa about 5)."""
    from krypy_amd import linsys
    from oracle.inputs import dense_spd_system

    n = 32768
    A, b = dense_spd_system(n)
    t0 = time.perf_counter()
    sol = linsys.Cg(linsys.LinearSystem(A, b, self_adjoint=True, positive_definite=True), tol=1e-8, maxiter=200)
    dt = time.perf_counter() - t0
    # (size-independent properties: kappa(A) is about 5 - twenty-odd monotone iterations, the residual identity)
    assert sol.resnorms[-1] <= 1e-8 and 10 < sol.iter < 40
    assert np.all(np.diff(sol.resnorms) < 0)
    assert np.linalg.norm(b - A.dot(sol.xk[:, 0])) <= 1.001e-8 * np.linalg.norm(b)
    print("config 4: %d CG iterations, %.1f iterations/s (incl. the 8.6 GB upload)" % (sol.iter, sol.iter / dt))
    t0 = time.perf_counter()
    want = ref.cg(A, b, tol=1e-8, maxiter=200)
    print("oracle: %d CG steps at n = %d in %.1f s" % (len(want.resnorms) - 1, n, time.perf_counter() - t0))
    got, wres = np.array(sol.resnorms), np.array(want.resnorms)
    assert len(got) == len(wres)
    # what "the same algorithm with its 32768-term row sums taken in another order" looks like from outside: the
    # oracle again with every row sum split in two halves (the residual recurrence at 1e-8 ||b|| keeps eight digits
    # less than the iterate, so its tail moves by 1e-9 relative under rounding alone)
    sens = xsens = 0.0
    for parts in (2, 3, 5):          # (the device adds 64 interleaved partial sums per row: more pieces than any of these)
        cuts = [n * i // parts for i in range(parts + 1)]

        def split_matvec(x, cuts=cuts):
            y = A[:, cuts[0]:cuts[1]].dot(x[cuts[0]:cuts[1]])
            for i in range(1, len(cuts) - 1):
                y = y + A[:, cuts[i]:cuts[i + 1]].dot(x[cuts[i]:cuts[i + 1]])
            return y

        other = ref.cg(split_matvec, b, tol=1e-8, maxiter=200)
        ores = np.array(other.resnorms)
        assert len(ores) == len(wres)
        sens = max(sens, float(np.max(np.abs(ores - wres) / wres)))
        xsens = max(xsens, float(np.linalg.norm(other.xk - want.xk) / np.linalg.norm(want.xk)))
    print("oracle's own movement under other summation orders: resnorms %.1e, xk %.1e" % (sens, xsens))
    assert sens < 1e-8
    _report("config 4, the whole CG solve at n = 32768 vs the oracle", resnorms_max_rel=np.max(np.abs(got - wres) / wres),
            bar=max(1e-10, 30.0 * sens), first_ten_max_rel=np.max(np.abs(got[:10] - wres[:10]) / wres[:10]),
            xk_rel=np.linalg.norm(sol.xk[:, 0] - want.xk) / np.linalg.norm(want.xk), bar_xk=max(1e-10, 30.0 * xsens),
            iterations=len(got) - 1)
    # thirty times the oracle's own movement (the factor tools/solve_fuzz.py uses), never below 1e-10; the first ten
    # iterations - residuals well above the rounding floor - at 1e-10 flat
    assert np.max(np.abs(got - wres) / wres) < max(1e-10, 30.0 * sens)
    assert np.max(np.abs(got[:10] - wres[:10]) / wres[:10]) < 1e-10
    assert np.linalg.norm(sol.xk[:, 0] - want.xk) < max(1e-10, 30.0 * xsens) * np.linalg.norm(want.xk)
    assert all(t[5] == 0 for t in sol.cg_trace)              # every fused step's sanity word is clean


def test_config5_shape_deflated_gmres_single_gpu(hip):
    """3-D 7-pt Laplacian 200^3 (N = 8*10^6, the per-GPU share of config 5 is 1.25*10^7), plain
    GMRES(60) to harvest 16 smallest-magnitude Ritz vectors, then DeflatedGmres with them:
    deflation identities E = <U,AU>, C = <U, A V_n>, projected residual orthogonal to U, and the
    deflated solve reduces the residual further than the plain one in the same number of steps."""
    from krypy_amd import deflation, linsys, utils

    A = ref.laplace3d(200)
    N = A.shape[0]
    b = np.random.default_rng(0).standard_normal(N)
    ls = linsys.LinearSystem(A, b, self_adjoint=True)
    try:
        s0 = deflation.DeflatedGmres(ls, tol=1e-12, maxiter=60, store_arnoldi=True)
    except utils.ConvergenceError as e:
        s0 = e.solver
    ritz = deflation.Ritz(s0)
    idx = np.argsort(np.abs(ritz.values))[:16]
    Ud = ritz._get_vectors_dev(idx)                  # stays on the device
    try:
        s1 = deflation.DeflatedGmres(ls, U=Ud, tol=1e-12, maxiter=60, store_arnoldi=True)
    except utils.ConvergenceError as e:
        s1 = e.solver
    assert s1.resnorms[-1] < s0.resnorms[-1]
    ctx = ls._ctx
    U, AU = s1.projection._Ud, s1.projection._AUd
    E = ctx.gemm_tn(U, 0, 16, AU, 0, 16)
    assert np.linalg.norm(E - s1.E) < 1e-10 * np.linalg.norm(E)
    assert np.linalg.norm(ctx.gemm_tn(U, 0, 16, U, 0, 16) - np.eye(16)) < 1e-12
    n = s1.H.shape[1]
    T = ctx.alloc(N, 1)
    ctx.apply(ls.A._device_matrix(), s1.arnoldi._V, 3, T, 0, 1)
    c3 = ctx.gemm_tn(U, 0, 16, T, 0, 1)[:, 0]
    assert np.linalg.norm(c3 - s1.C[:, 3]) < 1e-9 * max(np.linalg.norm(c3), 1e-30)
    # the Krylov basis of the projected operator is orthogonal to U^* A (range of P = ker <U, .>)
    G = ctx.gemm_tn(U, 0, 16, s1.arnoldi._V, 0, n)
    assert np.linalg.norm(G) < 1e-9


def test_config5_flow_against_the_oracle(hip):
    """Config 5's flow - plain GMRES(60), the 16 Ritz vectors of smallest magnitude harvested on the device,
    DeflatedGmres(60) with them (recycling/linsys.py:51-103, deflation.py:93-163) - iterate for iterate against the
    CPU oracle (gmres -> ritz_vectors_smallest -> deflated_gmres) on a 130^3 grid (N = 2.2 * 10^6: long enough for the
    fused step, the device projector and the SpMM of the set-up to run as they do at full size, short enough for the
    oracle).  The deflated solve depends on span(U) only; tolerances from the oracle's own movement when the Ritz
    vectors are perturbed by one rounding error per entry."""
    import bench
    from krypy_amd import deflation, linsys, utils

    A = bench.laplace3d(130, 130, 130)       # (2.2 M rows: 16 rows per lane - the one-launch projector of proj_reg.h serves it)
    N = A.shape[0]
    b = np.random.default_rng(0).standard_normal(N)
    m, d = 60, 16
    pr0 = hip.get("n_proj_reg")
    ls = linsys.LinearSystem(A, b, self_adjoint=True)

    def run(U):
        try:
            return deflation.DeflatedGmres(ls, U=U, tol=1e-12, maxiter=m, store_arnoldi=U is None)
        except utils.ConvergenceError as e:
            return e.solver

    s0 = run(None)
    ritz = deflation.Ritz(s0)
    idx = np.argsort(np.abs(ritz.values))[:d]
    U = ritz._get_vectors_dev(idx)
    s1 = run(U)
    o0 = ref.gmres(A, b, tol=1e-12, maxiter=m)
    vals, Uo = ref.ritz_vectors_smallest(o0, d, self_adjoint=True)
    o1 = ref.deflated_gmres(A, b, Uo, tol=1e-12, maxiter=m)
    o1p = ref.deflated_gmres(A, b, Uo * (1.0 + 1e-15 * np.random.default_rng(1).standard_normal(Uo.shape)), tol=1e-12,
                             maxiter=m)
    r0, w0 = np.array(s0.resnorms), np.array(o0.resnorms)
    assert len(r0) == len(w0) and np.max(np.abs(r0[:-1] - w0[:-1]) / w0[:-1]) < 1e-10
    assert np.allclose(np.sort(np.abs(ritz.values[idx])), np.sort(np.abs(vals)), rtol=1e-8)
    r1, w1, w1p = np.array(s1.resnorms), np.array(o1.resnorms), np.array(o1p.resnorms)
    sens = float(np.max(np.abs(w1p[:-1] - w1[:-1]) / w1[:-1]))
    print("deflated history: deviation %.2e, oracle's own movement %.2e" % (np.max(np.abs(r1[:-1] - w1[:-1]) / w1[:-1]), sens))
    assert len(r1) == len(w1)
    assert hip.get("n_proj_reg") - pr0 >= m - 1          # every deflated step projected with the vector in registers
    _report("config 5 flow (GMRES -> Ritz vectors on the device -> DeflatedGmres) at N = 2.2e6 vs the oracle",
            plain_resnorms_max_rel=np.max(np.abs(r0[:-1] - w0[:-1]) / w0[:-1]),
            deflated_resnorms_max_rel=np.max(np.abs(r1[:-1] - w1[:-1]) / w1[:-1]), bar_deflated=max(1e-10, 30.0 * sens))
    assert np.max(np.abs(r1[:-1] - w1[:-1]) / w1[:-1]) < max(1e-10, 30.0 * sens)          # (measured: 2e-12)
    assert r1[-1] < r0[-1]
    # the same subspace: the oracle's Ritz vectors lie in the span of the device's
    Ud = U.download()
    Q = np.linalg.qr(Ud)[0]
    assert np.linalg.norm(Uo - Q.dot(Q.T.dot(Uo))) < 1e-7 * np.linalg.norm(Uo)


def test_config5_slab_at_its_stated_size_through_the_sharded_path(hip):
    """Config 5 puts 12.5 M rows (a 500 x 500 x 50 slab of the 500 x 500 x 400 grid) on every GPU.  That is beyond
    what the register file holds (10.48 M): the 48-rows-per-lane kernels keep eight rows of w in LDS.  One such
    slab through the code path a rank takes on 8 GPUs (1-rank RCCL communicator in forced mode, ShardedCSROperator,
    panel Gram-Schmidt with all-reduced coefficients, split SpMV) - DeflatedGmres with 16 Ritz vectors harvested
    from a plain cycle: deflation identities, orthogonality of the basis, and the register-resident kernels ran."""
    import os
    from krypy_amd import _hip, deflation, dist as kdist, linsys, utils

    A = ref.laplace3d(500, 500, 50)
    N = A.shape[0]
    assert N == 12_500_000
    b = np.random.default_rng(0).standard_normal(N)
    os.environ["KRYPY_AMD_FORCE_MULTI"] = "1"
    try:
        ctx = _hip.Context(0)
        ctx.comm_init(0, 1, ctx.comm_unique_id())
    finally:
        del os.environ["KRYPY_AMD_FORCE_MULTI"]
    old = _hip._install_context_for_testing(ctx)
    try:
        op = kdist.ShardedCSROperator(A, 0, N, ctx)
        ls = linsys.LinearSystem(op, b, self_adjoint=True)
        try:
            s0 = deflation.DeflatedGmres(ls, tol=1e-12, maxiter=40, store_arnoldi=True, ortho="cgs")
        except utils.ConvergenceError as e:
            s0 = e.solver
        ritz = deflation.Ritz(s0)
        Ud = ritz._get_vectors_dev(np.argsort(np.abs(ritz.values))[:16])
        before = ctx.counters()["cgs_register"]
        try:
            s1 = deflation.DeflatedGmres(ls, U=Ud, tol=1e-12, maxiter=40, store_arnoldi=True, ortho="cgs")
        except utils.ConvergenceError as e:
            s1 = e.solver
        assert ctx.counters()["cgs_register"] - before >= 40          # k_cgs_dots / k_cgs_update<48, ., 8>
        assert ctx.get("n_spmv_split") > 0
        assert s1.resnorms[-1] < s0.resnorms[-1]
        U, AU = s1.projection._Ud, s1.projection._AUd
        E = ctx.gemm_tn(U, 0, 16, AU, 0, 16)
        assert np.linalg.norm(E - s1.E) < 1e-10 * np.linalg.norm(E)
        n = s1.H.shape[1]
        G = ctx.gemm_tn(s1.arnoldi._V, 0, n + 1, s1.arnoldi._V, 0, n + 1)
        assert np.linalg.norm(G - np.eye(n + 1)) < 1e-11
        assert np.linalg.norm(ctx.gemm_tn(U, 0, 16, s1.arnoldi._V, 0, n)) < 1e-9
    finally:
        _hip._install_context_for_testing(old)
        ctx.close()


def test_bench_config5_leg_one_rank_slab(hip):
    """bench.py --config 5 as a rank of the 8-GPU run sees it: the 500 x 500 x 50 slab (12.5 M rows) through the sharded
    code path on a 1-rank RCCL communicator (--force-sharded), plain GMRES(100) -> 16 Ritz vectors on the device ->
    DeflatedGmres(100) timed.  The JSON contract of the leg, and the deflated solve beats the plain one."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "bench.py", "--config", "5", "--force-sharded", "--nz", "50", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline"], cwd=root, env=env, capture_output=True, text=True,
                         timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    o = json.loads(out.stdout.strip().splitlines()[-1])
    print(json.dumps(o)[:1500])
    c = o["config"]
    assert o["n_gpus"] == 1 and o["unit"] == "iterations/s" and o["dtype"] == "f64" and o["scaling"] == "strong"
    assert c["n"] == 12_500_000 and c["rows_per_gpu"] == 12_500_000 and c["iterations_timed"] == 200
    assert c["ortho"] == "cgs" and c["deflation_vectors"] == 16
    assert c["deflated_relres"] < c["plain_relres"] < 1.0
    assert o["value"] > 50 and 0 < o["roofline"]["frac"] < 1
