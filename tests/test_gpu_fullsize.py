"""Full-size GPU runs of BASELINE.json configs 2, 3, 4 against fixtures made from the REFERENCE ITSELF at these sizes
(tests/golden/config{2,3,4}_full.npz, oracle/gen_golden_full.py; no CPU oracle run on the GPU box any more) and of the
single-GPU shapes of config 5 against the oracle / through size-independent properties.  Marked gpu (and slow)."""
import time

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import krylov_ref as ref

from tests.support.kernel_expect import expect_kernel

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


def _column_checks(block, ncols, stride, chunk=8):
    """sum, sum|.| and a strided sample of every column of a device block (downloaded a few columns at a time)."""
    sums, asums, samp = [], [], []
    for c0 in range(0, ncols, chunk):
        c = block.download(c0, min(chunk, ncols - c0))
        sums.append(c.sum(axis=0))
        asums.append(np.abs(c).sum(axis=0))
        samp.append(np.ascontiguousarray(c[::stride, :]))
        del c
    return np.concatenate(sums), np.concatenate(asums), np.concatenate(samp, axis=1)


def test_config2_whole_cycle_against_the_reference_at_full_size(hip, golden):
    """BASELINE.json config 2 at its stated size, the instantiation bench.py times (GMRES(100) through
    linsys.Gmres: operator fused into the prologue of the 40-rows-per-lane chain kernel): ONE whole restart
    cycle against the REFERENCE ITSELF - tests/golden/config2_full.npz, made by oracle/gen_golden_full.py from the
    unmodified /root/reference/krypy/linsys.py:951-997 on the same seeded inputs - iterate for iterate: every one of
    the 101 residual norms, the whole Hessenberg matrix and columns 0 / 25 / 50 / 99 of it, the iterate (norm, sum,
    strided sample), and per basis column its sum and a strided sample, at north_star's 1e-10.  (Rounds 1-4 ran the
    CPU oracle on the GPU box for this - two minutes of host time; oracle vs this fixture:
    tests/test_oracle_golden.py, profiles/r05_fullsize_parity.log.)"""
    from krypy_amd import linsys, utils

    g = golden("config2_full")
    nx, ny, stride = int(g["nx"]), int(g["ny"]), int(g["stride"])
    A = ref.laplace2d(nx, ny)
    N = A.shape[0]
    assert N == 10_000_000 and A.nnz == 49_987_000
    b = np.random.default_rng(0).standard_normal(N)
    ctx = hip
    before = ctx.counters()
    try:
        sol = linsys.Gmres(linsys.LinearSystem(A, b), maxiter=100, tol=1e-8, store_arnoldi=True)
        raise AssertionError("tolerance cannot be reached in one cycle at this N")
    except utils.ConvergenceError as e:
        sol = e.solver
    after = ctx.counters()
    H = np.array(sol.H)
    xk = np.array(sol.xk[:, 0])
    res = np.array(sol.resnorms)
    Vsum, Vabs, Vsamp = _column_checks(sol.arnoldi._V, 101, stride)
    vhead = sol.arnoldi._V.download(100, 1)[:4096, 0]
    del sol
    wres, wH = g["resnorms"], g["H"]
    assert len(res) == len(wres) == 101
    xn, wxn = float(np.linalg.norm(xk)), float(g["xk_norm"])
    _report("config 2, one GMRES(100) cycle at N = 1e7 vs the reference (fixture config2_full; bar 1e-10)",
            resnorms_max_rel=np.max(np.abs(res - wres) / wres), H_rel_fro=np.linalg.norm(H - wH) / np.linalg.norm(wH),
            xk_norm_rel=abs(xn - wxn) / wxn, xk_sample_rel=np.linalg.norm(xk[::stride] - g["xk_sample"]) / np.linalg.norm(g["xk_sample"]),
            Vsum_over_abssum_max=np.max(np.abs(Vsum - g["Vsum"]) / g["Vabssum"]),
            Vsample_abs_max_over_columns=np.max(np.linalg.norm(Vsamp - g["Vsample"], axis=0)),
            v101_head_abs=np.linalg.norm(vhead - g["v_last_head"]))
    assert np.max(np.abs(res - wres) / wres) < 1e-10
    for k in (0, 25, 50, 99):
        assert np.linalg.norm(H[: k + 2, k] - wH[: k + 2, k]) < 1e-10 * np.linalg.norm(wH[: k + 2, k]), k
    assert np.linalg.norm(H - wH) < 1e-10 * np.linalg.norm(wH)
    assert abs(xn - wxn) < 1e-10 * wxn
    assert np.linalg.norm(xk[::stride] - g["xk_sample"]) < 1e-10 * np.linalg.norm(g["xk_sample"])
    assert abs(xk.sum() - float(g["xk_sum"])) < 1e-10 * np.abs(xk).sum()
    # every basis column: its sum against the sum of moduli (the condition of a sum), its sampled entries in norm - a
    # unit vector's 501 sampled entries carry sqrt(501 / N) of its error norm; ten times that share of 1e-10
    assert np.max(np.abs(Vsum - g["Vsum"]) / g["Vabssum"]) < 1e-10
    assert np.max(np.abs(Vabs - g["Vabssum"]) / g["Vabssum"]) < 1e-10
    assert np.max(np.linalg.norm(Vsamp - g["Vsample"], axis=0)) < 1e-10 * 10.0 * np.sqrt(Vsamp.shape[0] / float(N))
    assert np.linalg.norm(vhead - g["v_last_head"]) < 1e-10 * 10.0 * np.sqrt(4096 / float(N))
    expect_kernel(after["chain_fused"] - before["chain_fused"] >= 100, "the bench's kernel (operator in the chain's prologue) ran 100 times")


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


def test_config3_against_the_reference_at_full_size(hip, golden):
    """BASELINE.json config 3 at its stated size against the REFERENCE (tests/golden/config3_full.npz: 60 steps of the
    unmodified /root/reference/krypy/linsys.py:791-853 with the Jacobi preconditioner on the N = 10^7 Laplacian,
    ortho='lanczos'), iterate for iterate.  What runs here and in no small fixture: the one-column Lanczos chain launch
    with the operator in its prologue at 40 rows per lane and the fused MINRES update on 80 MB vectors.  Residual
    history, the tridiagonal Lanczos matrix, the iterate's norm / sum / strided sample at 1e-10 - or ten times what
    the reference's own output moves under one rounding error per datum (measured by the generator WITH the reference:
    sens_* in the fixture; un-reorthogonalised Lanczos)."""
    from krypy_amd import linsys, utils
    from tests.parity_cases import ptol

    g = golden("config3_full")
    steps, stride = int(g["steps"]), int(g["stride"])
    A = ref.laplace2d(int(g["nx"]), int(g["ny"]))
    N = A.shape[0]
    b = np.random.default_rng(0).standard_normal(N)
    d = A.diagonal()
    M, Minv = sp.diags(1.0 / d).tocsr(), sp.diags(d).tocsr()
    try:
        sol = linsys.Minres(linsys.LinearSystem(A, b, M=M, Minv=Minv, self_adjoint=True), ortho="lanczos", tol=1e-8,
                            maxiter=steps, store_arnoldi=False)
        raise AssertionError("tolerance cannot be reached in 60 steps at this N")
    except utils.ConvergenceError as e:
        sol = e.solver
    res, H = np.array(sol.resnorms), np.array(sol.lanczos.H[: steps + 1, :steps])
    xk = np.array(sol.xk[:, 0])
    del sol
    sens = dict(resnorms=float(g["sens_resnorms"]), H=float(g["sens_H"]), xnorm=float(g["sens_xnorm"]))
    wres, wH, wxn = g["resnorms"], g["H"], float(g["xk_norm"])
    assert len(res) == len(wres) == steps + 1
    # (the last entry is the explicitly computed residual the failing solve ends with; the recurrence's are compared)
    dev = np.max(np.abs(res[:-1] - wres[:-1]) / wres[:-1])
    xn = float(np.linalg.norm(xk))
    _report("config 3, 60 MINRES + Jacobi steps at N = 1e7 vs the reference (fixture config3_full)", resnorms_max_rel=dev,
            bar_resnorms=ptol(sens, "resnorms"), H_rel_fro=np.linalg.norm(H - wH) / np.linalg.norm(wH), bar_H=ptol(sens, "H"),
            xk_norm_rel=abs(xn - wxn) / wxn, bar_xnorm=ptol(sens, "xnorm"),
            xk_sample_rel=np.linalg.norm(xk[::stride] - g["xk_sample"]) / np.linalg.norm(g["xk_sample"]))
    assert dev < ptol(sens, "resnorms"), (dev, sens)
    assert np.linalg.norm(H - wH) < ptol(sens, "H") * np.linalg.norm(wH)
    assert abs(xn - wxn) < ptol(sens, "xnorm") * wxn
    assert np.linalg.norm(xk[::stride] - g["xk_sample"]) < 10.0 * ptol(sens, "xnorm") * np.linalg.norm(g["xk_sample"])
    assert max(ptol(sens, k) for k in sens) < 1e-7, sens          # (the bar stays a bar)


def test_config4_against_the_reference_at_full_size(hip, golden):
    """BASELINE.json config 4 at its stated size against the REFERENCE (tests/golden/config4_full.npz: the whole CG
    solve of the unmodified /root/reference/krypy/linsys.py:593-689 on the dense SPD matrix of order 32768; 8.6 GB
    streamed per step by k_gemv_dense here).  Same number of iterations, residual history and the whole iterate at
    1e-10 - or thirty times what the reference's own output moves when its 32768-term row sums are taken in 2 / 3 / 5
    pieces (measured by the generator with the reference: sens_* in the fixture; the residual recurrence at 1e-8 ||b||
    keeps eight digits less than the iterate)."""
    from krypy_amd import linsys
    from oracle.inputs import dense_spd_system_blocked

    g = golden("config4_full")
    n = int(g["n"])
    A, b = dense_spd_system_blocked(n)
    # same inputs: b bit for bit (a seeded stream), A = G G^T / n + I to the rounding of this host's dgemm blocking
    assert np.array_equal(b[:64], g["b_head"]) and np.allclose(np.diag(A)[:64], g["A_diag_head"], rtol=1e-13, atol=0)
    t0 = time.perf_counter()
    sol = linsys.Cg(linsys.LinearSystem(A, b, self_adjoint=True, positive_definite=True), tol=1e-8, maxiter=200)
    dt = time.perf_counter() - t0
    # (size-independent properties: kappa(A) is about 5 - twenty-odd monotone iterations, the residual identity)
    assert sol.resnorms[-1] <= 1e-8 and 10 < sol.iter < 40
    assert np.all(np.diff(sol.resnorms) < 0)
    assert np.linalg.norm(b - A.dot(sol.xk[:, 0])) <= 1.001e-8 * np.linalg.norm(b)
    print("config 4: %d CG iterations, %.1f iterations/s (incl. the 8.6 GB upload)" % (sol.iter, sol.iter / dt))
    got, wres, wxk = np.array(sol.resnorms), g["resnorms"], g["xk"]
    sens, xsens = float(g["sens_resnorms"]), float(g["sens_xk"])
    assert len(got) == len(wres) and sol.iter == int(g["iter"])
    assert sens < 1e-8
    _report("config 4, the whole CG solve at n = 32768 vs the reference (fixture config4_full)",
            resnorms_max_rel=np.max(np.abs(got - wres) / wres), bar=max(1e-10, 30.0 * sens),
            first_ten_max_rel=np.max(np.abs(got[:10] - wres[:10]) / wres[:10]),
            xk_rel=np.linalg.norm(sol.xk[:, 0] - wxk) / np.linalg.norm(wxk), bar_xk=max(1e-10, 30.0 * xsens),
            iterations=len(got) - 1)
    # thirty times the reference's own movement (the factor tools/solve_fuzz.py uses), never below 1e-10; the first ten
    # iterations - residuals well above the rounding floor - at 1e-10 flat
    assert np.max(np.abs(got - wres) / wres) < max(1e-10, 30.0 * sens)
    assert np.max(np.abs(got[:10] - wres[:10]) / wres[:10]) < 1e-10
    assert np.linalg.norm(sol.xk[:, 0] - wxk) < max(1e-10, 30.0 * xsens) * np.linalg.norm(wxk)
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


def test_config5_flow_against_the_reference(hip, golden):
    """Config 5's flow - plain GMRES(60), the 16 Ritz vectors of smallest magnitude harvested on the device,
    DeflatedGmres(60) with them (recycling/linsys.py:51-103, deflation.py:738-847, 93-163) - iterate for iterate against
    the REFERENCE ITSELF run on the same problem (tests/golden/config5_flow.npz, made by oracle/gen_golden_full.py: a
    130^3 grid, N = 2.2 * 10^6 - long enough for the fused step, the device projector and the SpMM of the set-up to run
    as they do at full size).  The deflated solve depends on span(U) only; its bar is the reference's own movement when
    the Ritz vectors are perturbed by one rounding error per entry (stored with the fixture).  Rounds 1-4 ran the CPU
    oracle here (150 s of the GPU box's host time); the oracle is held to the same fixture in tests/test_oracle_golden.py."""
    import bench
    from krypy_amd import deflation, linsys, utils

    g = golden("config5_flow")
    n1, m, d = int(g["n1"]), int(g["m"]), int(g["d"])
    A = bench.laplace3d(n1, n1, n1)       # (2.2 M rows: 16 rows per lane - the one-launch projector of proj_reg.h serves it)
    N = A.shape[0]
    b = np.random.default_rng(0).standard_normal(N)
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
    r0, w0 = np.array(s0.resnorms), g["plain_resnorms"]
    assert len(r0) == len(w0) and np.max(np.abs(r0[:-1] - w0[:-1]) / w0[:-1]) < 1e-10
    assert np.allclose(np.sort(np.abs(ritz.values[idx])), g["ritz_values_abs"], rtol=1e-8)
    r1, w1, sens = np.array(s1.resnorms), g["deflated_resnorms"], float(g["sens_deflated"])
    assert len(r1) == len(w1)
    # the same subspace: the reference's Ritz vectors lie in the span of the device's (on the fixture's 2178 sampled rows, with
    # the coefficients the sampled rows themselves determine - 16 unknowns per column)
    S = U.download()[g["U_rows"], :]
    C = np.linalg.lstsq(S, g["U_sample"], rcond=None)[0]
    span = float(np.linalg.norm(S.dot(C) - g["U_sample"]) / np.linalg.norm(g["U_sample"]))
    _report("config 5 flow (GMRES -> Ritz vectors on the device -> DeflatedGmres) at N = 2.2e6 vs the reference (fixture config5_flow)",
            plain_resnorms_max_rel=np.max(np.abs(r0[:-1] - w0[:-1]) / w0[:-1]),
            deflated_resnorms_max_rel=np.max(np.abs(r1[:-1] - w1[:-1]) / w1[:-1]), bar_deflated=max(1e-10, 30.0 * sens),
            span_on_sampled_rows=span)
    assert np.max(np.abs(r1[:-1] - w1[:-1]) / w1[:-1]) < max(1e-10, 30.0 * sens)          # (measured: 2e-12)
    assert r1[-1] < r0[-1]
    assert span < 1e-7
    expect_kernel(hip.get("n_proj_reg") - pr0 >= m - 1, "hip.get(\"n_proj_reg\") - pr0 >= m - 1")          # every deflated step projected with the vector in registers


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
        expect_kernel(ctx.counters()["cgs_register"] - before >= 40, "ctx.counters()[\"cgs_register\"] - before >= 40")          # k_cgs_dots / k_cgs_update<48, ., 8>
        expect_kernel(ctx.get("n_spmv_split") > 0, "ctx.get(\"n_spmv_split\") > 0")
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


def test_config5_at_its_stated_size_on_one_device(hip):
    """BASELINE.json configs[4] at N = 10^8 (500 x 500 x 400, nnz = 698,700,000) as ONE device holds it - the N = 1 point of
    config 5's curve (/root/reference/krypy/deflation.py:93-163, recycling/linsys.py:51-103): the int32 CSR index arithmetic
    within a factor three of its limit (the last rows' products against the host's), plain GMRES(40) -> 16 Ritz vectors on the
    device -> DeflatedGmres(40): the deflation identity E = <U, A U>, an orthogonal basis, U orthogonal to the deflated basis,
    the deflated residual below the plain one, and which kernels served a vector eight times the register file.  Sizes the
    oracle cannot reach in seconds: properties, not a comparison (the flow itself is pinned at 2.2 M rows against the
    reference's fixture above)."""
    import gc
    import time
    import bench
    from krypy_amd import deflation, linsys, utils

    t0 = time.perf_counter()
    A = bench.laplace3d(500, 500, 400)
    N = A.shape[0]
    assert N == 100_000_000 and A.nnz == 698_700_000 and A.indices.dtype == np.int32
    b = np.random.default_rng(0).standard_normal(N)
    ls = linsys.LinearSystem(A, b, self_adjoint=True)
    dm = ls.A._device_matrix()
    # the operator at both ends of the index range, bit for bit against SciPy (rows 0 .. 999 and the last 1000)
    x = np.random.default_rng(1).standard_normal(N)
    X, Y = hip.upload(x), hip.alloc(N, 1)
    hip.apply(dm, X, 0, Y, 0, 1)
    head, tail = Y.get(0, 0, 1000), Y.get(0, N - 1000, 1000)
    assert np.array_equal(head, A[:1000].dot(x)) and np.array_equal(tail, A[N - 1000:].dot(x))
    del X, Y
    chain0 = hip.counters()["chain"]

    def solve(U=None):
        try:
            return deflation.DeflatedGmres(ls, U=U, tol=1e-12, maxiter=40, store_arnoldi=True, ortho="mgs")
        except utils.ConvergenceError as e:
            return e.solver
    s0 = solve()
    ritz = deflation.Ritz(s0)
    Ud = ritz._get_vectors_dev(np.argsort(np.abs(ritz.values))[:16])
    r0 = np.array(s0.resnorms)
    del s0, ritz
    gc.collect()
    pr0 = hip.get("n_proj_reg")
    s1 = solve(Ud)
    r1 = np.array(s1.resnorms)
    assert len(r0) == len(r1) == 41 and r1[-1] < r0[-1]
    U, AU = s1.projection._Ud, s1.projection._AUd
    E = hip.gemm_tn(U, 0, 16, AU, 0, 16)
    assert np.linalg.norm(E - s1.E) < 1e-10 * np.linalg.norm(E)
    n = s1.H.shape[1]
    G = hip.gemm_tn(s1.arnoldi._V, 0, n + 1, s1.arnoldi._V, 0, n + 1)
    assert np.linalg.norm(G - np.eye(n + 1)) < 1e-11
    assert np.linalg.norm(hip.gemm_tn(U, 0, 16, s1.arnoldi._V, 0, n)) < 1e-9
    expect_kernel(dm.diagonals == 7, "the banded copy of the 7-point operator exists at 698.7 M entries: %r" % (dm.diagonals,))
    expect_kernel(hip.counters()["chain"] == chain0 and hip.get("n_proj_reg") == pr0,
                  "a vector eight times the register file: per-column kernels and the four-launch projector, no register-resident launch")
    del s1, U, AU, Ud
    gc.collect()
    hip._pool_flush()
    took = time.perf_counter() - t0
    assert took < 240.0, "the full-size config-5 test took %.0f s" % took


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
    # --ortho auto: one untimed solve per (form, transport) candidate, the fastest that ran is timed - with the mailboxes on in
    # loopback the reference order runs as the chain kernel with the cross-rank stage inside its sums (csrc/chain_xr.hip)
    auto = c["ortho_auto"]
    assert c["ortho"] in ("cgs", "mgs") and c["deflation_vectors"] == 16 and auto["chosen"]["ortho"] == c["ortho"]
    assert all("ms" in cand for cand in auto["candidates"]) and c["timed_region_fallback"] is None
    mgs_xr = [cand for cand in auto["candidates"] if (cand["ortho"], cand["transport"]) == ("mgs", "xr")]
    # (a deflated step: the projector's sums - four launches per sweep on N ranks - go through the mailboxes as calls)
    expect_kernel(len(mgs_xr) == 1 and mgs_xr[0]["kernels"]["chain_in_launch_sums"] >= 100 and
                  mgs_xr[0]["per_iteration"]["allreduce_calls"] < 10,
                  "the reference-order candidate over the mailboxes took the chain kernel with the stage: %r" % (mgs_xr,))
    assert c["deflated_relres"] < c["plain_relres"] < 1.0
    assert o["value"] > 50 and 0 < o["roofline"]["frac"] < 1
