"""Pin the CPU oracle (oracle/krylov_ref.py) against the golden vectors emitted by the
real reference (oracle/gen_golden.py) and the reference's own known-answer scalars
(/root/reference/test/test_convenience_wrappers.py:10-12,37-39, quoted in BASELINE.md)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import krylov_ref as ref
from oracle.inputs import (dense_spd_system, kernel_panel, lap2d_system, lap3d_system,
                           minres_jacobi_system, toy_system)

RTOL = 1e-10  # north_star tolerance (fp64, relative)


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300)


def relmax(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.max(np.abs(a - b) / np.abs(b))


KNOWN = {  # reference test/test_convenience_wrappers.py:10-12 and 37-39
    "cg": [1004.1873775173957, 1000.0003174916551, 999.9999999997555],
    "gmres": [1004.1873724888546, 1000.0003124630923, 999.999994971191],
    "minres": [1004.187372488912, 1000.0003124632159, 999.9999949713145],
    "cg_defl": [1004.1873775173271, 1000.0003174918709, 1000.0],
    "minres_defl": [1004.1873774950692, 1000.0003174918709, 1000.0],
    "gmres_defl": [1004.1873774950692, 1000.0003174918709, 1000.0],
}


def check_resnorms(got, want, tol=RTOL, explicit_last=True, explicit_tol=1e-7):
    """Residual norms from the recurrences agree to ``tol``.  When the cycle ended with an
    explicit residual (``b - A x_k`` formed from an x of norm ~1e5 and compared at the 1e-7
    level: pure cancellation, error ~ eps*|A||x|/|r|) that single entry is pinned at 1e-7;
    the iterate it is formed from is compared at ``tol`` by the caller."""
    got, want = np.asarray(got, float), np.asarray(want, float)
    assert len(got) == len(want)
    if explicit_last:
        assert relmax(got[:-1], want[:-1]) < tol
        assert relmax(got[-1:], want[-1:]) < explicit_tol
    else:
        assert relmax(got, want) < tol


def _three(x):
    return [np.sum(np.abs(x)), np.sqrt(np.dot(x, x)), np.max(np.abs(x))]


@pytest.mark.parametrize("method", ["cg", "gmres", "minres"])
def test_toy_known_answers(golden, method):
    A, b = toy_system()
    g = golden("toy")
    fn = {"cg": ref.cg, "gmres": ref.gmres, "minres": ref.minres}[method]
    kw = {"ortho": "mgs"} if method == "minres" else {}  # krypy.minres default (_convenience.py:91)
    r = fn(A, b, **kw)
    for got, want in zip(_three(r.xk), KNOWN[method]):
        assert abs(got - want) < 1e-11 * want
    check_resnorms(r.resnorms, g[method + "_resnorms"], tol=1e-9)
    assert rel(r.xk, g[method + "_x"]) < RTOL
    # the golden fixture itself reproduces the reference's literals
    for got, want in zip(_three(g[method + "_x"]), KNOWN[method]):
        assert abs(got - want) < 1e-11 * want


def test_toy_gmres_arnoldi_and_failure(golden):
    A, b = toy_system()
    g = golden("toy")
    r = ref.gmres(A, b)
    # A is symmetric: H is tridiagonal up to rounding; the band is well determined, the
    # (noise-level, 1e-7) upper triangle is not
    band = lambda H: np.triu(np.tril(H, 1), -1)  # noqa: E731
    assert rel(band(r.H), band(g["gmres_H"])) < 1e-9
    assert rel(r.H, g["gmres_H"]) < 1e-7
    assert rel(r.V, g["gmres_V"]) < 1e-6
    r = ref.gmres(A, b, maxiter=10)
    assert r.failed and r.iter == int(g["gmres_m10_iter"]) == 9
    check_resnorms(r.resnorms, g["gmres_m10_resnorms"])
    assert rel(r.xk, g["gmres_m10_xk"][:, 0]) < RTOL


def test_toy_deflated_gmres(golden):
    A, b = toy_system()
    g = golden("toy")
    U = np.zeros((100, 1))
    U[0] = 1.0
    r = ref.deflated_gmres(A, b, U)
    for got, want in zip(_three(r.xk), KNOWN["gmres_defl"]):
        assert abs(got - want) < 1e-11 * want
    assert len(r.resnorms) == len(g["gmres_defl_resnorms"])
    assert relmax(r.resnorms, g["gmres_defl_resnorms"]) < 1e-9
    assert rel(r.E, g["gmres_defl_E"]) < RTOL


def test_givens_grid(golden):
    for a, b, c, s, r in golden("kernels")["givens"]:
        c2, s2, r2 = ref.drotg(a, b)
        assert abs(c2 - c) <= 1e-15 and abs(s2 - s) <= 1e-15
        assert abs(c2 * a + s2 * b - r) <= 1e-15 * max(1.0, abs(r))


@pytest.mark.parametrize("ortho", ["mgs", "dmgs", "lanczos"])
def test_arnoldi_steps(golden, ortho):
    g = golden("kernels")
    A, b = lap2d_system(40, rhs="rng1")
    st = ref.arnoldi_init(A, b, 12, ortho=ortho)
    for _ in range(12):
        ref.arnoldi_step(st)
    assert rel(st.H, g["arn_%s_H" % ortho]) < RTOL
    assert rel(st.V, g["arn_%s_V" % ortho]) < RTOL


@pytest.mark.parametrize("ortho", ["mgs", "lanczos"])
def test_arnoldi_steps_with_M(golden, ortho):
    g = golden("kernels")
    A, b = lap2d_system(40, rhs="rng1")
    M = sp.diags(np.linspace(0.5, 1.5, A.shape[0])).tocsr()
    st = ref.arnoldi_init(A, b, 12, ortho=ortho, M=M)
    for _ in range(12):
        ref.arnoldi_step(st)
    assert rel(st.H, g["arn_%sM_H" % ortho]) < RTOL
    assert rel(st.V, g["arn_%sM_V" % ortho]) < RTOL
    assert rel(st.P, g["arn_%sM_P" % ortho]) < RTOL


def test_qr_and_projection(golden):
    g = golden("kernels")
    X, a = kernel_panel(2000, 16, seed=7)
    Y, _ = kernel_panel(2000, 16, seed=8)
    Q, R = ref.mgs_qr(X, reorthos=1)
    assert rel(Q, g["qr_Q"]) < RTOL and rel(R, g["qr_R"]) < RTOL
    Q0, R0 = ref.mgs_qr(X, reorthos=0)
    assert rel(Q0, g["qr0_Q"]) < RTOL and rel(R0, g["qr0_R"]) < RTOL
    P = ref.Projection(X, Y)
    z, Ya = P.apply_complement(a[:, 0], return_Ya=True)
    assert rel(z, g["proj_z"][:, 0]) < 1e-9
    assert rel(Ya, g["proj_Ya"][:, 0]) < RTOL


@pytest.mark.parametrize("nx,rhs", [(64, "ones"), (64, "rng1"), (128, "ones")])
def test_restarted_gmres_ladder(golden, nx, rhs):
    g = golden("lap2d_restart_nx%d_%s" % (nx, rhs))
    A, b = lap2d_system(nx, rhs=rhs)
    # open loop: same total iteration count and final residual
    r = ref.restarted_gmres(A, b, tol=1e-8, maxiter=100, max_restarts=50)
    assert len(r.resnorms) - 1 == int(g["total_iters"])
    assert not r.failed
    assert abs(r.resnorms[-1] - g["resnorms"][-1]) < 1e-6 * g["resnorms"][-1]
    # closed loop: each cycle re-seeded with the reference's x0, 1e-10 relative
    for c in range(int(g["ncycles"])):
        s = ref.gmres(A, b, x0=g["c%d_x0" % c], tol=1e-8, maxiter=100)
        check_resnorms(s.resnorms, g["c%d_resnorms" % c])
        # b = ones excites only the grid-symmetric subspace: late H columns are ill-determined
        assert rel(s.H, g["c%d_H" % c]) < (RTOL if rhs != "ones" else 1e-6)
        assert rel(s.xk, g["c%d_xk" % c]) < RTOL


@pytest.mark.parametrize("ortho", ["mgs", "dmgs"])
def test_one_cycle_nx200(golden, ortho):
    g = golden("lap2d_cycle_nx200")
    A, b = lap2d_system(200, rhs="rng1")
    s = ref.gmres(A, b, tol=1e-8, maxiter=100, ortho=ortho)
    assert s.failed and s.iter == int(g[ortho + "_iter"])
    check_resnorms(s.resnorms, g[ortho + "_resnorms"])
    assert rel(s.H, g[ortho + "_H"]) < RTOL
    assert rel(s.xk, g[ortho + "_xk"]) < RTOL
    assert rel(s.V.sum(axis=0), g[ortho + "_Vsum"]) < 1e-9
    assert rel(s.V[::997, :], g[ortho + "_Vsample"]) < 1e-9


def test_minres_jacobi(golden):
    g = golden("minres_jacobi_nx100")
    A, b, M, _ = minres_jacobi_system(100)
    s = ref.minres(A, b, M=M, tol=1e-8, maxiter=2000)
    assert s.iter == int(g["iter"]) and len(s.resnorms) == len(g["resnorms"])
    assert relmax(s.resnorms[:60], g["resnorms"][:60]) < RTOL
    assert relmax(s.resnorms, g["resnorms"]) < 1e-6   # Lanczos drift beyond ~60 steps
    assert rel(s.xk, g["xk"]) < 1e-8
    assert tuple(s.V.shape) == tuple(g["Vshape"])
    assert rel(s.H[:60, :59], g["H"][:60, :59]) < RTOL


def test_minres_cg_sparse(golden):
    g = golden("lap2d_minres_cg_nx100")
    A, b = lap2d_system(100, rhs="rng1")
    s = ref.minres(A, b, tol=1e-8, maxiter=2000)
    assert s.iter == int(g["minres_iter"])
    assert relmax(s.resnorms[:60], g["minres_resnorms"][:60]) < RTOL
    assert rel(s.xk, g["minres_xk"]) < 1e-8
    s = ref.cg(A, b, tol=1e-8, maxiter=2000)
    assert s.iter == int(g["cg_iter"])
    assert relmax(s.resnorms[:60], g["cg_resnorms"][:60]) < RTOL
    assert rel(s.xk, g["cg_xk"]) < 1e-8


def test_cg_dense(golden):
    g = golden("cg_dense_n512")
    A, b = dense_spd_system(512)
    s = ref.cg(A, b, tol=1e-8)
    assert s.iter == int(g["iter"])
    assert relmax(s.resnorms, g["resnorms"]) < RTOL
    assert rel(s.xk, g["xk"]) < RTOL
    g = golden("cg_dense_jacobi_n512")
    M = sp.diags(1.0 / np.diag(A)).tocsr()
    s = ref.cg(A, b, M=M, tol=1e-8)
    assert s.iter == int(g["iter"])
    assert relmax(s.resnorms, g["resnorms"]) < RTOL


def test_deflated_gmres_recycling(golden):
    g = golden("deflation_lap3d_nx24")
    A, b = lap3d_system(24, rhs="ones")
    # solve 0: no deflation vectors
    s0 = ref.deflated_gmres(A, b, np.zeros((A.shape[0], 0)), tol=1e-8, maxiter=300)
    assert len(s0.resnorms) - 1 == int(g["s0_iters"])
    check_resnorms(s0.resnorms, g["s0_resnorms"], tol=1e-9, explicit_tol=1e-5)
    # Ritz vectors handed to solve 1 span the same space as the reference's
    vals, U1 = ref.ritz_vectors_smallest(s0, 16, self_adjoint=True)
    Uref = g["s0_U_next"]
    Q, _ = np.linalg.qr(Uref)
    assert np.linalg.norm(U1 - Q.dot(Q.T.dot(U1))) / np.linalg.norm(U1) < 1e-6
    # solve 1 with the reference's U: iterate-for-iterate
    s1 = ref.deflated_gmres(A, b, Uref, tol=1e-8, maxiter=300)
    assert len(s1.resnorms) - 1 == int(g["s1_iters"])
    check_resnorms(s1.resnorms, g["s1_resnorms"], tol=1e-8, explicit_tol=1e-5)
    assert rel(s1.E, g["s1_E"]) < RTOL
    assert rel(s1.C, g["s1_C"]) < 1e-8
    assert rel(s1.UMlr, g["s1_UMlr"][:, 0]) < RTOL
    assert rel(s1.xk, g["s1_xk"]) < 1e-9


def test_complex_oracle_pinned(golden):
    """The complex part of the oracle (oracle/krylov_ref_c.py) against the reference's outputs on the
    seeded complex systems (tests/golden/complex_nx24.npz, made by oracle/gen_golden.py)."""
    from oracle import krylov_ref_c as kc
    from oracle.inputs import complex_systems

    g = golden("complex_nx24")
    c = complex_systems(24)
    b = c["b"]

    def check(tag, x, res, tol=RTOL):
        want = g[tag + "_resnorms"]
        assert len(res) == len(want), tag
        assert np.max(np.abs(res[:-1] - want[:-1]) / want[:-1]) < tol, tag
        assert abs(res[-1] - want[-1]) / want[-1] < 1e-6, tag            # explicit residual: cancellation
        assert np.linalg.norm(x - g[tag + "_xk"]) / np.linalg.norm(x) < 1e-10, tag

    x, res, H, R = kc.gmres(c["nonh"], b, tol=1e-10, maxiter=300)
    check("gmres", x, res)
    # (leading columns: the tiny top entries of late columns are rounding noise of MGS itself)
    assert np.abs(H[:26, :25] - g["gmres_H"][:26, :25]).max() < 1e-10
    assert np.abs(R[:25, :25] - g["gmres_R"][:25, :25]).max() < 1e-10
    x, res, _, _ = kc.gmres(c["nonh"], b, x0=c["x0"], tol=1e-10, maxiter=300)
    check("gmres_x0", x, res)
    x, res, _, _ = kc.gmres(c["L"], b, tol=1e-10, maxiter=300)
    check("gmres_realA", x, res)
    x, res, _, _ = kc.gmres(c["nonh"], b.real.copy(), tol=1e-10, maxiter=300)
    check("gmres_realb", x, res)
    x, res = kc.minres(c["hind"], b, tol=1e-10, maxiter=600)
    check("minres", x, res)
    x, res = kc.cg(c["hpd"], b, tol=1e-10, maxiter=300)
    check("cg", x, res)
    d = np.asarray(c["hpd"].diagonal()).real
    M = sp.diags(1.0 / d).tocsr()
    x, res = kc.cg(c["hpd"], b, tol=1e-10, maxiter=300, M=M)
    check("cg_jacobi", x, res)
    x, res = kc.minres(c["hind"], b, tol=1e-10, maxiter=600, M=M)
    check("minres_jacobi", x, res)
    x, res, _, _ = kc.gmres(c["nonh"], b, tol=1e-10, maxiter=300, M=M)
    check("gmres_jacobi", x, res)
    for ortho, A in (("mgs", c["nonh"]), ("dmgs", c["nonh"]), ("lanczos", c["hind"])):
        V, H, _, _ = kc.arnoldi(A, b, 12, ortho)
        assert np.abs(V - g["arn_%s_V" % ortho]).max() < 1e-12
        assert np.abs(H - g["arn_%s_H" % ortho]).max() < 1e-12
    for row in g["givens"]:
        cc, s, r = kc.givens(row[0], row[1])
        assert abs(cc - row[2]) < 1e-15 and abs(s - row[3]) < 1e-15
        assert abs(r - row[4]) <= 1e-15 * max(1.0, abs(row[4]))


# ---- BASELINE.json configs 2, 3, 4 at their stated sizes (tests/golden/config{2,3,4}_full.npz, oracle/gen_golden_full.py) ----
def test_fullsize_fixtures_are_well_formed(golden):
    """The full-size fixtures hold what tests/test_gpu_fullsize.py compares the device with - made from the unmodified
    reference at N = 10^7 / n = 32768 - and are internally consistent: Hessenberg / tridiagonal structure, residual
    histories that start at 1 and decrease, the measured movements of the reference's own output small enough to be bars."""
    g2, g3, g4 = golden("config2_full"), golden("config3_full"), golden("config4_full")
    assert int(g2["nx"]) * int(g2["ny"]) == 10_000_000 and g2["H"].shape == (101, 100) and g2["resnorms"].shape == (101,)
    assert np.all(np.tril(g2["H"], -2) == 0) and np.all(np.diag(g2["H"], -1) > 0)
    assert g2["resnorms"][0] == 1.0 and np.all(np.diff(g2["resnorms"][:-1]) <= 0)
    assert g2["Vsum"].shape == g2["Vabssum"].shape == (101,) and g2["Vsample"].shape == (501, 101)
    # the sampled rows of an orthonormal basis: every column has norm one, so the 501-row sample has about sqrt(501 / N)
    assert np.all(np.abs(np.linalg.norm(g2["Vsample"], axis=0) / np.sqrt(501 / 1e7) - 1.0) < 0.35)
    steps = int(g3["steps"])
    assert g3["H"].shape == (steps + 1, steps) and np.all(np.triu(g3["H"], 2) == 0) and np.all(np.tril(g3["H"], -2) == 0)
    assert np.allclose(np.diag(g3["H"], 1), np.diag(g3["H"], -1)[:-1], rtol=1e-12)
    assert np.all(np.diff(g3["resnorms"][:-1]) <= 1e-14)
    for key in ("sens_resnorms", "sens_H", "sens_xnorm"):
        assert 0 <= float(g3[key]) < 1e-9, key
    assert int(g4["n"]) == 32768 and len(g4["resnorms"]) == int(g4["iter"]) + 1 and g4["resnorms"][-1] <= 1e-8
    assert np.all(np.diff(g4["resnorms"]) < 0) and g4["xk"].shape == (32768,)
    assert 0 <= float(g4["sens_resnorms"]) < 1e-8 and 0 <= float(g4["sens_xk"]) < 1e-10


def _fullsize_report(line):
    import os
    print(line)
    path = os.environ.get("KRYPY_AMD_PARITY_LOG")
    if path:
        with open(path, "a") as fh:
            fh.write(line + "\n")


_FULL = __import__("os").environ.get("KRYPY_AMD_FULLSIZE_ORACLE", "0") == "1"


@pytest.mark.skipif(not _FULL, reason="ten minutes and 45 GB of host work: run with KRYPY_AMD_FULLSIZE_ORACLE=1 "
                                      "(its output is committed as profiles/r05_fullsize_parity.log)")
@pytest.mark.parametrize("config", [2, 3, 4])
def test_oracle_against_the_fullsize_fixtures(golden, config):
    """The CPU oracle (oracle/krylov_ref.py) against the reference's own output at the stated sizes of configs 2, 3, 4:
    the restatement is pinned where the benchmark runs, not only on the small fixtures."""
    try:
        from threadpoolctl import threadpool_limits
        lim = threadpool_limits(limits=1)
    except ImportError:
        lim = None
    try:
        if config == 2:
            g = golden("config2_full")
            A = ref.laplace2d(int(g["nx"]), int(g["ny"]))
            b = np.random.default_rng(0).standard_normal(A.shape[0])
            o = ref.gmres(A, b, tol=1e-8, maxiter=100)
            st = int(g["stride"])
            dev = dict(resnorms_max_rel=relmax(np.array(o.resnorms), g["resnorms"]), H_rel_fro=rel(o.H, g["H"]),
                       xk_norm_rel=abs(np.linalg.norm(o.xk) - float(g["xk_norm"])) / float(g["xk_norm"]),
                       Vsum_over_abssum_max=float(np.max(np.abs(o.V.sum(axis=0) - g["Vsum"]) / g["Vabssum"])),
                       Vsample_abs_max=float(np.max(np.linalg.norm(o.V[::st, :] - g["Vsample"], axis=0))))
            _fullsize_report("ORACLE vs REFERENCE config 2 (one GMRES(100) cycle, N = 1e7): " + ", ".join("%s = %.3e" % kv for kv in dev.items()))
            assert dev["resnorms_max_rel"] < RTOL and dev["H_rel_fro"] < RTOL and dev["xk_norm_rel"] < RTOL
            assert dev["Vsum_over_abssum_max"] < RTOL and dev["Vsample_abs_max"] < RTOL
        elif config == 3:
            g = golden("config3_full")
            A = ref.laplace2d(int(g["nx"]), int(g["ny"]))
            b = np.random.default_rng(0).standard_normal(A.shape[0])
            o = ref.minres(A, b, tol=1e-8, maxiter=int(g["steps"]), M=sp.diags(1.0 / A.diagonal()).tocsr())
            dev = dict(resnorms_max_rel=relmax(np.array(o.resnorms)[:-1], g["resnorms"][:-1]), H_rel_fro=rel(o.H, g["H"]),
                       xk_norm_rel=abs(np.linalg.norm(o.xk) - float(g["xk_norm"])) / float(g["xk_norm"]))
            _fullsize_report("ORACLE vs REFERENCE config 3 (60 MINRES + Jacobi steps, N = 1e7): " + ", ".join("%s = %.3e" % kv for kv in dev.items())
                             + ", reference's own movement: resnorms %.1e H %.1e xnorm %.1e" % (g["sens_resnorms"], g["sens_H"], g["sens_xnorm"]))
            assert dev["resnorms_max_rel"] < max(RTOL, 10 * float(g["sens_resnorms"])) and dev["H_rel_fro"] < RTOL and dev["xk_norm_rel"] < RTOL
        else:
            g = golden("config4_full")
            from oracle.inputs import dense_spd_system_blocked
            A, b = dense_spd_system_blocked(int(g["n"]))
            o = ref.cg(A, b, tol=1e-8, maxiter=200)
            assert len(o.resnorms) == len(g["resnorms"])
            dev = dict(resnorms_max_rel=relmax(np.array(o.resnorms), g["resnorms"]), xk_rel=rel(o.xk, g["xk"]))
            _fullsize_report("ORACLE vs REFERENCE config 4 (whole CG solve, n = 32768): " + ", ".join("%s = %.3e" % kv for kv in dev.items())
                             + ", reference's own movement: resnorms %.1e xk %.1e" % (g["sens_resnorms"], g["sens_xk"]))
            assert dev["resnorms_max_rel"] < max(RTOL, 30 * float(g["sens_resnorms"])) and dev["xk_rel"] < max(RTOL, 30 * float(g["sens_xk"]))
    finally:
        if lim is not None:
            lim.restore_original_limits()


@pytest.mark.skipif(not _FULL, reason="five minutes of host work: run with KRYPY_AMD_FULLSIZE_ORACLE=1 "
                                      "(its output is committed as profiles/r05_fullsize_parity.log)")
def test_oracle_against_the_config5_flow_fixture(golden):
    """The CPU oracle's gmres -> ritz_vectors_smallest -> deflated_gmres against the reference's own run of config 5's flow at
    N = 2.2 M (tests/golden/config5_flow.npz, oracle/gen_golden_full.py): histories, Ritz values, the span of the vectors."""
    g = golden("config5_flow")
    n1, m, d = int(g["n1"]), int(g["m"]), int(g["d"])
    A = ref.laplace3d(n1, n1, n1)
    b = np.random.default_rng(0).standard_normal(A.shape[0])
    o0 = ref.gmres(A, b, tol=1e-12, maxiter=m)
    vals, Uo = ref.ritz_vectors_smallest(o0, d, self_adjoint=True)
    o1 = ref.deflated_gmres(A, b, Uo, tol=1e-12, maxiter=m)
    S = Uo[g["U_rows"], :]
    C = np.linalg.lstsq(S, g["U_sample"], rcond=None)[0]
    dev = dict(plain_resnorms_max_rel=relmax(np.array(o0.resnorms)[:-1], g["plain_resnorms"][:-1]),
               ritz_values_max_rel=relmax(np.sort(np.abs(vals)), g["ritz_values_abs"]),
               deflated_resnorms_max_rel=relmax(np.array(o1.resnorms)[:-1], g["deflated_resnorms"][:-1]),
               span_on_sampled_rows=float(np.linalg.norm(S.dot(C) - g["U_sample"]) / np.linalg.norm(g["U_sample"])))
    _fullsize_report("ORACLE vs REFERENCE config 5 flow (GMRES(60) -> 16 Ritz vectors -> DeflatedGmres(60), N = 2.2e6): " +
                     ", ".join("%s = %.3e" % kv for kv in dev.items()) + ", reference's own movement of the deflated history: %.1e" % g["sens_deflated"])
    assert dev["plain_resnorms_max_rel"] < RTOL and dev["ritz_values_max_rel"] < 1e-8
    assert dev["deflated_resnorms_max_rel"] < max(RTOL, 30.0 * float(g["sens_deflated"])) and dev["span_on_sampled_rows"] < 1e-7
